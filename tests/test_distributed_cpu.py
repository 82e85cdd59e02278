"""The N > 1 path on CPU: two gloo ranks exercise the camera sharding, the parameter broadcast and the
single flat-buffer gradient all-reduce (mc_nerf_amd/distributed.py) with the real MC_Model on the CPU
(stage 1 of training - camera parameters only - needs no render, so its forward/backward run here)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_cameras_partitions_like_distributed_sampler():
    from mc_nerf_amd.distributed import shard_cameras
    for world in (1, 2, 4, 8):
        seen = []
        lens = set()
        for r in range(world):
            ids = shard_cameras(110, epoch=3, rank=r, world=world, seed=42)
            lens.add(len(ids))
            seen += ids
        assert len(lens) == 1                                  # every rank takes the same number of steps
        assert set(seen) == set(range(110))                    # all cameras covered (padding repeats a few)
        assert len(seen) == ((110 + world - 1) // world) * world
    assert shard_cameras(110, 0, 0, 2, seed=1) != shard_cameras(110, 1, 0, 2, seed=1)   # set_epoch reshuffles


def _build_model(batch=64):
    from mc_nerf_amd import synthetic as S
    from mc_nerf_amd.model import MC_Model
    sp = S.make_sys_param("cpu", samples=32, scale=2, batch=batch, H=16, W=16, coarse=(4, 32, [2]), fine=(8, 64, [4]))
    torch.manual_seed(0)
    m = MC_Model(sp)
    # the per-step validation rays come from the HIP ray generator (no CPU path, by design): stub them out here,
    # this file tests the data-parallel plumbing only
    m.get_rays = lambda pose, img_id, intr_inv: (torch.zeros(256, 3), torch.zeros(256, 3))
    return m, sp


def _stage1_grads(model, sp, cam, calib):
    from mc_nerf_amd.model import MC_NeRF_Loss
    wpts, pts = calib
    data = (torch.zeros(1, 256, 3), torch.tensor([cam]), wpts, pts, wpts, pts)
    for p in model.parameters():
        p.grad = None
    loss_dict, *_ = model(data, 0, "CAM_PARAM_EPOCH", 0.0)
    MC_NeRF_Loss(sp)(loss_dict, "CAM_PARAM_EPOCH").backward()
    return {n: (p.grad.clone() if p.grad is not None else None) for n, p in model.named_parameters()}


def _worker(rank, world, port, q):
    try:
        _worker_body(rank, world, port, q)
    except Exception as e:                       # surface the failure instead of a queue timeout
        import traceback
        q.put((rank, "error", traceback.format_exc(), None, None))
        raise


def _worker_body(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from mc_nerf_amd import distributed as D
    from mc_nerf_amd import synthetic as S
    r, w, dev = D.init_distributed(backend="gloo")
    assert (r, w, dev.type) == (rank, world, "cpu")
    model, sp = _build_model()
    S.init_cameras_near_gt(model, noise=0.01, seed=rank)       # ranks start different ...
    with torch.no_grad():
        for p in model.nerf.parameters():
            p.add_(float(rank))
    sync = D.FlatGradSync(model, world)
    sync.broadcast_parameters()                                # ... and must leave equal to rank 0
    chk = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    ref = chk.clone()
    dist.broadcast(ref, src=0)
    same_after_bcast = bool(torch.equal(chk, ref))
    calib = S.calibration_points(sp["gt_pose"], sp["intr_mat"][0])
    cam = D.shard_cameras(model.train_numb, 0, rank, world, seed=1)[0]
    sync.prepare()
    mine = _stage1_grads(model, sp, cam, calib)
    # NeRF weights have no gradient in stage 1; give two of them synthetic ones to exercise the gather path
    for name, p in list(model.nerf.named_parameters())[:2]:
        p.grad = torch.full_like(p, float(rank + 1))
        mine["nerf." + name] = p.grad.clone()
    sync.sync()
    out = {n: (p.grad.clone() if p.grad is not None else None) for n, p in model.named_parameters()}
    as_np = lambda d: {k: (None if v is None else v.numpy().copy()) for k, v in d.items()}   # pickle by value
    sym_count = sync.asymmetric_steps()                        # both ranks had gradients for the same tensors so far
    # an ASYMMETRIC step: only rank 1 has a gradient for one NeRF tensor
    name3, p3 = list(model.nerf.named_parameters())[2]
    strict = D.FlatGradSync(model, world, check_flags=True)    # reads the flags back: DDP's any-rank semantics
    strict.prepare()
    if rank == 1:
        p3.grad = torch.full_like(p3, 4.0)
    strict.sync()
    any_rank = None if p3.grad is None else float(p3.grad.mean())
    # default mode: a local pattern seen for the first time is verified against the reduced flags -> the disagreement is found,
    # the any-rank rule applies and the object reads the flags back from then on
    sync.prepare()
    if rank == 1:
        p3.grad = torch.full_like(p3, 4.0)
    sync.sync()
    lazy_any_rank = None if p3.grad is None else float(p3.grad.mean())
    extra = dict(sym_count=sym_count, any_rank=any_rank, lazy_any_rank=lazy_any_rank, switched=sync.check_flags,
                 asym_count=sync.asymmetric_steps())
    q.put((rank, same_after_bcast, as_np(mine), as_np(out), extra))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_flat_allreduce_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for r in res:
        assert r[1] != "error", r[2]
    res = sorted(res, key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, b0, mine0, out0, x0), (_, b1, mine1, out1, x1) = res
    assert b0 and b1
    # flag handling: with check_flags a gradient present on ONE rank reaches both (4 / 2 ranks); the default mode verifies each
    # new local pattern once, finds the disagreement, applies the same rule and keeps reading the flags back
    assert x0["sym_count"] == 0 and x1["sym_count"] == 0
    assert x0["any_rank"] == 2.0 and x1["any_rank"] == 2.0
    assert x0["lazy_any_rank"] == 2.0 and x1["lazy_any_rank"] == 2.0
    assert x0["switched"] and x1["switched"] and x0["asym_count"] == 0 and x1["asym_count"] == 0
    n_checked = 0
    for name in out0:
        if mine0[name] is None and mine1[name] is None:
            assert out0[name] is None
            continue
        mean = (mine0[name] + mine1[name]) / 2                 # DDP semantics: SUM then divide by the world size
        assert abs(out0[name] - mean).max() <= 1e-6, name
        assert (out0[name] == out1[name]).all(), name           # both ranks hold identical averaged gradients
        n_checked += 1
    assert n_checked >= 8                                       # 6 camera tensors + the 2 synthetic NeRF gradients


def test_single_process_sync_is_identity():
    from mc_nerf_amd import distributed as D
    from mc_nerf_amd import synthetic as S
    model, sp = _build_model()
    S.init_cameras_near_gt(model)
    sync = D.FlatGradSync(model, 1)
    calib = S.calibration_points(sp["gt_pose"], sp["intr_mat"][0])
    sync.prepare()
    g = _stage1_grads(model, sp, 3, calib)
    sync.sync()
    for n, p in model.named_parameters():
        if g[n] is not None:
            assert torch.equal(p.grad, g[n])
    # one message: every gradient + one "some rank has a gradient" flag per parameter tensor
    assert sync.arena.numel() == sum(p.numel() for _, p in model.named_parameters() if not _.startswith("nerf.")) + \
        sum(n.flat_params().numel() for n in sync.nets) + len(list(model.parameters()))
    # stage 1 touches no NeRF parameter: they stay without a gradient, as in the reference (the optimiser skips them)
    assert all(p.grad is None for n, p in model.named_parameters() if n.startswith("nerf."))


def test_prepare_drops_stale_grads_no_doubling():
    """after sync() the grads are views of the arena; a following backward into the same arena must not be
    accumulated onto them (zero_grad(set_to_none=False) keeps `.grad` tensors alive)."""
    from mc_nerf_amd import distributed as D
    from mc_nerf_amd import synthetic as S
    model, sp = _build_model()
    S.init_cameras_near_gt(model)
    sync = D.FlatGradSync(model, 1)
    calib = S.calibration_points(sp["gt_pose"], sp["intr_mat"][0])
    sync.prepare()
    g1 = _stage1_grads(model, sp, 3, calib)
    sync.sync()
    first = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    for p in model.parameters():                     # what optimizer.zero_grad(set_to_none=False) does
        if p.grad is not None:
            p.grad.zero_()
    sync.prepare()
    _stage1_grads(model, sp, 3, calib)
    sync.sync()
    for n, p in model.named_parameters():
        if n in first:
            assert torch.allclose(p.grad, first[n], rtol=0, atol=1e-7), n      # same step twice: same gradient, not 2x


def _ddp_worker(rank, world, port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
        torch.set_num_threads(1)
        from torch.nn.parallel import DistributedDataParallel as DDP
        from mc_nerf_amd import distributed as D
        from mc_nerf_amd import synthetic as S
        from mc_nerf_amd.model import MC_NeRF_Loss
        D.init_distributed(backend="gloo")
        model, sp = _build_model()
        S.init_cameras_near_gt(model, noise=0.01, seed=rank)       # ranks start different: DDP's constructor broadcasts rank 0's
        calib = S.calibration_points(sp["gt_pose"], sp["intr_mat"][0])
        ddp = DDP(model, find_unused_parameters=True)              # exactly main.py:61
        same = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
        ref = same.clone()
        dist.broadcast(ref, src=0)
        cam = D.shard_cameras(model.train_numb, 0, rank, world, seed=1)[0]
        local = _stage1_grads(model, sp, cam, calib)               # this rank's own gradients (plain module call, no reducer)
        for p in model.parameters():
            p.grad = None
        wpts, pts = calib
        data = (torch.zeros(1, 256, 3), torch.tensor([cam]), wpts, pts, wpts, pts)
        loss_dict, *_ = ddp(data, 0, "CAM_PARAM_EPOCH", 0.0)       # main.py:82 through the DDP wrapper
        MC_NeRF_Loss(sp)(loss_dict, "CAM_PARAM_EPOCH").backward()
        out = {n: (None if p.grad is None else p.grad.numpy().copy()) for n, p in model.named_parameters()}
        keys = list(ddp.state_dict().keys())
        q.put((rank, bool(torch.equal(same, ref)), {k: (None if v is None else v.numpy().copy()) for k, v in local.items()}, out, keys))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        import traceback
        q.put((rank, "error", traceback.format_exc(), None, None))
        raise


@pytest.mark.timeout(300)
def test_stock_ddp_wrapper_camera_stage_gloo():
    """The reference wraps MC_Model in torch's DistributedDataParallel(find_unused_parameters=True) (main.py:61).  The
    parameters here are views into flat buffers; this checks on two gloo ranks that the stock wrapper accepts them:
    constructor broadcast, a camera-stage step through `ddp(...)`, DDP's averaged gradients == the mean of the ranks' own,
    and the "module."-prefixed state-dict keys the reference's checkpoints carry."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ddp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for r in res:
        assert r[1] != "error", r[2]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, b0, l0, o0, k0), (_, b1, l1, o1, k1) = res
    assert b0 and b1
    assert all(k.startswith("module.") for k in k0) and any(k.startswith("module.nerf.nerf_fine.") for k in k0)
    n = 0
    for name in o0:
        if l0[name] is None and l1[name] is None:
            continue                                             # untouched by the stage (DDP leaves zeros or None)
        mean = (l0[name] + l1[name]) / 2
        assert abs(o0[name] - mean).max() <= 1e-6 and (o0[name] == o1[name]).all(), name
        n += 1
    assert n >= 6                                               # the six camera parameter tensors
