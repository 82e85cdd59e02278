"""Two data-parallel ranks sharing ONE GPU over gloo: exercises the real HIP backward into the FlatGradSync
arena, the single all-reduce and the fused optimiser on the N > 1 path (RCCL itself needs >= 2 GPUs).
Checks: parameters stay bit-identical across ranks after several steps, and the averaged gradient equals
the mean of the two ranks' local gradients."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from _ranks import PG_TIMEOUT, run_ranks        # noqa: E402

BUDGET_S = 300      # < the pytest timeout of tests/test_z_multirank_gpu.py


def worker(rank, world, port):
    from mc_nerf_amd import distributed as D, synthetic as S
    from mc_nerf_amd.model import MC_Model, MC_NeRF_Loss, RAdam
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=PG_TIMEOUT)
    torch.manual_seed(42 + rank)
    H = W = 64
    sp = S.make_sys_param(dev, samples=64, scale=2, batch=2048, H=H, W=W)
    model = MC_Model(sp).to(dev)
    S.init_cameras_near_gt(model, noise=1e-3, seed=rank)
    loss_fn = MC_NeRF_Loss(sp)
    opt = RAdam(model.parameters(), lr=5e-4, weight_decay=4e-4)
    sync = D.FlatGradSync(model, world)
    sync.broadcast_parameters()
    wpts, pts = S.calibration_points(sp["gt_pose"], sp["intr_mat"][0])
    wpts, pts = wpts.to(dev), pts.to(dev)
    img = torch.rand(1, H * W, 3, device=dev)
    cams = D.shard_cameras(model.train_numb, 0, rank, world, seed=1)
    ok = True
    for step in range(4):
        data = (img, torch.tensor([cams[step]]), wpts, pts, wpts, pts)
        loss_dict, *_ = model(data, 20, "GLOBAL_OPTIM_EPOCH", 0.6)
        loss = loss_fn(loss_dict, "GLOBAL_OPTIM_EPOCH")
        opt.zero_grad(set_to_none=True)
        sync.prepare()
        loss.backward()
        assert model.nerf.grad_arena_used, "HIP backward did not use the arena"
        local = sync.arena.clone()                      # nets' local gradients (camera part is appended in sync())
        sync.sync()
        n_net = sum(sync.net_sizes)
        gathered = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(gathered, local)
        mean = sum(gathered) / world
        ok &= bool(torch.allclose(sync.arena[:n_net], mean[:n_net], rtol=1e-6, atol=1e-9))
        ok &= all(p.grad.data_ptr() == v.data_ptr() for p, v in
                  zip(sync.nets[1].ordered_parameters(), sync.nets[1].grad_views(sync.arena[sync.net_sizes[0]:n_net])))
        opt.step()
    flat = torch.cat([n.flat_params() for n in sync.nets] + [p.detach().reshape(-1) for p in sync.cam_params])
    ref = flat.clone()
    dist.broadcast(ref, src=0)
    res = (rank, ok, bool(torch.equal(flat, ref)), float(loss.detach()))
    dist.barrier()
    dist.destroy_process_group()
    return res


if __name__ == "__main__":
    res = run_ranks(worker, 2, BUDGET_S)
    print(res)
    assert all(r[1] and r[2] for r in res), "two-rank check FAILED"
    print("two-rank one-GPU data-parallel check: OK")
