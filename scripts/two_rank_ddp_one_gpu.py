"""main.py's training protocol under STOCK DistributedDataParallel on the HIP path, two gloo ranks sharing ONE GPU
(RCCL itself needs one GPU per rank): `DistributedDataParallel(MC_Model, find_unused_parameters=True)` (main.py:60-62), the
three RAdam / ExponentialLR sets built with the requires_grad_ toggles of main.py:176-207, and main.py:78-89's step
(optimizer = opt_list[model.opt_idx]; zero_grad; forward; loss; backward; step; sched.step) for two steps of EACH stage.
Then the same steps with this build's FlatGradSync (one flat all-reduce) from the same initial state and draws.
Checks: parameters bit-identical across the ranks on both paths, and equal between the two paths up to the summation order
of the weight-gradient atomics."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from _ranks import PG_TIMEOUT, run_ranks        # noqa: E402

BUDGET_S = 420      # < the pytest timeout of tests/test_z_multirank_gpu.py
STAGES = ["CAM_PARAM_EPOCH"] * 2 + ["GLOBAL_OPTIM_EPOCH"] * 2 + ["FINE_TUNE_EPOCH"] * 2


def generate_optimizer(model, steps_per_epoch, RAdam):
    """main.py:176-207 literally (stage lrs of config/config.yaml: 5e-3 camera stage, 5e-4 the NeRF stages; one epoch per stage)."""
    cam_lr, optim_lr, fine_lr, wd = 5e-3, 5e-4, 5e-4, 4e-4
    for name, p in model.named_parameters():
        p.requires_grad_(name.split(".")[0] != "nerf")
    opt_cam = RAdam(filter(lambda p: p.requires_grad, model.parameters()), lr=cam_lr, eps=1e-8, weight_decay=wd)
    sched_cam = torch.optim.lr_scheduler.ExponentialLR(opt_cam, (0.005 / cam_lr) ** (1.0 / steps_per_epoch))
    for _, p in model.named_parameters():
        p.requires_grad_(True)
    opt_global = RAdam(filter(lambda p: p.requires_grad, model.parameters()), lr=optim_lr, eps=1e-8, weight_decay=wd)
    sched_global = torch.optim.lr_scheduler.ExponentialLR(opt_global, 1.0)
    model.weights_pose.requires_grad_(False)
    opt_fine = RAdam(filter(lambda p: p.requires_grad, model.parameters()), lr=fine_lr, eps=1e-8, weight_decay=wd)
    sched_fine = torch.optim.lr_scheduler.ExponentialLR(opt_fine, 1.0)
    for _, p in model.named_parameters():
        p.requires_grad_(True)
    return [opt_cam, opt_global, opt_fine], [sched_cam, sched_global, sched_fine]


def worker(rank, world, port):
    from mc_nerf_amd import distributed as D, synthetic as S
    from mc_nerf_amd.model import MC_Model, MC_NeRF_Loss, RAdam
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=PG_TIMEOUT)
    H = W = 48
    sp = S.make_sys_param(dev, samples=32, scale=2, batch=1024, H=H, W=W, coarse=(4, 32, [2]), fine=(8, 64, [4]), precision="f16x3")
    wpts, pts = S.calibration_points(sp["gt_pose"], sp["intr_mat"][0])
    wpts, pts = wpts.to(dev), pts.to(dev)
    img = torch.rand(1, H * W, 3, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
    loss_fn = MC_NeRF_Loss(sp)

    def fresh_model():
        torch.manual_seed(7)                                    # the same initial parameters on both paths (and ranks: broadcast below)
        m = MC_Model(sp).to(dev)
        S.init_cameras_near_gt(m, noise=1e-3, seed=0)
        return m

    def flat(m):
        return torch.cat([p.detach().reshape(-1) for p in m.parameters()])

    results = {}
    for path in ("ddp", "flat"):
        model = fresh_model()
        cams = D.shard_cameras(model.train_numb, 0, rank, world, seed=1)
        if path == "ddp":
            wrapped = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0], find_unused_parameters=True)    # main.py:61
            sync = None
        else:
            wrapped = model
            sync = D.FlatGradSync(model, world)
            sync.broadcast_parameters()
        opt_list, sched_list = generate_optimizer(model, len(STAGES) // 3, RAdam)
        torch.manual_seed(100 + rank)                           # main.py:274-277: seed + rank; the same draws on both paths
        for step, stage in enumerate(STAGES):
            optimizer = opt_list[model.opt_idx] if step else opt_list[0]     # (main.py:79 reads the index the previous forward left)
            data = (img, torch.tensor([cams[step]]), wpts, pts, wpts, pts)
            optimizer.zero_grad(set_to_none=True)
            if sync:
                sync.prepare()
            loss_dict, *_ = wrapped(data, step, stage, step / len(STAGES))
            optimizer = opt_list[model.opt_idx]                 # the stage's optimiser (the first step of a stage in main.py uses the
            loss = loss_fn(loss_dict, stage)                    # previous stage's index for zero_grad only)
            loss.backward()
            if sync:
                sync.sync()
            optimizer.step()
            sched_list[model.opt_idx].step()
        f = flat(model)
        ref = f.clone()
        dist.broadcast(ref, src=0)
        results[path] = (f.cpu(), bool(torch.equal(f, ref)), int(sum(o.skipped_steps() for o in opt_list)))
        if sync:
            results["asym"] = sync.asymmetric_steps()
        del wrapped
    a, b = results["ddp"][0], results["flat"][0]
    rel = float((a - b).abs().max() / a.abs().max())
    res = (rank, results["ddp"][1], results["flat"][1], rel, results["ddp"][2] + results["flat"][2], results["asym"])
    dist.barrier()
    dist.destroy_process_group()
    return res


if __name__ == "__main__":
    res = run_ranks(worker, 2, BUDGET_S)
    print(res)
    ok = all(r[1] and r[2] and r[3] < 1e-5 and r[4] == 0 and r[5] == 0 for r in res)
    assert ok, "stock-DDP vs FlatGradSync check FAILED"
    print("stock DDP (find_unused_parameters) == FlatGradSync over 2 steps of each stage, replicas bit-identical: check: OK")
