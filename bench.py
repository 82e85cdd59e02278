#!/usr/bin/env python3
"""Headline benchmark: train rays/sec of the MC-NeRF hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--rays R]

Workload = BASELINE.json configs[1]: Ball_Lego-shaped rig (110 cameras, 800x800), coarse 4x128 (64 samples)
+ fine 8x256 (128-sample grid), joint intrinsic/extrinsic optimisation stage (GLOBAL_OPTIM_EPOCH), synthetic
images and random-init weights (no dataset / network here).  One step = one full pass of the hot path over a
batch of R rays of one camera per rank: camera parametrisation -> ray generation -> coarse MLP -> composite ->
selection -> fine MLP -> composite -> loss -> backward (composite, dX chain, dW, ray-gen) -> ONE gradient
all-reduce (N > 1) -> RAdam step.  Inputs are resident in HBM before the timed region.

For N > 1 launch as `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`.
Rank 0 prints ONE JSON line (see DESIGN.md "Measurement").
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic FLOPs per evaluated sample, forward: 2 x weight MACs
F_FINE, F_COARSE = 2 * 629248, 2 * 101632
PEAK_F32_MFMA_TFLOPS = 157.3       # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_F16_MFMA_TFLOPS = 2500.0      # ibid., "Peak BF16/FP16 MFMA ~2.5 PF dense"
PEAK_HBM_GBS = 8000.0              # ibid., "HBM3E peak BW 8.0 TB/s spec" (6.29 TB/s measured float4 copy)
# algorithmic HBM bytes per evaluated fine sample: fp32 operands the training step keeps for the
# weight-gradient GEMMs (10 layer slots x 256 x 4 B) + encoded input (256 B) + SH coefficients (128 B) +
# 1-bit ReLU masks (10 x 32 B) + output (16 B) + index (8 B)
B_FWD = 10 * 1024 + 256 + 128 + 320 + 16 + 8
B_BWD = 10 * 1024 + 128 + 320 + 256 + 128 + 32 + 8      # dY slots + dsh written; masks, enc, sh, out/d_out, index read
B_DW = 2 * 1024 * 9 + 2 * (1024 + 256) + 2 * (1024 + 128)   # 9 WxW segments, 2 enc segments, sh.2 and sigma.2 segments


class KernelTimer:
    """HIP-event timing of selected C-ABI calls on the stream they are launched on (torch's current stream)."""

    def __init__(self):
        self.records = {}
        self.enabled = False

    def install(self):
        from mc_nerf_amd import _lib
        orig = _lib.call
        timer = self

        def timed(name, *args):
            base = name.replace("_f16x3", "")
            if not timer.enabled or base not in ("mcnerf_mlp_fwd", "mcnerf_mlp_bwd", "mcnerf_mlp_dw"):
                return orig(name, *args)
            key = (base, args[1])           # (entry point, net width)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            r = orig(name, *args)
            b.record()
            timer.records.setdefault(key, []).append((a, b))
            return r

        _lib.call = timed
        from mc_nerf_amd import ops
        ops._lib.call = timed

    def summary(self):
        return {k: sum(a.elapsed_time(b) for a, b in v) / len(v) for k, v in self.records.items()}


def cpu_baseline(n_rays=1536):
    """The oracle (a plain-PyTorch CPU port of the reference path, parity-pinned in tests/) timed on the host
    cores on a bounded sample of the same workload: train forward+backward of cfg-2 nets on n_rays rays."""
    from oracle import mcnerf_oracle as O
    torch.manual_seed(0)
    cfg = O.RenderCfg(samples=64, scale=2)
    pc = {k: v.requires_grad_(True) for k, v in O.init_params(cfg.coarse, 1).items()}
    pf = {k: v.requires_grad_(True) for k, v in O.init_params(cfg.fine, 2).items()}

    def run(n):
        g = torch.Generator().manual_seed(n)
        o = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1) * 3.0
        d = torch.nn.functional.normalize(-o + 0.5 * torch.randn(n, 3, generator=g), dim=-1)
        jit = torch.rand(n, 1, generator=g) * 7.0 / 64
        e = [torch.randn(n, s, generator=g) for s in (64, 64, 128)]
        t0 = time.perf_counter()
        r = O.render_rays_train(pc, pf, cfg, d, o, 1.0, jit, e[0], e[1], e[2])
        O.rgb_loss(r["rgb_c"], r["rgb_f"], torch.rand(n, 3, generator=g)).backward()
        return time.perf_counter() - t0, r["idx_f"].shape[0]

    run(128)
    dt, k = run(n_rays)
    return {"value": n_rays / dt, "unit": "rays/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle train fwd+bwd, cfg-2 nets, {n_rays} rays, {k} fine samples, 1 timed run after warm-up"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rays", type=int, default=32768, help="rays per step per GPU (config `batch`)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", default="f16x3", choices=["f32", "f16x3"],
                    help="MFMA mode of the MLP chain kernels: exact fp32, or split-f16 (fp32-grade, 3 f16 MFMAs per product)")
    args = ap.parse_args()

    from mc_nerf_amd import distributed as D
    rank, world, dev = D.init_distributed()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    from mc_nerf_amd import synthetic as S
    from mc_nerf_amd.model import MC_Model, MC_NeRF_Loss, RAdam

    torch.manual_seed(42 + rank)                   # main.py:274-277: seed + rank
    H = W = 800
    sp = S.make_sys_param(dev, samples=64, scale=2, batch=args.rays, H=H, W=W, barf_mask=False, precision=args.precision)
    model = MC_Model(sp).to(dev)
    S.init_cameras_near_gt(model, noise=1e-3)
    loss_fn = MC_NeRF_Loss(sp)
    opt = RAdam(model.parameters(), lr=5e-4, weight_decay=4e-4)
    sync = D.FlatGradSync(model, world)
    sync.broadcast_parameters()
    C = model.train_numb
    wpts, pts = S.calibration_points(sp["gt_pose"], sp["intr_mat"][0])
    wpts, pts = wpts.to(dev), pts.to(dev)
    from mc_nerf_amd.data import DeviceImageSet
    images = DeviceImageSet.synthetic(C, H, W, dev, channels=4, seed=7)     # uint8 RGBA resident in HBM (282 MB)
    cams = D.shard_cameras(C, 0, rank, world, seed=42)
    timer = KernelTimer()
    timer.install()
    counts = []

    def step(i):
        cam = cams[i % len(cams)]
        data = (images, torch.tensor([cam]), wpts, pts, wpts, pts)
        loss_dict, _, _, _ = model(data, 20, "GLOBAL_OPTIM_EPOCH", 0.6)
        loss = loss_fn(loss_dict, "GLOBAL_OPTIM_EPOCH")
        opt.zero_grad(set_to_none=True)
        sync.prepare()
        loss.backward()
        sync.sync()
        opt.step()
        if timer.enabled:
            counts.append(model.nerf.last_selection[1].clone())

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    barrier()
    timer.enabled = True
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    barrier()
    dt = time.perf_counter() - t0
    timer.enabled = False
    tmax = torch.tensor([dt], device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    if rank == 0:
        k_mean = float(torch.stack(counts).float().mean())
        ks = timer.summary()
        # dominant kernel = the slowest of the fused fine-net forward / backward chain / weight-gradient kernels.
        # Both roofs are evaluated with ALGORITHMIC work per launch (FLOPs: 2 x MACs x evaluated samples; bytes:
        # the operands the kernel's contract makes it move) over the HIP-event launch time; the binding one
        # (larger fraction) is reported.
        cand = {"mlp_fwd<256>": (ks.get(("mcnerf_mlp_fwd", 256)), B_FWD),
                "mlp_bwd<256>": (ks.get(("mcnerf_mlp_bwd", 256)), B_BWD),
                "mlp_dw<256>": (ks.get(("mcnerf_mlp_dw", 256)), B_DW)}
        kern_ms = {k: v[0] for k, v in cand.items() if v[0]}
        mfma_peak = PEAK_F32_MFMA_TFLOPS if args.precision == "f32" else PEAK_F16_MFMA_TFLOPS

        def roof_of(name):
            secs = cand[name][0] * 1e-3
            ach_tf = F_FINE * k_mean / secs / 1e12
            ach_gbs = cand[name][1] * k_mean / secs / 1e9
            if ach_tf / mfma_peak >= ach_gbs / PEAK_HBM_GBS:
                r = {"bound": "mfma", "kernel": name, "achieved": ach_tf, "peak": mfma_peak, "unit": "TFLOP/s", "frac": ach_tf / mfma_peak}
            else:
                r = {"bound": "hbm", "kernel": name, "achieved": ach_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": ach_gbs / PEAK_HBM_GBS}
            r["other_roof"] = {"mfma_TFLOPs": ach_tf, "mfma_frac": ach_tf / mfma_peak, "hbm_GBs": ach_gbs, "hbm_frac": ach_gbs / PEAK_HBM_GBS}
            return r

        # The dominant KERNEL is the longest single launch: the fused forward or the fused backward chain (one launch
        # each per call).  mcnerf_mlp_dw is 15 launches per call (one per weight segment, <= 1.5 ms each); its
        # aggregate is reported beside them in `per_call`.
        dom = max((k for k in kern_ms if k != "mlp_dw<256>"), key=kern_ms.get)
        roof = roof_of(dom)
        per_call = {k: {kk: vv for kk, vv in roof_of(k).items() if kk in ("bound", "achieved", "unit", "frac")} for k in kern_ms}
        # HBM traffic of the dominant kernel per launch: rocprofv3 PMC passes of this same command, collected and
        # corrected as MI355X_MICROARCH.md prescribes (scripts/pmc_traffic.py; see profiles/*_pmc_traffic.json)
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01i_pmc_traffic.json" if args.precision == "f16x3" else "r01c_pmc_traffic.json")
        if os.path.exists(tpath) and args.rays == 32768:
            sfx = "_h" if args.precision == "f16x3" else ""
            names = {"mlp_fwd<256>": f"mlp_fwd{sfx}_kernel<256, true>", "mlp_bwd<256>": f"mlp_bwd{sfx}_kernel<256>"}
            kern = json.load(open(tpath))["kernels"]
            if names[dom] in kern:
                traffic = kern[names[dom]]["hbm_bytes_per_launch"]
        total_rays = args.rays * world * args.steps
        out = {
            "metric": "train rays/sec (coarse+fine, 64+128 samples)", "value": total_rays / dt, "unit": "rays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.precision == "f32" else "f16x3 (split-f16 MFMA operands, fp32 accumulate/storage)", "data": "synthetic",
            "config": {"workload": "Ball_Lego-shaped 110-view 800x800, coarse 4x128 @64 + fine 8x256 @128-grid, "
                                   "GLOBAL_OPTIM stage, fwd+bwd+allreduce+RAdam",
                       "precision": args.precision, "rays_per_step_per_gpu": args.rays, "fine_samples_per_ray": k_mean / args.rays,
                       "parallelism": f"dp{world} (cameras sharded, 1 flat all-reduce/step)"},
            "roofline": dict(roof, traffic=traffic,
                             traffic_unit="HBM bytes per launch (rocprofv3 PMC, profiles/*_pmc_traffic.json)",
                             kernel_ms=kern_ms, per_call=per_call,
                             step_algorithmic_tflops=3 * (F_FINE * k_mean + F_COARSE * args.rays * 64) / 1e12),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
