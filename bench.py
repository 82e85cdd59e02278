#!/usr/bin/env python3
"""Headline benchmark: train rays/sec of the MC-NeRF hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--rays R] [--precision P] [--also P,P,...] [--mode train|render]

Workload = BASELINE.json configs[1]: Ball_Lego-shaped rig (110 cameras, 800x800), coarse 4x128 (64 samples)
+ fine 8x256 (128-sample grid), joint intrinsic/extrinsic optimisation stage (GLOBAL_OPTIM_EPOCH), synthetic
images and random-init weights (no dataset / network here).  One step = one full pass of the hot path over a
batch of R rays of one camera per rank: camera parametrisation -> ray generation -> coarse MLP -> composite ->
selection -> fine MLP -> composite -> loss -> backward (composite, dX chain, dW, ray-gen) -> ONE gradient
all-reduce (N > 1) -> RAdam step.  Inputs are resident in HBM before the timed region.

`--gpus N` with N > 1 and no torchrun environment: this process (which never touches the GPU) starts N fresh rank
processes with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set and waits for them; under
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` the ranks are the given processes.
Rank 0 prints ONE JSON line (see DESIGN.md "Measurement"): the headline precision mode (`--precision`; default the
fp32-grade split-f16 mode `f16x3`: the reference's arithmetic is fp32) timed for `--steps` steps with nothing but the
step itself inside the timed region, the per-kernel HIP-event times from a separate untimed pass right after it, plus
`by_precision` with the other modes of `--also` timed in the same run (>= 20 steps each).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic FLOPs per evaluated sample, forward: 2 x weight MACs.  The backward chain (dX) and the
# weight gradients (dW) are one forward-equivalent each: a training step is 3x forward.
F_FINE, F_COARSE = 2 * 629248, 2 * 101632
PEAK_F32_MFMA_TFLOPS = 157.3       # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_F16_MFMA_TFLOPS = 2500.0      # ibid., "Peak BF16/FP16 MFMA ~2.5 PF dense"
PEAK_HBM_GBS = 8000.0              # ibid., "HBM3E peak BW 8.0 TB/s spec" (6.29 TB/s measured float4 copy)
# HBM bytes per evaluated fine sample that each kernel's CONTRACT makes it move.
#  fp32 / split-f16 words: 10 layer slots x 256 x 4 B (+ encoding 256, SH 128, 1-bit masks 320, output 16, index 8)
B32 = {"fwd": 10 * 1024 + 256 + 128 + 320 + 16 + 8,
       "bwd": 10 * 1024 + 128 + 320 + 256 + 128 + 32 + 8,
       "dw": 2 * 1024 * 9 + 2 * (1024 + 256) + 2 * (1024 + 128)}
#  16-bit modes: 10 slots x 256 x 2 B (+ encoding 128, sh 64, masks 320, output 16, index 8)
B16 = {"fwd": 10 * 512 + 128 + 64 + 320 + 16 + 8,
       "bwd": 10 * 512 + 64 + 320 + 64 + 32 + 8,
       "dw": 2 * 512 * 8 + (2 * 512 + 128) + (512 + 128) + 2 * (512 + 64)}    # the skip layer reads [hidden | encoded] against its dY in one segment
#  f16x3 (hi + lo planes): 10 slots x 256 x 4 B (+ encoding 256, sh.2 outputs 128 fp32, masks 320, output 16, index 8);
#  backward: d(sh.2) 128 written, sh.2 outputs 128 read; weight gradients: twice the 16-bit kernel's fragments
BX3 = {"fwd": 10 * 1024 + 256 + 128 + 320 + 16 + 8,
       "bwd": 10 * 1024 + 128 + 320 + 128 + 32 + 8,
       "dw": 2 * (2 * 512 * 8 + (2 * 512 + 128) + (512 + 128) + 2 * (512 + 64))}
DTYPE_TEXT = {"f32": "f32", "f16x3": "f16x3 (fp32-grade: every operand hi + lo f16 = 22 significand bits, 3 f16 MFMAs per product, fp32 accumulate / bias / epilogues, 4-byte saved operands)",
              "f16": "f16 (single-pass f16 MFMA operands, fp32 accumulate, 2-byte workspaces)",
              "bf16": "bf16 (single-pass bf16 MFMA operands, fp32 accumulate, 2-byte workspaces)"}
RIG_NAMES = {"ball": "Ball_Lego", "array": "Array_Ficus", "halfball": "HalfBall_Materials", "room": "Room_Statue"}


def workload_text(args):
    return (f"{RIG_NAMES[args.rig]}-shaped 110-view {args.img}x{args.img}, coarse 4x128 @{args.samples} + fine 8x256 "
            f"@{args.samples * args.scale}-grid{' (<= 128 kept per ray, random cap)' if args.samples * args.scale > 128 else ''}, "
            "GLOBAL_OPTIM stage, fwd+bwd+allreduce+RAdam")


def launch_ranks(args, argv):
    """Parent of a self-launched multi-rank run: starts `--gpus` fresh rank processes (never initialises the GPU,
    never re-execs), forwards their output, exits with the worst return code.  Mirrors what
    utils/distributed_init.py:11-34 + `torchrun` do for the reference's main.py."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    # poll all ranks: a rank that dies (import error, out of memory) leaves the others blocked in the rendezvous or in
    # the next collective until the RCCL timeout, so the first failure ends the run
    deadline = time.time() + float(os.environ.get("BENCH_LAUNCH_TIMEOUT_S", "3600"))
    rc = 0
    live = list(procs)
    while live:
        for p in list(live):
            r = p.poll()
            if r is not None:
                live.remove(p)
                if r != 0:
                    rc = rc or (r if 0 < r < 256 else 1)
        if rc or time.time() > deadline:
            for p in live:
                p.terminate()
            for p in live:
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
            return rc or 124
        time.sleep(0.05)
    return rc


class KernelTimer:
    """HIP-event timing of selected C-ABI calls on the stream they are launched on (torch's current stream)."""
    BASES = ("mcnerf_mlp_fwd", "mcnerf_mlp_bwd", "mcnerf_mlp_dw")

    def __init__(self):
        self.records = {}
        self.enabled = False

    def install(self):
        import torch
        from mc_nerf_amd import _lib, ops
        orig = _lib.call
        timer = self

        def timed(name, *args):
            base = name.replace("_16", "")
            if not timer.enabled or base not in timer.BASES:
                return orig(name, *args)
            key = (base[len("mcnerf_mlp_"):], args[1])           # (fwd | bwd | dw, net width)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            r = orig(name, *args)
            b.record()
            timer.records.setdefault(key, []).append((a, b))
            return r

        _lib.call = timed
        ops._lib.call = timed

    def reset(self):
        self.records = {}

    def summary(self):
        return {k: sum(a.elapsed_time(b) for a, b in v) / len(v) for k, v in self.records.items()}


def cpu_info():
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        import psutil
        phys = psutil.cpu_count(logical=False) or os.cpu_count()
    except Exception:
        phys = os.cpu_count()
    return model, int(phys)


def cpu_baseline(n_rays=2048, reps=3):
    """The oracle (a plain-PyTorch CPU port of the reference path, parity-pinned to the reference in tests/) timed
    on the host cores on a bounded sample of the same workload: cfg-2 nets, `n_rays` rays, one
    full-size warm-up then best of `reps`, train (forward + backward) and render (no-grad) separately, all
    physical cores."""
    import torch
    from oracle import mcnerf_oracle as O
    model, phys = cpu_info()
    torch.set_num_threads(phys)
    torch.manual_seed(0)
    cfg = O.RenderCfg(samples=64, scale=2)
    pc = {k: v.requires_grad_(True) for k, v in O.init_params(cfg.coarse, 1).items()}
    pf = {k: v.requires_grad_(True) for k, v in O.init_params(cfg.fine, 2).items()}
    g = torch.Generator().manual_seed(n_rays)
    o = torch.nn.functional.normalize(torch.randn(n_rays, 3, generator=g), dim=-1) * 3.0
    d = torch.nn.functional.normalize(-o + 0.5 * torch.randn(n_rays, 3, generator=g), dim=-1)
    jit = torch.rand(n_rays, 1, generator=g) * 7.0 / 64
    e = [torch.randn(n_rays, s, generator=g) for s in (64, 64, 128)]
    gt = torch.rand(n_rays, 3, generator=g)
    k_fine = [0]

    def train():
        t0 = time.perf_counter()
        r = O.render_rays_train(pc, pf, cfg, d, o, 1.0, jit, e[0], e[1], e[2])
        O.rgb_loss(r["rgb_c"], r["rgb_f"], gt).backward()
        k_fine[0] = int(r["idx_f"].shape[0])
        return time.perf_counter() - t0

    def render():
        t0 = time.perf_counter()
        with torch.no_grad():
            O.render_rays_test(pc, pf, cfg, d, o, e[0], e[1], e[2])
        return time.perf_counter() - t0

    train()
    t_train = min(train() for _ in range(reps))
    render()
    t_render = min(render() for _ in range(reps))
    return {"value": n_rays / t_train, "unit": "rays/s", "cores": phys, "cpu_model": model, "kind": "port",
            "render_value": n_rays / t_render, "render_unit": "rays/s (no-grad render_rays_test)",
            "sample": f"oracle train fwd+bwd and no-grad render, cfg-2 nets, {n_rays} rays ({k_fine[0]} fine samples), "
                      f"full-size warm-up then best of {reps}, {phys} threads"}


def parity_probe(precision, dev):
    """Max |rgb - oracle| of one small train render in `precision` (256 rays, same draws): what the mode's arithmetic
    costs against the fp32 CPU oracle on the bench's own nets (random init)."""
    import torch
    from mc_nerf_amd.model import NeRF_Model
    from mc_nerf_amd import synthetic as S
    from oracle import mcnerf_oracle as O
    cfg = O.RenderCfg(samples=64, scale=2)
    pc, pf = O.init_params(cfg.coarse, 1), O.init_params(cfg.fine, 2)
    g = torch.Generator().manual_seed(0)
    n = 256
    o = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1) * 3.0
    d = torch.nn.functional.normalize(-o + 0.4 * torch.randn(n, 3, generator=g), dim=-1)
    jit = torch.rand(n, 1, generator=g) * 7.0 / 64
    e = [torch.randn(n, s, generator=g) for s in (64, 64, 128)]
    with torch.no_grad():
        r = O.render_rays_train(pc, pf, cfg, d, o, 1.0, jit, e[0], e[1], e[2])
        sp = S.make_sys_param(dev, samples=64, scale=2, batch=256, H=64, W=64, precision=precision)
        m = NeRF_Model(sp).to(dev)
        m.nerf_coarse.load_state_dict(pc)
        m.nerf_fine.load_state_dict(pf)
        rgb_c, rgb_f = m.render_rays_train(d.to(dev), o.to(dev), 0, 1.0, jitter=jit.to(dev), eps_c=e[0].to(dev),
                                           eps_sel=e[1].to(dev), eps_f=e[2].to(dev))
    return {"rgb_c_max_abs_err": float((rgb_c.cpu() - r["rgb_c"]).abs().max()),
            "rgb_f_max_abs_err": float((rgb_f.cpu() - r["rgb_f"]).abs().max()),
            "what": "train render of 256 rays, random-init cfg-2 nets, same draws, vs the fp32 CPU oracle"}


def run_precision(precision, args, steps, warmup, rank, world, dev, timer, images_cache):
    """Builds the cfg-2 model in `precision`, runs warmup + `steps` timed steps; returns the per-mode record."""
    import torch
    import torch.distributed as dist
    from mc_nerf_amd import distributed as D
    from mc_nerf_amd import synthetic as S
    from mc_nerf_amd.model import MC_Model, MC_NeRF_Loss, RAdam
    from mc_nerf_amd.data import DeviceImageSet

    torch.manual_seed(42 + rank)                   # main.py:274-277: seed + rank
    H = W = args.img
    sp = S.make_sys_param(dev, samples=args.samples, scale=args.scale, batch=args.rays, H=H, W=W, barf_mask=False,
                          precision=precision, rig=args.rig)
    model = MC_Model(sp).to(dev)
    S.init_cameras_near_gt(model, noise=1e-3)
    loss_fn = MC_NeRF_Loss(sp)
    opt = RAdam(model.parameters(), lr=5e-4, weight_decay=4e-4)
    sync = D.FlatGradSync(model, world)
    sync.broadcast_parameters()
    C = model.train_numb
    if "wpts" not in images_cache:
        wpts, pts = S.calibration_points(sp["gt_pose"], sp["intr_mat"][0])
        images_cache["wpts"], images_cache["pts"] = wpts.to(dev), pts.to(dev)
        images_cache["images"] = DeviceImageSet.synthetic(C, H, W, dev, channels=4, seed=7)     # uint8 RGBA resident in HBM (282 MB)
    wpts, pts, images = images_cache["wpts"], images_cache["pts"], images_cache["images"]
    cams = D.shard_cameras(C, 0, rank, world, seed=42)
    counts, ar_events = [], []

    def step(i, timed):
        cam = cams[i % len(cams)]
        data = (images, torch.tensor([cam]), wpts, pts, wpts, pts)
        loss_dict, _, _, _ = model(data, 20, "GLOBAL_OPTIM_EPOCH", 0.6)
        loss = loss_fn(loss_dict, "GLOBAL_OPTIM_EPOCH")
        opt.zero_grad(set_to_none=True)
        sync.prepare()
        loss.backward()
        if timed and world > 1:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            sync.sync()
            b.record()
            ar_events.append((a, b))
        else:
            sync.sync()
        opt.step()
        if timed:
            counts.append(model.nerf.last_selection[1].clone())

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(warmup):
        step(i, False)
    barrier()
    t0 = time.perf_counter()
    for i in range(steps):                          # the timed region: nothing but the steps
        step(warmup + i, True)
    barrier()
    dt = time.perf_counter() - t0
    # per-kernel launch durations (HIP events on the launch stream around the three fine / coarse MLP calls): a separate,
    # untimed pass over the same workload
    timer.reset()
    timer.enabled = True
    for i in range(min(5, steps)):
        step(warmup + steps + i, False)
    barrier()
    timer.enabled = False
    tmax = torch.tensor([dt], device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    # parameters must be identical on every rank after synchronised steps (DDP invariant)
    in_sync = True
    if world > 1:
        flat = torch.cat([p.detach().reshape(-1).float() for p in model.parameters()])
        lo, hi = flat.clone(), flat.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        in_sync = bool(torch.equal(lo, hi))
    finite = bool(all(torch.isfinite(p).all() for p in model.parameters()))
    skipped = int(opt.skipped_steps())              # optimiser steps the overflow guard refused (inf / NaN gradients): must be 0
    k_mean = float(torch.stack(counts).float().mean())
    ks = timer.summary()
    rec = {"precision": precision, "value": args.rays * world * steps / dt, "unit": "rays/s", "steps": steps,
           "ms_per_step": dt / steps * 1e3, "fine_samples_per_ray": k_mean / args.rays, "finite": finite,
           "skipped_optimizer_steps": skipped, "valid": bool(finite and skipped == 0),
           "kernel_ms": {f"mlp_{k}<{w}>": v for (k, w), v in sorted(ks.items())}}
    if world > 1:
        rec["allreduce_ms"] = sum(a.elapsed_time(b) for a, b in ar_events) / max(1, len(ar_events))
        rec["params_identical_across_ranks"] = in_sync
        rec["asymmetric_grad_steps"] = sync.asymmetric_steps()      # ranks disagreeing on which tensors have gradients: must be 0
    # roofline of the fine-net kernels with ALGORITHMIC work per launch over the HIP-event launch time
    mfma_peak = PEAK_F32_MFMA_TFLOPS if precision == "f32" else PEAK_F16_MFMA_TFLOPS
    contract = B16 if precision in ("f16", "bf16") else (BX3 if precision == "f16x3" else B32)
    per_call = {}
    for k in ("fwd", "bwd", "dw"):
        ms = ks.get((k, 256))
        if ms:
            tf = F_FINE * k_mean / (ms * 1e-3) / 1e12
            gbs = contract[k] * k_mean / (ms * 1e-3) / 1e9
            per_call[f"mlp_{k}<256>"] = {"ms": ms, "mfma_TFLOPs": tf, "mfma_frac": tf / mfma_peak, "contract_bytes_per_launch": contract[k] * k_mean,
                                         "hbm_GBs": gbs, "hbm_frac": gbs / PEAK_HBM_GBS}
    rec["per_call"] = per_call
    rec["step_algorithmic_tflop"] = 3 * (F_FINE * k_mean + F_COARSE * args.rays * args.samples) / 1e12
    rec["step_mfma_frac"] = rec["step_algorithmic_tflop"] / (dt / steps) / mfma_peak
    del model, opt, sync
    torch.cuda.empty_cache()
    return rec, k_mean


# the sources that define the MLP kernels whose traffic is recorded (everything they include)
MLP_KERNEL_SOURCES = ("mcnerf_common.h", "mcnerf_kernels.h", "mcnerf_16.h", "mcnerf_x3.h", "mlp16_fwd.hip", "mlp16_bwd.hip", "mlp16_dw.hip",
                      "mlp_x3_fwd.hip", "mlp_x3_bwd.hip", "mlp_x3_dw.hip", "mlp_fwd.hip", "mlp_bwd.hip", "mlp_dw.hip")


def csrc_digest():
    """Content digest of the MLP kernels' sources (the GPU box has no .git): a recorded profile belongs to the kernels it names."""
    import hashlib
    h = hashlib.sha1()
    d = os.path.join(ROOT, "mc_nerf_amd", "csrc")
    for f in sorted(MLP_KERNEL_SOURCES):
        h.update(f.encode())
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(precision, kernel_key):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this command
    (scripts/pmc_bench.sh -> scripts/pmc_traffic.py: separate --pmc FETCH_SIZE / WRITE_SIZE runs, KB units, FETCH doubled on
    gfx950 as MI355X_MICROARCH.md prescribes).  Pre-recorded, not measured in this run; dropped (None) when the kernel
    sources have changed since the recording.  Returns (bytes | None, source text | None)."""
    path = os.path.join(ROOT, "profiles", f"pmc_traffic_{precision}.json")
    if not os.path.isfile(path):
        return None, None
    names = {"f16": {"mlp_fwd<256>": "mlp16_fwd_kernel<256, true, false>", "mlp_bwd<256>": "mlp16_bwd_kernel<256, false>",
                     "mlp_dw<256>": "dw16_stream_kernel<256, false>"},
             "bf16": {"mlp_fwd<256>": "mlp16_fwd_kernel<256, true, true>", "mlp_bwd<256>": "mlp16_bwd_kernel<256, true>",
                      "mlp_dw<256>": "dw16_stream_kernel<256, true>"},
             "f16x3": {"mlp_fwd<256>": "mlp_x3_fwd_kernel<256, true>", "mlp_bwd<256>": "mlp_x3_bwd_kernel<256>",
                       "mlp_dw<256>": "dwx3_stream_kernel<256>"},
             "f32": {"mlp_fwd<256>": "mlp_fwd_kernel<256, true>", "mlp_bwd<256>": "mlp_bwd_kernel<256>"}}[precision]
    rec = json.load(open(path))
    if rec.get("csrc_digest") != csrc_digest():
        return None, f"profiles/{os.path.basename(path)} was recorded for other kernel sources (digest {rec.get('csrc_digest')}): not used"
    k = names.get(kernel_key)
    kern = rec.get("kernels", {})
    if k in kern:
        return kern[k]["hbm_bytes_per_launch"], (f"pre-recorded rocprofv3 PMC passes of this command: profiles/{os.path.basename(path)} "
                                                  f"(N = 32768, kernel sources {rec.get('csrc_digest')})")
    return None, None


def run_rank(args):
    import torch
    import torch.distributed as dist
    from mc_nerf_amd import distributed as D
    rank, world, dev = D.init_distributed()
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the process group has {world} rank(s)")
    if args.selftest:                              # launcher / rendezvous check without kernels (runs on CPU over gloo)
        t = torch.ones(4, device=dev) * (rank + 1)
        if world > 1:
            dist.all_reduce(t)
        if rank == 0:
            print(json.dumps({"selftest": True, "n_gpus": world, "allreduce_ok": bool(float(t[0]) == world * (world + 1) / 2),
                              "backend": dist.get_backend() if world > 1 else None}))
        if world > 1:
            dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    timer = KernelTimer()
    timer.install()
    cache = {}
    if args.mode == "render":
        out = run_render(args, rank, world, dev)
        if rank == 0:
            print(json.dumps(out))
        if world > 1:
            dist.destroy_process_group()
        return
    head, k_mean = run_precision(args.precision, args, args.steps, args.warmup, rank, world, dev, timer, cache)
    others = {}
    for p in [q for q in args.also.split(",") if q and q != args.precision]:
        others[p], _ = run_precision(p, args, max(20, args.steps // 5), 3, rank, world, dev, timer, cache)
    if rank == 0:
        pc = head["per_call"]
        # The dominant KERNEL is the longest single launch.  In the register-chain modes (f16x3, f16, bf16) every fine-net call
        # is one launch and the weight-gradient kernel is the longest; in f32 mlp_dw is 15 launches per call (aggregate in
        # `per_call`) and the longest launch is a chain.  `roofline` prices that kernel against the roof that BINDS it (the
        # larger of its two fractions): the weight-gradient kernel streams its GEMM operands once and is HBM-bound -- achieved =
        # contract (algorithmic) bytes per launch / its duration, `traffic` = the PMC-measured bytes -- the chains are MFMA-bound
        # (algorithmic FLOPs against the dense f16 peak).  The other roof of the same kernel sits in `other_roof`.
        single = pc if args.precision != "f32" else {k: v for k, v in pc.items() if k != "mlp_dw<256>"}
        dom = max(single, key=lambda k: single[k]["ms"])
        mfma_peak = PEAK_F32_MFMA_TFLOPS if args.precision == "f32" else PEAK_F16_MFMA_TFLOPS
        traffic, tsrc = pmc_traffic(args.precision, dom) if args.rays == 32768 else (None, None)
        hbm_roof = {"bound": "hbm", "achieved": pc[dom]["hbm_GBs"], "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": pc[dom]["hbm_frac"],
                    "contract_bytes_per_launch": pc[dom]["contract_bytes_per_launch"],
                    "pmc_over_contract": (traffic / pc[dom]["contract_bytes_per_launch"]) if traffic else None}
        mfma_roof = {"bound": "mfma", "achieved": pc[dom]["mfma_TFLOPs"], "peak": mfma_peak, "unit": "TFLOP/s", "frac": pc[dom]["mfma_frac"]}
        first, second = (hbm_roof, mfma_roof) if pc[dom]["hbm_frac"] > pc[dom]["mfma_frac"] else (mfma_roof, hbm_roof)
        roof = {"bound": first["bound"], "kernel": dom, "ms": pc[dom]["ms"], "achieved": first["achieved"], "peak": first["peak"],
                "unit": first["unit"], "frac": first["frac"], "traffic": traffic, "traffic_unit": "HBM bytes per launch",
                "traffic_source": tsrc, "contract_bytes_per_launch": hbm_roof["contract_bytes_per_launch"],
                "pmc_over_contract": hbm_roof["pmc_over_contract"], "other_roof": second,
                "per_call": pc, "step_algorithmic_tflop": head["step_algorithmic_tflop"], "step_mfma_frac": head["step_mfma_frac"]}
        out = {
            "metric": "train rays/sec (coarse+fine, 64+128 samples)", "value": head["value"], "unit": "rays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": head["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPE_TEXT[args.precision], "data": "synthetic",
            "config": {"workload": workload_text(args), "precision": args.precision, "rays_per_step_per_gpu": args.rays,
                       "fine_samples_per_ray": head["fine_samples_per_ray"],
                       "parallelism": f"dp{world} (cameras sharded, 1 flat all-reduce/step)"},
            "roofline": roof,
            "by_precision": {p: {k: v for k, v in r.items() if k != "precision"} for p, r in others.items()},
        }
        for k in ("allreduce_ms", "params_identical_across_ranks", "asymmetric_grad_steps", "finite", "skipped_optimizer_steps", "valid"):
            if k in head:
                out[k] = head[k]
        if world == 1 and not args.no_cpu_baseline:
            out["parity"] = parity_probe(args.precision, dev)
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
        bad = [p for p, r in [(args.precision, head)] + list(others.items()) if not r["valid"]]
        if bad:                 # a mode whose steps were skipped by the overflow guard (or went non-finite) measured nothing
            print(f"bench.py: INVALID measurement in mode(s) {bad}: non-finite parameters or optimiser steps skipped", file=sys.stderr)
            if world > 1:
                dist.destroy_process_group()
            sys.exit(3)
    if world > 1:
        dist.destroy_process_group()


def run_render(args, rank, world, dev):
    """No-grad demo / validation render (model/mc_nerf.py:106-122): whole 800x800 images of the
    cfg-2 model in `batch`-ray chunks, uncapped fine selection; rays/s of render_rays_test."""
    import torch
    import torch.distributed as dist
    from mc_nerf_amd import distributed as D
    from mc_nerf_amd import synthetic as S
    from mc_nerf_amd.model import MC_Model
    torch.manual_seed(42 + rank)
    H = W = 800
    import tempfile
    kw = dict(samples=64, scale=2, batch=args.rays, H=H, W=W, barf_mask=False, precision=args.precision)
    tmp = tempfile.mkdtemp(prefix=f"mcnerf_bench_r{rank}_")
    trained = MC_Model(S.make_sys_param(dev, root_weight=tmp, **kw)).to(dev)        # random-init weights, written in the
    ckpt = trained.nerf.save_model(trained, 0)                                      # reference's checkpoint format ...
    del trained
    model = MC_Model(S.make_sys_param(dev, mode=1, demo_ckpt=ckpt, **kw)).to(dev).eval()    # ... and loaded the way the demo does
    cams = D.shard_cameras(model.train_numb, 0, rank, world, seed=42)
    n_img = max(1, args.steps)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    with torch.no_grad():
        model.render_image_device(cams[0])
        barrier()
        t0 = time.perf_counter()
        for i in range(n_img):
            model.render_image_device(cams[i % len(cams)])
        barrier()
        dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    rays = H * W * n_img * world
    return {"metric": "render rays/sec (no-grad, coarse+fine 64+128, uncapped selection)", "value": rays / dt, "unit": "rays/s",
            "n_gpus": world, "steps": n_img, "warmup": 1, "ms_per_step": dt / n_img * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": DTYPE_TEXT[args.precision], "data": "synthetic",
            "config": {"workload": "Ball_Lego-shaped 800x800 images rendered in chunks of `batch` rays (demo / validation path)",
                       "precision": args.precision, "rays_per_chunk": args.rays, "images": n_img}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rays", type=int, default=32768, help="rays per step per GPU (config `batch`)")
    ap.add_argument("--samples", type=int, default=64, help="coarse samples per ray (config `samples`; reference default 128)")
    ap.add_argument("--scale", type=int, default=2, help="fine grid = samples x scale (config `scale`; reference default 5), capped at 128 kept per ray")
    ap.add_argument("--img", type=int, default=800, help="image side (BASELINE cfg 5: 1600)")
    ap.add_argument("--rig", default="ball", choices=["ball", "array", "halfball", "room"], help="camera rig of the synthetic scene")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", default="f16x3", choices=["f32", "f16x3", "f16", "bf16"],
                    help="MFMA mode of the MLP kernels: split-f16 f16x3 (fp32-grade, the headline: the reference computes in fp32) or "
                         "exact f32 -- the 1e-4 parity modes; single-pass f16 / bf16 -- 16-bit operand modes with their own stated error")
    ap.add_argument("--also", default="f16,bf16,f32", help="other precision modes measured in the same run (by_precision)")
    ap.add_argument("--mode", default="train", choices=["train", "render"])
    ap.add_argument("--selftest", action="store_true", help="rendezvous / launcher check only (no kernels; works on CPU)")
    args = ap.parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args, sys.argv[1:]))
    run_rank(args)


if __name__ == "__main__":
    main()
