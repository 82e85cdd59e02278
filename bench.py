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
Rank 0 prints ONE JSON line (see DESIGN.md "Measurement"): the headline precision mode (`--precision`; default `f16x3h`: the
split-f16 forward and dX chains of `f16x3` -- 22-bit operands, fp32-grade colours: the reference's arithmetic is fp32 -- with the
weight gradients taken from the hi planes of the saved operands; `f16x3`, all operands 22-bit, is in `by_precision` beside it;
DESIGN.md 2 has the parity evidence of both) timed for `--steps` steps with nothing but the
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
# (this pool's host driver supports dmabuf IPC only: RCCL between rank processes needs it, and it has to be in the environment before
#  anything touches HIP; the image exports it already -- this is for a launcher that scrubs the environment)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

# algorithmic FLOPs per evaluated sample, forward: 2 x weight MACs.  The backward chain (dX) and the
# weight gradients (dW) are one forward-equivalent each: a training step is 3x forward.
F_FINE, F_COARSE = 2 * 629248, 2 * 101632


def net_flops(net):
    """2 x weight MACs of a (depth, width, [skip]) CorseFine_NeRF (model/net_block.py:40-65): forward FLOPs per sample."""
    d, w, skips = net
    macs = 63 * w + (d - 1) * w * w + sum(63 * w for k in skips if 0 < k < d) + (w * w + w) + (w * w + 27 * w)
    return 2 * macs


# SURVEY 8(d) "Bound": minimum HBM traffic of a step if activations never left the CU -- per ray o, d 24 B + gt 12 B + pixel id 8 B in,
# rgb_c + rgb_f 24 B out (+ the composite's per-ray scalars) ~ 100 B, plus both nets' parameters and their gradients once.
ALGO_BYTES_PER_RAY = 100
PEAK_F32_MFMA_TFLOPS = 157.3       # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_F16_MFMA_TFLOPS = 2500.0      # ibid., "Peak BF16/FP16 MFMA ~2.5 PF dense"
PEAK_HBM_GBS = 8000.0              # ibid., "HBM3E peak BW 8.0 TB/s spec" (6.29 TB/s measured float4 copy)
# The two unit costs of NOTES.md §3.6's energy budget, measured on this pool (NOT vendor figures): what a bare
# v_mfma_f32_32x32x16_f16 loop sustains on random operands under the chip's power management (profiles/r05k_mfma_shape.txt,
# r05_mfma_peak.txt: 1.71-1.74 PF/s), and what a gigabyte written by a chain costs at unchanged cycle counts (saving minus
# no-save forward: 1.6 ms for 18.3 GB in f16x3h, 1.6 ms for 18.1 GB in f16; profiles/r05f_ablate_storewin.txt, r05_cfg5_bf16_spread.txt)
BARE_MFMA_TFLOPS_RANDOM = 1720.0
CHAIN_WRITE_MS_PER_GB = 0.085
MFMAS_PER_PRODUCT = {"f16x3h": (3, 3, 1), "f16x3": (3, 3, 3), "f16": (1, 1, 1), "bf16": (1, 1, 1)}      # forward chain, dX chain, weight gradient
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
              "f16x3h": "f16x3h (forward and dX chains as f16x3: 22-bit operands, 3 f16 MFMAs per product; the WEIGHT-gradient operands are the hi planes only = 11 significand bits, 1 MFMA per product, 2-byte saved operands)",
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


def parity_probe(precisions, dev):
    """One small train render + backward (256 rays, cfg-2 nets at random init, the same draws) in each of `precisions` against the fp32
    CPU oracle: max |rgb - oracle|, and of the 40 parameter gradients the worst and the median error relative to the tensor's
    largest oracle gradient -- what each mode's arithmetic costs on the bench's own nets.  (The reference's own fp32 gradients move
    by 2.5e-5 median / 1.7e-3 worst of a tensor's max under a re-ordering of its sums at 2048 rays: NOTES.md §4.)
    -> the first mode's record, with the other modes' records under their names."""
    import torch
    from mc_nerf_amd.model import NeRF_Model, MC_NeRF_Loss
    from mc_nerf_amd import synthetic as S
    from oracle import mcnerf_oracle as O
    cfg = O.RenderCfg(samples=64, scale=2)
    pc = {k: v.requires_grad_(True) for k, v in O.init_params(cfg.coarse, 1).items()}
    pf = {k: v.requires_grad_(True) for k, v in O.init_params(cfg.fine, 2).items()}
    g = torch.Generator().manual_seed(0)
    n = 256
    o = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1) * 3.0
    d = torch.nn.functional.normalize(-o + 0.4 * torch.randn(n, 3, generator=g), dim=-1)
    jit = torch.rand(n, 1, generator=g) * 7.0 / 64
    e = [torch.randn(n, s, generator=g) for s in (64, 64, 128)]
    gt = torch.rand(n, 3, generator=g)
    r = O.render_rays_train(pc, pf, cfg, d, o, 1.0, jit, e[0], e[1], e[2])
    O.rgb_loss(r["rgb_c"], r["rgb_f"], gt).backward()
    recs = {}
    for precision in precisions:
        sp = S.make_sys_param(dev, samples=64, scale=2, batch=256, H=64, W=64, precision=precision)
        m = NeRF_Model(sp).to(dev)
        m.nerf_coarse.load_state_dict({k: v.detach() for k, v in pc.items()})
        m.nerf_fine.load_state_dict({k: v.detach() for k, v in pf.items()})
        rgb_c, rgb_f = m.render_rays_train(d.to(dev), o.to(dev), 0, 1.0, jitter=jit.to(dev), eps_c=e[0].to(dev),
                                           eps_sel=e[1].to(dev), eps_f=e[2].to(dev))
        MC_NeRF_Loss(dict(data_img_h=800, data_img_w=800)).get_rgb_loss([rgb_c, rgb_f, gt.to(dev)]).backward()
        rel = []
        for net, ref in ((m.nerf_coarse, pc), (m.nerf_fine, pf)):
            for k, p_ in net.named_parameters():
                rel.append(float((p_.grad.cpu() - ref[k].grad).abs().max()) / max(float(ref[k].grad.abs().max()), 1e-30))
        rel.sort()
        recs[precision] = {"rgb_c_max_abs_err": float((rgb_c.detach().cpu() - r["rgb_c"].detach()).abs().max()),
                           "rgb_f_max_abs_err": float((rgb_f.detach().cpu() - r["rgb_f"].detach()).abs().max()),
                           "grad_worst_rel_to_tensor_max": rel[-1], "grad_median_rel_to_tensor_max": rel[len(rel) // 2]}
    out = dict(recs[precisions[0]])
    # the yardstick of the gradient columns: what the ACTUAL reference's fp32 gradients do under a re-ordering of its own sums
    # (fixture data of tests/golden/g7_train_s64x2_full2048.npz: worst / median over the same 40 tensors); smoke() gates on multiples of it
    from __graft_entry__ import reorder_noise, PARITY_GATES
    n_worst, n_median = reorder_noise()
    out["reference_reorder_noise"] = {"worst": n_worst, "median": n_median}
    for p_ in precisions:
        recs[p_]["grad_worst_over_noise"] = recs[p_]["grad_worst_rel_to_tensor_max"] / n_worst
        recs[p_]["grad_median_over_noise"] = recs[p_]["grad_median_rel_to_tensor_max"] / n_median
        recs[p_]["gate_multiples"] = list(PARITY_GATES[p_][1:])
    out.update({k: recs[precisions[0]][k] for k in ("grad_worst_over_noise", "grad_median_over_noise", "gate_multiples")})
    out["what"] = ("train render + backward of 256 rays, random-init cfg-2 nets, same draws, vs the fp32 CPU oracle: colours; ALL 40 parameter "
                   "gradients (max error of a tensor / its largest oracle gradient: worst and median tensor), also as multiples of the "
                   "reference's own reorder noise on its at-size fixture")
    for p_ in precisions[1:]:
        out[p_] = recs[p_]
    return out


def compact_line(full, full_name="bench_full.json"):
    """The ONE stdout line: bench.py's contract fields + roofline + cpu_baseline, and every secondary measurement of the full record
    as a scalar (the fp32-grade `f16x3` mode beside the headline, the pinned-occupancy lines, N = 65536 / 7000, render)."""
    sig = lambda v, n=5: (None if v is None else float(f"{v:.{n}g}"))
    r = full["roofline"]
    line = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                 "vs_baseline") if k in full}
    line["value"], line["ms_per_step"] = sig(full["value"], 7), sig(full["ms_per_step"], 6)
    line["dtype"] = full["dtype"].split(" (")[0]
    line["data"] = full["data"]
    c = full["config"]
    line["config"] = {"workload": c["workload"], "precision": c["precision"], "rays_per_step_per_gpu": c["rays_per_step_per_gpu"],
                      "fine_samples_per_ray": sig(c["fine_samples_per_ray"]), "parallelism": c["parallelism"].split(" (")[0]}
    line["roofline"] = {"bound": r["bound"], "kernel": r["kernel"], "ms": sig(r["ms"]), "achieved": sig(r["achieved"]), "peak": r["peak"],
                        "unit": r["unit"], "frac": sig(r["frac"]), "traffic": sig(r["traffic"]),
                        "step_mfma_frac": sig(r["step_mfma_frac"]), "step_traffic_bytes": sig(r.get("step_traffic_bytes")),
                        "algorithmic_bytes_per_step": r.get("algorithmic_bytes_per_step")}
    if "cpu_baseline" in full:
        cb = full["cpu_baseline"]
        line["cpu_baseline"] = {"value": sig(cb["value"]), "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                                "sample": cb["sample"][:120], "render_value": sig(cb.get("render_value"))}
    for p, rec in full.get("by_precision", {}).items():
        line[f"{p}_value"], line[f"{p}_ms_per_step"] = sig(rec["value"]), sig(rec["ms_per_step"])
        if p == "f16x3":
            line["f16x3_step_mfma_frac"] = sig(rec.get("step_mfma_frac"))
    head = full["config"]["precision"]
    for rho, modes in full.get("by_occupancy", {}).items():
        tag = "rho" + rho.replace("0.", "0").replace(".", "")
        for p, rec in modes.items():
            line[f"{tag}_value" if p == head else f"{tag}_{p}_value"] = sig(rec["value"])
    ex = full.get("extra_lines", {})
    for key, name in (("rays_65536", "n65536_value"), ("rays_7000", "n7000_value"), ("coarse_8x256x4", "coarse8x256_value"),
                      ("reference_default_128x5_rays_7000", "ref128x5_n7000_value"), ("render", "render_value")):
        if key in ex:
            line[name] = sig(ex[key]["value"])
    if "parity" in full:        # what the headline mode's arithmetic costs against the fp32 oracle (256 rays), beside f16x3 / f32
        pr = full["parity"]
        gp = {head: [sig(pr["grad_worst_rel_to_tensor_max"], 2), sig(pr["grad_median_rel_to_tensor_max"], 2)]}
        for p in ("f16x3", "f32"):
            if p in pr:
                gp[p] = [sig(pr[p]["grad_worst_rel_to_tensor_max"], 2), sig(pr[p]["grad_median_rel_to_tensor_max"], 2)]
        line["parity"] = {"rgb_max_abs_err": sig(max(pr["rgb_c_max_abs_err"], pr["rgb_f_max_abs_err"]), 2),
                          "gradient_parity": gp, "what": "256 rays vs fp32 CPU oracle; gradients of all 40 tensors: [worst, median] tensor error / tensor max"}
        if "reference_reorder_noise" in pr:
            line["parity"]["reference_reorder_noise"] = [sig(pr["reference_reorder_noise"]["worst"], 2), sig(pr["reference_reorder_noise"]["median"], 2)]
    for k in ("allreduce_ms", "rank_ms_per_step", "params_identical_across_ranks", "asymmetric_grad_steps", "valid", "skipped_optimizer_steps"):
        if k in full:
            v = full[k]["all"] if isinstance(full[k], dict) and "all" in full[k] else full[k]      # (rank_ms_per_step: every rank's own time)
            line[k] = [sig(x, 4) for x in v] if isinstance(v, (list, tuple)) else (sig(v, 4) if isinstance(v, float) else v)
    line["dist"] = full.get("dist")
    line["full_record"] = full_name
    return fit_line(line)


COMPACT_LINE_BUDGET = 1900      # bytes of the stdout line (the driver's record keeps 2 KB of it)
# what goes first when the line is over budget: secondary scalars (every one of them is in the full record), then the long strings
DROP_ORDER = ("ref128x5_n7000_value", "coarse8x256_value", "n7000_value", "render_value", "f32_ms_per_step", "bf16_ms_per_step",
              "f16_ms_per_step", "rho025_f16_value", "rho005_f16_value", "f32_value", "bf16_value", "f16_value", "n65536_value",
              "rank_ms_per_step", "rho025_value", "rho005_value")


def fit_line(line, budget=COMPACT_LINE_BUDGET):
    """Bounds the stdout line: drops the lowest-priority secondary scalars, then shortens the free-text fields, until it fits.
    The contract's fields, `roofline`, `cpu_baseline`, the fp32-grade `f16x3_value` and `dist` are never dropped."""
    size = lambda: len(json.dumps(line))
    for k in DROP_ORDER:
        if size() <= budget:
            return line
        line.pop(k, None)
    if size() > budget and "parity" in line:
        line["parity"].pop("what", None)
    if size() > budget and "cpu_baseline" in line:
        line["cpu_baseline"]["sample"] = line["cpu_baseline"]["sample"][:48]
    if size() > budget:
        line["config"]["workload"] = line["config"]["workload"][:64]
    return line


def parse_net(text):
    """'8x256x4' -> (8, 256, [4]) (depth x width x skip layer; config/config.yaml:76-81)."""
    d, w, k = (int(t) for t in text.lower().split("x"))
    return (d, w, [k])


def calibrate_occupancy(model, step_nograd, target, n_grid):
    """SURVEY 8(d) "occupancy control": shifts the coarse net's sigma-head bias until the selected fraction of the fine grid
    (model/mc_nerf.py:619-632: `weights >= min(thresh, max)`, x `scale`) is `target`.  The fraction falls monotonically with
    the bias below the random-init value (every alpha falls), so a bisection on the shift over no-grad renders of the bench's
    own rays finds it; outside the timed region (it reads the count back).  Returns (shift, measured fraction)."""
    import torch
    bias = model.nerf.nerf_coarse.sigma[2].bias
    base = bias.detach().clone()

    def frac(shift):
        with torch.no_grad():
            bias.copy_(base + shift)
            tot = 0.0
            for i in range(3):
                step_nograd(i)
                tot += float(model.nerf.last_selection[1].item())
        return tot / 3 / n_grid

    lo, hi = -20.0, 0.0
    if frac(hi) <= target:
        return 0.0, frac(0.0)
    for _ in range(14):
        mid = 0.5 * (lo + hi)
        if frac(mid) > target:
            hi = mid
        else:
            lo = mid
    shift = 0.5 * (lo + hi)
    return shift, frac(shift)


def run_precision(precision, args, steps, warmup, rank, world, dev, timer, images_cache, *, rays=None, samples=None, scale=None,
                  coarse=None, occupancy=None):
    """Builds the model in `precision` (cfg-2 shape unless overridden), runs warmup + `steps` timed steps; returns the per-mode
    record.  `occupancy` = target selected fraction of the fine grid (None: random-init weights as they are)."""
    import torch
    import torch.distributed as dist
    from mc_nerf_amd import distributed as D
    from mc_nerf_amd import synthetic as S
    from mc_nerf_amd.model import MC_Model, MC_NeRF_Loss, RAdam
    from mc_nerf_amd.data import DeviceImageSet

    torch.manual_seed(42 + rank)                   # main.py:274-277: seed + rank
    H = W = args.img
    rays = rays or args.rays
    samples, scale = samples or args.samples, scale or args.scale
    coarse = coarse or parse_net(args.coarse)
    sp = S.make_sys_param(dev, samples=samples, scale=scale, batch=rays, H=H, W=W, barf_mask=False,
                          precision=precision, rig=args.rig, coarse=coarse)
    model = MC_Model(sp).to(dev)
    S.init_cameras_near_gt(model, noise=1e-3)
    loss_fn = MC_NeRF_Loss(sp)
    opt = RAdam(model.parameters(), lr=5e-4, weight_decay=4e-4)
    sync = D.FlatGradSync(model, world)
    sync.broadcast_parameters()
    model.nerf.reserve_workspaces(rays)              # the kernels' workspaces are sized here, once, and re-used by every step
    C = model.train_numb
    if "wpts" not in images_cache:
        wpts, pts = S.calibration_points(sp["gt_pose"], sp["intr_mat"][0])
        images_cache["wpts"], images_cache["pts"] = wpts.to(dev), pts.to(dev)
        images_cache["images"] = DeviceImageSet.synthetic(C, H, W, dev, channels=4, seed=7)     # uint8 RGBA resident in HBM (282 MB)
    wpts, pts, images = images_cache["wpts"], images_cache["pts"], images_cache["images"]
    cams = D.shard_cameras(C, 0, rank, world, seed=42)
    counts, ar_events = [], []
    occ = None
    if occupancy is not None and args.sigma_bias_shift is not None:      # (a shift calibrated by an earlier run: no calibration renders in a profiled run)
        with torch.no_grad():
            model.nerf.nerf_coarse.sigma[2].bias.add_(args.sigma_bias_shift)
        occ = {"target": occupancy, "sigma_bias_shift": args.sigma_bias_shift, "selected_fraction_at_calibration": None}
    elif occupancy is not None:
        def nograd_step(i):
            model((images, torch.tensor([cams[i % len(cams)]]), wpts, pts, wpts, pts), 20, "GLOBAL_OPTIM_EPOCH", 0.6)
        shift, got = calibrate_occupancy(model, nograd_step, occupancy, rays * samples * scale)
        if world > 1:                                  # one shift for the job (rank 0's): the parameters stay identical on every rank
            t = torch.tensor([shift], device=dev, dtype=torch.float64)
            dist.broadcast(t, 0)
            with torch.no_grad():
                model.nerf.nerf_coarse.sigma[2].bias.add_(float(t.item()) - shift)
            shift = float(t.item())
        occ = {"target": occupancy, "sigma_bias_shift": shift, "selected_fraction_at_calibration": got}

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
    torch.cuda.synchronize()
    dt_local = time.perf_counter() - t0             # this rank's own completion (before it waits for the others): a straggler shows here
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
    rec = {"precision": precision, "value": rays * world * steps / dt, "unit": "rays/s", "steps": steps,
           "ms_per_step": dt / steps * 1e3, "rays_per_step_per_gpu": rays, "fine_samples_per_ray": k_mean / rays,
           "selected_fraction": k_mean / (rays * samples * scale), "finite": finite,
           "skipped_optimizer_steps": skipped, "valid": bool(finite and skipped == 0),
           "kernel_ms": {f"mlp_{k}<{w}>": v for (k, w), v in sorted(ks.items())}}
    if occ is not None:
        rec["occupancy"] = occ
    if world > 1:
        mine = torch.tensor([dt_local / steps * 1e3, sum(a.elapsed_time(b) for a, b in ar_events) / max(1, len(ar_events))], device=dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank = torch.stack(every).cpu()
        rec["rank_ms_per_step"] = {"min": float(per_rank[:, 0].min()), "max": float(per_rank[:, 0].max()), "all": [round(float(v), 3) for v in per_rank[:, 0]],
                                   "what": "each rank's own completion time of the timed steps / steps, before the closing barrier"}
        rec["rank_allreduce_ms"] = {"min": float(per_rank[:, 1].min()), "max": float(per_rank[:, 1].max())}
        rec["allreduce_ms"] = sum(a.elapsed_time(b) for a, b in ar_events) / max(1, len(ar_events))
        rec["params_identical_across_ranks"] = in_sync
        rec["asymmetric_grad_steps"] = sync.asymmetric_steps()      # ranks disagreeing on which tensors have gradients: must be 0
    # roofline of the fine-net kernels with ALGORITHMIC work per launch over the HIP-event launch time
    mfma_peak = PEAK_F32_MFMA_TFLOPS if precision == "f32" else PEAK_F16_MFMA_TFLOPS
    contract = B16 if precision in ("f16", "bf16", "f16x3h") else (BX3 if precision == "f16x3" else B32)
    per_call = {}
    for k in ("fwd", "bwd", "dw"):
        ms = ks.get((k, 256))
        if ms:
            tf = F_FINE * k_mean / (ms * 1e-3) / 1e12
            gbs = contract[k] * k_mean / (ms * 1e-3) / 1e9
            per_call[f"mlp_{k}<256>"] = {"ms": ms, "mfma_TFLOPs": tf, "mfma_frac": tf / mfma_peak, "contract_bytes_per_launch": contract[k] * k_mean,
                                         "hbm_GBs": gbs, "hbm_frac": gbs / PEAK_HBM_GBS}
    rec["per_call"] = per_call
    rec["step_algorithmic_tflop"] = 3 * (F_FINE * k_mean + net_flops(coarse) * rays * samples) / 1e12
    n_param = sum(p.numel() for p in model.parameters())
    rec["algorithmic_bytes_per_step"] = ALGO_BYTES_PER_RAY * rays + 2 * 4 * n_param
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
             "f16x3": {"mlp_fwd<256>": "mlp_x3_fwd_kernel<256, 1>", "mlp_bwd<256>": "mlp_x3_bwd_kernel<256, false>",
                       "mlp_dw<256>": "dwx3_stream_kernel<256>"},
             "f16x3h": {"mlp_fwd<256>": "mlp_x3_fwd_kernel<256, 2>", "mlp_bwd<256>": "mlp_x3_bwd_kernel<256, true>",
                        "mlp_dw<256>": "dw16_stream_kernel<256, false>"},
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


def design_floor(precision, head, k_mean, rays, samples, coarse):
    """NOTES.md §3.6: the floor THIS design has on this chip -- the chains' executed MFMA work at the bare loop's rate + their PMC-measured
    written bytes at the measured cost per gigabyte + the (HBM-bound) weight-gradient kernels and everything else as measured in this
    run.  None without a PMC recording of the current kernel sources, and for the exact-fp32 mode (other instructions)."""
    m = MFMAS_PER_PRODUCT.get(precision)
    path = os.path.join(ROOT, "profiles", f"pmc_traffic_{precision}.json")
    if m is None or not os.path.isfile(path):
        return None
    rec = json.load(open(path))
    if rec.get("csrc_digest") != csrc_digest():
        return None
    alg = (F_FINE * k_mean + net_flops(coarse) * rays * samples) / 1e12                 # algorithmic TFLOP of one kernel family
    chain_tflop = (m[0] + m[1]) * alg
    written = sum(v.get("write_bytes", 0.0) for k, v in rec["kernels"].items() if "mlp" in k and ("fwd_kernel" in k or "bwd_kernel" in k))
    km = head["kernel_ms"]
    dw_ms = sum(v for k, v in km.items() if k.startswith("mlp_dw"))
    mlp_ms = sum(km.values())
    other_ms = max(0.0, head["ms_per_step"] - mlp_ms)
    chain_ms = chain_tflop / BARE_MFMA_TFLOPS_RANDOM * 1e3
    write_ms = CHAIN_WRITE_MS_PER_GB * written / 1e9
    floor = chain_ms + write_ms + dw_ms + other_ms
    return {"chains_executed_tflop": chain_tflop, "chains_mfma_ms": chain_ms, "chains_written_GB": written / 1e9, "chains_write_ms": write_ms,
            "dw_measured_ms": dw_ms, "other_measured_ms": other_ms, "floor_ms": floor, "measured_ms": head["ms_per_step"],
            "measured_over_floor": head["ms_per_step"] / floor, "floor_over_measured": floor / head["ms_per_step"],
            "what": "builder's energy budget of the step (NOTES.md §3.6), NOT the contract's roofline: executed MFMA work of the chains at "
                    f"{BARE_MFMA_TFLOPS_RANDOM:.0f} TFLOP/s (bare loop, random operands, this pool) + their written bytes at {CHAIN_WRITE_MS_PER_GB} ms/GB "
                    "+ the HBM-bound weight-gradient kernels and the small kernels as measured"}


def pmc_step_traffic(precision):
    """Sum of the PMC-measured HBM bytes of all six MLP kernels of a step from the same committed recording (None when stale)."""
    path = os.path.join(ROOT, "profiles", f"pmc_traffic_{precision}.json")
    if not os.path.isfile(path):
        return None, None
    rec = json.load(open(path))
    if rec.get("csrc_digest") != csrc_digest():
        return None, None
    # (one launch per step each: fwd / bwd / dW of the coarse and of the fine net; everything else moves < 0.1 % of a step's bytes)
    tot = sum(v["hbm_bytes_per_launch"] for k, v in rec.get("kernels", {}).items() if "mlp" in k or "dw" in k)
    return (tot or None), path


def run_rank(args):
    import torch
    import torch.distributed as dist
    from mc_nerf_amd import distributed as D
    rank, world, dev = D.init_distributed()
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the process group has {world} rank(s)")
    if world > 1 and torch.cuda.is_available() and os.environ.get("MCNERF_SHARE_GPU") != "1":
        # one GPU per rank over RCCL, or the scaling numbers mean something else: refuse to measure otherwise (utils/distributed_init.py:28-33
        # binds LOCAL_RANK the same way).  Test rigs that share one GPU say so with MCNERF_SHARE_GPU=1 (+ MCNERF_DIST_BACKEND=gloo).
        local = int(os.environ.get("LOCAL_RANK", rank))
        if torch.cuda.current_device() != local or dev.index != local:
            raise SystemExit(f"bench.py: rank {rank} is on cuda:{torch.cuda.current_device()}, LOCAL_RANK is {local}")
        if dist.get_backend() != "nccl":
            raise SystemExit(f"bench.py: backend {dist.get_backend()!r} with one GPU per rank: the multi-GPU line is measured over RCCL (backend nccl)")
    if args.also is None:          # defaults: every mode beside the headline at N = 1; at N > 1 the headline + the all-22-bit mode only,
        args.also = "f16x3,f16,bf16,f32" if world == 1 else "f16x3"        # so that the line arrives well inside a driver's per-run timeout
    if args.occupancy is None:
        args.occupancy = "0.25,0.05" if world == 1 else ""
    if args.extra is None:
        args.extra = world == 1
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
    head, k_mean = run_precision(args.precision, args, args.steps, args.warmup, rank, world, dev, timer, cache, occupancy=args.rho)
    others = {}
    for p in [q for q in args.also.split(",") if q and q != args.precision]:
        others[p], _ = run_precision(p, args, max(20, args.steps // 5), 3, rank, world, dev, timer, cache)
    slim = lambda r: {k: v for k, v in r.items() if k not in ("per_call",)}
    # ---- SURVEY 8(d) "Synthetic inputs": the same step with the selected fraction of the fine grid pinned (sigma-head bias of the
    # coarse net shifted; K reported with each), in the headline mode and in f16; and the "coarse also 8x256/[4]" variant
    by_occ, extra = {}, {}
    sub_steps = max(20, args.steps // 2)
    default_shape = (args.rays == 32768 and args.samples == 64 and args.scale == 2 and args.img == 800 and args.coarse == "4x128x2" and args.rho is None)
    if args.occupancy:
        for rho in [float(t) for t in args.occupancy.split(",") if t]:
            by_occ[f"{rho:g}"] = {p: slim(run_precision(p, args, sub_steps, 3, rank, world, dev, timer, cache, occupancy=rho)[0])
                                 for p in dict.fromkeys([args.precision, "f16"])}
    if args.extra and default_shape:
        ex = lambda **kw: slim(run_precision(args.precision, args, sub_steps, 3, rank, world, dev, timer, cache, **kw)[0])
        extra["coarse_8x256x4"] = dict(ex(coarse=(8, 256, [4])), what="both nets 8x256/[4] (SURVEY 8: the variant reported beside the default); "
                                       "kernel_ms<256> averages the coarse and the fine call")
        extra["rays_7000"] = dict(ex(rays=7000), what="N = 7000, the reference's `batch` (config/config.yaml:30)")
        extra["rays_65536"] = dict(ex(rays=65536), what="N = 65536, SURVEY 8(d)'s throughput batch (the default line is N = 32768)")
        extra["reference_default_128x5_rays_7000"] = dict(ex(rays=7000, samples=128, scale=5),
                                                          what="the reference's default sampling: 128 coarse x 5 (fine grid 640, random cap at 128 kept per ray), N = 7000")
        if world == 1:
            rargs = argparse.Namespace(**dict(vars(args), steps=3))
            r = run_render(rargs, rank, world, dev)
            extra["render"] = {"value": r["value"], "unit": r["unit"], "ms_per_image": r["ms_per_step"], "images": r["steps"],
                               "what": r["metric"] + ", " + r["config"]["workload"]}
    if rank == 0:
        pc = head["per_call"]
        # the path is MFMA-bound by construction (> 10^6 FLOP per algorithmic byte), so `roofline` prices the dominant
        # kernel -- the longest single launch -- by its ALGORITHMIC FLOPs (2 x weight MACs per evaluated sample; the three MFMAs per
        # product of the split-f16 mode and any recomputation are not counted) over its HIP-event duration against the dense f16
        # MFMA peak (fp32 MFMA peak in the exact-fp32 mode).  What the kernel's own design moves through HBM (its contract bytes:
        # the saved operands of the weight-gradient GEMMs) is in `other_roof`, with the PMC-measured bytes as `traffic`; the bytes a
        # step would move if activations never left the CU are `algorithmic_bytes_per_step`.
        single = pc if args.precision != "f32" else {k: v for k, v in pc.items() if k != "mlp_dw<256>"}    # (f32: mlp_dw is 15 launches)
        dom = max(single, key=lambda k: single[k]["ms"])
        mfma_peak = PEAK_F32_MFMA_TFLOPS if args.precision == "f32" else PEAK_F16_MFMA_TFLOPS
        traffic, tsrc = pmc_traffic(args.precision, dom) if default_shape else (None, None)
        step_traffic, _ = pmc_step_traffic(args.precision) if default_shape else (None, None)
        hbm_roof = {"bound": "hbm", "achieved": pc[dom]["hbm_GBs"], "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": pc[dom]["hbm_frac"],
                    "what": "the kernel's CONTRACT bytes (saved GEMM operands it is designed to move; not algorithmic) / duration",
                    "contract_bytes_per_launch": pc[dom]["contract_bytes_per_launch"],
                    "pmc_over_contract": (traffic / pc[dom]["contract_bytes_per_launch"]) if traffic else None}
        roof = {"bound": "mfma", "kernel": dom, "ms": pc[dom]["ms"], "achieved": pc[dom]["mfma_TFLOPs"], "peak": mfma_peak,
                "unit": "TFLOP/s", "frac": pc[dom]["mfma_frac"], "traffic": traffic, "traffic_unit": "HBM bytes per launch (PMC)",
                "traffic_source": tsrc, "other_roof": hbm_roof,
                "algorithmic_bytes_per_step": head["algorithmic_bytes_per_step"],
                "step_traffic_bytes": step_traffic,
                "traffic_over_algorithmic": (step_traffic / head["algorithmic_bytes_per_step"]) if step_traffic else None,
                "per_call": pc, "step_algorithmic_tflop": head["step_algorithmic_tflop"], "step_mfma_frac": head["step_mfma_frac"],
                "design_floor": design_floor(args.precision, head, k_mean, args.rays, args.samples, parse_net(args.coarse)) if default_shape else None}
        out = {
            "metric": "train rays/sec (coarse+fine, 64+128 samples)", "value": head["value"], "unit": "rays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": head["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPE_TEXT[args.precision], "data": "synthetic",
            "config": {"workload": workload_text(args), "precision": args.precision, "rays_per_step_per_gpu": args.rays,
                       "fine_samples_per_ray": head["fine_samples_per_ray"], "selected_fraction": head["selected_fraction"],
                       "occupancy": head.get("occupancy"),
                       "parallelism": f"dp{world} (cameras sharded, 1 flat all-reduce/step)"},
            "roofline": roof,
            "kernel_ms": head["kernel_ms"],
            "by_precision": {p: {k: v for k, v in r.items() if k != "precision"} for p, r in others.items()},
            "by_occupancy": by_occ,
            "extra_lines": extra,
        }
        for k in ("allreduce_ms", "rank_ms_per_step", "rank_allreduce_ms", "params_identical_across_ranks", "asymmetric_grad_steps", "finite", "skipped_optimizer_steps", "valid"):
            if k in head:
                out[k] = head[k]
        # the process group the numbers were taken on: `backend` "nccl" IS RCCL on ROCm (gloo only on the test rigs that share a GPU)
        out["dist"] = {"world_size": world, "backend": (dist.get_backend() if world > 1 else None),
                       "rccl_ranks": (world if (world > 1 and dist.get_backend() == "nccl") else 0),      # (no process group at world 1: none exercised)
                       "one_gpu_per_rank": os.environ.get("MCNERF_SHARE_GPU") != "1"}
        if world == 1 and not args.no_cpu_baseline:
            # (the default mode beside the mode whose every operand is 22-bit, and the exact-fp32 mode: the same probe in each)
            out["parity"] = parity_probe(list(dict.fromkeys([args.precision, "f16x3", "f32"])), dev)
            out["cpu_baseline"] = cpu_baseline()
        # the whole record goes to a file; stdout gets ONE compact line (< 2 KB) that carries the contract's fields and the secondary
        # lines as scalars -- the driver keeps the stdout line, and round 4's 8 KB line was cut in its record
        for path in dict.fromkeys([args.full_json] + ([os.path.join("gpurun_out", "bench_full.json")] if os.path.isdir("gpurun_out") else [])):
            try:
                with open(path, "w") as fh:
                    json.dump(out, fh)
            except OSError as e:
                print(f"bench.py: could not write {path}: {e}", file=sys.stderr)
        print(json.dumps(compact_line(out, os.path.basename(args.full_json))))
        allrecs = [(args.precision, head)] + list(others.items()) + [(f"occupancy {o}/{p}", r) for o, d in by_occ.items() for p, r in d.items()] \
            + [(k, r) for k, r in extra.items() if "valid" in r]
        bad = [p for p, r in allrecs if not r["valid"]]
        if bad:                 # a mode whose steps were skipped by the overflow guard (or went non-finite) measured nothing
            from mc_nerf_amd import ops
            print(f"bench.py: INVALID measurement in mode(s) {bad}: non-finite parameters or optimiser steps skipped; weight tensors "
                  f"outside their mode's operand range: {ops.range_report() or 'none'}", file=sys.stderr)
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
    ap.add_argument("--coarse", default="4x128x2", help="coarse net depth x width x skip (config/config.yaml:76-78); 8x256x4 = SURVEY 8's second variant")
    ap.add_argument("--occupancy", default=None, help="selected fractions of the fine grid measured beside the random-init line "
                    "(by_occupancy; sigma-head bias shift, SURVEY 8(d)); empty = none; default 0.25,0.05 at --gpus 1, none at --gpus > 1")
    ap.add_argument("--rho", type=float, default=None, help="pin the selected fraction of the HEADLINE run itself (profiling a low-occupancy step: "
                    "scripts/evidence.sh); the default line leaves the random-init weights as they are")
    ap.add_argument("--sigma-bias-shift", type=float, default=None, help="with --rho: apply this shift (from an earlier run's "
                    "`occupancy.sigma_bias_shift`) instead of calibrating")
    ap.add_argument("--no-extra", dest="extra", action="store_false", default=None, help="skip extra_lines (8x256 coarse, N = 7000, 128x5, render)")
    ap.add_argument("--extra", dest="extra", action="store_true", help="extra_lines also at --gpus > 1 (default: only at --gpus 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", default="f16x3h", choices=["f32", "f16x3", "f16x3h", "f16", "bf16"],
                    help="MFMA mode of the MLP kernels: split-f16 f16x3h (the headline: fp32-grade forward and dX chains, weight gradients from "
                         "the hi operand planes), f16x3 (every operand 22-bit) or exact f32 -- the 1e-4 parity modes; single-pass f16 / bf16 -- "
                         "16-bit operand modes with their own stated error")
    ap.add_argument("--also", default=None, help="other precision modes measured in the same run (by_precision); default f16x3,f16,bf16,f32 "
                    "at --gpus 1, f16x3 at --gpus > 1")
    ap.add_argument("--mode", default="train", choices=["train", "render"])
    ap.add_argument("--full-json", default=os.path.join(ROOT, "bench_full.json"), help="where rank 0 writes the whole record (stdout gets the compact line)")
    ap.add_argument("--selftest", action="store_true", help="rendezvous / launcher check only (no kernels; works on CPU)")
    args = ap.parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args, sys.argv[1:]))
    run_rank(args)


if __name__ == "__main__":
    main()
