"""Builds mc_nerf_amd/libmcnerf.so (hipcc, gfx950 only) in-tree.

    python -m mc_nerf_amd.build [--force]

hipcc cross-compiles without a GPU; the .so travels to the GPU box with the repo snapshot.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libmcnerf.so")
SOURCES = ["api.hip", "pack.hip", "mlp_fwd.hip", "mlp_bwd.hip", "mlp_dw.hip", "pack16.hip", "mlp16_fwd.hip", "mlp16_bwd.hip", "mlp16_dw.hip", "mlp_x3_fwd.hip", "mlp_x3_bwd.hip", "mlp_x3_dw.hip", "composite.hip", "select_raygen.hip", "optim.hip", "camera.hip"]
HEADERS = ["mcnerf_common.h", "mcnerf_kernels.h", "mcnerf_16.h", "mcnerf_x3.h", os.path.join("..", "..", "include", "mcnerf.h")]
# the f16x3 chains: a layer body is ~400 MFMAs with its epilogue slices, fully unrolled (beyond hipcc's default pragma-unroll budget);
# their 256-wide instantiations run one wave per SIMD with 512 registers, where hipcc would otherwise put the MFMA
# accumulators in AGPRs (every epilogue read then costs a v_accvgpr_read behind a full MFMA drain)
FILE_FLAGS = {s: ["-mllvm", "-pragma-unroll-threshold=200000", "-mllvm", "-amdgpu-mfma-vgpr-form"] for s in ("mlp_x3_fwd.hip", "mlp_x3_bwd.hip")}
# -fno-slp-vectorize: hipcc's SLP vectoriser packs adjacent scalar fp32 multiplies / adds of the layer epilogues into v_pk_mul_f32 /
# v_pk_add_f32, which cost more beside MFMAs than the scalar pairs they replace (MI355X_MICROARCH.md, issue-cost table); same-box A/B:
# f16x3 saving forward 12.03 -> 11.60 ms (256-wide), 3.17 -> 3.08 (128-wide), 128-wide backward 2.99 -> 2.83 ms, results bit-identical
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-ffp-contract=off", "-fno-slp-vectorize", "-Wall", "-Wno-unused-function"]


def _newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force=False, verbose=True, extra_flags=(), tag=""):
    """tag / extra_flags build a VARIANT (libmcnerf_<tag>.so with extra -D flags) for kernel ablations; the
    product library is the untagged one."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(HERE, "build" + ("_" + tag if tag else ""))
    out = os.path.join(HERE, f"libmcnerf_{tag}.so") if tag else OUT
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS]
    jobs = []
    for s in SOURCES:
        src, obj = os.path.join(CSRC, s), os.path.join(objdir, s.replace(".hip", ".o"))
        if force or _newer(src, obj) or any(_newer(h, obj) for h in hdrs):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [hipcc] + FLAGS + FILE_FLAGS.get(os.path.basename(src), []) + list(extra_flags) + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        return src, r.returncode, r.stdout + r.stderr

    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        for src, rc, log in ex.map(cc, jobs):
            if verbose and log.strip():
                print(log)
            if rc != 0:
                raise RuntimeError(f"hipcc failed on {src}\n{log}")
            if verbose:
                print(f"[mc_nerf_amd.build] compiled {os.path.basename(src)}")
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES]
    if jobs or not os.path.exists(out):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed\n" + r.stdout + r.stderr)
        if verbose:
            print(f"[mc_nerf_amd.build] linked {out}")
    return out


if __name__ == "__main__":
    tag = next((a.split("=", 1)[1] for a in sys.argv if a.startswith("--tag=")), "")
    build(force="--force" in sys.argv, extra_flags=[a for a in sys.argv[1:] if a.startswith("-") and not a.startswith("--")], tag=tag)
