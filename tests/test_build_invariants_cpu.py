"""Invariants of the COMPILED gfx950 code that the kernels rely on but the compiler does not enforce (checked on the objects
`python -m mc_nerf_amd.build` leaves under mc_nerf_amd/build/; hipcc cross-compiles here, no GPU needed):

  * M0 belongs to the weight ring in the split-f16 chains.  mcnx3_before_mfma_spread (csrc/mcnerf_x3.h) writes the LDS
    destination of a slab's refill into M0 once and the bare `global_load_lds_dwordx4 ... offset:N` pieces issued over the next
    MFMA gaps rely on it surviving compiler-scheduled code in between.  Every M0 write in those kernels must therefore be one of
    the inline-asm forms -- `s_mov_b32 m0, sN` (the refill; the save / set / restore sandwich of the one-off LDS-DMA helpers) --
    and nothing may use M0 implicitly (`s_set_gpr_idx*`, `v_movrel*`, `s_sendmsg`, GWS / `ds_*` addressing through M0).
  * No scratch in the MFMA chains: a spilled fragment array is a silent 10x (it happened twice: NOTES.md §3.1 / 3.2).  The
    kernels that DO spill are listed with their counts so that a change is seen.
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "mc_nerf_amd", "build")
LLVM = "/opt/rocm/lib/llvm/bin"


def _device_elf(obj, tmp_path):
    src = os.path.join(BUILD, obj)
    if not os.path.isfile(src) or not os.path.isfile(os.path.join(LLVM, "llvm-objdump")):
        pytest.skip("no built objects / llvm tools (run python -m mc_nerf_amd.build)")
    local = os.path.join(tmp_path, obj)
    shutil.copy(src, local)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], check=True, capture_output=True)
    elf = [f for f in os.listdir(tmp_path) if f.startswith(obj) and "gfx950" in f]
    assert elf, "no gfx950 code object in " + obj
    return os.path.join(tmp_path, elf[0])


@pytest.mark.parametrize("obj", ["mlp_x3_fwd.o", "mlp_x3_bwd.o"])
def test_m0_is_written_only_by_the_ring_asm(obj, tmp_path):
    elf = _device_elf(obj, str(tmp_path))
    dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", elf], check=True, capture_output=True, text=True).stdout
    assert dis.count("v_mfma_f32_32x32x16_f16") > 1000
    bad = []
    for line in dis.splitlines():
        ins = line.split("//")[0].strip()
        if re.search(r"\b(s_set_gpr_idx|v_movrel|s_sendmsg|ds_gws|s_movrel)", ins):
            bad.append(ins)
        elif re.search(r"\bm0\b", ins):
            if not (re.fullmatch(r"s_mov_b32 m0, s\d+", ins) or re.fullmatch(r"s_mov_b32 s\d+, m0", ins)):
                bad.append(ins)
    assert not bad, f"{obj}: M0 is touched outside the ring's inline asm (mcnerf_x3.h): {bad[:5]}"
    assert dis.count("global_load_lds_dwordx4") > 100


def _kernel_meta(elf):
    notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", elf], check=True, capture_output=True, text=True).stdout
    out = {}
    for blk in notes.split("- .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        out[name] = {k: int(re.search(rf"\.{k}:\s+(\d+)", blk).group(1)) for k in ("private_segment_fixed_size", "vgpr_spill_count", "vgpr_count")}
    return out


# kernels allowed to use scratch, with what they use today (bytes of scratch, spilled VGPRs).  Both x3 chains and the 16-bit
# chains / dW kernels at every width the bench uses are NOT in this list.
KNOWN_SCRATCH = {
    # 29 VGPRs: segment descriptors / source pointers of the one-launch job, stored in the kernel prologue and reloaded at each
    # segment's set-up and flush -- test_scratch_stays_out_of_the_mfma_loops shows none of it inside a tile loop
    "mlp_x3_dw.o": {"_Z18dwx3_stream_kernelILi256EEv7DwX3JobPKiiPKj": 128},
    # the hi-plane saving forward (f16x3h): 5 VGPRs (two lane pointers and a dword), 12 scratch operations per pass of 2 640 MFMAs
    "mlp_x3_fwd.o": {"_Z17mlp_x3_fwd_kernelILi256ELi2EEv12Mcn16FwdArgs": 32},
    "mlp16_bwd.o": {"_Z16mlp16_bwd_kernelILi256ELb1EEv12Mcn16BwdArgs": 160, "_Z16mlp16_bwd_kernelILi256ELb0EEv12Mcn16BwdArgs": 160},
    "mlp16_fwd.o": {"_Z16mlp16_fwd_kernelILi128ELb1ELb1EEv12Mcn16FwdArgs": 96},
    "mlp_fwd.o": {"_Z14mlp_fwd_kernelILi256ELb1EEv13McnMlpFwdArgs": 256, "_Z14mlp_fwd_kernelILi256ELb0EEv13McnMlpFwdArgs": 96},
}


@pytest.mark.parametrize("obj", ["mlp_x3_fwd.o", "mlp_x3_bwd.o", "mlp_x3_dw.o", "mlp16_fwd.o", "mlp16_bwd.o", "mlp16_dw.o", "mlp_fwd.o", "mlp_bwd.o", "mlp_dw.o"])
def test_no_unexpected_scratch_in_the_mlp_kernels(obj, tmp_path):
    meta = _kernel_meta(_device_elf(obj, str(tmp_path)))
    assert meta
    allowed = KNOWN_SCRATCH.get(obj, {})
    for name, m in meta.items():
        lim = allowed.get(name, 0)
        # (the x3 no-save forward keeps 2 "spilled" VGPRs in AGPRs: spill count without scratch bytes is not scratch traffic)
        assert m["private_segment_fixed_size"] <= lim, f"{obj}: {name} uses {m['private_segment_fixed_size']} B of scratch ({m['vgpr_spill_count']} VGPRs spilled), allowed {lim}"


def _loops_with_mfma(body_lines):
    """[(first, last)] line ranges of backward-branch loops of one kernel's disassembly that contain an MFMA."""
    addr = {}
    for i, l in enumerate(body_lines):
        m = re.search(r"//\s*([0-9A-Fa-f]{12}):", l)
        if m:
            addr[int(m.group(1), 16)] = i
    base = min(addr) if addr else 0
    loops = []
    for i, l in enumerate(body_lines):
        if "s_cbranch" in l or "s_branch" in l:
            m = re.search(r"<[^>]+\+0x([0-9a-fA-F]+)>", l)
            if not m:
                continue
            tgt = addr.get(base + int(m.group(1), 16))
            if tgt is not None and tgt < i and any("v_mfma" in x for x in body_lines[tgt:i]):
                loops.append((tgt, i))
    return loops


@pytest.mark.parametrize("obj", ["mlp_x3_dw.o", "mlp16_bwd.o", "mlp16_fwd.o"])
def test_scratch_stays_out_of_the_mfma_loops(obj, tmp_path):
    """The kernels of KNOWN_SCRATCH that the bench runs: whatever they spill must live outside every loop that issues MFMAs (the
    per-tile streaming loops of the weight-gradient kernel, the layer loops of the chains).  A chain's pass loop contains
    everything, so there the number of scratch operations per pass (thousands of MFMAs) is bounded as well."""
    elf = _device_elf(obj, str(tmp_path))
    dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", elf], check=True, capture_output=True, text=True).stdout
    parts = re.split(r"\n[0-9a-f]+ <(_Z\w+)>:\n", dis)
    seen = 0
    for name, body in zip(parts[1::2], parts[2::2]):
        if name not in KNOWN_SCRATCH[obj]:
            continue
        lines = body.split("\n")
        loops = _loops_with_mfma(lines)
        assert loops, name
        in_loop = [i for i, l in enumerate(lines) if "scratch_" in l and any(a <= i <= b for a, b in loops)]
        seen += 1
        # innermost MFMA loops: the tile loops of the weight-gradient kernel, the layer loops of the chains
        inner = [(a, b) for a, b in loops if not any((c > a or d < b) and a <= c and d <= b for c, d in loops if (c, d) != (a, b))]
        hot = [i for i, l in enumerate(lines) if "scratch_" in l and any(a <= i <= b for a, b in inner)]
        assert not hot, f"{name}: {len(hot)} scratch operations inside an innermost MFMA loop"
        if obj != "mlp_x3_dw.o":      # chains: the pass loop contains everything; <= 20 scratch operations per pass (of >= 600 MFMAs)
            assert len(in_loop) <= 20, f"{name}: {len(in_loop)} scratch operations per pass"
    assert seen == len(KNOWN_SCRATCH[obj])


# ---- product-build hygiene (review item 7): the library is ONE build of ONE source text -- no timing-only / ablation / experiment
#      switches live in the product sources or in the build flags, and nothing but the declared C ABI is exported
ALLOWED_PP_NAMES = {"__HIPCC__"}


def test_no_ablation_or_experiment_switches_in_the_product_sources():
    """Every preprocessor conditional of mc_nerf_amd/csrc tests a name of ALLOWED_PP_NAMES and nothing else (round 5 carried 32
    `ABL*` / `MCN*_EXP_*` / `*_STAMPS` knobs in the kernel sources; their records are in the history and under profiles/)."""
    csrc = os.path.join(ROOT, "mc_nerf_amd", "csrc")
    bad = []
    for fn in sorted(f for f in os.listdir(csrc) if f.endswith((".hip", ".h"))):
        for n, line in enumerate(open(os.path.join(csrc, fn)).read().split("\n"), 1):
            m = re.match(r"\s*#\s*(ifdef|ifndef|if|elif)\b(.*)", line)
            if not m:
                continue
            names = set(re.findall(r"[A-Za-z_]\w*", re.sub(r"//.*", "", m.group(2)))) - {"defined"}
            if not names <= ALLOWED_PP_NAMES:
                bad.append(f"{fn}:{n}: {line.strip()}")
    assert not bad, "preprocessor switches in the product sources:\n" + "\n".join(bad)


def test_build_flags_define_nothing():
    from mc_nerf_amd import build
    flags = list(build.FLAGS) + [f for fl in build.FILE_FLAGS.values() for f in fl]
    assert not [f for f in flags if f.startswith("-D") or f.startswith("-U")], flags


def test_library_exports_only_the_declared_abi():
    """`nm -D` of the built library: every exported mcnerf_* symbol is declared in include/mcnerf.h (no mcnerf_debug_* / stamp
    read-back entry points), and no variant library or variant object directory ships beside it."""
    lib = os.path.join(ROOT, "mc_nerf_amd", "libmcnerf.so")
    if not os.path.isfile(lib) or shutil.which("nm") is None:
        pytest.skip("no built library / nm")
    out = subprocess.run(["nm", "-D", "--defined-only", lib], check=True, capture_output=True, text=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if l.strip()}
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "mcnerf.h")).read(), flags=re.S)
    declared = set(re.findall(r"\b(mcnerf_[a-z_0-9]+)\s*\(", hdr))
    extra = {s for s in exported if s.startswith("mcnerf_")} - declared
    assert not extra, f"exported but not declared: {sorted(extra)}"
    assert not [s for s in exported if "debug" in s.lower() or "stamp" in s.lower()]
    strings = subprocess.run(["strings", "-n", "6", lib], capture_output=True, text=True).stdout if shutil.which("strings") else ""
    assert "mcnerf_debug_stamps" not in strings and "g_mcnx3_fstamps" not in strings
    pkg = os.path.join(ROOT, "mc_nerf_amd")
    tracked = subprocess.run(["git", "ls-files", pkg], capture_output=True, text=True, cwd=ROOT).stdout
    assert "libmcnerf_" not in tracked and "build_" not in tracked
