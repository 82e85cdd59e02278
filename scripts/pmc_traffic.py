"""Turns two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; each `--pmc X --kernel-trace --output-format csv`)
into profiles/<tag>_pmc_traffic.json: mean HBM bytes per launch and kernel.

  python scripts/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01e_pmc_traffic.json "<command profiled>"

Corrections per /opt/skills/guides/MI355X_MICROARCH.md: counter unit is KB; FETCH_SIZE is doubled on gfx950 (a wide
coalesced read is reported at half its size); WRITE_SIZE is taken as is.
"""
import csv, glob, json, os, re, sys
from collections import defaultdict
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def per_kernel(d, counter):
    acc = defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]) * 1024.0)
    return {k: sum(v) / len(v) for k, v in acc.items()}


def short(name):
    name = re.sub(r"^void\s+", "", name)
    return re.sub(r"\(.*$", "", name)


def main():
    fetch_dir, write_dir, out, cmd = sys.argv[1:5]
    fe, wr = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
    kernels = {}
    for k in fe:
        if any(t in k for t in ("at::native", "rocclr", "ncclDevKernel", "rocprim", "Cijk_")):
            continue
        w = wr.get(k, 0.0)
        kernels[short(k)] = {"fetch_bytes_raw": fe[k], "fetch_bytes_corrected": 2 * fe[k], "write_bytes": w,
                             "hbm_bytes_per_launch": 2 * fe[k] + w}
    json.dump({"method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (each with --kernel-trace only), "
                         f"command: {cmd}; counter unit KB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of a "
                         "wide coalesced read), WRITE_SIZE taken as is; mean over dispatches",
               "csrc_digest": __import__("bench").csrc_digest(),
               "kernels": kernels}, open(out, "w"), indent=1)
    for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:12]:
        print(f"{k:40s} {v['hbm_bytes_per_launch'] / 1e9:9.3f} GB/launch (fetch {v['fetch_bytes_corrected'] / 1e9:.3f}, write {v['write_bytes'] / 1e9:.3f})")


if __name__ == "__main__":
    main()
