"""Per-kernel means of the SQ counters collected by scripts/pmc_sq.sh:  python scripts/pmc_sq_summary.py gpurun_out/sq_base [filter]"""
import csv, glob, sys, re
from collections import defaultdict
d = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else "mlp|dw"
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*$", "", re.sub(r"^void\s+", "", r["Kernel_Name"]))
        if re.search(flt, k):
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in sorted(acc.items()):
    print(k)
    m = {n: sum(v) / len(v) for n, v in c.items()}
    for n in sorted(m):
        extra = ""
        if n != "SQ_WAVE_CYCLES" and "SQ_WAVE_CYCLES" in m and m["SQ_WAVE_CYCLES"]:
            extra = f"   ({m[n] / m['SQ_WAVE_CYCLES']:.3f} of WAVE_CYCLES)"
        if n != "SQ_BUSY_CYCLES" and "SQ_BUSY_CYCLES" in m and m["SQ_BUSY_CYCLES"]:
            extra += f"   ({m[n] / m['SQ_BUSY_CYCLES']:.3f} of BUSY_CYCLES)"
        print(f"   {n:34s} {m[n]:16.4g}{extra}")
