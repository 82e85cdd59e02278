"""Prints a rocprofv3 --kernel-trace --stats kernel_stats.csv as a per-step table."""
import csv, glob, sys
f = sys.argv[1] if len(sys.argv) > 1 else sorted(glob.glob("gpurun_out/prof*/**/*_kernel_stats.csv", recursive=True))[-1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"{f}: total {tot/1e6/steps:.2f} ms/step over {steps:g} steps")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 14]:
    print(f"{r['Name'][:60]:60s} calls/step={float(r['Calls'])/steps:7.1f} ms/step={float(r['TotalDurationNs'])/1e6/steps:8.2f} "
          f"avg_us={float(r['AverageNs'])/1e3:9.1f} pct={float(r['Percentage']):5.2f}")
