#!/bin/bash
# usage (GPU box, repo root): scripts/evidence.sh <tag>
# One snapshot of the measurement evidence at the current sources: the default bench line, rocprofv3 kernel stats of the same
# command (headline mode f16x3h, the all-22-bit f16x3 and f16), HBM traffic (separate PMC passes) and SQ counters of the fine-net kernels.
TAG=${1:-r04}
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
# (stdout = the compact line the driver keeps; --full-json = the whole record, committed as profiles/<tag>_bench_default.json)
python3 bench.py --steps 50 --warmup 5 --full-json gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err | tail -1 > gpurun_out/${TAG}_bench_default_line.json
for P in f16x3h f16x3 f16; do
  scripts/prof_bench.sh ${TAG}_$P --steps 10 --warmup 3 --also= --occupancy= --no-extra --no-cpu-baseline --precision $P > gpurun_out/${TAG}_kernel_table_$P.txt 2>&1
  cp gpurun_out/prof_${TAG}_$P/kernel_stats.csv gpurun_out/${TAG}_kernel_stats_$P.csv
  scripts/pmc_bench.sh $TAG $P > gpurun_out/${TAG}_pmc_table_$P.txt 2>&1
done
# the step at a pinned selected fraction of 0.05 (SURVEY 8(d)'s low-occupancy regime): per-kernel table
SHIFT=$(python3 -c "import json; print(json.load(open('gpurun_out/${TAG}_bench_default.json'))['by_occupancy']['0.05']['f16x3h']['occupancy']['sigma_bias_shift'])")
for P in f16x3h f16x3 f16; do
  scripts/prof_bench.sh ${TAG}_rho005_$P --steps 10 --warmup 3 --rho 0.05 --sigma-bias-shift $SHIFT --also= --occupancy= --no-extra --no-cpu-baseline --precision $P > gpurun_out/${TAG}_kernel_table_rho005_$P.txt 2>&1
  cp gpurun_out/prof_${TAG}_rho005_$P/kernel_stats.csv gpurun_out/${TAG}_kernel_stats_rho005_$P.csv
done
mkdir -p gpurun_out/sq_$TAG
for P in f16x3h f16x3; do
  rm -rf gpurun_out/sq_$TAG; mkdir -p gpurun_out/sq_$TAG
  scripts/pmc_sq.sh $TAG $P 25600 256 > /dev/null 2>&1
  python3 scripts/pmc_sq_summary.py gpurun_out/sq_$TAG > gpurun_out/${TAG}_sq_counters_$P.txt 2>&1
done
ls -la gpurun_out | tail -20
# ---- the other workload shapes of BASELINE.json / NOTES.md §5 (headline mode unless said otherwise), one JSON line each
B="python3 bench.py --also= --occupancy= --no-extra --no-cpu-baseline --steps 20 --warmup 3"
for RIG in array halfball room; do $B --rig $RIG 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_rig_$RIG.json; done
# (N = 7000, the reference's 128 x 5 shape, the 8x256 coarse variant and render mode are `extra_lines` of the default line)
$B --rig room --img 1600 --samples 64 --scale 4 --precision bf16 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_cfg5_bf16.json
$B --rig room --img 1600 --samples 64 --scale 4 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_cfg5_f16x3h.json
$B --rig room --img 1600 --samples 64 --scale 4 --precision f16x3 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_cfg5_f16x3.json
python3 bench.py --mode render --steps 3 --precision f16 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_render_f16.json
scripts/probe/mfma_peak > gpurun_out/${TAG}_mfma_peak.txt 2>&1
scripts/probe/store_cost > gpurun_out/${TAG}_store_cost.txt 2>&1
python3 scripts/sparsity_probe.py 2>/dev/null > gpurun_out/${TAG}_sparsity.txt
ls gpurun_out | grep ${TAG}_
