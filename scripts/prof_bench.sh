#!/bin/bash
# usage (on the GPU box, from the repo root): scripts/prof_bench.sh <tag> [bench args...]
# rocprofv3 --kernel-trace --stats of the bench command; prints the per-step kernel table (steps = timed + warm-up + the
# KernelTimer's min(5, steps) untimed extra steps; profile a --rho run with --sigma-bias-shift, or its calibration renders are in the table)
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py "$@" > $OUT/bench.log 2>&1
tail -c 400 $OUT/bench.log; echo
CSV=$(find $OUT -name "*_kernel_stats.csv" | head -1)
STEPS=$(python3 -c "import sys; a=sys.argv[1:]; s=int(a[a.index('--steps')+1]) if '--steps' in a else 50; w=int(a[a.index('--warmup')+1]) if '--warmup' in a else 5; print(s + w + min(5, s))" "$@")
python3 scripts/prof_summary.py "$CSV" $STEPS 32
cp "$CSV" $OUT/kernel_stats.csv
