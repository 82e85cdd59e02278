#!/bin/bash
# usage (GPU box, repo root): scripts/pmc_bench.sh <tag> <precision>
# separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the bench command (kernel-trace only) -> profiles/<tag>_pmc_traffic_<precision>.json
TAG=$1; PREC=$2
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
CMD="python3 bench.py --steps 4 --warmup 2 --also= --occupancy= --no-extra --no-cpu-baseline --precision $PREC"
for c in FETCH_SIZE WRITE_SIZE; do
  mkdir -p gpurun_out/pmc_${TAG}_${PREC}_$c
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_${PREC}_$c -- $CMD > gpurun_out/pmc_${TAG}_${PREC}_$c/log.txt 2>&1
done
python3 scripts/pmc_traffic.py gpurun_out/pmc_${TAG}_${PREC}_FETCH_SIZE gpurun_out/pmc_${TAG}_${PREC}_WRITE_SIZE gpurun_out/pmc_traffic_${PREC}.json "$CMD"
