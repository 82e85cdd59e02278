mkdir -p gpurun_out/r06a
timeout 900 python -m pytest tests/test_mlpx3_gpu.py -x -q -m gpu 2>&1 | tail -6 | tee gpurun_out/r06a/test_mlpx3.log
for rep in 1 2; do for t in r05 base NOEPI NOLDS NODMA; do
  if [ "$t" = base ]; then unset MCNERF_LIB; else export MCNERF_LIB=$PWD/mc_nerf_amd/libmcnerf_$t.so; fi
  printf "%-6s " $t; python scripts/time_kernels.py f16x3h 25600 256 fwd,fwd_nosave 2>&1 | tail -1
done; done 2>&1 | tee gpurun_out/r06a/ab_fwd_variants.txt
for t in r05 base NOEPI NOLDS NODMA; do
  if [ "$t" = base ]; then unset MCNERF_LIB; else export MCNERF_LIB=$PWD/mc_nerf_amd/libmcnerf_$t.so; fi
  printf "%-6s " $t; CLOCKPROBE=1 python scripts/time_kernels.py f16x3h 25600 256 fwd,fwd_nosave 2>&1 | tail -1
done 2>&1 | tee gpurun_out/r06a/ab_fwd_variants_clock.txt
