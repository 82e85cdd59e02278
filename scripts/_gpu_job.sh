mkdir -p gpurun_out/r06c
timeout 900 python -m pytest tests/test_mlpx3_gpu.py tests/test_model_gpu.py -x -q -m gpu 2>&1 | tail -4 | tee gpurun_out/r06c/tests.log
for rep in 1 2 3; do for t in r05 base; do
  if [ "$t" = base ]; then unset MCNERF_LIB; else export MCNERF_LIB=$PWD/mc_nerf_amd/libmcnerf_$t.so; fi
  printf "%-5s " $t; python scripts/time_kernels.py f16x3h 25600 256 fwd,fwd_nosave,bwd 2>&1 | tail -1
  printf "%-5s " $t; python scripts/time_kernels.py f16x3h 65536 128 fwd,bwd 2>&1 | tail -1
done; done 2>&1 | tee gpurun_out/r06c/ab_mix.txt
