for w in 256 128; do for lib in old new; do
  if [ $lib = old ]; then export MCNERF_LIB=$PWD/mc_nerf_amd/libmcnerf_old.so; else unset MCNERF_LIB; fi
  n=25600; [ $w = 128 ] && n=32768
  echo "== width $w lib $lib"; python scripts/time_kernels.py f16x3h $n $w fwd,fwd_nosave,bwd 2>&1 | tail -4
done; done
