# usage: bash scripts/ab_variants.sh "<tags>" <precision> <rays> <width> <which>   (tag "" = the product library)
tags="$1"; prec=${2:-f16x3h}; n=${3:-25600}; w=${4:-256}; which=${5:-fwd,bwd}
for rep in 1 2; do for t in $tags; do
  if [ "$t" = base ]; then unset MCNERF_LIB; else export MCNERF_LIB=$PWD/mc_nerf_amd/libmcnerf_$t.so; fi
  printf "%-8s " $t; python scripts/time_kernels.py $prec $n $w $which 2>&1 | tail -1
done; done
