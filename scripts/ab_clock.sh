# usage: bash scripts/ab_clock.sh "<tags>" <precision> <rays> <width> <which>: time + in-kernel clock of each variant
tags="$1"; prec=${2:-f16x3h}; n=${3:-25600}; w=${4:-256}; which=${5:-fwd_nosave}
for t in $tags; do
  if [ "$t" = base ]; then unset MCNERF_LIB; else export MCNERF_LIB=$PWD/mc_nerf_amd/libmcnerf_$t.so; fi
  printf "%-8s " $t; CLOCKPROBE=1 python scripts/time_kernels.py $prec $n $w $which 2>&1 | tail -1
done
