#!/bin/bash
# Builds the dW ablation variants from the product source (sed-generated copies under gpurun_out/) and runs them.
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/ablate; mkdir -p $O
SRC=$R/mc_nerf_amd/csrc/mlp_dw.hip
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -ffp-contract=off -I$R/mc_nerf_amd/csrc"
cp $SRC $O/dw0.hip
# 1: no loads inside the loop (keeps the prologue loads, waits become trivially satisfied)
sed 's|if ((i \* PP) / NP == pp) piece(i, base + 2 \* RS, b_fill);|;|' $SRC > $O/dw1.hip
# 2: no MFMA: fold the operands into one accumulator element with VALU
sed 's|acc\[t\]\[kt\] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b_c\[kt\], acc\[t\]\[kt\], 0, 0, 0);|acc[t][kt][0] += a1 * b_c[kt];|' $SRC > $O/dw2.hip
# 3: no slab loads and no LDS reads (operands synthesised in registers): the bare MFMA + barrier structure
sed -e 's|if ((i \* PP) / NP == pp) piece(i, base + 2 \* RS, b_fill);|;|' \
    -e 's|a_n = \*reinterpret_cast<const AV\*>(sY + rown \* N + nbase + VN \* r);|a_n = AV((float)(rown + lane));|' \
    -e 's|b_n\[kt\] = sX\[rown \* K + kbase + 32 \* kt + r\];|b_n[kt] = (float)(rown + kt);|' $SRC > $O/dw3.hip
# 4: full kernel + in-kernel clock stamps (s_memtime cycles vs s_memrealtime 100 MHz) from wave 0 of every workgroup
sed -e 's|^template <int N, int K>$|__device__ long long g_dbg[2048];\ntemplate <int N, int K>|' \
    -e 's|    const int tid = threadIdx.x, lane = tid \& 63, wave = tid >> 6;|    const int tid = threadIdx.x, lane = tid \& 63, wave = tid >> 6;\n    const long long t0 = clock64(), w0 = wall_clock64();|' \
    -e 's|    // accumulators -> global (float atomics; one register = two 128-byte row segments)|    if (tid == 0) { g_dbg[2 * blockIdx.x] = clock64() - t0; g_dbg[2 * blockIdx.x + 1] = wall_clock64() - w0; }|' $SRC > $O/dw4.hip
for v in 0 1 2 3 4; do
  hipcc $FLAGS -DDW_ABLATE=$v -DDW_SOURCE="\"$O/dw$v.hip\"" $R/scripts/bench_dw.hip -o $O/bench_dw$v
  $O/bench_dw$v ${1:-2097152} | tail -n +$([ $v = 0 ] && echo 1 || echo 2)
done
