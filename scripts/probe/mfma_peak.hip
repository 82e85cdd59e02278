// Speed-of-light probe for DESIGN.md: what v_mfma_f32_32x32x16_f16 sustains on THIS chip on random operands (the chip lowers its
// clock under matrix load: MI355X_MICROARCH.md, "DVFS give-back"), with the operands in registers (a bare MFMA loop: the practical
// ceiling of any kernel) -- one wave per SIMD as the f16x3 chains run, and two.  Prints TFLOP/s and the in-kernel clock.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/probe/mfma_peak scripts/probe/mfma_peak.hip && scripts/probe/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512) void mfma_loop(const f16x8* __restrict__ in, float* __restrict__ out, int iters, unsigned long long* clk) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    f16x8 a0 = in[tid], a1 = in[tid + 131072], b0 = in[tid + 262144], b1 = in[tid + 393216];
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, c3, 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int e = 0; e < 16; ++e) s += c0[e] + c1[e] + c2[e] + c3[e];
    out[tid] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

int main() {
    const int cus = 256;
    std::vector<_Float16> h((size_t)4 * 131072 * 8);
    srand(1);
    for (auto& v : h) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 0.25f);
    f16x8* din; float* dout; unsigned long long* dclk;
    hipMalloc(&din, h.size() * 2); hipMalloc(&dout, 4 * cus * 512); hipMalloc(&dclk, 16 * cus);
    hipMemcpy(din, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    for (int wps = 1; wps <= 2; ++wps) {
        const int threads = 256 * wps, iters = 400000 / wps;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(mfma_loop, dim3(cus), dim3(threads), 0, 0, din, dout, iters / 8, dclk);      // warm-up
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(mfma_loop, dim3(cus), dim3(threads), 0, 0, din, dout, iters, dclk);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> c(2 * cus);
        hipMemcpy(c.data(), dclk, 16 * cus, hipMemcpyDeviceToHost);
        std::vector<double> mhz;
        for (int b = 0; b < cus; ++b) mhz.push_back((double)c[2 * b] / (double)c[2 * b + 1] * 100.0);
        std::sort(mhz.begin(), mhz.end());
        const double flop = 3.0 * cus * (threads / 64) * (double)iters * 4 * 2.0 * 32 * 32 * 16;
        printf("v_mfma_f32_32x32x16_f16, random operands in registers, %d wave(s) per SIMD: %.0f TFLOP/s over %.2f s, in-kernel clock %.0f MHz (median over workgroups)\n",
               wps, flop / (ms * 1e-3) / 1e12, ms * 1e-3, mhz[cus / 2]);
    }
    return 0;
}
