// Which MFMA shape does the power limit favour?  Bare loops on random operands in registers, one wave per SIMD, every CU:
// v_mfma_f32_32x32x16_f16 (what the chains use) against v_mfma_f32_16x16x32_f16 (same FLOPs per cycle: 16 cycles each).
// MI355X_MICROARCH.md "DVFS give-back" (7) reports 1.15 x for the bf16 forms; this is the f16 check on this pool's chips.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/probe/mfma_shape scripts/probe/mfma_shape.hip && scripts/probe/mfma_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(256) void mfma_loop(const f16x8* __restrict__ in, float* __restrict__ out, int iters, unsigned long long* clk) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    f16x8 a0 = in[tid], a1 = in[tid + 131072], b0 = in[tid + 262144], b1 = in[tid + 393216];
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    if (SHAPE == 32) {
        f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, c3, 0, 0, 0);
        }
        for (int e = 0; e < 16; ++e) s += c0[e] + c1[e] + c2[e] + c3[e];
    } else {
        // inline asm: hipcc's own allocation of eight 16x16 accumulators (AGPR shuffles, s_nops) halves the issue rate
        f32x4 c0 = {}, c1 = {}, c2 = {}, c3 = {}, c4 = {}, c5 = {}, c6 = {}, c7 = {};
        for (int i = 0; i < iters; ++i) {          // 8 x 16x16x32 = the FLOPs of 4 x 32x32x16
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %8, %10, %0\n\t"
                         "v_mfma_f32_16x16x32_f16 %1, %9, %10, %1\n\t"
                         "v_mfma_f32_16x16x32_f16 %2, %8, %11, %2\n\t"
                         "v_mfma_f32_16x16x32_f16 %3, %9, %11, %3\n\t"
                         "v_mfma_f32_16x16x32_f16 %4, %8, %10, %4\n\t"
                         "v_mfma_f32_16x16x32_f16 %5, %9, %10, %5\n\t"
                         "v_mfma_f32_16x16x32_f16 %6, %8, %11, %6\n\t"
                         "v_mfma_f32_16x16x32_f16 %7, %9, %11, %7"
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
                         : "v"(a0), "v"(a1), "v"(b0), "v"(b1));
        }
        for (int e = 0; e < 4; ++e) s += c0[e] + c1[e] + c2[e] + c3[e] + c4[e] + c5[e] + c6[e] + c7[e];
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    out[tid] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int SHAPE>
static void run(const f16x8* din, float* dout, unsigned long long* dclk, const char* data) {
    const int cus = 256, threads = 256, iters = 300000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_loop<SHAPE>, dim3(cus), dim3(threads), 0, 0, din, dout, iters, dclk);      // warm-up (thermal / DVFS settle)
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(mfma_loop<SHAPE>, dim3(cus), dim3(threads), 0, 0, din, dout, iters, dclk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> c(2 * cus);
    hipMemcpy(c.data(), dclk, 16 * cus, hipMemcpyDeviceToHost);
    std::vector<double> mhz, cyc;
    for (int b = 0; b < cus; ++b) { mhz.push_back((double)c[2 * b] / (double)c[2 * b + 1] * 100.0); cyc.push_back((double)c[2 * b]); }
    std::sort(mhz.begin(), mhz.end()); std::sort(cyc.begin(), cyc.end());
    const double flop = 3.0 * cus * (threads / 64) * (double)iters * 4 * 2.0 * 32 * 32 * 16;
    printf("%s operands, %s: %.0f TFLOP/s, in-kernel clock %.0f MHz, %.1f cycles per 32x32x16-equivalent\n", data,
           SHAPE == 32 ? "v_mfma_f32_32x32x16_f16" : "v_mfma_f32_16x16x32_f16", flop / (ms * 1e-3) / 1e12, mhz[cus / 2], cyc[cus / 2] / (4.0 * iters));
}

int main() {
    const int cus = 256;
    std::vector<_Float16> h((size_t)4 * 131072 * 8);
    f16x8* din; float* dout; unsigned long long* dclk;
    hipMalloc(&din, h.size() * 2); hipMalloc(&dout, 4 * cus * 512); hipMalloc(&dclk, 16 * cus);
    for (int pass = 0; pass < 2; ++pass) {
        srand(1);
        for (auto& v : h) v = pass == 0 ? (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 0.25f) : (_Float16)0.f;
        hipMemcpy(din, h.data(), h.size() * 2, hipMemcpyHostToDevice);
        for (int rep = 0; rep < 2; ++rep) {
            run<32>(din, dout, dclk, pass == 0 ? "random" : "zero");
            run<16>(din, dout, dclk, pass == 0 ? "random" : "zero");
        }
    }
    return 0;
}
