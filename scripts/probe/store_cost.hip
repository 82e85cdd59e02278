// What does a 1 KiB fragment store cost a wave that is alone on its SIMD and otherwise issues MFMAs back to back (the saving f16x3
// chains)?  A "tile" = 48 v_mfma_f32_32x32x16_f16 (1536 matrix-pipe cycles) + 4 global_store_dwordx4 nt of 64 lanes x 16 B.
// Prints shader cycles per tile for: no stores; the four stores back to back; one every 12 MFMAs; stores from one wave of the
// workgroup only (no same-CU contention); the same bytes as 8 dwordx2 stores; default-policy stores.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/probe/store_cost scripts/probe/store_cost.hip && scripts/probe/store_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void tile_loop(const f16x8* __restrict__ in, char* __restrict__ ws, size_t ws_mask, float* __restrict__ out,
                                                 int iters, unsigned long long* clk) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f16x8 a0 = in[tid], b0 = in[tid + 131072];
    f32x16 c0 = {}, c1 = {};
    u32x4 v = {(unsigned)tid, 1u, 2u, 3u};
    const size_t gw = (size_t)blockIdx.x * 4 + wave;                   // global wave id
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        char* p = ws + ((((size_t)i * 1024 + gw) * 4096) & ws_mask) + lane * 16;
#pragma unroll
        for (int m = 0; m < 48; ++m) {
            const bool st_here = MODE == 1 ? (m >= 24 && m < 28) : (MODE != 0 && (m % 12) == 6);
            if (st_here) {
                const int k = MODE == 1 ? m - 24 : m / 12;
                if (MODE == 3) { if (wave == 0) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p + k * 1024)); }
                else if (MODE == 4) {
                    __builtin_nontemporal_store(u32x2{v[0], v[1]}, reinterpret_cast<u32x2*>(ws + ((((size_t)i * 1024 + gw) * 4096) & ws_mask) + k * 1024 + lane * 8));
                    __builtin_nontemporal_store(u32x2{v[2], v[3]}, reinterpret_cast<u32x2*>(ws + ((((size_t)i * 1024 + gw) * 4096) & ws_mask) + k * 1024 + 512 + lane * 8));
                } else if (MODE == 5) *reinterpret_cast<u32x4*>(p + k * 1024) = v;
                else __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p + k * 1024));
            }
            __builtin_amdgcn_sched_barrier(0);
            if (m & 1) c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, c1, 0, 0, 0);
            else c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, c0, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        v[1] += 1u;
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int e = 0; e < 16; ++e) s += c0[e] + c1[e];
    out[tid] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

int main() {
    const int cus = 256, iters = 9000;
    std::vector<_Float16> h((size_t)2 * 131072 * 8);
    srand(1);
    for (auto& x : h) x = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 0.25f);
    f16x8* din; float* dout; unsigned long long* dclk; char* ws;
    const size_t ws_bytes = (size_t)4 << 30;
    hipMalloc(&din, h.size() * 2); hipMalloc(&dout, 4 * cus * 256); hipMalloc(&dclk, 16 * cus);
    if (hipMalloc(&ws, ws_bytes) != hipSuccess) { printf("no workspace\n"); return 1; }
    hipMemcpy(din, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    const char* names[6] = {"no stores", "4 stores back to back", "1 store every 12 MFMAs", "spread, wave 0 of the workgroup only", "spread, as 8 dwordx2", "spread, default cache policy"};
    void (*kerns[6])(const f16x8*, char*, size_t, float*, int, unsigned long long*) = {tile_loop<0>, tile_loop<1>, tile_loop<2>, tile_loop<3>, tile_loop<4>, tile_loop<5>};
    for (int rep = 0; rep < 2; ++rep)
    for (int mode = 0; mode < 6; ++mode) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(kerns[mode], dim3(cus), dim3(256), 0, 0, din, ws, ws_bytes - 1, dout, iters / 8, dclk);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(kerns[mode], dim3(cus), dim3(256), 0, 0, din, ws, ws_bytes - 1, dout, iters, dclk);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> c(2 * cus);
        hipMemcpy(c.data(), dclk, 16 * cus, hipMemcpyDeviceToHost);
        std::vector<double> cyc, mhz;
        for (int b = 0; b < cus; ++b) { cyc.push_back((double)c[2 * b] / iters); mhz.push_back((double)c[2 * b] / (double)c[2 * b + 1] * 100.0); }
        std::sort(cyc.begin(), cyc.end()); std::sort(mhz.begin(), mhz.end());
        const double gb = mode == 0 ? 0.0 : (double)iters * 4096.0 * (mode == 3 ? 256 : 1024) / 1e9;
        printf("%-40s %7.0f cycles per tile (1536 of MFMA), %.2f ms, clock %.0f MHz, %.1f GB written = %.2f TB/s\n", names[mode], cyc[cus / 2], ms, mhz[cus / 2],
               gb, gb / ms);
    }
    return 0;
}
