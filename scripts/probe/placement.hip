// Which workgroups share a CU, and which hardware wave slots do their waves get?  (2 WGs of 256 threads and 77 KB LDS
// per CU, like the split-f16 MLP kernels.)   hipcc --offload-arch=gfx950 -O2 -o placement placement.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
#include <algorithm>
struct Rec { unsigned hwid, xcc; unsigned long long t0, t1; };
__global__ void probe(Rec* out, int spin) {
    extern __shared__ float lds[];
    const unsigned long long t0 = __builtin_readcyclecounter();
    float acc = threadIdx.x;
    for (int i = 0; i < spin; ++i) { acc = acc * 1.0001f + lds[(threadIdx.x + i) & 1023]; }
    lds[threadIdx.x] = acc;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        Rec r;
        r.hwid = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));       // HW_ID, 32 bits
        r.xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));        // XCC_ID
        r.t0 = t0; r.t1 = __builtin_readcyclecounter();
        out[blockIdx.x * 4 + (threadIdx.x >> 6)] = r;
    }
}
int main() {
    const int grid = 1024, spin = 20000;
    Rec* d; hipMalloc(&d, sizeof(Rec) * grid * 4);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 77312);
    hipLaunchKernelGGL(probe, dim3(grid), dim3(256), 77312, 0, d, spin);
    hipDeviceSynchronize();
    std::vector<Rec> h(grid * 4);
    hipMemcpy(h.data(), d, sizeof(Rec) * grid * 4, hipMemcpyDeviceToHost);
    // HW_ID: wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ...
    std::map<unsigned, std::vector<int>> cu;
    for (int b = 0; b < grid; ++b) {
        const Rec& r = h[b * 4];
        const unsigned key = ((r.xcc & 0xf) << 16) | (((r.hwid >> 13) & 7) << 8) | (((r.hwid >> 12) & 1) << 7) | ((r.hwid >> 8) & 0xf);
        cu[key].push_back(b);
    }
    printf("distinct CUs seen: %zu\n", cu.size());
    int shown = 0;
    for (auto& kv : cu) {
        if (shown++ >= 6) break;
        printf("CU key %05x:", kv.first);
        for (int b : kv.second) {
            printf("  [wg %d slots", b);
            for (int w = 0; w < 4; ++w) printf(" s%u/w%u", (h[b * 4 + w].hwid >> 4) & 3, h[b * 4 + w].hwid & 0xf);
            printf(" t0=%llu]", (h[b * 4].t0 - h[0].t0) / 100);
        }
        printf("\n");
    }
    return 0;
}
