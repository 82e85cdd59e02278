// Debug-only clock probe (not part of the product library): one wave samples the shader-clock counter (s_memtime) against the
// constant 100 MHz counter (s_memrealtime) while other kernels run, so a timing script can report the effective
// shader clock under that kernel's load.   hipcc --offload-arch=gfx950 -shared -fPIC -o libclockprobe.so clockprobe.hip
#include <hip/hip_runtime.h>
__global__ void clock_probe_kernel(unsigned long long* out, int n, unsigned gap) {
    if (threadIdx.x != 0) return;
    for (int i = 0; i < n; ++i) {
        const unsigned long long r0 = wall_clock64(), c0 = clock64();
        unsigned long long r1;
        do { __builtin_amdgcn_s_sleep(32); r1 = wall_clock64(); } while (r1 - r0 < gap);
        const unsigned long long c1 = clock64();
        out[2 * i] = c1 - c0; out[2 * i + 1] = r1 - r0;
    }
}
extern "C" int clock_probe(unsigned long long* out, int n, unsigned gap, void* stream) {
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out, n, gap);
    return (int)hipGetLastError();
}
