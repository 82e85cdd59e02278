#include <hip/hip_runtime.h>
#include "mcnerf_common.h"
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const float* x, float* s, float* c, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) mcn_sincos(x[i], s[i], c[i]); }
int main() {
    const int n = 1 << 22; std::vector<float> x(n), s(n), c(n);
    unsigned long long st = 88172645463325252ull;
    for (int i = 0; i < n; ++i) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; double u = (st >> 11) * (1.0 / 9007199254740992.0); x[i] = (float)((u * 2 - 1) * (i % 3 == 0 ? 6000.0 : (i % 3 == 1 ? 20.0 : 1.0))); }
    float *dx, *ds, *dc; hipMalloc(&dx, n * 4); hipMalloc(&ds, n * 4); hipMalloc(&dc, n * 4);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, ds, dc, n); hipDeviceSynchronize();
    hipMemcpy(s.data(), ds, n * 4, hipMemcpyDeviceToHost); hipMemcpy(c.data(), dc, n * 4, hipMemcpyDeviceToHost);
    double es = 0, ec = 0;
    for (int i = 0; i < n; ++i) { es = fmax(es, fabs((double)s[i] - sin((double)x[i]))); ec = fmax(ec, fabs((double)c[i] - cos((double)x[i]))); }
    printf("max abs err sin %.3e cos %.3e\n", es, ec);
    return (es < 2.5e-7 && ec < 2.5e-7) ? 0 : 1;
}
