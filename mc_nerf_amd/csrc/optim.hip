// Fused multi-tensor Rectified-Adam step for gfx950: ONE launch updates every parameter tensor of a
// param group (46 MLP tensors + 6 camera tensors in MC-NeRF), replacing the reference's per-tensor Python
// loop of ~10 ATen ops each (model/net_utils.py:38-99).  Pure streaming: 16 B/element read + 12 B written.
//
// Semantics per element (reference :62-63, 88-99), with the scalar step size / N_sma computed on the host
// exactly like the reference's 10-slot cache (:67-86):
//   v = beta2 v + (1-beta2) g^2 ;  m = beta1 m + (1-beta1) g
//   rectified (N_sma >= 5):  p -= wd*lr*p ;  p -= step_size*lr * m / (sqrt(v) + eps)
//   otherwise (step_size>0): p -= wd*lr*p ;  p -= step_size*lr * m
#include "mcnerf_kernels.h"

// Overflow guard: the 16-bit precision modes carry range limits (f16 operands saturate to inf, which the fp32
// accumulators turn into NaN gradients); one non-finite gradient would poison the flat parameter buffers for good.
// radam_check_kernel raises guard[0] if any gradient of the group is non-finite; radam_kernel then leaves parameters
// and moments untouched and counts the skipped step in guard[1] (read by the host whenever it likes: no sync per step).
__global__ __launch_bounds__(256) void radam_check_kernel(McnRadamTable t, unsigned* guard) {
    int ti = 0;
    while (ti + 1 < t.n_tensors && (int)blockIdx.x >= t.first_block[ti + 1]) ++ti;
    const long long base = (long long)((int)blockIdx.x - t.first_block[ti]) * MCN_RADAM_CHUNK;
    const float* __restrict__ g = t.g[ti];
    const long long n = t.n[ti];
    bool bad = false;
    for (long long i = base + threadIdx.x; i < base + MCN_RADAM_CHUNK && i < n; i += 256) {
        const unsigned b = __float_as_uint(g[i]);
        bad |= (b & 0x7F800000u) == 0x7F800000u;              // inf or NaN
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(guard, 1u);
}

__global__ __launch_bounds__(256) void radam_kernel(McnRadamTable t, unsigned* guard, int count_skip) {
    if (guard && guard[0]) {
        if (count_skip && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&guard[1], 1u);
        return;
    }
    // block -> (tensor, chunk): blocks are dealt to tensors by their precomputed first-block index
    int ti = 0;
    while (ti + 1 < t.n_tensors && (int)blockIdx.x >= t.first_block[ti + 1]) ++ti;
    const long long base = (long long)((int)blockIdx.x - t.first_block[ti]) * MCN_RADAM_CHUNK;
    float* __restrict__ p = t.p[ti];
    const float* __restrict__ g = t.g[ti];
    float* __restrict__ m = t.m[ti];
    float* __restrict__ v = t.v[ti];
    const long long n = t.n[ti];
    const float b1 = t.beta1, b2 = t.beta2, decay = t.wd * t.lr, ss = t.step_size * t.lr;
    for (long long i = base + threadIdx.x; i < base + MCN_RADAM_CHUNK && i < n; i += 256) {
        const float gi = g[i];
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        v[i] = vi; m[i] = mi;
        float pi = p[i];
        if (t.rectified) {
            pi += -decay * pi;
            pi += -ss * (mi / (sqrtf(vi) + t.eps));
        } else if (t.step_size > 0.f) {
            pi += -decay * pi;
            pi += -ss * mi;
        }
        p[i] = pi;
    }
}

// phase bits (include/mcnerf.h): 1 = clear guard[0] first, 2 = check the gradients, 4 = update, 8 = this launch counts a refused step
hipError_t mcn_launch_radam(const McnRadamTable& t, int n_blocks, unsigned* guard, int phase, hipStream_t st) {
    if (guard && (phase & 1)) {
        hipError_t e = hipMemsetAsync(guard, 0, sizeof(unsigned), st);
        if (e != hipSuccess) return e;
    }
    if (n_blocks <= 0) return hipSuccess;
    if (guard && (phase & 2)) hipLaunchKernelGGL(radam_check_kernel, dim3(n_blocks), dim3(256), 0, st, t, guard);
    if (phase & 4) hipLaunchKernelGGL(radam_kernel, dim3(n_blocks), dim3(256), 0, st, t, guard, (phase & 8) ? 1 : 0);
    return hipGetLastError();
}

// ---- finish of the step's one gradient all-reduce (mc_nerf_amd/distributed.py, FlatGradSync.sync): the arena holds
// [n_grad summed gradient floats | n_flags summed "this rank produced a gradient" flags].  One launch averages the gradients
// (arena / world, the division DistributedDataParallel applies: /root/reference main.py:60-62 wraps the model in it) and checks
// that every rank set the same flags (sum == world * local flag), counting a disagreeing step in *asym.
__global__ __launch_bounds__(256) void sync_finish_kernel(float* arena, long long n_grad, int n_flags, float world, const float* local, int* asym) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_grad) arena[i] = arena[i] / world;
    if (blockIdx.x == 0) {
        int bad = 0;
        for (int k = threadIdx.x; k < n_flags; k += blockDim.x) bad |= arena[n_grad + k] != local[k] * world;
        bad = __syncthreads_or(bad);
        if (bad && threadIdx.x == 0) atomicAdd(asym, 1);
    }
}
hipError_t mcn_launch_sync_finish(float* arena, long long n_grad, int n_flags, float world, const float* local, int* asym, hipStream_t st) {
    const long long blocks = (n_grad + 255) / 256;
    hipLaunchKernelGGL(sync_finish_kernel, dim3((unsigned)(blocks > 0 ? blocks : 1)), dim3(256), 0, st, arena, n_grad, n_flags, world, local, asym);
    return hipGetLastError();
}
