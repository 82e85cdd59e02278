// Alpha compositing along rays for gfx950: one 64-lane wavefront per ray, samples strided over the
// lanes (coalesced float4 reads), transmittance by wavefront prefix scans (DPP/shuffle), no LDS in
// the forward pass.
//
// Replaces the two composites of NeRF_Model.inference (model/mc_nerf.py:705-727) and
// NeRF_Model.sigma2weights (model/mc_nerf.py:729-736), plus the weights used for the fine-sample
// selection (model/mc_nerf.py:613-621).  The N(0,1) draws are inputs.
#include "mcnerf_kernels.h"

__device__ __forceinline__ float softplus_t(float x) {      // torch.nn.Softplus(beta=1, threshold=20)
    return x > 20.f ? x : log1pf(expf(x));
}
__device__ __forceinline__ float wave_incl_prod(float v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const float t = __shfl_up(v, o); if (lane >= o) v *= t; }
    return v;
}
__device__ __forceinline__ float wave_incl_sum(float v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const float t = __shfl_up(v, o); if (lane >= o) v += t; }
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// Running maximum of non-negative floats (as bit patterns) in ONE device word.  Every ray's wave contributes, and tens of
// thousands of same-address atomics serialise in the L2 (~150 us per launch at 32768 rays): read the word first with an
// L1-bypassing load and skip the atomic unless this wave would raise it -- after the first few waves almost none does.
// (A stale, lower value only costs a redundant atomic; the maximum never decreases.)
__device__ __forceinline__ void running_max_bits(unsigned* word, float v) {
    const unsigned mine = __float_as_uint(v);
    if (mine > __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(word, mine);
}

__global__ __launch_bounds__(256) void composite_fwd_kernel(McnCompositeArgs a) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= a.N) return;
    const int S = a.S;
    const float jit = a.jitter ? a.jitter[n] : 0.f;
    const float dx = a.rays_d[n * 3], dy = a.rays_d[n * 3 + 1], dz = a.rays_d[n * 3 + 2];
    const float rlen = sqrtf(dx * dx + dy * dy + dz * dz);
    const f32x4* sr = reinterpret_cast<const f32x4*>(a.sig_rgb) + (size_t)n * S;
    float carryT = 1.f, carryTs = 1.f, carryCum = 0.f;
    float ar = 0.f, ag = 0.f, ab = 0.f, aw = 0.f, aop = 0.f, adep = 0.f, wmx = 0.f;
    for (int base = 0; base < S; base += 64) {
        const int j = base + lane;
        const bool ok = j < S;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        float z = 0.f, delta = 0.f, e1 = 0.f, e2 = 0.f;
        if (ok) {
            v = sr[j];
            z = __fadd_rn(a.zgrid[j], jit);
            delta = (j + 1 < S) ? __fsub_rn(__fadd_rn(a.zgrid[j + 1], jit), z) : 1e10f;
            e1 = a.eps[(size_t)n * S + j];
            if (a.eps_sel) e2 = a.eps_sel[(size_t)n * S + j];
        }
        // rgb composite (sigma2weights)
        const float alpha = ok ? 1.f - expf(-delta * softplus_t(v[0] + e1)) : 0.f;
        const float u = ok ? (1.f - alpha) + 1e-10f : 1.f;
        const float inc = wave_incl_prod(u, lane);
        float excl = __shfl_up(inc, 1);
        if (lane == 0) excl = 1.f;
        const float w = alpha * (carryT * excl);
        carryT *= __shfl(inc, 63);
        ar += w * v[1]; ag += w * v[2]; ab += w * v[3]; aw += w;
        // selection weights (independent noise draw)
        if (a.eps_sel) {
            const float al2 = ok ? 1.f - expf(-delta * softplus_t(v[0] + e2)) : 0.f;
            const float u2 = ok ? (1.f - al2) + 1e-10f : 1.f;
            const float inc2 = wave_incl_prod(u2, lane);
            float ex2 = __shfl_up(inc2, 1);
            if (lane == 0) ex2 = 1.f;
            const float w2 = al2 * (carryTs * ex2);
            carryTs *= __shfl(inc2, 63);
            if (ok) { a.w_sel[(size_t)n * S + j] = w2; wmx = fmaxf(wmx, w2); }
        }
        // depth / opacity: noise-free, delta scaled by |d|, exp(-cumsum) transmittance
        if (a.depth) {
            const float sd = ok ? softplus_t(v[0]) * (delta * rlen) : 0.f;
            const float al3 = 1.f - expf(-sd);
            const float incs = wave_incl_sum(sd, lane);
            float exs = __shfl_up(incs, 1);      // exclusive sum by shift, never by subtraction:
            if (lane == 0) exs = 0.f;            // the last sample's sd (delta = 1e10) would cancel everything
            const float T = expf(-(carryCum + exs));
            carryCum += __shfl(incs, 63);
            const float p = ok ? T * al3 : 0.f;
            aop += p; adep += z * p;
        }
    }
    ar = wave_sum(ar); ag = wave_sum(ag); ab = wave_sum(ab); aw = wave_sum(aw);
    if (a.depth) { aop = wave_sum(aop); adep = wave_sum(adep); }
    if (a.eps_sel && a.wmax_bits) {
        wmx = wave_max(wmx);
        if (lane == 0) running_max_bits(a.wmax_bits, wmx);
    }
    if (lane == 0) {
        if (a.white_back) { ar = (ar + 1.f) - aw; ag = (ag + 1.f) - aw; ab = (ab + 1.f) - aw; }
        a.rgb[n * 3] = ar; a.rgb[n * 3 + 1] = ag; a.rgb[n * 3 + 2] = ab;
        if (a.depth) { a.depth[n] = adep; a.opacity[n] = aop; }
    }
}

hipError_t mcn_launch_composite_fwd(const McnCompositeArgs& a, hipStream_t st) {
    if (a.N <= 0) return hipSuccess;
    hipLaunchKernelGGL(composite_fwd_kernel, dim3((a.N + 3) / 4), dim3(256), 0, st, a);
    return hipGetLastError();
}

// Backward of the rgb composite wrt (sigma_raw, rgb) of every sample:
//   rgb_out = sum_j w_j c_j (+ 1 - sum_j w_j),  w_j = alpha_j T_j,  T_j = prod_{k<j} u_k,  u = 1 - alpha + 1e-10
//   d alpha_j = dw_j T_j - (sum_{k>j} dw_k w_k) / u_j      (the cumprod gradient, division form)
// Per-ray intermediates are staged in LDS (5 floats per sample per wave) between the prefix and the
// suffix pass.
__global__ __launch_bounds__(256) void composite_bwd_kernel(McnCompositeBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int n = blockIdx.x * 4 + wv;
    if (n >= a.N) return;
    const int S = a.S;
    float* sT = sm + (size_t)wv * 5 * S;
    float* sU = sT + S; float* sE = sU + S; float* sDW = sE + S; float* sDWW = sDW + S;
    const float jit = a.jitter ? a.jitter[n] : 0.f;
    const float g0 = a.d_rgb[n * 3], g1 = a.d_rgb[n * 3 + 1], g2 = a.d_rgb[n * 3 + 2];
    const float wb = a.white_back ? 1.f : 0.f;
    const f32x4* sr = reinterpret_cast<const f32x4*>(a.sig_rgb) + (size_t)n * S;
    f32x4* dst = reinterpret_cast<f32x4*>(a.d_sig_rgb) + (size_t)n * S;
    float carryT = 1.f;
    float gmx = 0.f;                       // max |gradient| written by this lane (scale of the split-f16 backward)
    for (int base = 0; base < S; base += 64) {
        const int j = base + lane;
        const bool ok = j < S;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        float delta = 0.f, e1 = 0.f;
        if (ok) {
            v = sr[j];
            const float z = __fadd_rn(a.zgrid[j], jit);
            delta = (j + 1 < S) ? __fsub_rn(__fadd_rn(a.zgrid[j + 1], jit), z) : 1e10f;
            e1 = a.eps[(size_t)n * S + j];
        }
        const float sx = v[0] + e1;
        const float sp = softplus_t(sx);
        const float ex = ok ? expf(-delta * sp) : 1.f;
        const float alpha = 1.f - ex;
        const float u = ok ? (1.f - alpha) + 1e-10f : 1.f;
        const float inc = wave_incl_prod(u, lane);
        float excl = __shfl_up(inc, 1);
        if (lane == 0) excl = 1.f;
        const float T = carryT * excl;
        carryT *= __shfl(inc, 63);
        if (ok) {
            const float w = alpha * T;
            const float dw = g0 * (v[1] - wb) + g1 * (v[2] - wb) + g2 * (v[3] - wb);
            const float dsp = sx > 20.f ? 1.f : 1.f / (1.f + expf(-sx));      // d softplus
            sT[j] = T; sU[j] = u; sE[j] = ex * delta * dsp; sDW[j] = dw; sDWW[j] = dw * w;
            f32x4 o; o[0] = 0.f; o[1] = w * g0; o[2] = w * g1; o[3] = w * g2;
            dst[j] = o;
            gmx = fmaxf(gmx, fmaxf(fabsf(o[1]), fmaxf(fabsf(o[2]), fabsf(o[3]))));
        }
    }
    // suffix pass (each lane re-reads only what it wrote: no barrier needed within the wave)
    float carryR = 0.f;
    const int nch = (S + 63) / 64;
    for (int c = nch - 1; c >= 0; --c) {
        const int j = c * 64 + lane;
        const bool ok = j < S;
        float x = ok ? sDWW[j] : 0.f;
        float inc = x;     // inclusive suffix sum within the chunk
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const float t = __shfl_down(inc, o); if (lane + o < 64) inc += t; }
        float after = __shfl_down(inc, 1);           // strictly-after sum inside the chunk
        if (lane == 63) after = 0.f;
        const float R = carryR + after;
        carryR += __shfl(inc, 0);
        if (ok) {
            const float dalpha = sDW[j] * sT[j] - R / sU[j];
            const float dsig = dalpha * sE[j];
            a.d_sig_rgb[((size_t)n * S + j) * 4] = dsig;
            gmx = fmaxf(gmx, fabsf(dsig));
        }
    }
    if (a.gmax_bits) {
        gmx = wave_max(gmx);
        if (lane == 0 && gmx < 3e38f) running_max_bits(a.gmax_bits, gmx);
    }
}

hipError_t mcn_launch_composite_bwd(const McnCompositeBwdArgs& a, hipStream_t st) {
    if (a.N <= 0) return hipSuccess;
    const size_t lds = (size_t)4 * 5 * a.S * sizeof(float);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(composite_bwd_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(composite_bwd_kernel, dim3((a.N + 3) / 4), dim3(256), lds, st, a);
    return hipGetLastError();
}
