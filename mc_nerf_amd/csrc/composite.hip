// Alpha compositing along rays for gfx950: one 64-lane wavefront per ray, samples strided over the
// lanes (coalesced float4 reads), transmittance by wavefront prefix scans (DPP), no LDS in
// the forward pass.
//
// Replaces the two composites of NeRF_Model.inference (model/mc_nerf.py:705-727) and
// NeRF_Model.sigma2weights (model/mc_nerf.py:729-736), plus the weights used for the fine-sample
// selection (model/mc_nerf.py:613-621).  The N(0,1) draws are inputs.
#include "mcnerf_kernels.h"

#define MCN_COMPOSITE_BLOCKS 2048      // 4 waves each: 32 waves per CU, every ray-wave resident at once

__device__ __forceinline__ float softplus_t(float x) {      // torch.nn.Softplus(beta=1, threshold=20)
    return x > 20.f ? x : log1pf(expf(x));
}
// Wavefront scans on the DPP path (row shifts inside each 16-lane row, then row_bcast:15 / row_bcast:31 carry the row totals
// across: six full-rate vector ops with a DPP operand) instead of six ds_bpermute round trips through the LDS crossbar; lane 63 of
// an inclusive scan is the reduction (v_readlane).  A lane whose DPP source is out of range keeps `old` = the identity.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float ident, float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(ident), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
#define MCN_WAVE_SCAN(OP, IDENT)                                                         \
    v = OP(v, dpp_f<0x111, 0xf>(IDENT, v));   /* row_shr:1 */                             \
    v = OP(v, dpp_f<0x112, 0xf>(IDENT, v));   /* row_shr:2 */                             \
    v = OP(v, dpp_f<0x114, 0xf>(IDENT, v));   /* row_shr:4 */                             \
    v = OP(v, dpp_f<0x118, 0xf>(IDENT, v));   /* row_shr:8 */                             \
    v = OP(v, dpp_f<0x142, 0xa>(IDENT, v));   /* row_bcast:15 into rows 1, 3 */           \
    v = OP(v, dpp_f<0x143, 0xc>(IDENT, v));   /* row_bcast:31 into rows 2, 3 */
__device__ __forceinline__ float op_mul(float a, float b) { return a * b; }
__device__ __forceinline__ float op_add(float a, float b) { return a + b; }
__device__ __forceinline__ float wave_incl_prod(float v) { MCN_WAVE_SCAN(op_mul, 1.f) return v; }
__device__ __forceinline__ float wave_incl_sum(float v) { MCN_WAVE_SCAN(op_add, 0.f) return v; }
__device__ __forceinline__ float wave_shr1(float v, float ident) { return dpp_f<0x138, 0xf>(ident, v); }      // lane l <- lane l - 1, lane 0 <- ident (wave_shr:1)
__device__ __forceinline__ float wave_last(float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63)); }
__device__ __forceinline__ float wave_sum(float v) { return wave_last(wave_incl_sum(v)); }
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_f<0x111, 0xf>(v, v)); v = fmaxf(v, dpp_f<0x112, 0xf>(v, v)); v = fmaxf(v, dpp_f<0x114, 0xf>(v, v));
    v = fmaxf(v, dpp_f<0x118, 0xf>(v, v)); v = fmaxf(v, dpp_f<0x142, 0xa>(v, v)); v = fmaxf(v, dpp_f<0x143, 0xc>(v, v));
    return wave_last(v);
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
    const int S = a.S;
    float wmx = 0.f;
    // a wave walks several rays (grid capped at MCN_COMPOSITE_BLOCKS): ONE running-maximum operation per wave, not per ray --
    // the waves' reads of that single device word serialise at its memory channel (32768 of them were most of the 45-65 us this
    // kernel took at 32768 rays)
    for (int n = blockIdx.x * 4 + (threadIdx.x >> 6); n < a.N; n += gridDim.x * 4) {
    const float jit = a.jitter ? a.jitter[n] : 0.f;
    const float dx = a.rays_d[n * 3], dy = a.rays_d[n * 3 + 1], dz = a.rays_d[n * 3 + 2];
    const float rlen = sqrtf(dx * dx + dy * dy + dz * dz);
    const f32x4* sr = reinterpret_cast<const f32x4*>(a.sig_rgb) + (size_t)n * S;
    float carryT = 1.f, carryTs = 1.f, carryCum = 0.f;
    float ar = 0.f, ag = 0.f, ab = 0.f, aw = 0.f, aop = 0.f, adep = 0.f;
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
        const float inc = wave_incl_prod(u);
        const float excl = wave_shr1(inc, 1.f);
        const float w = alpha * (carryT * excl);
        carryT *= wave_last(inc);
        ar += w * v[1]; ag += w * v[2]; ab += w * v[3]; aw += w;
        // selection weights (independent noise draw)
        if (a.eps_sel) {
            const float al2 = ok ? 1.f - expf(-delta * softplus_t(v[0] + e2)) : 0.f;
            const float u2 = ok ? (1.f - al2) + 1e-10f : 1.f;
            const float inc2 = wave_incl_prod(u2);
            const float ex2 = wave_shr1(inc2, 1.f);
            const float w2 = al2 * (carryTs * ex2);
            carryTs *= wave_last(inc2);
            if (ok) { a.w_sel[(size_t)n * S + j] = w2; wmx = fmaxf(wmx, w2); }
        }
        // depth / opacity: noise-free, delta scaled by |d|, exp(-cumsum) transmittance
        if (a.depth) {
            const float sd = ok ? softplus_t(v[0]) * (delta * rlen) : 0.f;
            const float al3 = 1.f - expf(-sd);
            const float incs = wave_incl_sum(sd);
            const float exs = wave_shr1(incs, 0.f);   // exclusive sum by shift, never by subtraction: the last sample's sd (delta = 1e10) would cancel everything
            const float T = expf(-(carryCum + exs));
            carryCum += wave_last(incs);
            const float p = ok ? T * al3 : 0.f;
            aop += p; adep += z * p;
        }
    }
    ar = wave_sum(ar); ag = wave_sum(ag); ab = wave_sum(ab); aw = wave_sum(aw);
    if (a.depth) { aop = wave_sum(aop); adep = wave_sum(adep); }
    if (lane == 0) {
        if (a.white_back) { ar = (ar + 1.f) - aw; ag = (ag + 1.f) - aw; ab = (ab + 1.f) - aw; }
        a.rgb[n * 3] = ar; a.rgb[n * 3 + 1] = ag; a.rgb[n * 3 + 2] = ab;
        if (a.depth) { a.depth[n] = adep; a.opacity[n] = aop; }
    }
    }
    if (a.eps_sel && a.wmax_bits) {
        wmx = wave_max(wmx);
        if (lane == 0) running_max_bits(a.wmax_bits, wmx);
    }
}

hipError_t mcn_launch_composite_fwd(const McnCompositeArgs& a, hipStream_t st) {
    if (a.N <= 0) return hipSuccess;
    hipLaunchKernelGGL(composite_fwd_kernel, dim3(min((a.N + 3) / 4, MCN_COMPOSITE_BLOCKS)), dim3(256), 0, st, a);
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
    const int S = a.S;
    float* sT = sm + (size_t)wv * 5 * S;
    float* sU = sT + S; float* sE = sU + S; float* sDW = sE + S; float* sDWW = sDW + S;
    float gmx = 0.f;                       // max |gradient| written by this lane (scale of the split-f16 backward)
    for (int n = blockIdx.x * 4 + wv; n < a.N; n += gridDim.x * 4) {      // (several rays per wave: see the forward kernel)
    const float jit = a.jitter ? a.jitter[n] : 0.f;
    const float g0 = a.d_rgb[n * 3], g1 = a.d_rgb[n * 3 + 1], g2 = a.d_rgb[n * 3 + 2];
    const float wb = a.white_back ? 1.f : 0.f;
    const f32x4* sr = reinterpret_cast<const f32x4*>(a.sig_rgb) + (size_t)n * S;
    f32x4* dst = reinterpret_cast<f32x4*>(a.d_sig_rgb) + (size_t)n * S;
    float carryT = 1.f;
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
        const float inc = wave_incl_prod(u);
        const float excl = wave_shr1(inc, 1.f);
        const float T = carryT * excl;
        carryT *= wave_last(inc);
        if (ok) {
            const float w = alpha * T;
            const float dw = g0 * (v[1] - wb) + g1 * (v[2] - wb) + g2 * (v[3] - wb);
            const float dsp = sx > 20.f ? 1.f : 1.f / (1.f + expf(-sx));      // d softplus
            sT[j] = T; sU[j] = u; sE[j] = ex * delta * dsp; sDW[j] = dw; sDWW[j] = dw * w;
            // components 1..3 only: component 0 (d sigma) belongs to the lane that takes sample j in the suffix pass
            float* o = reinterpret_cast<float*>(dst + j);
            const float o1 = w * g0, o2 = w * g1, o3 = w * g2;
            o[1] = o1;
            *reinterpret_cast<f32x2*>(o + 2) = (f32x2){o2, o3};
            gmx = fmaxf(gmx, fmaxf(fabsf(o1), fmaxf(fabsf(o2), fabsf(o3))));
        }
    }
    // pass 1's LDS writes are read below by OTHER lanes of this wave: the hardware executes one wave's LDS operations in issue
    // order, the fence + wave barrier keep the compiler from moving them across this line
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // suffix pass: lane l takes sample 63 - l of the chunk, so the sum of the samples AFTER a sample is an exclusive PREFIX scan over
    // the lanes (the DPP scans run one way).
    float carryR = 0.f;
    const int nch = (S + 63) / 64;
    for (int c = nch - 1; c >= 0; --c) {
        const int j = c * 64 + (63 - lane);
        const bool ok = j < S;
        const float inc = wave_incl_sum(ok ? sDWW[j] : 0.f);      // this sample and every later one of the chunk
        const float R = carryR + wave_shr1(inc, 0.f);             // strictly later samples (of the chunk + of the chunks behind it)
        carryR += wave_last(inc);
        if (ok) {
            const float dalpha = sDW[j] * sT[j] - R / sU[j];
            const float dsig = dalpha * sE[j];
            a.d_sig_rgb[((size_t)n * S + j) * 4] = dsig;
            gmx = fmaxf(gmx, fabsf(dsig));
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");         // the next ray's pass 1 rewrites the same LDS words
    __builtin_amdgcn_wave_barrier();
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
    hipLaunchKernelGGL(composite_bwd_kernel, dim3(min((a.N + 3) / 4, MCN_COMPOSITE_BLOCKS)), dim3(256), lds, st, a);
    return hipGetLastError();
}
