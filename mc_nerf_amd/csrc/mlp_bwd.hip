// Fused NeRF MLP backward (activation-gradient chain) for gfx950.
//
// Replaces autograd through CorseFine_NeRF.forward / eval_sh / sigmoid (model/net_block.py:67-78,
// model/net_utils.py:103-191), SinCosEmbedding.forward (model/net_block.py:20-35) and the sample
// position arithmetic (model/mc_nerf.py:602, 635, 690-691).
//
// One workgroup = one tile of MT samples, same geometry as the forward kernel.  The LDS tile holds
// dY (gradient wrt a layer's pre-activation); dX = W^T dY is an MFMA GEMM against the TRANSPOSED
// packed weights; the ReLU mask of the producing layer comes from the saved forward activations.
// Every dY is also written to dy_save: it is the A operand of the weight-gradient kernel (mlp_dw.hip).
#include "mcnerf_common.h"
#include "mcnerf_kernels.h"

template <int WIDTH>
struct BwdSmem {
    using G = McnGeom<WIDTH>;
    static constexpr int MT = G::WM * G::MI * 32;
    static constexpr int NT = G::WN * G::WM * 64;
    static constexpr int XW = WIDTH > 64 ? WIDTH : 64;
    static constexpr int oX = 0;
    static constexpr int oDsig = oX + MT * XW;        // [MT] d sigma_raw
    static constexpr int oDdir = oDsig + MT;          // [MT][4] d view-direction from the SH colour
    static constexpr int oGo = oDdir + MT * 4;        // [MT][4] d sample position (= d ray origin contribution)
    static constexpr int oAddr = oGo + MT * 4;        // [MT] int ray id (or -1)
    static constexpr int oZ = oAddr + MT;             // [MT] z value
    static constexpr int total = oZ + MT;
    static constexpr size_t bytes = (size_t)total * 4;
};

// acc (gradient wrt a post-ReLU activation) -> masked by the forward's ReLU bit mask -> LDS tile + dy_save.
// The mask words are read with UNCONDITIONAL loads (row clamped): a predicated load is an exec-masked branch
// that hipcc follows with a vmcnt(0) drain, i.e. one serialised memory round trip per load.
template <int WIDTH, int NI, int MI>
__device__ __forceinline__ void mask_store(f32x16 (&acc)[NI][MI], const unsigned int* __restrict__ msave, float* __restrict__ dysave,
                                           float* X, int xw, int mrow0, int ncol0, long long row0, long long total, int lane) {
    const int r = lane & 31, h = lane >> 5;
    unsigned wm[NI][MI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const long long g = row0 + mrow0 + mi * 32 + r;
            const long long gc = g < total ? g : total - 1;
            wm[ni][mi] = msave[(size_t)gc * (WIDTH / 32) + (ncol0 >> 5) + ni];
        }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k4 = ncol0 + ni * 32 + 8 * q + 4 * h;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int m = mrow0 + mi * 32 + r;
                const bool ok = row0 + m < total;
                const unsigned w = ok ? wm[ni][mi] >> (8 * q + 4 * h) : 0u;
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = ((w >> e) & 1u) ? acc[ni][mi][4 * q + e] : 0.f;
                *reinterpret_cast<f32x4*>(&X[mcn_swz_chunk(m, k4 >> 2, xw)]) = v;
                if (ok) *reinterpret_cast<f32x4*>(dysave + (size_t)(row0 + m) * WIDTH + k4) = v;
            }
        }
}

// Per-sample prologue for an SH degree other than 2 (templated on the degree: every index is a constant): sigmoid and SH
// backward -> the dY row of sh.2 (3 (DEG + 1)^2 columns, d sigma in the spare column after them) into the X tile and dsh_save,
// d sigma, d colour / d direction through the basis derivatives (model/net_utils.py:152-179).
template <int WIDTH, int DEG>
__device__ __forceinline__ void mcn_sh_bwd_general(const McnMlpBwdArgs& a, float* X, int xw, int m, long long g, long long total,
                                                   float* sdsig, float* sddir, int* sray, float* sz) {
    constexpr int NB = (DEG + 1) * (DEG + 1), NSH = 3 * NB, NSHP = NSH < 32 ? 32 : 64;
    float dsg = 0.f, ddx = 0.f, ddy = 0.f, ddz = 0.f, zv = 0.f;
    int ray = -1;
    float dpre[3] = {0.f, 0.f, 0.f};
    float b[MCN_NBMAX];
#pragma unroll
    for (int i = 0; i < MCN_NBMAX; ++i) b[i] = 0.f;
    if (g < total) {
        int j;
        if (a.idx) { const int2 rj = a.idx[g]; ray = rj.x; j = rj.y; }
        else { ray = (int)(g / a.S); j = (int)(g - (long long)ray * a.S); }
        zv = a.zgrid[j];
        if (a.jitter) zv = __fadd_rn(zv, a.jitter[ray]);
        const size_t addr = (size_t)ray * a.S + j;
        const f32x4 o = *reinterpret_cast<const f32x4*>(a.out + addr * 4);
        const f32x4 go = *reinterpret_cast<const f32x4*>(a.d_out + addr * 4);
        dsg = go[0];
        const float x = a.rays_d[ray * 3], y = a.rays_d[ray * 3 + 1], z = a.rays_d[ray * 3 + 2];
        mcn_sh_basis16(DEG, x, y, z, b);
        float gx[MCN_NBMAX], gy[MCN_NBMAX], gz[MCN_NBMAX];
        mcn_sh_dbasis16(DEG, x, y, z, gx, gy, gz);
        const float* sh = a.sh_save + (size_t)g * NSHP;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            dpre[c] = go[1 + c] * o[1 + c] * (1.f - o[1 + c]);
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const float s = sh[NB * c + i];
                ddx += dpre[c] * (gx[i] * s); ddy += dpre[c] * (gy[i] * s); ddz += dpre[c] * (gz[i] * s);
            }
        }
    }
    float* dst = g < total ? a.dsh_save + (size_t)g * NSHP : nullptr;
#pragma unroll
    for (int n = 0; n < NSHP; ++n) {
        const float v = n < NSH ? dpre[n / NB] * b[n % NB] : (n == NSH ? dsg : 0.f);
        X[mcn_swz(m, n, xw)] = v;
        if (dst) dst[n] = v;
    }
    sdsig[m] = dsg; sray[m] = ray; sz[m] = zv;
    sddir[m * 4] = ddx; sddir[m * 4 + 1] = ddy; sddir[m * 4 + 2] = ddz; sddir[m * 4 + 3] = 0.f;
}

template <int WIDTH>
__global__ __launch_bounds__(McnGeom<WIDTH>::WN * McnGeom<WIDTH>::WM * 64, 2) void mlp_bwd_kernel(McnMlpBwdArgs a) {
    using G = McnGeom<WIDTH>;
    using SM = BwdSmem<WIDTH>;
    constexpr int MT = SM::MT, XW = SM::XW, NI = G::NI, MI = G::MI, WN = G::WN, NT = SM::NT, WAVES = NT / 64;
    constexpr int NSH = WIDTH / 8;            // reduction steps over a hidden-wide dY
    constexpr int W4 = WIDTH / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* X = smem + SM::oX;
    float* sdsig = smem + SM::oDsig;
    float* sddir = smem + SM::oDdir;
    float* sgo = smem + SM::oGo;
    int* sray = reinterpret_cast<int*>(smem + SM::oAddr);
    float* sz = smem + SM::oZ;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave % WN, wm = wave / WN;
    const int mrow0 = wm * MI * 32, ncol0 = wn * NI * 32;
    const long long total = a.count ? (long long)min(*a.count, a.max_rows) : (long long)a.n_rays * a.S;
    const long long row0 = (long long)blockIdx.x * MT;
    if (row0 >= total) return;
    const McnLayout& L = a.lay;
    const int D = L.depth;
    const float* __restrict__ prm = a.params;
    const f32x4* __restrict__ pk = reinterpret_cast<const f32x4*>(a.packed);
    const size_t AS = a.act_stride;

    // ---- per-sample prologue: sigmoid and SH backward -> dsh (the dY of sh.2), d sigma, d dir
    for (int m = tid; m < MT; m += NT) {
        const long long g = row0 + m;
        if (L.sh_deg != 2) {          // (block-uniform) general SH degree
            switch (L.sh_deg) {
                case 0: mcn_sh_bwd_general<WIDTH, 0>(a, X, XW, m, g, total, sdsig, sddir, sray, sz); break;
                case 1: mcn_sh_bwd_general<WIDTH, 1>(a, X, XW, m, g, total, sdsig, sddir, sray, sz); break;
                default: mcn_sh_bwd_general<WIDTH, 3>(a, X, XW, m, g, total, sdsig, sddir, sray, sz); break;
            }
            continue;
        }
        __attribute__((aligned(16))) float dsh[MCN_NSHP];
#pragma unroll
        for (int i = 0; i < MCN_NSHP; ++i) dsh[i] = 0.f;
        float dsg = 0.f, ddx = 0.f, ddy = 0.f, ddz = 0.f, zv = 0.f;
        int ray = -1;
        if (g < total) {
            int j;
            if (a.idx) { const int2 rj = a.idx[g]; ray = rj.x; j = rj.y; }
            else { ray = (int)(g / a.S); j = (int)(g - (long long)ray * a.S); }
            zv = a.zgrid[j];
            if (a.jitter) zv = __fadd_rn(zv, a.jitter[ray]);
            const size_t addr = (size_t)ray * a.S + j;
            const f32x4 o = *reinterpret_cast<const f32x4*>(a.out + addr * 4);
            const f32x4 go = *reinterpret_cast<const f32x4*>(a.d_out + addr * 4);
            dsg = go[0];
            const float x = a.rays_d[ray * 3], y = a.rays_d[ray * 3 + 1], z = a.rays_d[ray * 3 + 2];
            float b[9];
            mcn_sh_basis(x, y, z, b);
            __attribute__((aligned(16))) float sh[MCN_NSHP];                 // the saved SH row: 7 vector loads
#pragma unroll
            for (int i = 0; i < MCN_NSH + 1; i += 4)
                *reinterpret_cast<f32x4*>(&sh[i]) = *reinterpret_cast<const f32x4*>(a.sh_save + (size_t)g * MCN_NSHP + i);
            const float C1 = 0.4886025119029199f, C20 = 1.0925484305920792f, C22 = 0.31539156525252005f,
                        C24 = 0.5462742152960396f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float dpre = go[1 + c] * o[1 + c] * (1.f - o[1 + c]);
#pragma unroll
                for (int i = 0; i < 9; ++i) dsh[9 * c + i] = dpre * b[i];
                const float* s = sh + 9 * c;
                // d(pre)/d(dir): derivatives of the nine basis polynomials
                ddx += dpre * (-C1 * s[3] + C20 * y * s[4] - 2.f * C22 * x * s[6] - C20 * z * s[7] + 2.f * C24 * x * s[8]);
                ddy += dpre * (-C1 * s[1] + C20 * x * s[4] - C20 * z * s[5] - 2.f * C22 * y * s[6] - 2.f * C24 * y * s[8]);
                ddz += dpre * (C1 * s[2] - C20 * y * s[5] + 4.f * C22 * z * s[6] - C20 * x * s[7]);
            }
            dsh[MCN_NSH] = dsg;      // spare column 27 carries d sigma: the dW kernel reduces d sigma.2.{weight,bias} from it
            float* dst = a.dsh_save + (size_t)g * MCN_NSHP;
#pragma unroll
            for (int i = 0; i < MCN_NSHP; i += 4) *reinterpret_cast<f32x4*>(dst + i) = *reinterpret_cast<f32x4*>(&dsh[i]);
        }
#pragma unroll
        for (int c4 = 0; c4 < MCN_NSHP / 4; ++c4)
            *reinterpret_cast<f32x4*>(&X[mcn_swz_chunk(m, c4, XW)]) = *reinterpret_cast<f32x4*>(&dsh[4 * c4]);
        sdsig[m] = dsg; sray[m] = ray; sz[m] = zv;
        sddir[m * 4] = ddx; sddir[m * 4 + 1] = ddy; sddir[m * 4 + 2] = ddz; sddir[m * 4 + 3] = 0.f;
    }
    __syncthreads();

    f32x16 acc[NI][MI];
    // ---- sh.2^T : dsh [MT][32] -> d hc ; mask with hc -> dY of sh.0
    mcn_zero<NI, MI>(acc);
    mcn_gemm_seg<NI, MI>(acc, X, XW, mrow0, 0, L.nshp / 8, pk + (L.bC2 >> 2) + (wn * NI) * (L.nshp / 8) * 64, lane);
    __syncthreads();
    mask_store<WIDTH, NI, MI>(acc, a.mask_save + (size_t)(D + 1) * (AS / 32), a.dy_save + (size_t)(D + 1) * AS, X, XW, mrow0, ncol0, row0, total, lane);
    __syncthreads();
    // ---- sh.0^T and sigma.0^T both feed d h_{D-1}
    mcn_zero<NI, MI>(acc);
    mcn_gemm_seg<NI, MI>(acc, X, XW, mrow0, 0, NSH, pk + (L.bC1 >> 2) + (wn * NI) * NSH * 64, lane);
    __syncthreads();
    {   // dY of sigma.0 = d sigma * w_sigma2 masked by hs > 0 (outer product, no GEMM)
        const unsigned int* hm = a.mask_save + (size_t)D * (AS / 32);
        float* dys = a.dy_save + (size_t)D * AS;
        const float* w2 = prm + L.pWs2;
        constexpr int MG = NT / W4;             // sample groups (threads / chunks per row)
        const int c4 = tid % W4, mg = tid / W4;
        const f32x4 ww = *reinterpret_cast<const f32x4*>(w2 + 4 * c4);
        for (int m = mg; m < MT; m += MG) {
            const bool ok = row0 + m < total;
            const long long gc = ok ? row0 + m : total - 1;
            const unsigned w = hm[(size_t)gc * (WIDTH / 32) + (c4 >> 3)] >> (4 * (c4 & 7));
            const float ds = ok ? sdsig[m] : 0.f;
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = ((w >> e) & 1u) ? ds * ww[e] : 0.f;
            *reinterpret_cast<f32x4*>(&X[mcn_swz_chunk(m, c4, XW)]) = v;
            if (ok) *reinterpret_cast<f32x4*>(dys + (size_t)(row0 + m) * WIDTH + 4 * c4) = v;
        }
    }
    __syncthreads();
    mcn_gemm_seg<NI, MI>(acc, X, XW, mrow0, 0, NSH, pk + (L.bS1 >> 2) + (wn * NI) * NSH * 64, lane);
    __syncthreads();
    mask_store<WIDTH, NI, MI>(acc, a.mask_save + (size_t)(D - 1) * (AS / 32), a.dy_save + (size_t)(D - 1) * AS, X, XW, mrow0, ncol0, row0, total, lane);
    __syncthreads();

    // ---- trunk, last layer to first.  X holds dY_l; the encoded-input gradient accumulates in denc.
    f32x16 denc[1][1];
    mcn_zero<1, 1>(denc);
    constexpr int ENC_TILES = 2 * (MT / 32);          // (2 k-tiles of the 64 encoded channels) x m-tiles
    static_assert(ENC_TILES <= 2 * WAVES, "at most two encoded-gradient tiles per wave");
    f32x16 denc2[1][1];                                // second tile when ENC_TILES > WAVES
    mcn_zero<1, 1>(denc2);
    for (int l = D - 1; l >= 0; --l) {
        if (l == 0 || ((L.skip_mask >> l) & 1u)) {
            const f32x4* pe = pk + ((l == 0 ? L.bEnc0 : L.bEncS[l]) >> 2);
            {
                const int t = wave;
                if (t < ENC_TILES) mcn_gemm_seg<1, 1>(denc, X, XW, (t >> 1) * 32, 0, NSH, pe + (t & 1) * NSH * 64, lane);
            }
            if (ENC_TILES > WAVES) {
                const int t = wave + WAVES;
                if (t < ENC_TILES) mcn_gemm_seg<1, 1>(denc2, X, XW, (t >> 1) * 32, 0, NSH, pe + (t & 1) * NSH * 64, lane);
            }
        }
        if (l == 0) break;
        mcn_zero<NI, MI>(acc);
        mcn_gemm_seg<NI, MI>(acc, X, XW, mrow0, 0, NSH, pk + (L.bH[l] >> 2) + (wn * NI) * NSH * 64, lane);
        __syncthreads();
        mask_store<WIDTH, NI, MI>(acc, a.mask_save + (size_t)(l - 1) * (AS / 32), a.dy_save + (size_t)(l - 1) * AS, X, XW, mrow0, ncol0, row0, total, lane);
        __syncthreads();
    }
    __syncthreads();
    // ---- encoded-input gradient -> LDS [MT][64]
    {
        const int r = lane & 31, h = lane >> 5;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int t = wave + WAVES * pass;
            if (t < ENC_TILES) {
                const int mt = t >> 1, kt = t & 1;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = pass == 0 ? denc[0][0][4 * q + e] : denc2[0][0][4 * q + e];
                    *reinterpret_cast<f32x4*>(&X[mcn_swz_chunk(mt * 32 + r, (kt * 32 + 8 * q + 4 * h) >> 2, XW)]) = v;
                }
            }
        }
    }
    __syncthreads();
    if (a.d_enc_out) {     // caller-supplied encodings (model/net_block.py:67-78 differentiated with respect to its input x)
        const int nenc = L.nenc;
        for (int it = tid; it < MT * nenc; it += NT) {
            const int m = it / nenc, ch = it - m * nenc;
            if (row0 + m < total) a.d_enc_out[(size_t)(row0 + m) * nenc + ch] = X[mcn_swz(m, ch, XW)];
        }
    }
    // ---- encoding backward -> d xyz -> d rays_o / d rays_d
    //   enc channel 3+2Fc+f = w_f sin(2^f x_c), 3+2Fc+F+f = w_f cos(2^f x_c)  (w_f already inside enc_save; F = 10 by default)
    if (a.d_rays_o || a.d_rays_d) {
        const int F = L.nfreq;
        for (int it = tid; it < MT * 3; it += NT) {
            const int m = it / 3, c = it - m * 3;
            const long long g = row0 + m;
            float dx = 0.f;
            if (g < total && !a.d_enc_out) {
                const float* en = a.enc_save + (size_t)g * MCN_ENCP;
                dx = X[mcn_swz(m, c, XW)];
                for (int f = 0; f < F; ++f) {
                    const float s = en[3 + 2 * F * c + f], co = en[3 + 2 * F * c + F + f];
                    const float ds = X[mcn_swz(m, 3 + 2 * F * c + f, XW)], dc = X[mcn_swz(m, 3 + 2 * F * c + F + f, XW)];
                    dx += (float)(1 << f) * (co * ds - s * dc);
                }
            }
            sgo[m * 4 + c] = dx;                                       // d origin
            sddir[m * 4 + c] = dx * sz[m] + sddir[m * 4 + c];          // d direction: through x = o + d z, plus the SH term
        }
        __syncthreads();
        // samples of one ray are contiguous in the tile: the first sample of each run sums the run, so a
        // ray costs 6 atomics per tile instead of 6 per sample (64-way same-address contention otherwise)
        for (int m = tid; m < MT; m += NT) {
            const int ray = sray[m];
            if (ray < 0 || (m > 0 && sray[m - 1] == ray)) continue;
            float so[3] = {0.f, 0.f, 0.f}, sd[3] = {0.f, 0.f, 0.f};
            for (int mm = m; mm < MT && sray[mm] == ray; ++mm) {
#pragma unroll
                for (int c = 0; c < 3; ++c) { so[c] += sgo[mm * 4 + c]; sd[c] += sddir[mm * 4 + c]; }
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                if (a.d_rays_o) atomicAdd(a.d_rays_o + ray * 3 + c, so[c]);
                if (a.d_rays_d) atomicAdd(a.d_rays_d + ray * 3 + c, sd[c]);
            }
        }
    }
}

template <int WIDTH>
static hipError_t launch_bwd(const McnMlpBwdArgs& a, long long max_rows, hipStream_t st) {
    using SM = BwdSmem<WIDTH>;
    const int grid = (int)((max_rows + SM::MT - 1) / SM::MT);
    if (grid <= 0) return hipSuccess;
    auto kern = mlp_bwd_kernel<WIDTH>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SM::bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(SM::NT), SM::bytes, st, a);
    return hipGetLastError();
}

hipError_t mcn_launch_mlp_bwd(const McnMlpBwdArgs& a, hipStream_t st) {
    const long long max_rows = a.count ? (long long)a.max_rows : (long long)a.n_rays * a.S;
    switch (a.lay.width) {
        case 256: return launch_bwd<256>(a, max_rows, st);
        case 128: return launch_bwd<128>(a, max_rows, st);
        case 64:  return launch_bwd<64>(a, max_rows, st);
        case 32:  return launch_bwd<32>(a, max_rows, st);
    }
    return hipErrorInvalidValue;
}
