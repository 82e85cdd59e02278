// Fused NeRF MLP forward for gfx950: sample generation -> positional encoding -> trunk -> sigma / SH
// heads -> SH colour -> sigmoid, one launch, activations never leave the CU (LDS + registers).
//
// Replaces (reference, eager PyTorch): NeRF_Model.inference's gather/encode/MLP/scatter part
// (model/mc_nerf.py:688-701), SinCosEmbedding.forward (model/net_block.py:20-35),
// CorseFine_NeRF.forward (model/net_block.py:67-78) and eval_sh (model/net_utils.py:103-191).
//
// One workgroup = WN x WM waves (4 at widths 128/256) = one tile of MT samples (64), two workgroups per CU.  Per layer each wave accumulates its
// (NI x MI) 32x32 output tiles with v_mfma_f32_32x32x2_f32 (exact fp32), weight fragments streamed
// from L2 in packed order, activation fragments read from the swizzled LDS tile; the layer output
// (bias + ReLU) is written back in place after a barrier.
#include "mcnerf_common.h"
#include "mcnerf_kernels.h"

template <int WIDTH>
struct FwdSmem {
    using G = McnGeom<WIDTH>;
    static constexpr int MT = G::WM * G::MI * 32;
    static constexpr int NT = G::WN * G::WM * 64;       // threads per workgroup
    static constexpr int XW = WIDTH > 64 ? WIDTH : 64;
    static constexpr int oX = 0;
    static constexpr int oXyz = oX + MT * XW;          // [MT][4]  x,y,z,z_val
    static constexpr int oDir = oXyz + MT * 4;         // [MT][4]  dx,dy,dz,-
    static constexpr int oSig = oDir + MT * 4;         // [WN][MT] partial sigma
    static constexpr int oSh = oSig + G::WN * MT;      // [MT][33] sh coefficients
    static constexpr int oAddr = oSh + MT * 33;        // [MT] int: ray*S + j  (or -1)
    static constexpr int total = oAddr + MT;
    static constexpr size_t bytes = (size_t)total * 4;
};

// Writes the 63(+1 pad) encoded channels of every tile row into X columns 0..63 (swizzled).
// Channel order (model/net_block.py:22-33): [x,y,z, per coord c: sin(2^f c) f=0..9, cos(2^f c) f=0..9],
// each multiplied by the BARF weight of its frequency (all ones when BARF is off).
template <int MT, int XW>
__device__ __forceinline__ void write_encoding(float* X, const float* sxyz, const float* barf_w, int tid, int nthreads,
                                               const float* enc_in = nullptr, long long row0 = 0, long long total = 0, int F = MCN_NFREQ) {
    // (F = `emb_freqs_xyz` <= 10: 3 + 6 F real channels, [x,y,z, per coord: sin f < F, cos f < F]; the columns up to 64 are zero)
    const int nenc = 3 + 6 * F;
    if (enc_in) {          // caller-supplied encodings [rows][63] (the stand-alone CorseFine_NeRF.forward, model/net_block.py:67-78)
        for (int it = tid; it < MT * MCN_ENCP; it += nthreads) {
            const int m = it / MCN_ENCP, ch = it - m * MCN_ENCP;
            X[mcn_swz(m, ch, XW)] = (ch < nenc && row0 + m < total) ? enc_in[(size_t)(row0 + m) * nenc + ch] : 0.f;
        }
        return;
    }
    for (int it = tid; it < MT * 3 * F; it += nthreads) {
        const int m = it / (3 * F), cf = it - m * (3 * F);
        const int c = cf / F, f = cf - c * F;
        const float v = sxyz[m * 4 + c] * (float)(1 << f);     // exact: power-of-two scale
        float s, co;
        mcn_sincos(v, s, co);
        const float w = barf_w[f];
        X[mcn_swz(m, 3 + c * 2 * F + f, XW)] = s * w;
        X[mcn_swz(m, 3 + c * 2 * F + F + f, XW)] = co * w;
    }
    const int npad = MCN_ENCP - nenc;                      // zero columns behind the real channels (1 at F = 10)
    for (int it = tid; it < MT * (3 + npad); it += nthreads) {
        const int m = it / (3 + npad), c = it - m * (3 + npad);
        X[mcn_swz(m, c < 3 ? c : nenc + (c - 3), XW)] = c < 3 ? sxyz[m * 4 + c] : 0.f;
    }
}

// Layer epilogue shared by the trunk layers and the two head hidden layers: v = relu(acc + bias);
//   TO_LDS : write v into the LDS tile (next layer's input)
//   SAVE   : store v (dW operand) and its 1-bit ReLU mask (backward chain) to the workspaces
//   DOT    : accumulate sum_n v[n] * w2[n] per sample (the 1-wide sigma output layer, lane-local)
// A lane holds 16 of the 32 columns of its row per tile (the other 16 sit in lane ^ 32), so the mask halves
// are combined with one cross-lane move.
template <int WIDTH, int NI, int MI, bool TO_LDS, bool SAVE, bool DOT>
__device__ __forceinline__ void layer_epilogue(f32x16 (&acc)[NI][MI], const float* __restrict__ bias, const float* __restrict__ w2,
                                               float* X, int xw, float* __restrict__ save, unsigned int* __restrict__ msave,
                                               float (&dot)[MI], int mrow0, int ncol0, long long row0, long long total, int lane) {
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) dot[mi] = 0.f;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int m = mrow0 + mi * 32 + r;
            const bool ok = row0 + m < total;
            unsigned bits = 0u;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n4 = ncol0 + ni * 32 + 8 * q + 4 * h;
                const f32x4 bb = *reinterpret_cast<const f32x4*>(bias + n4);
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = fmaxf(acc[ni][mi][4 * q + e] + bb[e], 0.f);
                    if (SAVE) bits |= (v[e] > 0.f ? 1u : 0u) << (8 * q + 4 * h + e);
                }
                if (DOT) {
                    const f32x4 ww = *reinterpret_cast<const f32x4*>(w2 + n4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) dot[mi] = fmaf(v[e], ww[e], dot[mi]);
                }
                if (TO_LDS) *reinterpret_cast<f32x4*>(&X[mcn_swz_chunk(m, n4 >> 2, xw)]) = v;
                if (SAVE && ok) *reinterpret_cast<f32x4*>(save + (size_t)(row0 + m) * WIDTH + n4) = v;
            }
            if (SAVE) {
                const unsigned w = bits | (unsigned)__shfl_xor((int)bits, 32);
                if (h == 0 && ok) msave[(size_t)(row0 + m) * (WIDTH / 32) + (ncol0 >> 5) + ni] = w;
            }
        }
}

// SH output layer + per-sample epilogue for an SH degree other than 2 (`MLP_deg`, model/net_block.py:43, 75-76; eval_sh,
// model/net_utils.py:103-191): 3 (DEG + 1)^2 outputs in one or two 32-row tiles of sh.2, each through the same [MT][33] LDS
// buffer; the colour pre-activations accumulate over the coefficients in index order, as the reference's expression does.
// (The degree-2 path of the kernel is untouched: same code, same summation order, same LDS budget.)
template <int WIDTH, bool SAVE, int DEG>
__device__ __forceinline__ void mcn_sh_head_general(const McnMlpFwdArgs& a, const float* X, float* ssh, const float* ssig, const float* sdir,
                                                    const int* saddr, const f32x4* __restrict__ pk, const float* __restrict__ prm,
                                                    long long row0, long long total, int tid, int lane, int wave) {
    using G = McnGeom<WIDTH>;
    using SM = FwdSmem<WIDTH>;
    constexpr int MT = SM::MT, XW = SM::XW, WN = G::WN, NT = SM::NT, WAVES = NT / 64, KSH = WIDTH / 8;
    constexpr int NB = (DEG + 1) * (DEG + 1), NSH = 3 * NB, TILES = NSH <= 32 ? 1 : 2, NSHP = 32 * TILES;
    static_assert(MT <= NT, "one sample per thread");
    const McnLayout& L = a.lay;
    const int r = lane & 31, h = lane >> 5;
    const int m = tid;
    const bool mine = m < MT && row0 + m < total;
    float b[MCN_NBMAX];
    float pre[3] = {0.f, 0.f, 0.f};
    if (mine) mcn_sh_basis16(DEG, sdir[m * 4 + 0], sdir[m * 4 + 1], sdir[m * 4 + 2], b);
#pragma unroll
    for (int nt = 0; nt < TILES; ++nt) {
        for (int mt = wave; mt < MT / 32; mt += WAVES) {
            f32x16 a1[1][1];
            mcn_zero<1, 1>(a1);
            mcn_gemm_seg<1, 1>(a1, X, XW, mt * 32, 0, KSH, pk + (L.fC2 >> 2) + nt * KSH * 64, lane);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = (e & 3) + 8 * (e >> 2) + 4 * h;
                ssh[(mt * 32 + r) * 33 + n] = a1[0][0][e];
            }
        }
        __syncthreads();
        if (mine) {
            float* dst = SAVE ? a.sh_save + (size_t)(row0 + m) * NSHP + 32 * nt : nullptr;
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                const int n = 32 * nt + k;                 // (compile-time: the colour and basis indices below are constants)
                float v = 0.f;
                if (n < NSH) {
                    v = ssh[m * 33 + k] + prm[L.pBc2 + n];
                    pre[n / NB] += b[n % NB] * v;
                }
                if (SAVE) dst[k] = v;
            }
        }
        __syncthreads();
    }
    if (mine) {
        float sigma = prm[L.pBs2];
#pragma unroll
        for (int w = 0; w < WN; ++w) sigma += ssig[w * MT + m];
        f32x4 o;
        o[0] = sigma;
#pragma unroll
        for (int c = 0; c < 3; ++c) o[1 + c] = 1.0f / (1.0f + expf(-pre[c]));
        *reinterpret_cast<f32x4*>(a.out + (size_t)saddr[m] * 4) = o;
    }
}

template <int WIDTH, bool SAVE>
__global__ __launch_bounds__(McnGeom<WIDTH>::WN * McnGeom<WIDTH>::WM * 64, 2) void mlp_fwd_kernel(McnMlpFwdArgs a) {
    using G = McnGeom<WIDTH>;
    using SM = FwdSmem<WIDTH>;
    constexpr int MT = SM::MT, XW = SM::XW, NI = G::NI, MI = G::MI, WN = G::WN, NT = SM::NT, WAVES = NT / 64;
    constexpr int KSH = WIDTH / 8;      // k-steps of a hidden segment
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* X = smem + SM::oX;
    float* sxyz = smem + SM::oXyz;
    float* sdir = smem + SM::oDir;
    float* ssig = smem + SM::oSig;
    float* ssh = smem + SM::oSh;
    int* saddr = reinterpret_cast<int*>(smem + SM::oAddr);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave % WN, wm = wave / WN;
    const int r = lane & 31, h = lane >> 5;
    const int mrow0 = wm * MI * 32;
    const int ncol0 = wn * NI * 32;
    const long long total = a.count ? (long long)min(*a.count, a.max_rows) : (long long)a.n_rays * a.S;
    const long long row0 = (long long)blockIdx.x * MT;
    if (row0 >= total) return;
    const McnLayout& L = a.lay;
    const float* __restrict__ prm = a.params;
    const f32x4* __restrict__ pk = reinterpret_cast<const f32x4*>(a.packed);

    // ---- per-sample setup: position, direction, output address
    for (int m = tid; m < MT; m += NT) {
        const long long g = row0 + m;
        float x = 0.f, y = 0.f, z = 0.f, dx = 0.f, dy = 0.f, dz = 1.f, zv = 0.f;
        int addr = -1;
        if (g < total) {
            int ray, j;
            if (a.idx) { const int2 rj = a.idx[g]; ray = rj.x; j = rj.y; }
            else { ray = (int)(g / a.S); j = (int)(g - (long long)ray * a.S); }
            zv = a.zgrid[j];
            if (a.jitter) zv = __fadd_rn(zv, a.jitter[ray]);
            dx = a.rays_d[ray * 3 + 0]; dy = a.rays_d[ray * 3 + 1]; dz = a.rays_d[ray * 3 + 2];
            // o + d*z with separate roundings, as the reference's broadcasted mul then add (mc_nerf.py:602)
            x = __fadd_rn(a.rays_o[ray * 3 + 0], __fmul_rn(dx, zv));
            y = __fadd_rn(a.rays_o[ray * 3 + 1], __fmul_rn(dy, zv));
            z = __fadd_rn(a.rays_o[ray * 3 + 2], __fmul_rn(dz, zv));
            addr = ray * a.S + j;
        }
        sxyz[m * 4 + 0] = x; sxyz[m * 4 + 1] = y; sxyz[m * 4 + 2] = z; sxyz[m * 4 + 3] = zv;
        sdir[m * 4 + 0] = dx; sdir[m * 4 + 1] = dy; sdir[m * 4 + 2] = dz; sdir[m * 4 + 3] = 0.f;
        saddr[m] = addr;
    }
    __syncthreads();
    write_encoding<MT, XW>(X, sxyz, a.barf_w, tid, NT, a.enc_in, row0, total, L.nfreq);
    __syncthreads();
    if (SAVE) {   // encoded inputs are the X operand of dW for layer 0 and the skip layer
        for (int it = tid; it < MT * 16; it += NT) {
            const int m = it >> 4, ch = it & 15;
            if (row0 + m < total)
                *reinterpret_cast<f32x4*>(a.enc_save + (size_t)(row0 + m) * MCN_ENCP + ch * 4) =
                    *reinterpret_cast<const f32x4*>(&X[mcn_swz_chunk(m, ch, XW)]);
        }
    }

    f32x16 acc[NI][MI];
    // ---- trunk
    for (int l = 0; l < L.depth; ++l) {
        mcn_zero<NI, MI>(acc);
        if (l == 0) {
            mcn_gemm_seg<NI, MI>(acc, X, XW, mrow0, 0, MCN_ENCP / 8, pk + (L.fEnc0 >> 2) + (wn * NI) * (MCN_ENCP / 8) * 64, lane);
        } else {
            mcn_gemm_seg<NI, MI>(acc, X, XW, mrow0, 0, KSH, pk + (L.fH[l] >> 2) + (wn * NI) * KSH * 64, lane);
            if ((L.skip_mask >> l) & 1u) {
                __syncthreads();                       // everyone finished reading h from X
                write_encoding<MT, XW>(X, sxyz, a.barf_w, tid, NT, a.enc_in, row0, total, L.nfreq);
                __syncthreads();
                mcn_gemm_seg<NI, MI>(acc, X, XW, mrow0, 0, MCN_ENCP / 8, pk + (L.fEncS[l] >> 2) + (wn * NI) * (MCN_ENCP / 8) * 64, lane);
            }
        }
        __syncthreads();
        float unused[MI];
        layer_epilogue<WIDTH, NI, MI, true, SAVE, false>(acc, prm + L.pB[l], nullptr, X, XW,
            SAVE ? a.act_save + (size_t)l * a.act_stride : nullptr,
            SAVE ? a.mask_save + (size_t)l * (a.act_stride / 32) : nullptr, unused, mrow0, ncol0, row0, total, lane);
        __syncthreads();
    }

    // ---- sigma head: hidden layer on MFMA, the 1-wide output layer lane-local on the VALU
    {
        mcn_zero<NI, MI>(acc);
        mcn_gemm_seg<NI, MI>(acc, X, XW, mrow0, 0, KSH, pk + (L.fS1 >> 2) + (wn * NI) * KSH * 64, lane);
        float s[MI];
        layer_epilogue<WIDTH, NI, MI, false, SAVE, true>(acc, prm + L.pBs1, prm + L.pWs2, X, XW,
            SAVE ? a.act_save + (size_t)L.depth * a.act_stride : nullptr,
            SAVE ? a.mask_save + (size_t)L.depth * (a.act_stride / 32) : nullptr, s, mrow0, ncol0, row0, total, lane);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            s[mi] += __shfl_xor(s[mi], 32);
            if (h == 0) ssig[wn * MT + mrow0 + mi * 32 + r] = s[mi];
        }
    }
    // ---- SH head hidden layer (reads the same trunk output still resident in X)
    {
        mcn_zero<NI, MI>(acc);
        mcn_gemm_seg<NI, MI>(acc, X, XW, mrow0, 0, KSH, pk + (L.fC1 >> 2) + (wn * NI) * KSH * 64, lane);
        __syncthreads();
        float unused[MI];
        layer_epilogue<WIDTH, NI, MI, true, SAVE, false>(acc, prm + L.pBc1, nullptr, X, XW,
            SAVE ? a.act_save + (size_t)(L.depth + 1) * a.act_stride : nullptr,
            SAVE ? a.mask_save + (size_t)(L.depth + 1) * (a.act_stride / 32) : nullptr, unused, mrow0, ncol0, row0, total, lane);
        __syncthreads();
    }
    if (L.sh_deg != 2) {          // (block-uniform) general SH degree: templated on the degree so that every index is a constant
        switch (L.sh_deg) {
            case 0: mcn_sh_head_general<WIDTH, SAVE, 0>(a, X, ssh, ssig, sdir, saddr, pk, prm, row0, total, tid, lane, wave); break;
            case 1: mcn_sh_head_general<WIDTH, SAVE, 1>(a, X, ssh, ssig, sdir, saddr, pk, prm, row0, total, tid, lane, wave); break;
            default: mcn_sh_head_general<WIDTH, SAVE, 3>(a, X, ssh, ssig, sdir, saddr, pk, prm, row0, total, tid, lane, wave); break;
        }
        return;
    }
    // ---- SH output layer (27 -> 32 padded outputs): one 32-row m-tile per wave
    for (int mt = wave; mt < MT / 32; mt += WAVES) {
        f32x16 a1[1][1];
        mcn_zero<1, 1>(a1);
        mcn_gemm_seg<1, 1>(a1, X, XW, mt * 32, 0, KSH, pk + (L.fC2 >> 2), lane);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int n = (e & 3) + 8 * (e >> 2) + 4 * h;
            ssh[(mt * 32 + r) * 33 + n] = a1[0][0][e];
        }
    }
    __syncthreads();
    // ---- per-sample epilogue: sigma, SH colour, sigmoid
    for (int m = tid; m < MT; m += NT) {
        const long long g = row0 + m;
        if (g >= total) continue;
        float sigma = prm[L.pBs2];
#pragma unroll
        for (int w = 0; w < WN; ++w) sigma += ssig[w * MT + m];
        float sh[MCN_NSH];
#pragma unroll
        for (int i = 0; i < MCN_NSH; ++i) sh[i] = ssh[m * 33 + i] + prm[L.pBc2 + i];
        float b[9];
        mcn_sh_basis(sdir[m * 4 + 0], sdir[m * 4 + 1], sdir[m * 4 + 2], b);
        f32x4 o;
        o[0] = sigma;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float pre = b[0] * sh[9 * c];
#pragma unroll
            for (int i = 1; i < 9; ++i) pre += b[i] * sh[9 * c + i];
            o[1 + c] = 1.0f / (1.0f + expf(-pre));
        }
        *reinterpret_cast<f32x4*>(a.out + (size_t)saddr[m] * 4) = o;
        if (SAVE) {
            float* dst = a.sh_save + (size_t)g * MCN_NSHP;             // 128-byte row: 7 vector stores (column 27 is padding)
#pragma unroll
            for (int i = 0; i < MCN_NSH + 1; i += 4) {
                f32x4 v4;
#pragma unroll
                for (int e = 0; e < 4; ++e) v4[e] = i + e < MCN_NSH ? sh[i + e] : 0.f;
                *reinterpret_cast<f32x4*>(dst + i) = v4;
            }
        }
    }
}

template <int WIDTH>
static hipError_t launch_fwd(const McnMlpFwdArgs& a, long long max_rows, hipStream_t st) {
    using SM = FwdSmem<WIDTH>;
    const int grid = (int)((max_rows + SM::MT - 1) / SM::MT);
    if (grid <= 0) return hipSuccess;
    const bool save = a.act_save != nullptr;
    auto kern = save ? mlp_fwd_kernel<WIDTH, true> : mlp_fwd_kernel<WIDTH, false>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SM::bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(SM::NT), SM::bytes, st, a);
    return hipGetLastError();
}

hipError_t mcn_launch_mlp_fwd(const McnMlpFwdArgs& a, hipStream_t st) {
    const long long max_rows = a.count ? (long long)a.max_rows : (long long)a.n_rays * a.S;
    switch (a.lay.width) {
        case 256: return launch_fwd<256>(a, max_rows, st);
        case 128: return launch_fwd<128>(a, max_rows, st);
        case 64:  return launch_fwd<64>(a, max_rows, st);
        case 32:  return launch_fwd<32>(a, max_rows, st);
    }
    return hipErrorInvalidValue;
}

int mcn_mlp_tile_rows(int width) {
    switch (width) {
        case 256: return FwdSmem<256>::MT;
        case 128: return FwdSmem<128>::MT;
        case 64:  return FwdSmem<64>::MT;
        case 32:  return FwdSmem<32>::MT;
    }
    return 0;
}
