// Split-f16 ("f16x3") GEMM building blocks for gfx950.
//
// Every fp32 operand x is represented as hi + lo with hi = f16(x * S), lo = f16(x * S - hi) (S a power of two
// that keeps lo a normal f16), and a product is accumulated in fp32 as hi*hi + hi*lo + lo*hi with
// v_mfma_f32_32x32x16_f16: 22 significand bits per operand, relative product error ~2^-21 (fp32 is 2^-24),
// at 3 x 32 cycles per 32x32x16 block instead of 8 x 64 cycles of the exact-fp32 MFMA (5.3x).
//   weights scale  MCN_SW = 2^8   (|w| <= 255 representable; lo stays normal down to |w| ~ 5e-4)
//   activation scale MCN_SX = 2^3 (|x| <= 8188)
// Layouts:
//   packed weights  PH[ntile][ks16][part(hi,lo)][lane][8]  = W*S [32 ntile + (lane&31)][16 ks16 + 8 (lane>>5) + j]
//   LDS tile        Xh[MT][XW] f16 followed by Xl[MT][XW] f16, 16-byte chunks XOR-swizzled by (row & SWZ)
#pragma once
#include "mcnerf_common.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

#define MCN_SW 256.0f
#define MCN_SX 8.0f

#ifdef __HIPCC__
__device__ __forceinline__ void mcn_split(float x, _Float16& hi, _Float16& lo) {
    hi = (_Float16)x;
    lo = (_Float16)(x - (float)hi);
}

// f16-element offset of 16-byte chunk c of row m in a tile with XW halves per row
template <int XW>
__device__ __forceinline__ int mcn_hoff(int m, int c) {
    constexpr int SWZ = (XW / 8 - 1) < 15 ? (XW / 8 - 1) : 15;
    return m * XW + ((c ^ (m & SWZ)) << 3);
}

// acc[ni][mi] += W-fragment x X-fragment over KS16 k-steps of 16, three MFMAs per product.
//   P : packed split weights for this wave's first n-tile, h8 units: [ni][KS16][2][64 lanes]
template <int XW, int NI, int MI>
__device__ __forceinline__ void mcn_gemm_seg_h(f32x16 (&acc)[NI][MI], const _Float16* Xh, const _Float16* Xl, int mrow0,
                                               int kchunk0, int KS16, const h8* __restrict__ P, int lane) {
    const int r = lane & 31, h = lane >> 5;
    constexpr int SWZ = (XW / 8 - 1) < 15 ? (XW / 8 - 1) : 15;
    const int sw = (mrow0 + r) & SWZ;
    int xoff[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) xoff[mi] = (mrow0 + mi * 32 + r) * XW;
    const h8* pa[NI];
    h8 a0h[NI], a0l[NI], a1h[NI], a1l[NI], bnh[MI], bnl[MI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        pa[ni] = P + (size_t)(ni * KS16) * 128 + lane;
        a0h[ni] = pa[ni][0]; a0l[ni] = pa[ni][64];
        a1h[ni] = pa[ni][KS16 > 1 ? 128 : 0]; a1l[ni] = pa[ni][KS16 > 1 ? 192 : 64];
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int o = xoff[mi] + (((kchunk0 + h) ^ sw) << 3);
        bnh[mi] = *reinterpret_cast<const h8*>(Xh + o);
        bnl[mi] = *reinterpret_cast<const h8*>(Xl + o);
    }
    for (int ks = 0; ks < KS16; ++ks) {
        h8 ach[NI], acl[NI], bh[MI], bl[MI];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) { ach[ni] = a0h[ni]; acl[ni] = a0l[ni]; a0h[ni] = a1h[ni]; a0l[ni] = a1l[ni]; }
        if (ks + 2 < KS16) {
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) { a1h[ni] = pa[ni][256]; a1l[ni] = pa[ni][320]; }
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) pa[ni] += 128;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) { bh[mi] = bnh[mi]; bl[mi] = bnl[mi]; }
        if (ks + 1 < KS16) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int o = xoff[mi] + (((kchunk0 + 2 * (ks + 1) + h) ^ sw) << 3);
                bnh[mi] = *reinterpret_cast<const h8*>(Xh + o);
                bnl[mi] = *reinterpret_cast<const h8*>(Xl + o);
            }
        }
        __builtin_amdgcn_sched_barrier(0);       // keep the prefetches above this step's MFMAs
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ach[ni], bh[mi], acc[ni][mi], 0, 0, 0);
                acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ach[ni], bl[mi], acc[ni][mi], 0, 0, 0);
                acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(acl[ni], bh[mi], acc[ni][mi], 0, 0, 0);
            }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// "Split word": the saved-operand format of the f16x3 mode: one 32-bit word per value, hi f16 in the low half,
// lo f16 in the high half (value = (hi + lo) / scale).  Same size and row-major layout as the fp32 workspaces,
// so the weight-gradient kernel builds its MFMA fragments with plain half-word packs instead of conversions.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned mcn_word(_Float16 hi, _Float16 lo) {
    return (unsigned)__builtin_bit_cast(unsigned short, hi) | ((unsigned)__builtin_bit_cast(unsigned short, lo) << 16);
}
__device__ __forceinline__ float mcn_unword(unsigned w, float inv_scale) {
    const _Float16 hi = __builtin_bit_cast(_Float16, (unsigned short)(w & 0xffffu));
    const _Float16 lo = __builtin_bit_cast(_Float16, (unsigned short)(w >> 16));
    return ((float)hi + (float)lo) * inv_scale;
}

// 4 consecutive values of row m (fp32, unscaled) -> split f16 (x scale) in the LDS tile; returns the split words
template <int XW>
__device__ __forceinline__ u32x4 mcn_store_split4(_Float16* Xh, _Float16* Xl, int m, int n4, const f32x4& v, float scale = MCN_SX) {
    h4 hi, lo;
    u32x4 w;
#pragma unroll
    for (int e = 0; e < 4; ++e) { _Float16 a, b; mcn_split(v[e] * scale, a, b); hi[e] = a; lo[e] = b; w[e] = mcn_word(a, b); }
    const int o = mcn_hoff<XW>(m, n4 >> 3) + (n4 & 7);
    *reinterpret_cast<h4*>(Xh + o) = hi;
    *reinterpret_cast<h4*>(Xl + o) = lo;
    return w;
}
// split words of 4 values without touching LDS
__device__ __forceinline__ u32x4 mcn_words4(const f32x4& v, float scale) {
    u32x4 w;
#pragma unroll
    for (int e = 0; e < 4; ++e) { _Float16 a, b; mcn_split(v[e] * scale, a, b); w[e] = mcn_word(a, b); }
    return w;
}
#endif
