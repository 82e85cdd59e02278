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

// Store waves (training forward / backward chains of the wider nets): the workgroup gets 4 extra waves that never
// load weights and do nothing but copy the finished LDS tiles (and the ReLU bit masks) to the HBM workspaces.  On
// gfx950 loads and stores retire through ONE in-order vmcnt, so a wave that stores a tile cannot consume a later
// weight load before the stores are acknowledged; with the stores on other waves the MFMA waves never wait for the
// ~35 GB workspace drain.  One such workgroup per CU (MCN_HELP_MIN_WIDTH = narrowest net that uses them; 0 = off).
#ifndef MCN_HELP_MIN_WIDTH
#define MCN_HELP_MIN_WIDTH 0
#endif
#ifndef MCN_HELP_THREADS
#define MCN_HELP_THREADS 256        // 256: one workgroup (4 MFMA + 4 store waves) per CU at 256 registers; 128: two per CU at 168
#endif
#define MCN_HELP_WGS (MCN_HELP_THREADS <= 128 ? 2 : 1)
#ifndef MCN_HELP_PRIO
#define MCN_HELP_PRIO 3
#endif
// Tile geometry of the split-f16 forward / backward chains (defaults = McnGeom).  MCN_H_WM256 = 2 gives the 256-wide
// net 128-row tiles on 8 waves, one workgroup per CU: the two waves that share an output slice fetch the same packed
// weight fragments, so the L2 -> CU weight stream per sample halves.
// workspace rows are written once and read by a later kernel: non-temporal stores keep them from evicting the packed
// weights (re-read by every workgroup) from the L2.  Only the row-coalesced tile copies: on the lane-local 32-byte
// pieces of the fp32 epilogues non-temporal stores lose the L2's write combining and run 5 % slower.
#ifndef MCN_NT_STORES
#define MCN_NT_STORES 1
#endif
#ifndef MCN_WGS128
#define MCN_WGS128 3
#endif
#ifndef MCN_H_WM256
#define MCN_H_WM256 1
#endif
template <int WIDTH> struct McnGeomH : McnGeom<WIDTH> { static constexpr int WGS = 2, WGS_BWD = 2; };
// 128-wide nets: the backward chain fits 168 registers, i.e. three workgroups (12 waves) per CU
template <> struct McnGeomH<128> : McnGeom<128> { static constexpr int WGS = MCN_WGS128, WGS_BWD = 3; };
template <> struct McnGeomH<256> { static constexpr int WN = 4, NI = 2, WM = MCN_H_WM256, MI = 2, WGS = 2 / MCN_H_WM256, WGS_BWD = WGS; };
#ifndef MCN_GEMM_UNROLL     // k-loop unrolling of mcn_gemm_seg_h: 0 = compiler's choice (full), 1 = rolled, n = by n
#define MCN_GEMM_UNROLL 0
#endif
#ifndef MCN_COPY_MODE
#define MCN_COPY_MODE 2
#endif
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
struct McnNoSide { __device__ __forceinline__ void operator()(int) const {} };

//   side(ks) : optional per-k-step side job issued with the prefetches (used to trickle the previous layer's
//              workspace rows out to HBM while the MFMAs run, instead of one burst per layer)
//   pre()    : optional job issued once, AFTER the first two k-steps' weight loads are in flight (stores issued
//              there do not sit in front of those loads in the in-order vmcnt queue)
struct McnNoPre { __device__ __forceinline__ void operator()() const {} };
template <int XW, int NI, int MI, class Side = McnNoSide, class Pre = McnNoPre>
__device__ __forceinline__ void mcn_gemm_seg_h(f32x16 (&acc)[NI][MI], const _Float16* Xh, const _Float16* Xl, int mrow0,
                                               int kchunk0, int KS16, const h8* __restrict__ P, int lane, Side side = Side(), Pre pre = Pre()) {
#ifdef ABL_NOGEMM      // (ablation: no matrix work at all -- what do the workspace streams alone cost?)
    pre();
    for (int ks = 0; ks < KS16; ++ks) side(ks);
    return;
#endif
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
    __builtin_amdgcn_sched_barrier(0);
    pre();
    __builtin_amdgcn_sched_barrier(0);
#if MCN_GEMM_UNROLL == 1
#pragma clang loop unroll(disable)      // (KS16 is a constant at every call site: hipcc otherwise unrolls all k-steps,
#elif MCN_GEMM_UNROLL > 1               //  hoists the weight loads far ahead and spills inside the MFMA stream)
#pragma clang loop unroll_count(MCN_GEMM_UNROLL)
#endif
    for (int ks = 0; ks < KS16; ++ks) {
        h8 ach[NI], acl[NI], bh[MI], bl[MI];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) { ach[ni] = a0h[ni]; acl[ni] = a0l[ni]; a0h[ni] = a1h[ni]; a0l[ni] = a1l[ni]; }
        if (ks + 2 < KS16) {
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) { a1h[ni] = pa[ni][256]; a1l[ni] = pa[ni][320]; }
        }
#ifndef ABL_W        // (ablation: keep re-reading the same fragments = L1 hits)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) pa[ni] += 128;
#endif
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
        side(ks);
        __builtin_amdgcn_sched_barrier(0);       // keep the prefetches above this step's MFMAs
#if defined(MCN_MFMA_ORDER) && MCN_MFMA_ORDER == 1
        // the three partial products of one accumulator issued NI*MI MFMAs apart (no back-to-back dependent MFMAs)
#pragma unroll
        for (int part = 0; part < 3; ++part)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(part == 2 ? acl[ni] : ach[ni], part == 1 ? bl[mi] : bh[mi], acc[ni][mi], 0, 0, 0);
#else
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ach[ni], bh[mi], acc[ni][mi], 0, 0, 0);
#ifndef ABL_MFMA1   // (ablation: one MFMA per product, the lo operands folded in by cheap VALU so that their loads stay)
                acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ach[ni], bl[mi], acc[ni][mi], 0, 0, 0);
                acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(acl[ni], bh[mi], acc[ni][mi], 0, 0, 0);
#else
                acc[ni][mi][0] += (float)bl[mi][0] + (float)acl[ni][0];
#endif
            }
#endif
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
// Cooperative, row-coalesced copy of the split LDS tile (columns 0 .. COLS-1 of every row) to a split-word
// workspace [rows][COLS]: one wave instruction stores whole contiguous rows (a lane-local store out of the MFMA
// accumulator layout would scatter 32-byte pieces over 32 rows per instruction).  The copy is MT*COLS/4/NT
// "trips" per thread; mcn_copy_tile_step does the share of k-step ks of a KS-step GEMM.
template <int MT, int XW, int COLS, int NT>
__device__ __forceinline__ void mcn_copy_tile_trip(const _Float16* Xh, const _Float16* Xl, float* __restrict__ dst,
                                                   long long row0, long long total, int tid, int k) {
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    constexpr int C4 = COLS / 4;
    const int it = tid + k * NT;
    const int m = it / C4, n4 = 4 * (it - m * C4);
    const int o = mcn_hoff<XW>(m, n4 >> 3) + (n4 & 7);
    const u2 H = *reinterpret_cast<const u2*>(Xh + o);
    const u2 L = *reinterpret_cast<const u2*>(Xl + o);
    u32x4 w;
    w[0] = __builtin_amdgcn_perm(L[0], H[0], 0x05040100u);
    w[1] = __builtin_amdgcn_perm(L[0], H[0], 0x07060302u);
    w[2] = __builtin_amdgcn_perm(L[1], H[1], 0x05040100u);
    w[3] = __builtin_amdgcn_perm(L[1], H[1], 0x07060302u);
#ifdef ABL_NOSTORE   // (ablation: LDS reads and packs only)
    if (total < 0) *reinterpret_cast<u32x4*>(dst + (size_t)(row0 + m) * COLS + n4) = w;
#elif defined(ABL_NOHBM)     // (ablation: every workgroup writes the same 64 rows = stores issue but stay in L2)
    if (row0 + m < total) *reinterpret_cast<u32x4*>(dst + (size_t)(m) * COLS + n4) = w;
#elif defined(ABL_NT)
    if (row0 + m < total) __builtin_nontemporal_store(w, reinterpret_cast<u32x4*>(dst + (size_t)(row0 + m) * COLS + n4));
#else
    if (row0 + m < total) *reinterpret_cast<u32x4*>(dst + (size_t)(row0 + m) * COLS + n4) = w;
#endif
}
template <int MT, int XW, int COLS, int NT, int U = 4>
__device__ __forceinline__ void mcn_copy_tile_words(const _Float16* Xh, const _Float16* Xl, float* __restrict__ dst,
                                                    long long row0, long long total, int tid) {
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    constexpr int C4 = COLS / 4, TRIPS = MT * C4 / NT, UU = TRIPS < U ? TRIPS : U;
    static_assert((MT * C4) % NT == 0 && TRIPS % UU == 0, "tile copy shape");
    if (row0 + MT <= total) {
        // whole tile inside the batch (all but the last workgroup): unpredicated, UU trips in flight -- the LDS reads of
        // a group are issued back to back and their latency is paid once per group, not once per row
#pragma unroll 1
        for (int k0 = 0; k0 < TRIPS; k0 += UU) {
            u2 H[UU], L[UU];
#pragma unroll
            for (int k = 0; k < UU; ++k) {
                const int it = tid + (k0 + k) * NT;
                const int m = it / C4, n4 = 4 * (it - m * C4);
                const int o = mcn_hoff<XW>(m, n4 >> 3) + (n4 & 7);
                H[k] = *reinterpret_cast<const u2*>(Xh + o);
                L[k] = *reinterpret_cast<const u2*>(Xl + o);
            }
#pragma unroll
            for (int k = 0; k < UU; ++k) {
                const int it = tid + (k0 + k) * NT;
                const int m = it / C4, n4 = 4 * (it - m * C4);
                u32x4 w;
                w[0] = __builtin_amdgcn_perm(L[k][0], H[k][0], 0x05040100u);
                w[1] = __builtin_amdgcn_perm(L[k][0], H[k][0], 0x07060302u);
                w[2] = __builtin_amdgcn_perm(L[k][1], H[k][1], 0x05040100u);
                w[3] = __builtin_amdgcn_perm(L[k][1], H[k][1], 0x07060302u);
#if defined(ABL_NOSTORE)
                if (total < 0)
#endif
#if MCN_NT_STORES
                __builtin_nontemporal_store(w, reinterpret_cast<u32x4*>(dst + (size_t)(row0 + m) * COLS + n4));
#else
                *reinterpret_cast<u32x4*>(dst + (size_t)(row0 + m) * COLS + n4) = w;
#endif
            }
        }
    } else {
#pragma unroll 1
        for (int k = 0; k < TRIPS; ++k) mcn_copy_tile_trip<MT, XW, COLS, NT>(Xh, Xl, dst, row0, total, tid, k);
    }
}
// ReLU bit masks of the split LDS tile (bit = activation > 0 = its hi half is non-zero), 32 columns per word, row-major
// [rows][WIDTH / 32] as the forward's epilogue used to write them: a thread takes 4 columns (one nibble), 8 neighbouring
// lanes assemble a word with three exchanges.
template <int MT, int XW, int WIDTH, int NT>
__device__ __forceinline__ void mcn_tile_masks(const _Float16* Xh, unsigned* __restrict__ msave, long long row0, long long total, int tid) {
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    constexpr int C4 = WIDTH / 4, TRIPS = MT * C4 / NT;
    static_assert((MT * C4) % NT == 0 && C4 % 8 == 0 && NT % 8 == 0, "mask tile shape");
#pragma unroll 4
    for (int k = 0; k < TRIPS; ++k) {
        const int it = tid + k * NT;
        const int m = it / C4, c4 = it - m * C4, n4 = 4 * c4;
        const u2 H = *reinterpret_cast<const u2*>(Xh + mcn_hoff<XW>(m, n4 >> 3) + (n4 & 7));
        unsigned x = ((H[0] & 0xffffu) ? 1u : 0u) | ((H[0] >> 16) ? 2u : 0u) | ((H[1] & 0xffffu) ? 4u : 0u) | ((H[1] >> 16) ? 8u : 0u);
        x |= (unsigned)__shfl_xor((int)x, 1) << 4;        // valid on even lanes
        x |= (unsigned)__shfl_xor((int)x, 2) << 8;        // valid on lanes = 0 mod 4
        x |= (unsigned)__shfl_xor((int)x, 4) << 16;       // valid on lanes = 0 mod 8
        if ((c4 & 7) == 0 && row0 + m < total) msave[(size_t)(row0 + m) * (WIDTH / 32) + (c4 >> 3)] = x;
    }
}

template <int MT, int XW, int COLS, int NT, int KS>
__device__ __forceinline__ void mcn_copy_tile_step(const _Float16* Xh, const _Float16* Xl, float* __restrict__ dst,
                                                   long long row0, long long total, int tid, int ks) {
    [[maybe_unused]] constexpr int TRIPS = MT * (COLS / 4) / NT;
    static_assert((MT * (COLS / 4)) % NT == 0, "tile copy shape");
#if MCN_COPY_MODE == 0          // one row group per k-step
    constexpr int TPS = (TRIPS + KS - 1) / KS;
#pragma unroll
    for (int j = 0; j < TPS; ++j) {
        const int k = ks * TPS + j;
        if (k < TRIPS) mcn_copy_tile_trip<MT, XW, COLS, NT>(Xh, Xl, dst, row0, total, tid, k);
    }
#elif MCN_COPY_MODE == 1        // everything in the last two k-steps (after the GEMM's last weight load was issued)
    constexpr int HALF = (TRIPS + 1) / 2;
    if (ks >= KS - 2 || KS < 2) {
        const int k0 = (KS < 2 || ks == KS - 2) ? 0 : HALF, k1 = (KS < 2) ? TRIPS : (ks == KS - 2 ? HALF : TRIPS);
#pragma unroll 4
        for (int k = k0; k < k1; ++k) mcn_copy_tile_trip<MT, XW, COLS, NT>(Xh, Xl, dst, row0, total, tid, k);
    }
#endif                          // (mode 2: the caller copies the whole tile before the GEMM, nothing here)
}
// split words of 4 values without touching LDS
__device__ __forceinline__ u32x4 mcn_words4(const f32x4& v, float scale) {
    u32x4 w;
#pragma unroll
    for (int e = 0; e < 4; ++e) { _Float16 a, b; mcn_split(v[e] * scale, a, b); w[e] = mcn_word(a, b); }
    return w;
}
#endif
