// Shared device/host definitions for the MC-NeRF gfx950 kernels.
//
// Data layout conventions:
//  * Network parameters live in ONE flat fp32 buffer per net in the reference's state-dict order
//    (model/net_block.py:51-65): xyz_encoding_{1..D}.0.{weight,bias}, sigma.0.*, sigma.2.*,
//    sh.0.*, sh.2.*; Linear weights are [out][in] row-major.
//  * Before use the weights are re-laid ("packed") into MFMA operand-fragment order so that one
//    wave's A-operand load for (n-tile, k-step) is a single contiguous 1 KiB global_load_dwordx4.
//  * Activations of a tile of MT samples live in LDS as X[MT][XW] fp32 with a 16-byte-chunk XOR
//    swizzle (chunk ^= row & 15) so that ds_read_b128 of 32 different rows at one k is conflict-free.
//  * All GEMMs use v_mfma_f32_32x32x2_f32 (exact fp32, == fmaf chain) in the "sample on the lane"
//    orientation: D[n][m] = sum_k W[n][k] * X[m][k]; lane&31 = sample m, accumulator regs = n.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MCN_MAXD 8          // max trunk depth
#define MCN_ENC 63          // 3 + 3*2*10 encoded channels (model/net_block.py:17)
#define MCN_ENCP 64         // padded
#define MCN_NFREQ 10
#define MCN_NSH 27          // 3 * (deg+1)^2 at deg = 2 (the register-chain families; the exact-fp32 family takes L.nsh)
#define MCN_NSHP 32
#define MCN_MAXDEG 3        // SH degrees 0 .. 3 (model/net_utils.py:103-191; 3, 12, 27, 48 sh.2 outputs)
#define MCN_NBMAX 16        // (MAXDEG + 1)^2 basis functions

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Offsets (in floats) of every tensor of one CorseFine_NeRF inside (a) the flat parameter buffer and
// (b) the packed buffer.  Built on the host by mcn_make_layout(), passed to kernels by value.
// `skip` at the C ABI (include/mcnerf.h): -1 = no skip layer, 0 .. depth-1 = that layer takes [encoding | hidden]
// (model/net_block.py:55-58, 71), >= MCN_SKIP_MASK = (bit mask of such layers) << 8 -- the reference's `skips` is a list.
// The exact-fp32 kernel family handles any mask; the register-chain families (f16 / bf16 / f16x3) one skip layer.
// In the mask form the low byte may also carry the SH degree of the colour head (`MLP_deg`, model/net_block.py:43, 75-76):
// bit 7 set -> bits 4..6 = degree (0 .. 3); otherwise (and in the index forms) the degree is 2.  Degree 3: exact-fp32
// family only.
#define MCN_SKIP_MASK 256
#define MCN_TOPO_HAS_DEG 0x80
static inline int mcn_topo_deg(int skip) { return (skip >= MCN_SKIP_MASK && (skip & MCN_TOPO_HAS_DEG)) ? ((skip >> 4) & 7) : 2; }
// ... and in bits 0..3 the number of encoding frequencies + 1 (`emb_freqs_xyz`, model/net_block.py:11-18: 3 + 6 F input channels;
// 0 = the default 10).  F <= 10 (the 64-column encoded-input tiles of every kernel).
static inline int mcn_topo_nfreq(int skip) { return (skip >= MCN_SKIP_MASK && (skip & 15)) ? (skip & 15) - 1 : MCN_NFREQ; }
static inline unsigned mcn_skip_mask(int depth, int skip) {
    if (skip >= MCN_SKIP_MASK) return ((unsigned)skip >> 8) & ((1u << depth) - 1u) & ~1u;      // (layer 0 takes the encoding alone)
    return (skip > 0 && skip < depth) ? (1u << skip) : 0u;
}
// Index maps of the register-chain families, whose kernels have ONE geometry (10 frequencies = 63 encoded channels, SH degree 2 =
// 27 sh.2 rows): a net with F < 10 frequencies / a degree below 2 runs on it with its tensors scattered into that geometry -- the
// channels / rows it does not have carry zero weights (and a zero bias), so they contribute nothing forward or backward, and
// their weight gradients are dropped.  -> the column (row) of the net's own tensor, -1 = none.
//   encoded channel order (model/net_block.py:20-35): [x, y, z, per axis: sin 2^0 .. 2^(F-1), cos 2^0 .. 2^(F-1)]
static inline __host__ __device__ int mcn_enc_col(int c, int F) {
    if (F == MCN_NFREQ || c < 3) return c;
    const int a = (c - 3) / (2 * MCN_NFREQ), r = (c - 3) % (2 * MCN_NFREQ), sc = r / MCN_NFREQ, k = r % MCN_NFREQ;
    return k < F ? 3 + a * 2 * F + sc * F + k : -1;
}
//   sh.2 rows: colour-major, (deg + 1)^2 coefficients per colour (model/net_block.py:75)
static inline __host__ __device__ int mcn_sh_row(int n, int deg) {
    if (deg == 2) return n;
    const int nb = (deg + 1) * (deg + 1), c = n / 9, i = n % 9;
    return i < nb ? c * nb + i : -1;
}
// the one skip layer of a mask, -1 for none, -2 for more than one
static inline int mcn_single_skip(unsigned mask) {
    if (!mask) return -1;
    if (mask & (mask - 1)) return -2;
    int l = 0;
    while (!((mask >> l) & 1u)) ++l;
    return l;
}

struct McnLayout {
    int depth, width, skip;      // skip: the single skip layer, -1 none, -2 several (skip_mask has them all)
    unsigned skip_mask;
    int sh_deg, nb, nsh, nshp;   // SH degree, (deg + 1)^2 basis functions, 3 nb outputs of sh.2, padded to 32 / 64 (+ 1 spare column for d sigma)
    int nfreq, nenc;             // encoding frequencies F and 3 + 6 F encoded channels (<= MCN_ENC; padded to MCN_ENCP columns everywhere)
    // flat parameter buffer (reference order)
    int pW[MCN_MAXD], pB[MCN_MAXD];
    int pWs1, pBs1, pWs2, pBs2, pWc1, pBc1, pWc2, pBc2;
    int n_params;
    // packed, forward orientation: P[ntile][kstep][lane][4] = W[32*ntile + (lane&31)][8*kstep + 4*(lane>>5) + i]
    int fEnc0;               // layer 0, K = 64 (63 padded)
    int fH[MCN_MAXD];        // layers >= 1: the hidden-input segment (K = width)
    int fEncS[MCN_MAXD];     // skip layers: the encoded-input segment (K = 64)
    int fS1, fC1;            // sigma.0, sh.0
    int fC2;                 // sh.2 (N = 32, 27 padded)
    // packed, transposed orientation (for dX = W^T dY): PT[ktile][nstep][lane][4] = W[8*nstep + 4*(lane>>5) + i][32*ktile + (lane&31)]
    int bEnc0, bH[MCN_MAXD], bEncS[MCN_MAXD], bS1, bC1, bC2;
    int n_packed;
};

static inline int mcn_in_features(int depth, int width, int skip, int i) {       // (`skip` in the ABI's encoding)
    const int nenc = 3 + 6 * mcn_topo_nfreq(skip);
    if (i == 0) return nenc;
    return ((mcn_skip_mask(depth, skip) >> i) & 1u) ? width + nenc : width;
}

static inline int mcn_layer_in(const McnLayout& L, int i) {                         // input features of trunk layer i
    return i == 0 ? L.nenc : (((L.skip_mask >> i) & 1u) ? L.width + L.nenc : L.width);
}

static inline McnLayout mcn_make_layout(int depth, int width, int skip) {
    McnLayout L;
    L.depth = depth; L.width = width;
    L.skip_mask = mcn_skip_mask(depth, skip);
    L.skip = mcn_single_skip(L.skip_mask);
    L.sh_deg = mcn_topo_deg(skip);
    L.nb = (L.sh_deg + 1) * (L.sh_deg + 1);
    L.nsh = 3 * L.nb;
    L.nshp = L.nsh < 32 ? 32 : 64;
    L.nfreq = mcn_topo_nfreq(skip);
    L.nenc = 3 + 6 * L.nfreq;
    int o = 0;
    // every tensor starts on a 16-byte boundary so that float4 loads of biases / weight rows are aligned
    auto al = [&o]() { o = (o + 3) & ~3; return o; };
    for (int i = 0; i < MCN_MAXD; ++i) { L.pW[i] = L.pB[i] = 0; L.fH[i] = L.bH[i] = 0; L.fEncS[i] = L.bEncS[i] = 0; }
    for (int i = 0; i < depth; ++i) {
        L.pW[i] = al(); o += width * mcn_in_features(depth, width, skip, i);
        L.pB[i] = al(); o += width;
    }
    L.pWs1 = al(); o += width * width; L.pBs1 = al(); o += width;
    L.pWs2 = al(); o += width;         L.pBs2 = al(); o += 1;
    L.pWc1 = al(); o += width * width; L.pBc1 = al(); o += width;
    L.pWc2 = al(); o += L.nsh * width; L.pBc2 = al(); o += L.nsh;
    al();
    L.n_params = o;
    int q = 0;
    L.fEnc0 = q; q += width * MCN_ENCP;
    for (int i = 1; i < depth; ++i) { L.fH[i] = q; q += width * width; }
    {   // (one encoded-input segment per skip layer; a net without a skip layer keeps one unused slot: the packed size of the
        //  single-skip nets is what it always was)
        int any = 0;
        for (int i = 1; i < depth; ++i) if ((L.skip_mask >> i) & 1u) { L.fEncS[i] = q; q += width * MCN_ENCP; any = 1; }
        if (!any) q += width * MCN_ENCP;
    }
    L.fS1 = q; q += width * width;
    L.fC1 = q; q += width * width;
    L.fC2 = q; q += L.nshp * width;
    L.bEnc0 = q; q += width * MCN_ENCP;
    for (int i = 1; i < depth; ++i) { L.bH[i] = q; q += width * width; }
    {
        int any = 0;
        for (int i = 1; i < depth; ++i) if ((L.skip_mask >> i) & 1u) { L.bEncS[i] = q; q += width * MCN_ENCP; any = 1; }
        if (!any) q += width * MCN_ENCP;
    }
    L.bS1 = q; q += width * width;
    L.bC1 = q; q += width * width;
    L.bC2 = q; q += L.nshp * width;
    L.n_packed = q;
    return L;
}

// Tile geometry per width: a workgroup is WN (along outputs) x WM (along samples) waves; each wave owns
// NI x MI MFMA tiles of 32x32.  Tiles are 64 samples (4 waves) for the production widths so that TWO
// workgroups are resident per CU (LDS ~77 KB each at width 256): while one is in a layer epilogue
// (bias/ReLU/LDS write/activation save, barriers) the other keeps the MFMA pipe busy.
template <int WIDTH> struct McnGeom;
template <> struct McnGeom<256> { static constexpr int WN = 4, NI = 2, WM = 1, MI = 2; };
template <> struct McnGeom<128> { static constexpr int WN = 4, NI = 1, WM = 1, MI = 2; };
template <> struct McnGeom<64>  { static constexpr int WN = 2, NI = 1, WM = 2, MI = 1; };
template <> struct McnGeom<32>  { static constexpr int WN = 1, NI = 1, WM = 4, MI = 1; };

#ifdef __HIPCC__
// Swizzled float offset of element (row m, column k) in an LDS tile with XW floats per row.
__device__ __forceinline__ int mcn_swz(int m, int k, int xw) {
    return m * xw + ((((k >> 2) ^ (m & 15)) << 2) | (k & 3));
}
// Same, for a 16-byte chunk index.
__device__ __forceinline__ int mcn_swz_chunk(int m, int chunk, int xw) {
    return m * xw + ((chunk ^ (m & 15)) << 2);
}

// acc[ni][mi] += W-fragment x X-fragment over KS k-steps of 8.
//   P  : packed weights for this wave's first n-tile, float4 units, laid out [ni][KS][64 lanes]
//   X  : LDS tile; rows mrow0 + mi*32 + (lane&31); the segment starts at 16-B chunk `kchunk0`
// The weight fragments are read straight from global memory (L2-resident) one k-step ahead.
template <int NI, int MI>
__device__ __forceinline__ void mcn_gemm_seg(f32x16 (&acc)[NI][MI], const float* X, int xw, int mrow0,
                                             int kchunk0, int KS, const f32x4* __restrict__ P, int lane) {
    const int r = lane & 31, h = lane >> 5;
    // software pipeline: weight fragments two k-steps ahead (L2 latency), activation fragments one ahead (LDS)
    f32x4 a0[NI], a1[NI], b_n[MI];
    int xoff[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) xoff[mi] = (mrow0 + mi * 32 + r) * xw;
    const int sw = (mrow0 + r) & 15;            // rows of all mi tiles share (row & 15): tiles are 32 rows apart
    // one running pointer per n-tile (advanced 1 KiB per k-step, loads use small immediate offsets): indexing P
    // with the unrolled k-step would keep dozens of precomputed 64-bit addresses live
    const f32x4* pa[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        pa[ni] = P + (ni * KS) * 64 + lane;
        a0[ni] = pa[ni][0];
        a1[ni] = pa[ni][KS > 1 ? 64 : 0];
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) b_n[mi] = *reinterpret_cast<const f32x4*>(&X[xoff[mi] + (((kchunk0 + h) ^ sw) << 2)]);
    for (int ks = 0; ks < KS; ++ks) {
        f32x4 a_c[NI], b[MI];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) { a_c[ni] = a0[ni]; a0[ni] = a1[ni]; }
        if (ks + 2 < KS) {
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) a1[ni] = pa[ni][128];
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) pa[ni] += 64;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) b[mi] = b_n[mi];
        if (ks + 1 < KS) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                b_n[mi] = *reinterpret_cast<const f32x4*>(&X[xoff[mi] + (((kchunk0 + 2 * (ks + 1) + h) ^ sw) << 2)]);
        }
        // pin the pipeline: hipcc otherwise sinks the prefetches next to their consumers (vmcnt(0)/lgkmcnt(0)
        // one MFMA after each load) and every k-step exposes the L2 and LDS latency
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_c[ni][i], b[mi][i], acc[ni][mi], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int NI, int MI>
__device__ __forceinline__ void mcn_zero(f32x16 (&acc)[NI][MI]) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[ni][mi][e] = 0.f;
}

// sin and cos of an fp32 angle up to a few thousand radians (positions x 2^9), accurate to ~2e-7 absolute.
// The library sincosf spends ~1500 cycles per call on such arguments (Payne-Hanek reduction); here the reduction
// runs in fp64 -- v / 2pi, nearest integer, nearest quarter turn -- which leaves |phi| <= pi/4 exact to 1e-16, and
// the kernels are the classic fp32 minimax polynomials on that interval (~40 instructions in all).
__device__ __forceinline__ void mcn_sincos(float v, float& s, float& c) {
    const double t = (double)v * 0.15915494309189535;         // turns
    const double fr = t - __builtin_rint(t);                   // [-0.5, 0.5]
    const double q = __builtin_rint(4.0 * fr);                 // nearest quarter turn: -2 .. 2
    const float phi = (float)((fr - 0.25 * q) * 6.283185307179586);   // [-pi/4, pi/4]
    const float z = phi * phi;
    const float sp = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f) * z, phi, phi);
    const float cp = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f) * z, z,
                          fmaf(-0.5f, z, 1.0f));
    const int qi = (int)q & 3;                                 // two's complement: -1 -> 3, -2 -> 2
    const float a = (qi & 1) ? cp : sp, b = (qi & 1) ? sp : cp;
    s = (qi & 2) ? -a : a;                                     // q=0: (s, c)  1: (c, -s)  2: (-s, -c)  3: (-c, s)
    c = ((qi + 1) & 2) ? -b : b;
}

// sin / cos of 2^f v for f = 0 .. NF-1, every one within an fp32 rounding of the true value: the octave-0 pair in fp64 (the
// reduction of mcn_sincos, Taylor kernels to x^13 / x^14 on [-pi/4, pi/4]: error < 1e-15), every further octave by the double-angle
// step s' = 2 s c, c' = 1 - 2 s^2 IN FP64 -- 3 fp64 operations + 2 conversions per octave (fp64 FMAs issue at the fp32 rate on
// gfx950) instead of a 9-operation fp64 reduction + two fp32 polynomials (~40 instructions) per octave.  Error growth: the
// step doubles a phase error and turns a radius error into at most 4 x as much phase error; from 1e-16 roundings that is < 1e-12
// after nine steps, against fp32's 6e-8.  (The 16-bit modes run the same recurrence in fp32, where it costs 3e-5 at 2^9.)
// Per sample: 3 x (~35 + 9 x 5) ~ 240 instructions instead of 30 x ~40; stamps (scripts/stamps_x3.py): the encoding was 6.0 k
// cycles of a 36 k (128-wide) / 193 k (256-wide) forward pass, and as much again in the backward.
template <int NF>
__device__ __forceinline__ void mcn_sincos_octaves(float v, float (&S)[NF], float (&C)[NF]) {
    const double t = (double)v * 0.15915494309189535;         // turns
    const double fr = t - __builtin_rint(t);                   // [-0.5, 0.5]
    const double q = __builtin_rint(4.0 * fr);                 // nearest quarter turn: -2 .. 2
    const double phi = (fr - 0.25 * q) * 6.283185307179586;    // [-pi/4, pi/4]
    const double z = phi * phi;
    double sp = __builtin_fma(z, 1.6059043836821613e-10, -2.505210838544172e-08);
    sp = __builtin_fma(sp, z, 2.7557319223985893e-06);
    sp = __builtin_fma(sp, z, -1.984126984126984e-04);
    sp = __builtin_fma(sp, z, 8.333333333333333e-03);
    sp = __builtin_fma(sp, z, -1.6666666666666666e-01);
    sp = __builtin_fma(sp * z, phi, phi);
    double cp = __builtin_fma(z, -1.1470745597729725e-11, 2.08767569878681e-09);
    cp = __builtin_fma(cp, z, -2.755731922398589e-07);
    cp = __builtin_fma(cp, z, 2.48015873015873e-05);
    cp = __builtin_fma(cp, z, -1.388888888888889e-03);
    cp = __builtin_fma(cp, z, 4.1666666666666664e-02);
    cp = __builtin_fma(cp, z, -0.5);
    cp = __builtin_fma(cp, z, 1.0);
    const int qi = (int)q & 3;                                 // two's complement: -1 -> 3, -2 -> 2
    const double a = (qi & 1) ? cp : sp, b = (qi & 1) ? sp : cp;
    double s = (qi & 2) ? -a : a;                              // q=0: (s, c)  1: (c, -s)  2: (-s, -c)  3: (-c, s)
    double c = ((qi + 1) & 2) ? -b : b;
    S[0] = (float)s; C[0] = (float)c;
#pragma unroll
    for (int f = 1; f < NF; ++f) {
        const double t2 = s + s;
        const double sn = t2 * c;                              // sin 2a = 2 sin a cos a
        c = __builtin_fma(-t2, s, 1.0);                        // cos 2a = 1 - 2 sin^2 a
        s = sn;
        S[f] = (float)s; C[f] = (float)c;
    }
}

// The nine signed deg-2 SH basis factors of eval_sh (model/net_utils.py:154-169).
__device__ __forceinline__ void mcn_sh_basis(float x, float y, float z, float (&b)[9]) {
    const float C0 = 0.28209479177387814f, C1 = 0.4886025119029199f;
    const float C20 = 1.0925484305920792f, C22 = 0.31539156525252005f, C24 = 0.5462742152960396f;
    b[0] = C0;
    b[1] = -C1 * y; b[2] = C1 * z; b[3] = -C1 * x;
    b[4] = C20 * (x * y); b[5] = -C20 * (y * z);
    b[6] = C22 * (2.0f * z * z - x * x - y * y);
    b[7] = -C20 * (x * z); b[8] = C24 * (x * x - y * y);
}
// The signed SH basis factors of eval_sh up to degree 3 (model/net_utils.py:152-179; the same association as the reference's
// expressions: (C * a) * (polynomial)), zero beyond (deg + 1)^2, and their derivatives with respect to the direction.
__device__ __forceinline__ void mcn_sh_basis16(int deg, float x, float y, float z, float (&b)[MCN_NBMAX]) {
#pragma unroll
    for (int i = 0; i < MCN_NBMAX; ++i) b[i] = 0.f;
    float b9[9];
    mcn_sh_basis(x, y, z, b9);
    b[0] = b9[0];
    if (deg >= 1) { b[1] = b9[1]; b[2] = b9[2]; b[3] = b9[3]; }
    if (deg >= 2) { b[4] = b9[4]; b[5] = b9[5]; b[6] = b9[6]; b[7] = b9[7]; b[8] = b9[8]; }
    if (deg >= 3) {
        const float xx = x * x, yy = y * y, zz = z * z;
        const float C30 = -0.5900435899266435f, C31 = 2.890611442640554f, C32 = -0.4570457994644658f, C33 = 0.3731763325901154f,
                    C35 = 1.445305721320277f;
        b[9] = C30 * y * (3.f * xx - yy);
        b[10] = C31 * (x * y) * z;
        b[11] = C32 * y * (4.f * zz - xx - yy);
        b[12] = C33 * z * (2.f * zz - 3.f * xx - 3.f * yy);
        b[13] = C32 * x * (4.f * zz - xx - yy);
        b[14] = C35 * z * (xx - yy);
        b[15] = C30 * x * (xx - 3.f * yy);
    }
}
__device__ __forceinline__ void mcn_sh_dbasis16(int deg, float x, float y, float z, float (&gx)[MCN_NBMAX], float (&gy)[MCN_NBMAX], float (&gz)[MCN_NBMAX]) {
#pragma unroll
    for (int i = 0; i < MCN_NBMAX; ++i) { gx[i] = 0.f; gy[i] = 0.f; gz[i] = 0.f; }
    const float C1 = 0.4886025119029199f, C20 = 1.0925484305920792f, C22 = 0.31539156525252005f, C24 = 0.5462742152960396f;
    if (deg >= 1) { gy[1] = -C1; gz[2] = C1; gx[3] = -C1; }
    if (deg >= 2) {
        gx[4] = C20 * y; gy[4] = C20 * x;
        gy[5] = -C20 * z; gz[5] = -C20 * y;
        gx[6] = -2.f * C22 * x; gy[6] = -2.f * C22 * y; gz[6] = 4.f * C22 * z;
        gx[7] = -C20 * z; gz[7] = -C20 * x;
        gx[8] = 2.f * C24 * x; gy[8] = -2.f * C24 * y;
    }
    if (deg >= 3) {
        const float xx = x * x, yy = y * y, zz = z * z;
        const float C30 = -0.5900435899266435f, C31 = 2.890611442640554f, C32 = -0.4570457994644658f, C33 = 0.3731763325901154f,
                    C35 = 1.445305721320277f;
        gx[9] = C30 * 6.f * x * y;           gy[9] = C30 * (3.f * xx - 3.f * yy);
        gx[10] = C31 * y * z;                gy[10] = C31 * x * z;                         gz[10] = C31 * x * y;
        gx[11] = C32 * (-2.f * x * y);       gy[11] = C32 * (4.f * zz - xx - 3.f * yy);    gz[11] = C32 * 8.f * y * z;
        gx[12] = C33 * (-6.f * x * z);       gy[12] = C33 * (-6.f * y * z);                gz[12] = C33 * (6.f * zz - 3.f * xx - 3.f * yy);
        gx[13] = C32 * (4.f * zz - 3.f * xx - yy); gy[13] = C32 * (-2.f * x * y);          gz[13] = C32 * 8.f * x * z;
        gx[14] = C35 * 2.f * x * z;          gy[14] = C35 * (-2.f * y * z);                gz[14] = C35 * (xx - yy);
        gx[15] = C30 * (3.f * xx - 3.f * yy); gy[15] = C30 * (-6.f * x * y);
    }
}
#endif  // __HIPCC__
