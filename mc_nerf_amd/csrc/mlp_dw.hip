// Weight / bias gradients of the NeRF MLPs for gfx950:  dW[n][k] = sum_m dY[m][n] X[m][k],
// db[n] = sum_m dY[m][n], over the (hundreds of thousands of) samples of one step.
//
// Replaces the dW half of autograd's addmm backward for every nn.Linear of CorseFine_NeRF
// (model/net_block.py:51-65).  The reduction runs over samples, so each workgroup takes a chunk of
// rows, keeps its full (wave-tiled) dW block in MFMA accumulators for the whole chunk and issues one
// float-atomic pass at the end (>= 512 FLOP per atomic byte, far above the atomic roofline).
// (Design notes at the kernel.)
#include "mcnerf_h.h"
#include "mcnerf_kernels.h"

struct DwSeg {
    const float* dY; int ldy;     // [rows][ldy], columns nbase.. are the outputs
    const float* X;  int ldx;     // [rows][ldx]
    int N, n_lo, n_real;          // padded output count; outputs n_lo <= n < n_real are real (row n - n_lo of dW)
    int K, k_real;                // padded / real input count
    float* dW; int ldw;           // destination (already offset to the segment's first column)
    float* db;                    // bias gradient or null
};

// cache policy of the operand stream (LDS-DMA aux bits on gfx950: 1 = sc0, 2 = nt, 16 = sc1): the rows are read once
#ifndef MCN_DW_AUX
#define MCN_DW_AUX 2
#endif
#define DW_SLAB_ROWS 16
template <int V> struct VecT;
template <> struct VecT<1> { typedef float T; };
template <> struct VecT<2> { typedef float T __attribute__((ext_vector_type(2))); };
template <> struct VecT<4> { typedef float T __attribute__((ext_vector_type(4))); };
template <int V> __device__ __forceinline__ float vget(const typename VecT<V>::T& v, int i) { return v[i]; }
template <> __device__ __forceinline__ float vget<1>(const float& v, int) { return v; }

// One workgroup = 8 waves, PERSISTENT: the grid is one workgroup per CU and each takes a contiguous
// chunk of ceil(rows / grid) rows, so the float-atomic epilogue (the full dW block, 256 KB at width 256)
// is paid once per CU instead of once per 1024 rows (the chip-wide atomic rate is only ~1.3 TB/s).
// Operands are staged through LDS in slabs of RS rows (LDS-DMA into a 3-deep ring, one raw barrier per
// slab, counted vmcnt): every dY / X row is fetched from HBM once per workgroup and shared by the 8 waves,
// instead of each wave re-fetching its fragments (4x / 2x redundant) through the L1.
//   wave tile = (32*VN) outputs x (32*KT) inputs; lane (r = lane&31, h = lane>>5) owns outputs
//   nbase + VN*r + t (interleaved, so its A fragment is ONE ds_read of VN floats) and inputs
//   kbase + 32*kt + r (contiguous per tile, so the final atomics are 128-byte row segments).
template <int N, int K>
__global__ __launch_bounds__(512) void dw_kernel(DwSeg s, const int* count, int rows_cap) {
    // N, K (padded output / input counts) are compile-time so that the slab copy is a fixed, branch-free
    // sequence (a runtime-predicated load is an exec-masked branch and serialises on vmcnt(0))
    constexpr int VN = N >= 128 ? 4 : N / 32;
    constexpr int KT = N == 32 ? (K >= 128 ? 4 : K / 32) : (K >= 64 ? 2 : 1);
    typedef typename VecT<VN>::T AV;
    constexpr int RS = DW_SLAB_ROWS;               // rows per slab (8 MFMA k-steps of 2 rows)
    // LDS ring of 3 slabs: slab s is consumed while slab s+2 is in flight
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int rows = count ? min(*count, rows_cap) : rows_cap;
    int chunk = (rows + (int)gridDim.x - 1) / (int)gridDim.x;
    chunk = (chunk + RS - 1) / RS * RS;
    const int r0 = blockIdx.x * chunk;
    if (r0 >= rows) return;
    const int r1 = min(r0 + chunk, rows);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    constexpr int NG = N / (32 * VN), KG = K / (32 * KT);
    constexpr int G = NG * KG;              // wave tiles per workgroup; the remaining factor splits the slab rows
    constexpr int MS = 8 / G;
    static_assert(G >= 1 && G <= 8 && 8 % G == 0, "wave tiling");
    constexpr int PP = RS / 2 / MS;         // row pairs per wave per slab
    static_assert(PP >= 1, "slab too small for the row split");
    const int gi = wave % G, ms = wave / G;
    const int nbase = (gi % NG) * 32 * VN, kbase = (gi / NG) * 32 * KT;
    constexpr int slab = RS * (N + K);      // floats per buffer: sY[RS][N] then sX[RS][K]
    constexpr int n4 = N / 4, k4 = K / 4;
    constexpr int tot4 = RS * (n4 + k4);    // float4 per slab
    constexpr int NP = (tot4 + 511) / 512;  // LDS-DMA pieces (wave-instructions) per wave per slab

    f32x16 acc[VN][KT];
    mcn_zero<VN, KT>(acc);
    float bsum[VN];
#pragma unroll
    for (int t = 0; t < VN; ++t) bsum[t] = 0.f;

    // Slab rows -> LDS by LDS-DMA (global_load_lds, 16 B per lane): no staging registers.  The destination of
    // one wave-instruction is wave-uniform base + lane*16, i.e. exactly the linear slab image (float4 index
    // q = tid + i*512).  Rows beyond the chunk are clamped to its last row (finite data) and neutralised by
    // zeroing the A fragment when it is read.  (Segment fields are copied to locals: a by-reference capture of
    // the kernel-argument struct makes hipcc re-read it with ordinary global loads inside the loop, and any
    // such load next to LDS-DMA drains vmcnt(0).)
    const float* const gY = s.dY;
    const float* const gX = s.X;
    const int ldy = s.ldy, ldx = s.ldx;
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
    auto piece = [=](int i, int base_row, float* buf) {
        const int q = tid + i * 512;
        if ((i + 1) * 512 <= tot4 || q < tot4) {
            const bool isY = q < RS * n4;
            const int qq = isY ? q : q - RS * n4;
            const int w4 = isY ? n4 : k4;
            const int row = qq / w4, c4 = qq - row * w4;
            const int grow = base_row + row;
            const int rc = grow < r1 ? grow : r1 - 1;
            const float* src = isY ? gY + (size_t)rc * ldy + 4 * c4 : gX + (size_t)rc * ldx + 4 * c4;
            float* dst = buf + 4 * (i * 512 + (tid & ~63));                // wave-uniform; the lane offset is implicit
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)dst, 16, 0, MCN_DW_AUX);
        }
    };
    // All VMEM operations of the main loop are these pieces, NP per slab per wave, in program order, so
    // "slab s+1 has landed" == at most NP (the pieces of slab s+2) still outstanding: counted vmcnt + raw
    // s_barrier (a __syncthreads() would drain vmcnt(0) and expose the HBM latency every slab).
    // (When tot4 is not a multiple of 512 the trailing waves issue only NP-1 pieces per slab, so the count
    // allowed in flight is per wave: `full` is wave-uniform.)
    const bool full = (tot4 % 512 == 0) || (__builtin_amdgcn_readfirstlane(tid >> 6) * 64 + (NP - 1) * 512 < tot4);
#define DW_WAIT_ASM(n) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(n) : "memory")
#define DW_WAIT_BARRIER(k) do { if (full) DW_WAIT_ASM((k) * NP); else DW_WAIT_ASM((k) * (NP - 1)); } while (0)
    float* b_cur = lds;
    float* b_nxt = lds + slab;
    float* b_fill = lds + 2 * slab;
#pragma unroll
    for (int i = 0; i < NP; ++i) piece(i, r0, b_cur);
    if (r0 + RS < r1) {
#pragma unroll
        for (int i = 0; i < NP; ++i) piece(i, r0 + RS, b_nxt);
        DW_WAIT_BARRIER(1);
    } else {
        DW_WAIT_BARRIER(0);
    }
    for (int base = r0; base < r1; base += RS) {
        const bool fill = base + 2 * RS < r1;          // workgroup-uniform
        const float* sY = b_cur;
        const float* sX = sY + RS * N;
        // operand fragments are read from LDS one row pair ahead of the MFMAs that use them
        AV a_n = *reinterpret_cast<const AV*>(sY + (2 * ms + h) * N + nbase + VN * r);
        float b_n[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) b_n[kt] = sX[(2 * ms + h) * K + kbase + 32 * kt + r];
#pragma unroll
        for (int pp = 0; pp < PP; ++pp) {              // row pair p: lanes h=0 take row 2p, h=1 row 2p+1
            const int row = 2 * (ms + pp * MS) + h;
            AV a_c = a_n;
            float b_c[KT];
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) b_c[kt] = b_n[kt];
            if (pp + 1 < PP) {
                const int rown = row + 2 * MS;
                a_n = *reinterpret_cast<const AV*>(sY + rown * N + nbase + VN * r);
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) b_n[kt] = sX[rown * K + kbase + 32 * kt + r];
            }
            if (fill) {                                // this slab's share of the DMA, spread between the MFMA groups
#pragma unroll
                for (int i = 0; i < NP; ++i)
                    if ((i * PP) / NP == pp) piece(i, base + 2 * RS, b_fill);
            }
            if (base + row >= r1) a_c = AV(0.f);
            __builtin_amdgcn_sched_barrier(0);         // keep the next pair's LDS reads ABOVE this pair's MFMAs
#pragma unroll
            for (int t = 0; t < VN; ++t) {
                const float a1 = vget<VN>(a_c, t);
                bsum[t] += a1;
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) acc[t][kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b_c[kt], acc[t][kt], 0, 0, 0);
            }
        }
        // everyone is done reading b_cur; slab base+RS (in b_nxt) has landed for every wave
        if (fill) DW_WAIT_BARRIER(1); else DW_WAIT_BARRIER(0);
        float* t = b_cur; b_cur = b_nxt; b_nxt = b_fill; b_fill = t;
    }
#undef DW_WAIT_BARRIER
#undef DW_WAIT_ASM
    // accumulators -> global (float atomics; one register = two 128-byte row segments)
#pragma unroll
    for (int t = 0; t < VN; ++t)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            const int k = kbase + 32 * kt + r;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = nbase + VN * ((e & 3) + 8 * (e >> 2) + 4 * h) + t;
                if (n >= s.n_lo && n < s.n_real && k < s.k_real) atomicAdd(s.dW + (size_t)(n - s.n_lo) * s.ldw + k, acc[t][kt][e]);
            }
        }
    if (s.db && kbase == 0) {
#pragma unroll
        for (int t = 0; t < VN; ++t) {
            const float b = bsum[t] + __shfl_xor(bsum[t], 32);
            const int n = nbase + VN * r + t;
            if (h == 0 && n >= s.n_lo && n < s.n_real) atomicAdd(s.db + (n - s.n_lo), b);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Split-f16 ("f16x3", mcnerf_h.h) variant: same persistent structure; the operands in HBM / LDS are the SPLIT
// WORDS (hi | lo << 16) written by the split-f16 forward (activations, encodings: scaled by MCN_SX) and
// backward (dY, dsh: scaled by the per-launch power of two SG derived from max|d_out|), so a wave builds its MFMA
// fragments with half-word packs only and issues three v_mfma_f32_32x32x16_f16 per 32x32x16 block.
// One slab = one MFMA k-step of 16 sample rows; LDS ring of 4 slabs; the accumulators are rescaled by
// 1 / (SG * MCN_SX) before the atomic pass.
// At 5.3x the MFMA rate this kernel is HBM-bound (2 KB of operands per sample row and W x W segment).
template <int V> struct VecU;
template <> struct VecU<1> { typedef unsigned T; };
template <> struct VecU<2> { typedef unsigned T __attribute__((ext_vector_type(2))); };
template <> struct VecU<4> { typedef unsigned T __attribute__((ext_vector_type(4))); };
template <int V> __device__ __forceinline__ unsigned uget(const typename VecU<V>::T& v, int i) { return v[i]; }
template <> __device__ __forceinline__ unsigned uget<1>(const unsigned& v, int) { return v; }

// 8 split words (8 consecutive sample rows of one column) -> the hi and lo MFMA fragments (4 packs each)
__device__ __forceinline__ void mcn_frag_from_words(const unsigned (&w)[8], h8& hi, h8& lo) {
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    u4 ph, pl;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        ph[p] = __builtin_amdgcn_perm(w[2 * p + 1], w[2 * p], 0x05040100u);     // {lo16(w[2p]), lo16(w[2p+1])}
        pl[p] = __builtin_amdgcn_perm(w[2 * p + 1], w[2 * p], 0x07060302u);     // {hi16(w[2p]), hi16(w[2p+1])}
    }
    hi = __builtin_bit_cast(h8, ph);
    lo = __builtin_bit_cast(h8, pl);
}

// Wave tile of the split-f16 kernel: (32*VN) outputs x (32*KT) inputs, chosen so that the N x K block splits into
// (up to) 8 wave tiles -- every wave then works on every slab; shapes too small for that (G < 8) let the waves
// that share a tile take alternate slabs.  Ties prefer the larger tile, then the wider A read.
constexpr int dwh_pick(int N, int K, int waves, bool want_vn) {
    int bestG = 0, bestVN = 1, bestKT = 1;
    for (int vn = 4; vn >= 1; vn /= 2)
        for (int kt = 4; kt >= 1; kt /= 2) {
            if (32 * vn > N || 32 * kt > K) continue;
            const int g = (N / (32 * vn)) * (K / (32 * kt));
            if (g > waves) continue;
            const bool better = g > bestG || (g == bestG && vn * kt > bestVN * bestKT) ||
                                (g == bestG && vn * kt == bestVN * bestKT && vn > bestVN);
            if (better) { bestG = g; bestVN = vn; bestKT = kt; }
        }
    return want_vn ? bestVN : bestKT;
}

// One 16-row slab (= one MFMA k-step) of the wave's tile: fragments straight from the split words in LDS.
template <int N, int K, int VN, int KT>
__device__ __forceinline__ void dwh_consume(f32x16 (&acc)[VN][KT], float (&bsum)[VN], const unsigned* wY, const unsigned* wX, bool bias) {
    typedef typename VecU<VN>::T UV;
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    h8 ah[VN], al[VN];
    {
        UV av[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) av[j] = *reinterpret_cast<const UV*>(wY + j * N);
#pragma unroll
        for (int t = 0; t < VN; ++t) {
            unsigned w[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) w[j] = uget<VN>(av[j], t);
            mcn_frag_from_words(w, ah[t], al[t]);
            if (bias) {      // (wave-uniform) bias gradient = column sums of dY (still scaled by sg): hi + lo in one v_dot2 per word
                const h2 ones = {(_Float16)1.0f, (_Float16)1.0f};
#pragma unroll
                for (int j = 0; j < 8; ++j) bsum[t] = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, w[j]), ones, bsum[t], false);
            }
        }
    }
    // B fragments two input tiles at a time (bounds the live registers: hipcc otherwise hoists every LDS read of
    // the slab above the first MFMA)
    constexpr int KB = KT >= 2 ? 2 : 1;
#pragma unroll
    for (int k0 = 0; k0 < KT; k0 += KB) {
        __builtin_amdgcn_sched_barrier(0);
        h8 bh[KB], bl[KB];
#pragma unroll
        for (int kk = 0; kk < KB; ++kk) {
            unsigned bv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) bv[j] = wX[j * K + 32 * (k0 + kk)];
            mcn_frag_from_words(bv, bh[kk], bl[kk]);
        }
#pragma unroll
        for (int t = 0; t < VN; ++t)
#pragma unroll
            for (int kk = 0; kk < KB; ++kk) {
                acc[t][k0 + kk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], bh[kk], acc[t][k0 + kk], 0, 0, 0);
                acc[t][k0 + kk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], bl[kk], acc[t][k0 + kk], 0, 0, 0);
                acc[t][k0 + kk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[t], bh[kk], acc[t][k0 + kk], 0, 0, 0);
            }
    }
}

// waves per workgroup (a 4-wave, 128 x 128-tile variant for the 256 x 256 segments spilled next to its 256 accumulator
// registers and was slower; those segments use dw_h_kernel_v1 below)
constexpr int dwh_waves(int, int) { return 8; }

template <int N, int K>
__global__ __launch_bounds__(64 * dwh_waves(N, K)) void dw_h_kernel(DwSeg s, const int* count, int rows_cap, const unsigned int* gmax_bits) {
    constexpr int WAVES = dwh_waves(N, K), NT = 64 * WAVES;
    constexpr int VN = dwh_pick(N, K, WAVES, true), KT = dwh_pick(N, K, WAVES, false);
    constexpr int RS = 16;                  // LDS ring of 4 slabs: slab s is consumed while slabs s+1..s+3 are in flight
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int rows = count ? min(*count, rows_cap) : rows_cap;
    int chunk = (rows + (int)gridDim.x - 1) / (int)gridDim.x;
    chunk = (chunk + RS - 1) / RS * RS;
    const int r0 = blockIdx.x * chunk;
    if (r0 >= rows) return;
    const int r1 = min(r0 + chunk, rows);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    constexpr int NG = N / (32 * VN), KG = K / (32 * KT);
    constexpr int G = NG * KG;
    constexpr int MS = WAVES / G;           // waves sharing one output tile take alternate slabs
    static_assert(G >= 1 && G <= WAVES && WAVES % G == 0, "wave tiling");
    const int gi = wave % G, ms = wave / G;
    const int nbase = (gi % NG) * 32 * VN, kbase = (gi / NG) * 32 * KT;
    constexpr int slab = RS * (N + K);
    constexpr int n4 = N / 4, k4 = K / 4;
    constexpr int tot4 = RS * (n4 + k4);
    constexpr int NP = (tot4 + NT - 1) / NT;

    const float gmax = gmax_bits ? __uint_as_float(*gmax_bits) : 1.f;
    const float sg = (gmax > 0.f && gmax < 3e38f) ? exp2f(4.f - ceilf(log2f(gmax))) : 1.f;

    f32x16 acc[VN][KT];
    mcn_zero<VN, KT>(acc);
    float bsum[VN];
#pragma unroll
    for (int t = 0; t < VN; ++t) bsum[t] = 0.f;

    // LDS-DMA pieces (see dw_kernel): the per-lane source pointers are set up once and advanced by one slab per
    // fill; only a slab that reaches past r1 (the chunk's last one) takes the clamped path.
    const float* const gY = s.dY;
    const float* const gX = s.X;
    const int ldy = s.ldy, ldx = s.ldx;
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
    // Every piece lies wholly in the dY part or wholly in the X part when both parts are multiples of NT float4
    // or the X part is the last, partial piece: a piece is then a constant row offset from its part's first
    // piece, and two running per-lane pointers (dY, X) serve all of them.  Other (small) shapes keep one pointer
    // per piece.
    constexpr int NPY = RS * n4 / NT;                             // pieces of the dY part (when regular)
    constexpr bool REG = (RS * n4) % NT == 0 && NT % n4 == 0 && NT % k4 == 0;
    constexpr int NPTR = REG ? 2 : NP;
    const float* srcp[NPTR];
    if (REG) {
        srcp[0] = gY + (size_t)(r0 + tid / n4) * ldy + 4 * (tid % n4);
        srcp[1] = gX + (size_t)(r0 + tid / k4) * ldx + 4 * (tid % k4);
    } else {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int q = tid + i * NT;
            const bool isY = q < RS * n4;
            const int qq = isY ? q : q - RS * n4;
            const int w4 = isY ? n4 : k4;
            const int row = qq / w4, c4 = qq - row * w4;
            srcp[i < NPTR ? i : 0] = isY ? gY + (size_t)(r0 + row) * ldy + 4 * c4 : gX + (size_t)(r0 + row) * ldx + 4 * c4;
        }
    }
    auto fill = [&](int base_row, float* buf) {
        if (base_row + RS <= r1) {                                   // workgroup-uniform
#pragma unroll
            for (int i = 0; i < NP; ++i)
                if ((i + 1) * NT <= tot4 || tid + i * NT < tot4) {
                    const float* src = !REG ? srcp[i < NPTR ? i : 0]
                                     : (i < NPY ? srcp[0] + (size_t)(i * (NT / n4)) * ldy : srcp[1] + (size_t)((i - NPY) * (NT / k4)) * ldx);
                    __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(buf + 4 * (i * NT + (tid & ~63))), 16, 0, MCN_DW_AUX);
                }
        } else {
            // (rare path, the chunk's last slabs: its address arithmetic is tied to this point by laundering tid,
            // otherwise hipcc hoists the loop-invariant parts of all NP pieces out of the main loop and spills them)
            int tl = tid;
            asm volatile("" : "+v"(tl));
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int q = tl + i * NT;
                if ((i + 1) * NT <= tot4 || q < tot4) {
                    const bool isY = q < RS * n4;
                    const int qq = isY ? q : q - RS * n4;
                    const int w4 = isY ? n4 : k4;
                    const int row = qq / w4, c4 = qq - row * w4;
                    const int rc = min(base_row + row, r1 - 1);
                    const float* src = isY ? gY + (size_t)rc * ldy + 4 * c4 : gX + (size_t)rc * ldx + 4 * c4;
                    __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(buf + 4 * (i * NT + (tid & ~63))), 16, 0, MCN_DW_AUX);
                }
            }
        }
        if (REG) {
            srcp[0] += RS * ldy;
            srcp[1] += RS * ldx;
        } else {
#pragma unroll
            for (int i = 0; i < NP; ++i) srcp[i < NPTR ? i : 0] += RS * ((tid + i * NT) < RS * n4 ? ldy : ldx);
        }
    };
    // k slabs may stay in flight; the trailing waves issue NP-1 pieces per slab when tot4 % NT != 0 (wave-uniform)
    const bool full = (tot4 % NT == 0) || (wave * 64 + (NP - 1) * NT < tot4);
#define DWH_WAIT_ASM(n) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(n) : "memory")
#define DWH_WAIT_BARRIER(k) do { if (full) DWH_WAIT_ASM((k) * NP); else DWH_WAIT_ASM((k) * (NP - 1)); } while (0)
    // prologue: slabs 0..2 in flight, slab 0 landed
    const int nslab = (r1 - r0 + RS - 1) / RS;
    fill(r0, lds);
    if (nslab > 1) fill(r0 + RS, lds + slab);
    if (nslab > 2) fill(r0 + 2 * RS, lds + 2 * slab);
    if (nslab > 2) DWH_WAIT_BARRIER(2); else if (nslab > 1) DWH_WAIT_BARRIER(1); else DWH_WAIT_BARRIER(0);
    const unsigned* const ldsw = reinterpret_cast<const unsigned*>(lds);
    const int offY = (8 * h) * N + nbase + VN * r;               // this lane's first A word / B word inside a slab
    const int offX = RS * N + (8 * h) * K + kbase + r;
    const bool bias = s.db && kbase == 0;                        // wave-uniform
    int cur = 0;
    for (int sidx = 0; sidx < nslab; ++sidx) {
        const int base = r0 + sidx * RS;
        if (sidx + 3 < nslab) fill(base + 3 * RS, lds + ((cur + 3) & 3) * slab);        // workgroup-uniform
        if (base + RS > r1) {
            // the chunk's last, partial slab: rows at or past r1 must contribute nothing.  The DMA clamped their
            // addresses to the last valid row (finite X); their dY words are zeroed in LDS here.
            unsigned* z = reinterpret_cast<unsigned*>(lds) + cur * slab;
            for (int it = (r1 - base) * N + tid; it < RS * N; it += NT) z[it] = 0u;
            __syncthreads();
        }
        if ((sidx % MS) == ms)                                     // wave-uniform
            dwh_consume<N, K, VN, KT>(acc, bsum, ldsw + cur * slab + offY, ldsw + cur * slab + offX, bias);
        // everyone is done reading `cur`; the next slab has landed for every wave (2 younger slabs may be in flight)
        const int left = nslab - 1 - sidx;       // slabs after this one
        if (left >= 3) DWH_WAIT_BARRIER(2); else if (left == 2) DWH_WAIT_BARRIER(1); else DWH_WAIT_BARRIER(0);
        cur = (cur + 1) & 3;
    }
#undef DWH_WAIT_BARRIER
#undef DWH_WAIT_ASM
    // (an opaque zero ties the epilogue's address arithmetic to this point: hipcc otherwise computes the output
    // addresses before the main loop and spills them across it)
    int opaque0;
    asm volatile("v_mov_b32 %0, 0" : "=v"(opaque0));
    const float inv = 1.0f / (sg * MCN_SX);
#pragma unroll
    for (int t = 0; t < VN; ++t)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            const int k = kbase + 32 * kt + r + opaque0;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = nbase + VN * ((e & 3) + 8 * (e >> 2) + 4 * h) + t;
                if (n >= s.n_lo && n < s.n_real && k < s.k_real) atomicAdd(s.dW + (size_t)(n - s.n_lo) * s.ldw + k, acc[t][kt][e] * inv);
            }
        }
    if (bias) {
#pragma unroll
        for (int t = 0; t < VN; ++t) {
            const float b = (bsum[t] + __shfl_xor(bsum[t], 32)) * (1.0f / sg);
            const int n = nbase + VN * r + t;
            if (h == 0 && n >= s.n_lo && n < s.n_real) atomicAdd(s.db + (n - s.n_lo), b);
        }
    }
}

// 256 x 256 segments keep the first formulation of the split-f16 kernel (8 waves, 128 x 64 wave tiles, addresses
// recomputed per slab): the restructured kernel below needs ~30 more registers next to the 128 accumulator
// registers and spills (measured 18.8 ms vs 17.4 ms per fine-net call).
template <int N, int K>
__global__ __launch_bounds__(512) void dw_h_kernel_v1(DwSeg s, const int* count, int rows_cap, const unsigned int* gmax_bits) {
    constexpr int VN = N >= 128 ? 4 : N / 32;
    constexpr int KT = N == 32 ? (K >= 128 ? 4 : K / 32) : (K >= 64 ? 2 : 1);
    constexpr int RS = 16;                  // LDS ring of 4 slabs: slab s is consumed while slabs s+1..s+3 are in flight
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int rows = count ? min(*count, rows_cap) : rows_cap;
    int chunk = (rows + (int)gridDim.x - 1) / (int)gridDim.x;
    chunk = (chunk + RS - 1) / RS * RS;
    const int r0 = blockIdx.x * chunk;
    if (r0 >= rows) return;
    const int r1 = min(r0 + chunk, rows);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    constexpr int NG = N / (32 * VN), KG = K / (32 * KT);
    constexpr int G = NG * KG;
    constexpr int MS = 8 / G;               // waves sharing one output tile take alternate slabs
    static_assert(G >= 1 && G <= 8 && 8 % G == 0, "wave tiling");
    const int gi = wave % G, ms = wave / G;
    const int nbase = (gi % NG) * 32 * VN, kbase = (gi / NG) * 32 * KT;
    constexpr int slab = RS * (N + K);
    constexpr int n4 = N / 4, k4 = K / 4;
    constexpr int tot4 = RS * (n4 + k4);
    constexpr int NP = (tot4 + 511) / 512;

    const float gmax = gmax_bits ? __uint_as_float(*gmax_bits) : 1.f;
    const float sg = (gmax > 0.f && gmax < 3e38f) ? exp2f(4.f - ceilf(log2f(gmax))) : 1.f;

    f32x16 acc[VN][KT];
    mcn_zero<VN, KT>(acc);
    float bsum[VN];
#pragma unroll
    for (int t = 0; t < VN; ++t) bsum[t] = 0.f;

    const float* const gY = s.dY;
    const float* const gX = s.X;
    const int ldy = s.ldy, ldx = s.ldx;
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
    auto piece = [=](int i, int base_row, float* buf) {
        const int q = tid + i * 512;
        if ((i + 1) * 512 <= tot4 || q < tot4) {
            const bool isY = q < RS * n4;
            const int qq = isY ? q : q - RS * n4;
            const int w4 = isY ? n4 : k4;
            const int row = qq / w4, c4 = qq - row * w4;
            const int grow = base_row + row;
            const int rc = grow < r1 ? grow : r1 - 1;
            const float* src = isY ? gY + (size_t)rc * ldy + 4 * c4 : gX + (size_t)rc * ldx + 4 * c4;
            float* dst = buf + 4 * (i * 512 + (tid & ~63));
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)dst, 16, 0, MCN_DW_AUX);
        }
    };
    // k slabs may stay in flight; the trailing waves issue NP-1 pieces per slab when tot4 % 512 != 0 (wave-uniform)
    const bool full = (tot4 % 512 == 0) || (__builtin_amdgcn_readfirstlane(tid >> 6) * 64 + (NP - 1) * 512 < tot4);
#define DWH_WAIT_ASM(n) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(n) : "memory")
#define DWH_WAIT_BARRIER(k) do { if (full) DWH_WAIT_ASM((k) * NP); else DWH_WAIT_ASM((k) * (NP - 1)); } while (0)
    // prologue: slabs 0..2 in flight, slab 0 landed
    int nslab = (r1 - r0 + RS - 1) / RS;
#pragma unroll
    for (int i = 0; i < NP; ++i) piece(i, r0, lds);
    if (nslab > 1) {
#pragma unroll
        for (int i = 0; i < NP; ++i) piece(i, r0 + RS, lds + slab);
    }
    if (nslab > 2) {
#pragma unroll
        for (int i = 0; i < NP; ++i) piece(i, r0 + 2 * RS, lds + 2 * slab);
    }
    if (nslab > 2) DWH_WAIT_BARRIER(2); else if (nslab > 1) DWH_WAIT_BARRIER(1); else DWH_WAIT_BARRIER(0);
    int cur = 0;
    for (int sidx = 0; sidx < nslab; ++sidx) {
        const int base = r0 + sidx * RS;
        const bool fill = sidx + 3 < nslab;                       // workgroup-uniform
        if (fill) {
#pragma unroll
            for (int i = 0; i < NP; ++i) piece(i, base + 3 * RS, lds + ((cur + 3) & 3) * slab);
        }
        if (base + RS > r1) {
            // the chunk's last, partial slab: rows at or past r1 must contribute nothing.  The DMA clamped their
            // addresses to the last valid row (finite X); their dY words are zeroed in LDS here (a per-lane select
            // in registers would cost 28 v_cndmask in EVERY slab).
            unsigned* z = reinterpret_cast<unsigned*>(lds) + cur * slab;
            for (int it = (r1 - base) * N + tid; it < RS * N; it += 512) z[it] = 0u;
            __syncthreads();
        }
        if ((sidx % MS) == ms) {                                   // wave-uniform
            const float* sY = lds + cur * slab;
            const float* sX = sY + RS * N;
            // fragments: rows 8h .. 8h+7 of the slab, column = this lane's output(s) / input(s).  The operands are
            // split words (hi | lo << 16): a fragment is 8 half-words picked from 8 rows, i.e. 4 packs per part.
            typedef typename VecU<VN>::T UV;
            const unsigned* wY = reinterpret_cast<const unsigned*>(sY);
            const unsigned* wX = reinterpret_cast<const unsigned*>(sX);
            UV av[8];
            unsigned bv[KT][8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int row = 8 * h + j;
                av[j] = *reinterpret_cast<const UV*>(wY + row * N + nbase + VN * r);
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) bv[kt][j] = wX[row * K + kbase + 32 * kt + r];
            }
            h8 ah[VN], al[VN], bh[KT], bl[KT];
#pragma unroll
            for (int t = 0; t < VN; ++t) {
                unsigned w[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) w[j] = uget<VN>(av[j], t);
                mcn_frag_from_words(w, ah[t], al[t]);
                if (s.db && kbase == 0) {                           // bias gradient = column sums of dY (still scaled by sg):
                    typedef _Float16 h2 __attribute__((ext_vector_type(2)));   // hi + lo of a word in one v_dot2
                    const h2 ones = {(_Float16)1.0f, (_Float16)1.0f};
#pragma unroll
                    for (int j = 0; j < 8; ++j) bsum[t] = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, w[j]), ones, bsum[t], false);
                }
            }
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) mcn_frag_from_words(bv[kt], bh[kt], bl[kt]);
#pragma unroll
            for (int t = 0; t < VN; ++t)
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) {
                    acc[t][kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], bh[kt], acc[t][kt], 0, 0, 0);
                    acc[t][kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], bl[kt], acc[t][kt], 0, 0, 0);
                    acc[t][kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[t], bh[kt], acc[t][kt], 0, 0, 0);
                }
        }
        // everyone is done reading `cur`; the next slab has landed for every wave (2 younger slabs may be in flight)
        const int left = nslab - 1 - sidx;       // slabs after this one
        if (left >= 3) DWH_WAIT_BARRIER(2); else if (left == 2) DWH_WAIT_BARRIER(1); else DWH_WAIT_BARRIER(0);
        cur = (cur + 1) & 3;
    }
#undef DWH_WAIT_BARRIER
#undef DWH_WAIT_ASM
    const float inv = 1.0f / (sg * MCN_SX);
#pragma unroll
    for (int t = 0; t < VN; ++t)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            const int k = kbase + 32 * kt + r;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = nbase + VN * ((e & 3) + 8 * (e >> 2) + 4 * h) + t;
                if (n >= s.n_lo && n < s.n_real && k < s.k_real) atomicAdd(s.dW + (size_t)(n - s.n_lo) * s.ldw + k, acc[t][kt][e] * inv);
            }
        }
    if (s.db && kbase == 0) {
#pragma unroll
        for (int t = 0; t < VN; ++t) {
            const float b = (bsum[t] + __shfl_xor(bsum[t], 32)) * (1.0f / sg);
            const int n = nbase + VN * r + t;
            if (h == 0 && n >= s.n_lo && n < s.n_real) atomicAdd(s.db + (n - s.n_lo), b);
        }
    }
}

static int dw_num_cus() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}

template <class Kern, class... Args>
static hipError_t dw_launch(Kern kern, int grid, int threads, size_t lds, hipStream_t st, Args... args) {
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, st, args...);
    return hipGetLastError();
}

// one segment: exact-fp32 kernel, or the split-f16 kernel for its shape (256 x 256: first formulation, see above)
template <int NN, int KK>
static hipError_t launch_seg_t(const DwSeg& s, const int* count, int rows_cap, int grid, size_t lds, hipStream_t st,
                               const unsigned int* gmax_bits, bool split16) {
    if (!split16) return dw_launch(dw_kernel<NN, KK>, grid, 512, lds, st, s, count, rows_cap);
    if constexpr (NN >= 256 && KK >= 256) return dw_launch(dw_h_kernel_v1<NN, KK>, grid, 512, lds, st, s, count, rows_cap, gmax_bits);
    else return dw_launch(dw_h_kernel<NN, KK>, grid, 64 * dwh_waves(NN, KK), lds, st, s, count, rows_cap, gmax_bits);
}

static hipError_t launch_seg(const DwSeg& s, const int* count, int rows_cap, hipStream_t st, const unsigned int* gmax_bits = nullptr, bool split16 = false) {
    if (rows_cap <= 0) return hipSuccess;
    int grid = dw_num_cus();                                   // persistent: one workgroup per CU
    const int max_wgs = (rows_cap + DW_SLAB_ROWS - 1) / DW_SLAB_ROWS;
    if (grid > max_wgs) grid = max_wgs;
    const size_t lds = (size_t)(split16 ? 4 : 3) * DW_SLAB_ROWS * (s.N + s.K) * sizeof(float);
#define DW_LAUNCH(NN, KK) return launch_seg_t<NN, KK>(s, count, rows_cap, grid, lds, st, gmax_bits, split16)
    switch (s.N * 1000 + s.K) {
        case 256256: DW_LAUNCH(256, 256);
        case 256064: DW_LAUNCH(256, 64);
        case 32256:  DW_LAUNCH(32, 256);
        case 128128: DW_LAUNCH(128, 128);
        case 128064: DW_LAUNCH(128, 64);
        case 32128:  DW_LAUNCH(32, 128);
        case 64064:  DW_LAUNCH(64, 64);
        case 32064:  DW_LAUNCH(32, 64);
        case 32032:  DW_LAUNCH(32, 32);
        default: return hipErrorInvalidValue;
    }
#undef DW_LAUNCH
}

hipError_t mcn_launch_dw(const McnDwArgs& a, hipStream_t st) {
    const McnLayout& L = a.lay;
    const int W = L.width, D = L.depth;
    const size_t AS = a.act_stride;
    auto act = [&](int slot) { return a.act_save + (size_t)slot * AS; };
    auto dy = [&](int slot) { return a.dy_save + (size_t)slot * AS; };
    hipError_t e;
    for (int l = 0; l < D; ++l) {
        const int ldw = mcn_in_features(D, W, L.skip, l);
        if (l == 0 || l == L.skip) {      // encoded-input columns
            DwSeg s = {dy(l), W, a.enc_save, MCN_ENCP, W, 0, W, MCN_ENCP, MCN_ENC, a.grads + L.pW[l], ldw, a.grads + L.pB[l]};
            if ((e = launch_seg(s, a.count, a.rows, st, a.gmax_bits, a.split16)) != hipSuccess) return e;
        }
        if (l > 0) {                      // hidden-input columns (after the 63 encoded ones at the skip layer)
            DwSeg s = {dy(l), W, act(l - 1), W, W, 0, W, W, W, a.grads + L.pW[l] + (l == L.skip ? MCN_ENC : 0), ldw,
                       l == L.skip ? nullptr : a.grads + L.pB[l]};
            if ((e = launch_seg(s, a.count, a.rows, st, a.gmax_bits, a.split16)) != hipSuccess) return e;
        }
    }
    {   // sigma.0 and sh.0 read the last trunk activation; sh.2 reads the sh hidden layer
        DwSeg s1 = {dy(D), W, act(D - 1), W, W, 0, W, W, W, a.grads + L.pWs1, W, a.grads + L.pBs1};
        if ((e = launch_seg(s1, a.count, a.rows, st, a.gmax_bits, a.split16)) != hipSuccess) return e;
        DwSeg c1 = {dy(D + 1), W, act(D - 1), W, W, 0, W, W, W, a.grads + L.pWc1, W, a.grads + L.pBc1};
        if ((e = launch_seg(c1, a.count, a.rows, st, a.gmax_bits, a.split16)) != hipSuccess) return e;
        DwSeg c2 = {a.dsh_save, MCN_NSHP, act(D + 1), W, MCN_NSHP, 0, MCN_NSH, W, W, a.grads + L.pWc2, W, a.grads + L.pBc2};
        if ((e = launch_seg(c2, a.count, a.rows, st, a.gmax_bits, a.split16)) != hipSuccess) return e;
        // sigma.2 (1 x W): d sigma sits in the spare column 27 of dsh_save, its input is the sigma hidden layer
        DwSeg s2 = {a.dsh_save, MCN_NSHP, act(D), W, MCN_NSHP, MCN_NSH, MCN_NSH + 1, W, W, a.grads + L.pWs2, W, a.grads + L.pBs2};
        if ((e = launch_seg(s2, a.count, a.rows, st, a.gmax_bits, a.split16)) != hipSuccess) return e;
    }
    return hipSuccess;
}
