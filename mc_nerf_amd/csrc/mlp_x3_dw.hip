// Weight / bias gradients of the NeRF MLPs, split-f16 ("f16x3") mode (mcnerf_x3.h):
//     dW[n][k] = sum_m dY[m][n] X[m][k],   db[n] = sum_m dY[m][n]
// over the fragment-major (hi plane, lo plane) workspaces written by mlp_x3_fwd.hip (X, scaled by SX) and mlp_x3_bwd.hip
// (dY, scaled by SG): dY_hi X_hi + dY_hi X_lo + dY_lo X_hi, three MFMAs per product into one fp32 accumulator.
// Replaces the dW half of autograd's addmm backward for every nn.Linear of CorseFine_NeRF (model/net_block.py:51-65).
//
// Same streaming structure as mlp16_dw.hip (one launch per net; the segments' 32-row tiles form one linear sequence cut
// into gridDim.x pieces; persistent 8-wave workgroup per CU; LDS ring of whole tiles filled by LDS-DMA with the
// non-temporal policy; both operands read with ds_read_b64_tr_b16; a block flushed with fp32 atomics when a workgroup's
// piece leaves a segment).  A tile is now 2 (N + K) / 16 pieces of 1 KiB, which leaves room for only 2 tile slots in the
// CU's 160 KiB at 256 x 256 -- so the slots are filled, consumed and freed by QUARTERS (dwx3_run).
#include "mcnerf_x3.h"
#include <cstdlib>

struct DwX3Seg {
    const char* dY; int ksn;      // fragment-major [tile][part][ksn][64][8]: N = 16 ksn columns
    const char* X;  int ksk;      // fragment-major [tile][part][ksk][64][8]: K = 16 ksk columns
    const char* X2; int ksk2;     // optional second input block (the skip layer: [hidden | encoded] in ONE pass); null / 0 otherwise
    int n_lo, n_real;             // outputs n_lo <= n < n_real are real (row n - n_lo of dW)
    int col, k_real;              // input k < K of X is real for k < k_real and lands in column col + k of the dW row
    int col2, k_real2;            // input K + k of X2: column col2 + k, real for k < k_real2
    float* dW; int ldw;
    float* db;
    int kmap, kmap2, nmap;        // index maps (mcnerf_common.h; 0 = none): encoded input columns of X / X2 through mcn_enc_col(k, F = kmap / kmap2),
                                  // sh.2 output rows through mcn_sh_row(n, nmap - 16): gradients of channels / rows the net does not have are dropped
};

constexpr int dwx3_pick(int N, int K, bool want_vn) {
    int bestG = 0, bestVN = 1, bestKT = 1;
    for (int vn = 4; vn >= 1; vn /= 2)
        for (int kt = 5; kt >= 1; --kt) {
            if (32 * vn > N || 32 * kt > K || K % (32 * kt) != 0) continue;
            const int g = (N / (32 * vn)) * (K / (32 * kt));
            if (g > MCN16_WAVES) continue;
            const bool better = g > bestG || (g == bestG && vn * kt > bestVN * bestKT) || (g == bestG && vn * kt == bestVN * bestKT && vn > bestVN);
            if (better) { bestG = g; bestVN = vn; bestKT = kt; }
        }
    return want_vn ? bestVN : bestKT;
}
constexpr int dwx3_slots(int pieces) {             // tile slots of `pieces` KiB that fit the CU's LDS (2 .. 4)
    const int s = 160 / pieces;
    return s > 4 ? 4 : (s < 2 ? 2 : s);
}

// s_waitcnt vmcnt(NX * cx + NY * cy) + s_barrier with this wave's piece counts per X / dY quarter (cx in {PWX, PWX - 1}, cy in
// {PWY, PWY - 1}; wave-uniform) as immediates
template <int NX, int NY, int PWX, int PWY>
__device__ __forceinline__ void dwx3_wait(bool full_x, bool full_y, bool steady) {
#define DWX3_W(n) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(n) : "memory")
    if (!steady) DWX3_W(0);
    else if (full_x && full_y) DWX3_W(NX * PWX + NY * PWY);
    else if (full_x) DWX3_W(NX * PWX + NY * (PWY > 0 ? PWY - 1 : 0));
    else if (full_y) DWX3_W(NX * (PWX > 0 ? PWX - 1 : 0) + NY * PWY);
    else DWX3_W(NX * (PWX > 0 ? PWX - 1 : 0) + NY * (PWY > 0 ? PWY - 1 : 0));
#undef DWX3_W
}

// tiles [t0, t1) of one segment: stream, accumulate, flush.
// A tile's four operand planes are QUARTERS of a tile slot, issued, consumed and freed in the order
//     X_lo, dY_hi, X_hi, dY_lo:   step A  acc += dY_hi X_lo  (frees X_lo)
//                                 step B  acc += dY_hi X_hi  (frees dY_hi)
//                                 step C  acc += dY_lo X_hi  (frees X_hi, dY_lo)
// so a freed quarter is refilled (tile + R) at once, one step = one third of a tile's MFMAs after it was last read: with the
// R = 2 slots that a 256 x 256 block leaves room for, 80-96 KiB are in flight per CU at any time instead of one whole tile
// (64 KiB) issued once per tile.  One counted vmcnt wait + raw barrier per step; leaves no LDS-DMA piece outstanding and every
// wave past the barrier that follows the last LDS read.
template <int N, int K1, int K2>
__device__ __forceinline__ void dwx3_run(const DwX3Seg& sg, const int t0, const int t1, const float sgs, char* smem) {
    constexpr int K = K1 + K2;
    constexpr int KSN = N / 16, KSK1 = K1 / 16, KSK2 = K2 / 16, KSK = KSK1 + KSK2, P2 = 2 * (KSN + KSK);   // 1 KiB pieces per tile
    constexpr int PWX = (KSK + MCN16_WAVES - 1) / MCN16_WAVES, PWY = (KSN + MCN16_WAVES - 1) / MCN16_WAVES;   // pieces per wave and quarter
    constexpr int VN = dwx3_pick(N, K, true), KT = dwx3_pick(N, K, false);
    constexpr int NG = N / (32 * VN), KG = K / (32 * KT), G = NG * KG, MS = MCN16_WAVES / G;
    static_assert(G >= 1 && MCN16_WAVES % G == 0, "wave tiling");
    constexpr int SLOT = P2 * 1024, R = dwx3_slots(P2);
    constexpr int oXlo = 0, oYhi = KSK * 1024, oXhi = (KSK + KSN) * 1024, oYlo = (2 * KSK + KSN) * 1024;
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) s16x4* lds_tr_ptr;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gi = wave % G, ms = wave / G;
    const int nbase = (gi % NG) * 32 * VN, kbase = (gi / NG) * 32 * KT;

    // ---- LDS-DMA pieces of this wave: piece i of a quarter = fragment wave + 8 i of that operand plane
    const unsigned lds_base = (unsigned)reinterpret_cast<size_t>((mcn16_lds_ptr_t)smem);
    const int hh = lane >> 5, mm = lane & 31;
    const int nx = (KSK > wave) ? (KSK - wave + MCN16_WAVES - 1) / MCN16_WAVES : 0;        // wave-uniform piece counts
    const int ny = (KSN > wave) ? (KSN - wave + MCN16_WAVES - 1) / MCN16_WAVES : 0;
    const bool full_x = nx == PWX, full_y = ny == PWY;
    // Source addressing: piece i of a quarter is fragment s = wave + 8 i, so (s & 1) = (wave & 1) for every piece of this wave and the
    // lane's byte offset inside a fragment, (32 hh + (mm ^ 4 (2 (s & 1) + hh))) * 16, is ONE value for all X and dY pieces; what
    // differs per piece is a wave-uniform base (scalar registers).  (A 64-bit lane pointer per piece was 16 vector registers of
    // this kernel's 256: 29 -> 17 spilled, none of either inside a tile loop.)
    const unsigned voff = (unsigned)(hh * 32 + (mm ^ (4 * (2 * (wave & 1) + hh)))) * 16;
    const char* sxl[PWX > 0 ? PWX : 1]; const char* sxh[PWX > 0 ? PWX : 1];
    const char* syl[PWY > 0 ? PWY : 1]; const char* syh[PWY > 0 ? PWY : 1];
    int stx[PWX > 0 ? PWX : 1];
#pragma unroll
    for (int i = 0; i < PWX; ++i) {
        const int f = wave + MCN16_WAVES * i;                          // fragment of [X | X2]
        const bool two = f >= KSK1;
        const int s = two ? f - KSK1 : f, ks = two ? KSK2 : KSK1;
        const char* base = (two ? sg.X2 : sg.X);
        const char* p0 = base ? base + ((size_t)t0 * 2 * ks + s) * 1024 : nullptr;
        sxh[i] = p0; sxl[i] = p0 ? p0 + (size_t)ks * 1024 : nullptr;
        stx[i] = 2 * ks * 1024;
    }
#pragma unroll
    for (int i = 0; i < PWY; ++i) {
        const int s = wave + MCN16_WAVES * i;
        const char* p0 = sg.dY + ((size_t)t0 * 2 * KSN + s) * 1024;
        syh[i] = p0; syl[i] = p0 + (size_t)KSN * 1024;
    }
    // quarter q of the NEXT tile to issue for it (each quarter keeps its own tile cursor through its source pointers)
    auto issue_x = [&](const char* (&src)[PWX > 0 ? PWX : 1], int slot, int off) {
#pragma unroll
        for (int i = 0; i < PWX; ++i) {
            if (i < nx) mcn16_dma16_nt_s(src[i], voff, lds_base + slot * SLOT + off + (wave + MCN16_WAVES * i) * 1024);
            src[i] += stx[i];
        }
    };
    auto issue_y = [&](const char* (&src)[PWY > 0 ? PWY : 1], int slot, int off) {
#pragma unroll
        for (int i = 0; i < PWY; ++i) {
            if (i < ny) mcn16_dma16_nt_s(src[i], voff, lds_base + slot * SLOT + off + (wave + MCN16_WAVES * i) * 1024);
            src[i] += 2 * KSN * 1024;
        }
    };

    f32x16 acc[VN][KT];
    mcn_zero<VN, KT>(acc);
    float bsum[VN];
#pragma unroll
    for (int t = 0; t < VN; ++t) bsum[t] = 0.f;
    const bool bias = sg.db && kbase == 0;                            // wave-uniform

    // ---- per-lane LDS offsets of the transposed reads (mlp16_dw.hip)
    const int g16 = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int hp = p & 1;
    const int cxor = 4 * (2 * (g16 & 1) + hp);
    int roff[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const int m = 16 * u + 8 * (g16 >> 1) + 4 * v + q;
            roff[u][v] = (hp * 32 + (m ^ cxor)) * 16 + 8 * (p >> 1);
        }
    const int fragA0 = (nbase / 16) + (g16 & 1);                      // first dY fragment of this lane
    const int fragB0 = (kbase / 16) + (g16 & 1);                      // first [X | X2] fragment of this lane
    auto frag = [&](const char* fp, int u) -> u32x4_t {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(fp + roff[u][0]));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(fp + roff[u][1]));
        const u32x2_t l2 = __builtin_bit_cast(u32x2_t, lo), h2 = __builtin_bit_cast(u32x2_t, hi);
        return u32x4_t{l2[0], l2[1], h2[0], h2[1]};
    };
    auto colsum = [&](const u32x4_t& v, float& sacc) {              // the 8 samples of this lane's dY column
        const f16x2_t ones = {(_Float16)1.0f, (_Float16)1.0f};
        const f16x8_t hv = __builtin_bit_cast(f16x8_t, v);
#pragma unroll
        for (int d = 0; d < 4; ++d) sacc = __builtin_amdgcn_fdot2(f16x2_t{hv[2 * d], hv[2 * d + 1]}, ones, sacc, false);
    };

    const int nt = t1 - t0;
#pragma unroll
    for (int i = 0; i < R; ++i)
        if (i < nt) { issue_x(sxl, i, oXlo); issue_y(syh, i, oYhi); issue_x(sxh, i, oXhi); issue_y(syl, i, oYlo); }
    int slot = 0;
    for (int it = 0; it < nt; ++it) {
        const bool mine = (it % MS) == ms;                            // wave-uniform: waves sharing an output tile alternate tiles
        const char* st = smem + slot * SLOT;
        const int pslot = (slot + R - 1) % R;                         // the previous tile's slot
        u32x4_t ah[2][VN], bx[2][KT];
        // ---- step A: dY_hi X_lo.  Landed: X_lo, dY_hi of this tile; every wave is past step C of the previous tile
        dwx3_wait<2 * R - 2, 2 * R - 2, PWX, PWY>(full_x, full_y, it + R - 1 < nt);
        if (it >= 1 && it - 1 + R < nt) { issue_x(sxh, pslot, oXhi); issue_y(syl, pslot, oYlo); }
        if (mine) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
#pragma unroll
                for (int t = 0; t < VN; ++t) ah[u][t] = frag(st + oYhi + (fragA0 + 2 * t) * 1024, u);
#pragma unroll
                for (int t = 0; t < KT; ++t) bx[u][t] = frag(st + oXlo + (fragB0 + 2 * t) * 1024, u);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int t = 0; t < VN; ++t) {
                    if (bias) colsum(ah[u][t], bsum[t]);
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt) acc[t][kt] = mcnx3_mfma(ah[u][t], bx[u][kt], acc[t][kt]);
                }
        }
        // ---- step B: dY_hi X_hi.  Landed: X_hi; every wave is past step A: X_lo of this slot is free
        dwx3_wait<2 * R - 2, 2 * R - 1, PWX, PWY>(full_x, full_y, it + R - 1 < nt);
        if (it + R < nt) issue_x(sxl, slot, oXlo);
        if (mine) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int t = 0; t < KT; ++t) bx[u][t] = frag(st + oXhi + (fragB0 + 2 * t) * 1024, u);
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int t = 0; t < VN; ++t)
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt) acc[t][kt] = mcnx3_mfma(ah[u][t], bx[u][kt], acc[t][kt]);
        }
        // ---- step C: dY_lo X_hi.  Landed: dY_lo; every wave is past step B: dY_hi of this slot is free
        dwx3_wait<2 * R - 1, 2 * R - 2, PWX, PWY>(full_x, full_y, it + R < nt);
        if (it + R < nt) issue_y(syh, slot, oYhi);
        if (mine) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int t = 0; t < VN; ++t) ah[u][t] = frag(st + oYlo + (fragA0 + 2 * t) * 1024, u);
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int t = 0; t < VN; ++t) {
                    if (bias) colsum(ah[u][t], bsum[t]);
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt) acc[t][kt] = mcnx3_mfma(ah[u][t], bx[u][kt], acc[t][kt]);
                }
        }
        slot = (slot + 1) % R;
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");     // every wave has finished its last LDS read: the next run may refill
    // ---- accumulators -> global (float atomics; one register = two 128-byte row segments)
    const int r = lane & 31, h = lane >> 5;
    const float inv = 1.0f / (sgs * MCNX3_SX), inv_b = 1.0f / sgs;
    if (!(sg.kmap | sg.kmap2 | sg.nmap)) {
#pragma unroll
        for (int t = 0; t < VN; ++t)
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                const int k = kbase + 32 * kt + r;
                const bool in2 = K2 > 0 && kbase + 32 * kt >= K1;
                const bool k_ok = in2 ? (k - K1 < sg.k_real2) : (k < sg.k_real);
                const int colk = in2 ? sg.col2 + (k - K1) : sg.col + k;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int n = nbase + 32 * t + (e & 3) + 8 * (e >> 2) + 4 * h;
                    if (n >= sg.n_lo && n < sg.n_real && k_ok) atomicAdd(sg.dW + (size_t)(n - sg.n_lo) * sg.ldw + colk, acc[t][kt][e] * inv);
                }
            }
        if (bias) {
#pragma unroll
            for (int t = 0; t < VN; ++t) {
                const float b = (bsum[t] + __shfl_xor(bsum[t], 32)) * inv_b;
                const int n = nbase + 32 * t + r;
                if (h == 0 && n >= sg.n_lo && n < sg.n_real) atomicAdd(sg.db + (n - sg.n_lo), b);
            }
        }
    } else {      // a net scattered into the kernel geometry (fewer frequencies / a lower SH degree: mcnerf_common.h): its own columns / rows
#pragma unroll
        for (int t = 0; t < VN; ++t)
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                const int k = kbase + 32 * kt + r;
                const bool in2 = K2 > 0 && kbase + 32 * kt >= K1;
                const int kk = in2 ? k - K1 : k, km = in2 ? sg.kmap2 : sg.kmap;
                const int kc = km ? mcn_enc_col(kk, km) : kk;         // (a net with fewer encoding frequencies: its own column, or none)
                const bool k_ok = kk < (in2 ? sg.k_real2 : sg.k_real) && kc >= 0;
                const int colk = (in2 ? sg.col2 : sg.col) + kc;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int n = nbase + 32 * t + (e & 3) + 8 * (e >> 2) + 4 * h;
                    const int nr = sg.nmap ? mcn_sh_row(n - sg.n_lo, sg.nmap - 16) : n - sg.n_lo;      // (an SH degree below 2: its own row, or none)
                    if (n >= sg.n_lo && n < sg.n_real && k_ok && nr >= 0) atomicAdd(sg.dW + (size_t)nr * sg.ldw + colk, acc[t][kt][e] * inv);
                }
            }
        if (bias) {
#pragma unroll
            for (int t = 0; t < VN; ++t) {
                const float b = (bsum[t] + __shfl_xor(bsum[t], 32)) * inv_b;
                const int n = nbase + 32 * t + r;
                const int nr = sg.nmap ? mcn_sh_row(n - sg.n_lo, sg.nmap - 16) : n - sg.n_lo;
                if (h == 0 && n >= sg.n_lo && n < sg.n_real && nr >= 0) atomicAdd(sg.db + nr, b);
            }
        }
    }
}

// ---- one launch per net ---------------------------------------------------------------------------------------------
#define DWX3_MAXSEG 15
struct DwX3Job {
    int n;
    int shape[DWX3_MAXSEG];       // 0: W x W   1: W x 64 (encoded-input columns)   2: 32 x W (sh.2 / sigma.2 rows)   3: W x (W + 64) (skip layer)
    DwX3Seg seg[DWX3_MAXSEG];
};
template <int W> struct DwX3SkipMerged { static constexpr bool value = (W == 256) || (W == 128); };

template <int W>
__global__ __launch_bounds__(64 * MCN16_WAVES) void dwx3_stream_kernel(DwX3Job job, const int* count, int rows_cap, const unsigned* gmax_bits) {
    constexpr bool SKIP_MERGED = DwX3SkipMerged<W>::value;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int rows = count ? min(*count, rows_cap) : rows_cap;
    const int ntiles = (rows + 31) / 32;
    if (ntiles <= 0) return;
    const float gmax = gmax_bits ? __uint_as_float(*gmax_bits) : 1.f;
    const float sgs = mcn16_grad_scale(gmax);
    auto pieces = [](int shape) { return shape == 0 ? 2 * W / 16 : shape == 1 ? W / 16 + MCN16_ENCKS : shape == 2 ? 2 + W / 16 : 2 * W / 16 + MCN16_ENCKS; };
    long long total = 0;
    for (int s = 0; s < job.n; ++s) total += (long long)pieces(job.shape[s]) * ntiles;
    const long long lo = total * blockIdx.x / gridDim.x, hi = total * (blockIdx.x + 1) / gridDim.x;
    long long base = 0;
    for (int s = 0; s < job.n; ++s) {
        const int shape = job.shape[s], P = pieces(shape);
        const long long a = (lo - base + P - 1) / P, e = (hi - base + P - 1) / P;
        const int t0 = (int)(a < 0 ? 0 : a > ntiles ? ntiles : a), t1 = (int)(e < 0 ? 0 : e > ntiles ? ntiles : e);
        base += (long long)P * ntiles;
        if (t0 >= t1) continue;                                        // (block-uniform)
        if (shape == 0) dwx3_run<W, W, 0>(job.seg[s], t0, t1, sgs, smem);
        else if (shape == 1) dwx3_run<W, 16 * MCN16_ENCKS, 0>(job.seg[s], t0, t1, sgs, smem);
        else if (shape == 2) dwx3_run<32, W, 0>(job.seg[s], t0, t1, sgs, smem);
        else if constexpr (SKIP_MERGED) dwx3_run<W, W, 16 * MCN16_ENCKS>(job.seg[s], t0, t1, sgs, smem);
    }
}

static int dwx3_num_cus() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}

template <int W>
static hipError_t dwx3_launch_job(const DwX3Job& job, const int* count, int rows_cap, const unsigned* gmax_bits, hipStream_t st) {
    if (rows_cap <= 0) return hipSuccess;
    const long long ntiles = (rows_cap + 31) / 32;
    long long grid = dwx3_num_cus();
    if (grid > ntiles * job.n) grid = ntiles * job.n;
    // LDS: the largest stages x stage product over the shapes of this width
    constexpr int KS = W / 16;
    constexpr int p0 = 2 * (2 * KS), p1 = 2 * (KS + MCN16_ENCKS), p2 = 2 * (2 + KS), p3 = 2 * (2 * KS + MCN16_ENCKS);
    constexpr int l0 = dwx3_slots(p0) * p0, l1 = dwx3_slots(p1) * p1, l2 = dwx3_slots(p2) * p2, l3 = DwX3SkipMerged<W>::value ? dwx3_slots(p3) * p3 : 0;
    constexpr int lmax = (l0 > l1 ? l0 : l1) > (l2 > l3 ? l2 : l3) ? (l0 > l1 ? l0 : l1) : (l2 > l3 ? l2 : l3);
    static_assert(lmax <= 160, "ring exceeds the CU's LDS");
    const size_t lds = (size_t)lmax * 1024;
    void (*kern)(DwX3Job, const int*, int, const unsigned*) = dwx3_stream_kernel<W>;
    static bool attr_set = false;                                      // (per width instantiation)
    if (lds > 64 * 1024 && !attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * MCN16_WAVES), lds, st, job, count, rows_cap, gmax_bits);
    return hipGetLastError();
}

hipError_t mcnx3_launch_dw(const Mcn16DwArgs& a, hipStream_t st) {
    const McnLayout& L = a.lay;
    const int W = L.width, D = L.depth, KS = W / 16;
    const int ME = L.nfreq == MCN_NFREQ ? 0 : L.nfreq, MS = L.sh_deg == 2 ? 0 : 16 + L.sh_deg;      // index maps (DwX3Seg)
    auto act = [&](int slot) { return reinterpret_cast<const char*>(a.act_ws) + (size_t)slot * a.slot_bytes; };
    auto dy = [&](int slot) { return reinterpret_cast<const char*>(a.dy_ws) + (size_t)slot * a.slot_bytes; };
    const char* enc = reinterpret_cast<const char*>(a.enc_ws);
    const char* dsh = reinterpret_cast<const char*>(a.dsh_ws);
    DwX3Job job;
    job.n = 0;
    auto add = [&](int shape, const DwX3Seg& s) { job.shape[job.n] = shape; job.seg[job.n] = s; ++job.n; };
    if (D + 5 > DWX3_MAXSEG) return hipErrorInvalidValue;
    const bool merged = (W == 256 || W == 128);       // (DwX3SkipMerged)
    for (int l = 0; l < D; ++l) {
        const int ldw = mcn_layer_in(L, l);
        float* dWl = a.grads + L.pW[l];
        float* dbl = a.grads + L.pB[l];
        if (l == 0)                       // encoded-input columns only
            add(1, DwX3Seg{dy(l), KS, enc, MCN16_ENCKS, nullptr, 0, 0, W, 0, MCN_ENC, 0, 0, dWl, ldw, dbl, ME, 0, 0});
        else if (l == L.skip && merged)   // [hidden | encoded] in one pass: hidden k -> column 63 + k, encoded k -> column k
            add(3, DwX3Seg{dy(l), KS, act(l - 1), KS, enc, MCN16_ENCKS, 0, W, L.nenc, W, 0, MCN_ENC, dWl, ldw, dbl, 0, ME, 0});
        else if (l == L.skip) {
            add(1, DwX3Seg{dy(l), KS, enc, MCN16_ENCKS, nullptr, 0, 0, W, 0, MCN_ENC, 0, 0, dWl, ldw, dbl, ME, 0, 0});
            add(0, DwX3Seg{dy(l), KS, act(l - 1), KS, nullptr, 0, 0, W, L.nenc, W, 0, 0, dWl, ldw, nullptr, 0, 0, 0});
        } else
            add(0, DwX3Seg{dy(l), KS, act(l - 1), KS, nullptr, 0, 0, W, 0, W, 0, 0, dWl, ldw, dbl, 0, 0, 0});
    }
    add(0, DwX3Seg{dy(D), KS, act(D - 1), KS, nullptr, 0, 0, W, 0, W, 0, 0, a.grads + L.pWs1, W, a.grads + L.pBs1, 0, 0, 0});
    add(0, DwX3Seg{dy(D + 1), KS, act(D - 1), KS, nullptr, 0, 0, W, 0, W, 0, 0, a.grads + L.pWc1, W, a.grads + L.pBc1, 0, 0, 0});
    add(2, DwX3Seg{dsh, 2, act(D + 1), KS, nullptr, 0, 0, MCN_NSH, 0, W, 0, 0, a.grads + L.pWc2, W, a.grads + L.pBc2, 0, 0, MS});
    // sigma.2 (1 x W): d sigma sits in column 27 of dsh, its input is the sigma hidden layer
    add(2, DwX3Seg{dsh, 2, act(D), KS, nullptr, 0, MCN_NSH, MCN_NSH + 1, 0, W, 0, 0, a.grads + L.pWs2, W, a.grads + L.pBs2, 0, 0, 0});
    switch (W) {
        case 256: return dwx3_launch_job<256>(job, a.count, a.rows, a.gmax_bits, st);
        case 128: return dwx3_launch_job<128>(job, a.count, a.rows, a.gmax_bits, st);
        case 64:  return dwx3_launch_job<64>(job, a.count, a.rows, a.gmax_bits, st);
        case 32:  return dwx3_launch_job<32>(job, a.count, a.rows, a.gmax_bits, st);
        default:  return hipErrorInvalidValue;
    }
}
