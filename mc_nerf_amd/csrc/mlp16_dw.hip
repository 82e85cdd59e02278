// Weight / bias gradients of the NeRF MLPs, single-pass 16-bit MFMA mode (mcnerf_16.h):
//     dW[n][k] = sum_m dY[m][n] X[m][k],   db[n] = sum_m dY[m][n]
// over the fragment-major 16-bit workspaces written by mlp16_fwd.hip (X) and mlp16_bwd.hip (dY, scaled by SG).
// Replaces the dW half of autograd's addmm backward for every nn.Linear of CorseFine_NeRF (model/net_block.py:51-65).
//
// HBM-bound streaming kernel, ONE launch per net: the 12 (depth + 4; depth + 5 for the 32- and 64-wide test nets, whose skip layer stays two segments) GEMM segments of a net are laid end to end as one
// linear sequence of 32-row tiles weighted by their bytes, the sequence is cut into gridDim.x equal pieces and the
// persistent workgroup (8 waves, one per CU) b streams piece b.  A piece lies inside one segment or crosses one
// boundary, so a segment's dW block is flushed (fp32 float atomics from the MFMA accumulators) by ~ grid / 13 + 1
// workgroups instead of by all of them: the flush (16.8 M lane-atomics per 256x256 block and workgroup round, measured
// 58 us per segment when every workgroup flushed every segment) drops by 12x and overlaps the other workgroups'
// streaming.  A 32-row tile of both
// operands ((N + K) / 16 fragments of 1 KiB) is one stage of a 4-stage LDS ring filled by LDS-DMA (inline asm, counted
// vmcnt, one raw barrier per tile).  The contraction runs over SAMPLES, which sit on the lanes of the saved fragments,
// so both MFMA operands are read with ds_read_b64_tr_b16 (the transposing LDS read): a fragment's 16-byte chunks are
// (lane half hh, sample m) -> 8 channels, and a 4-sample x 16-channel block of a transposed read touches the chunks
// (s, hh, m .. m+3) of two fragments s.  The DMA places chunk (hh, m) of fragment s at position
// 32 hh + (m ^ 4 (2 (s & 1) + hh)), which spreads the 16 chunks of one 32-lane half over all 64 banks.
#include "mcnerf_16.h"

struct Dw16Seg {
    const char* dY; int ksn;      // fragment-major [tile][ksn][64][8]: N = 16 ksn columns
    const char* X;  int ksk;      // fragment-major [tile][ksk][64][8]: K = 16 ksk columns
    const char* X2; int ksk2;     // optional second input block, K2 = 16 ksk2 more columns (the skip layer: [hidden | encoded] in ONE
                                  // pass, so its dY is read once); null / 0 otherwise
    int n_lo, n_real;             // outputs n_lo <= n < n_real are real (row n - n_lo of dW)
    int col, k_real;              // input k < K of X is real for k < k_real and lands in column col + k of the dW row
    int col2, k_real2;            // input K + k of X2: column col2 + k, real for k < k_real2
    float* dW; int ldw;
    float* db;
    int kmap, kmap2, nmap;        // index maps (mcnerf_common.h; 0 = none): encoded input columns of X / X2 through mcn_enc_col(k, F = kmap / kmap2),
                                  // sh.2 output rows through mcn_sh_row(n, nmap - 16): gradients of channels / rows the net does not have are dropped
};

constexpr int dw16_pick(int N, int K, bool want_vn) {
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

#define DW16_STAGES 4

// tiles [t0, t1) of one segment: stream, accumulate, flush.  Leaves no LDS-DMA piece outstanding and every wave past the
// barrier that follows the last LDS read, so the next segment's run may refill the ring at once.
template <int N, int K1, int K2, bool BF>
__device__ __forceinline__ void dw16_run(const Dw16Seg& sg, const int t0, const int t1, const float sgs, const float xs, char* smem) {
    using T = Mcn16T<BF>;
    constexpr int K = K1 + K2;                                        // the X2 columns follow the X columns
    constexpr int KSN = N / 16, KSK1 = K1 / 16, KSK2 = K2 / 16, P = KSN + KSK1 + KSK2;   // 1 KiB pieces per tile
    constexpr int PW = (P + MCN16_WAVES - 1) / MCN16_WAVES;           // pieces per wave (the trailing waves may issue PW - 1)
    constexpr int VN = dw16_pick(N, K, true), KT = dw16_pick(N, K, false);
    constexpr int NG = N / (32 * VN), KG = K / (32 * KT), G = NG * KG, MS = MCN16_WAVES / G;
    static_assert(G >= 1 && MCN16_WAVES % G == 0, "wave tiling");
    constexpr int STAGE = P * 1024;
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) s16x4* lds_tr_ptr;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gi = wave % G, ms = wave / G;
    const int nbase = (gi % NG) * 32 * VN, kbase = (gi / NG) * 32 * KT;

    // ---- LDS-DMA pieces of this wave: piece pi = wave + 8 i (dY fragments first, then the X and X2 fragments)
    const unsigned lds_base = (unsigned)reinterpret_cast<size_t>((mcn16_lds_ptr_t)smem);
    const char* src[PW];
    const int hh = lane >> 5, mm = lane & 31;
#pragma unroll
    for (int i = 0; i < PW; ++i) {
        const int pi = wave + MCN16_WAVES * i;
        const int which = pi < KSN ? 0 : (pi < KSN + KSK1 ? 1 : 2);
        const int s = which == 0 ? pi : (which == 1 ? pi - KSN : pi - KSN - KSK1);
        // (every piece of this wave has (s & 1) = (wave & 1): KSN and KSK1 are even -- so the lane's byte offset inside a fragment is
        //  one value, `voff`, and a piece is a wave-uniform base: SADDR-form LDS-DMA, scalar registers instead of a lane pointer each)
        const char* base = which == 0 ? sg.dY + ((size_t)t0 * KSN + s) * 1024
                         : which == 1 ? sg.X + ((size_t)t0 * KSK1 + s) * 1024
                                      : sg.X2 + ((size_t)t0 * KSK2 + s) * 1024;
        src[i] = base;
    }
    const unsigned voff = (unsigned)(hh * 32 + (mm ^ (4 * (2 * (wave & 1) + hh)))) * 16;
    const int np = (P % MCN16_WAVES == 0 || wave < P % MCN16_WAVES) ? PW : PW - 1;      // wave-uniform
    auto fill = [&](int stage) {
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            const int pi = wave + MCN16_WAVES * i;
            if (i < np) mcn16_dma16_nt_s(src[i], voff, lds_base + stage * STAGE + pi * 1024);     // read once: non-temporal (6.5 -> 6.06 ms per fine-net call)
            src[i] += (size_t)(pi < KSN ? KSN : (pi < KSN + KSK1 ? KSK1 : KSK2)) * 1024;
        }
    };
#define DW16_WAIT_ASM(n) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(n) : "memory")
#define DW16_WAIT(k) do { if (np == PW) DW16_WAIT_ASM((k) * PW); else DW16_WAIT_ASM((k) * (PW - 1)); } while (0)

    f32x16 acc[VN][KT];
    mcn_zero<VN, KT>(acc);
    float bsum[VN];
#pragma unroll
    for (int t = 0; t < VN; ++t) bsum[t] = 0.f;
    const bool bias = sg.db && kbase == 0;                            // wave-uniform

    // ---- per-lane LDS offsets of the transposed reads.  A fragment (n-tile tn, m-step u, half v): 16-lane group G16 = lane >> 4,
    //      group lane gl = 4 q + p: row m = 16 u + 8 (G16 >> 1) + 4 v + q, columns n0 + 4 p .. + 3 with n0 = 32 tn + 16 (G16 & 1):
    //      fragment s = 2 tn + (G16 & 1), chunk (hh' = p & 1, m), byte 8 (p >> 1) inside the chunk.
    const int g16 = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int hp = p & 1;
    const int cxor = 4 * (2 * (g16 & 1) + hp);                        // (s & 1) = (G16 & 1) for every tile
    int roff[2][2];                                                   // [u][v]: offset inside one fragment's 1 KiB piece
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const int m = 16 * u + 8 * (g16 >> 1) + 4 * v + q;
            roff[u][v] = (hp * 32 + (m ^ cxor)) * 16 + 8 * (p >> 1);
        }
    const int fragA0 = (nbase / 16) + (g16 & 1);                      // first dY fragment of this lane (n-tile 0 of the wave tile)
    const int fragB0 = KSN + (kbase / 16) + (g16 & 1);

    constexpr int AH = DW16_STAGES - 1;                              // tiles in flight ahead of the one being consumed
    int nt = t1 - t0;
#pragma unroll
    for (int i = 0; i < AH; ++i)
        if (i < nt) fill(i);
    {
        const int younger = min(nt, AH) - 1;
        if (younger >= 3) DW16_WAIT(3); else if (younger == 2) DW16_WAIT(2); else if (younger == 1) DW16_WAIT(1); else DW16_WAIT(0);
    }
    static_assert(AH <= 4, "wait ladder covers up to 3 younger tiles");
    int cur = 0;
    for (int it = 0; it < nt; ++it) {
        if (it + AH < nt) fill((cur + AH) % DW16_STAGES);
        if ((it % MS) == ms) {                                        // wave-uniform: waves sharing an output tile alternate tiles
            const char* st = smem + cur * STAGE;
            u32x4_t af[2][VN], bf[2][KT];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
#pragma unroll
                for (int t = 0; t < VN; ++t) {
                    const char* fp = st + (fragA0 + 2 * t) * 1024;
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(fp + roff[u][0]));
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(fp + roff[u][1]));
                    const u32x2_t l2 = __builtin_bit_cast(u32x2_t, lo), h2 = __builtin_bit_cast(u32x2_t, hi);
                    af[u][t] = u32x4_t{l2[0], l2[1], h2[0], h2[1]};
                }
#pragma unroll
                for (int t = 0; t < KT; ++t) {
                    const char* fp = st + (fragB0 + 2 * t) * 1024;
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(fp + roff[u][0]));
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(fp + roff[u][1]));
                    const u32x2_t l2 = __builtin_bit_cast(u32x2_t, lo), h2 = __builtin_bit_cast(u32x2_t, hi);
                    bf[u][t] = u32x4_t{l2[0], l2[1], h2[0], h2[1]};
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int t = 0; t < VN; ++t) {
                    if (bias) {          // column sums of dY: the 8 samples of this lane's column
                        if (BF) {
#pragma unroll
                            for (int d = 0; d < 4; ++d) bsum[t] += T::lo(af[u][t][d]) + T::hi(af[u][t][d]);
                        } else {
                            // (whole-vector bit cast, then pairs: hipcc 7.2 miscompiles bit_cast<f16x2>(u32x4[d]) feeding fdot2 -- it
                            //  reads element 0 four times)
                            const f16x2_t ones = {(_Float16)1.0f, (_Float16)1.0f};
                            const f16x8_t hv = __builtin_bit_cast(f16x8_t, af[u][t]);
#pragma unroll
                            for (int d = 0; d < 4; ++d) bsum[t] = __builtin_amdgcn_fdot2(f16x2_t{hv[2 * d], hv[2 * d + 1]}, ones, bsum[t], false);
                        }
                    }
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt) acc[t][kt] = T::mfma(af[u][t], bf[u][kt], acc[t][kt]);
                }
        }
        const int left = nt - 1 - it;                                 // tiles after this one; at most AH - 1 are younger than the next
        const int younger = min(left, AH) - 1;
        if (younger >= 3) DW16_WAIT(3); else if (younger == 2) DW16_WAIT(2); else if (younger == 1) DW16_WAIT(1); else DW16_WAIT(0);
        cur = (cur + 1) % DW16_STAGES;
    }
#undef DW16_WAIT
#undef DW16_WAIT_ASM
    // ---- accumulators -> global (float atomics; one register = two 128-byte row segments)
    const int r = lane & 31, h = lane >> 5;
    const float inv = 1.0f / (sgs * xs), inv_b = 1.0f / sgs;      // (xs: the scale the X planes carry -- 1, or the split-f16 chains' 2^3 in the hi-plane mode)
    if (!(sg.kmap | sg.kmap2 | sg.nmap)) {
#pragma unroll
        for (int t = 0; t < VN; ++t)
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                const int k = kbase + 32 * kt + r;
                // (a 32-column tile lies inside X or inside X2: K1 is a multiple of 32)
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
                // (a 32-column tile lies inside X or inside X2: K1 is a multiple of 32)
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
#define DW16_MAXSEG 15
struct Dw16Job {
    int n;
    float x_scale;                // power-of-two scale of the X planes (1; MCNX3_SX when the planes are the split-f16 chains' hi planes)
    int shape[DW16_MAXSEG];       // 0: W x W   1: W x 64 (encoded-input columns)   2: 32 x W (sh.2 / sigma.2 rows)   3: W x (W + 64) (skip layer)
    Dw16Seg seg[DW16_MAXSEG];
};

// the skip layer as ONE segment over [hidden | encoded] inputs (its dY read once): wide nets, where the W x (W + 64) block
// tiles the 8 waves evenly (2 x 5 tiles of 32 x 32 per wave at W = 256)
template <int W> struct Dw16SkipMerged { static constexpr bool value = (W == 256) || (W == 128); };   // (128 x 192: 1 x 3 tiles per wave)

template <int W, bool BF>
__global__ __launch_bounds__(64 * MCN16_WAVES) void dw16_stream_kernel(Dw16Job job, const int* count, int rows_cap, const unsigned* gmax_bits) {
    constexpr bool SKIP_MERGED = Dw16SkipMerged<W>::value;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int rows = count ? min(*count, rows_cap) : rows_cap;
    const int ntiles = (rows + 31) / 32;
    if (ntiles <= 0) return;
    const float gmax = gmax_bits ? __uint_as_float(*gmax_bits) : 1.f;
    const float sgs = mcn16_grad_scale(gmax);
    // the linear tile sequence, in units of 1 KiB pieces: segment s holds ntiles tiles of pieces(shape) each
    auto pieces = [](int shape) { return shape == 0 ? 2 * W / 16 : shape == 1 ? W / 16 + MCN16_ENCKS : shape == 2 ? 2 + W / 16 : 2 * W / 16 + MCN16_ENCKS; };
    long long total = 0;
    for (int s = 0; s < job.n; ++s) total += (long long)pieces(job.shape[s]) * ntiles;
    const long long lo = total * blockIdx.x / gridDim.x, hi = total * (blockIdx.x + 1) / gridDim.x;
    long long base = 0;
    for (int s = 0; s < job.n; ++s) {
        const int shape = job.shape[s], P = pieces(shape);
        // a tile belongs to the piece that holds its first 1 KiB piece
        const long long a = (lo - base + P - 1) / P, e = (hi - base + P - 1) / P;
        const int t0 = (int)(a < 0 ? 0 : a > ntiles ? ntiles : a), t1 = (int)(e < 0 ? 0 : e > ntiles ? ntiles : e);
        base += (long long)P * ntiles;
        if (t0 >= t1) continue;                                        // (block-uniform)
        if (shape == 0) dw16_run<W, W, 0, BF>(job.seg[s], t0, t1, sgs, job.x_scale, smem);
        else if (shape == 1) dw16_run<W, 16 * MCN16_ENCKS, 0, BF>(job.seg[s], t0, t1, sgs, job.x_scale, smem);
        else if (shape == 2) dw16_run<32, W, 0, BF>(job.seg[s], t0, t1, sgs, job.x_scale, smem);
        else if constexpr (SKIP_MERGED) dw16_run<W, W, 16 * MCN16_ENCKS, BF>(job.seg[s], t0, t1, sgs, job.x_scale, smem);
    }
}

static int dw16_num_cus() {
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
static hipError_t dw16_launch_job(const Dw16Job& job, const int* count, int rows_cap, int bf16, const unsigned* gmax_bits, hipStream_t st) {
    if (rows_cap <= 0) return hipSuccess;
    const long long ntiles = (rows_cap + 31) / 32;
    long long grid = dw16_num_cus();
    if (grid > ntiles * job.n) grid = ntiles * job.n;
    const int pmax = Dw16SkipMerged<W>::value ? 2 * W / 16 + MCN16_ENCKS : (2 * W / 16 > W / 16 + MCN16_ENCKS ? 2 * W / 16 : W / 16 + MCN16_ENCKS);
    const size_t lds = (size_t)DW16_STAGES * pmax * 1024;
    void (*kern)(Dw16Job, const int*, int, const unsigned*) = bf16 ? dw16_stream_kernel<W, true> : dw16_stream_kernel<W, false>;
    static bool attr_set[2] = {false, false};                          // (per width instantiation and precision)
    if (lds > 64 * 1024 && !attr_set[bf16 ? 1 : 0]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_set[bf16 ? 1 : 0] = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * MCN16_WAVES), lds, st, job, count, rows_cap, gmax_bits);
    return hipGetLastError();
}

hipError_t mcn16_launch_dw(const Mcn16DwArgs& a, hipStream_t st) {
    const McnLayout& L = a.lay;
    const int W = L.width, D = L.depth, KS = W / 16;
    const int ME = L.nfreq == MCN_NFREQ ? 0 : L.nfreq, MS = L.sh_deg == 2 ? 0 : 16 + L.sh_deg;      // index maps (Dw16Seg)
    auto act = [&](int slot) { return reinterpret_cast<const char*>(a.act_ws) + (size_t)slot * a.slot_bytes; };
    auto dy = [&](int slot) { return reinterpret_cast<const char*>(a.dy_ws) + (size_t)slot * a.slot_bytes; };
    const char* enc = reinterpret_cast<const char*>(a.enc_ws);
    const char* dsh = reinterpret_cast<const char*>(a.dsh_ws);
    Dw16Job job;
    job.n = 0;
    job.x_scale = a.x_scale > 0.f ? a.x_scale : 1.f;
    auto add = [&](int shape, const Dw16Seg& s) { job.shape[job.n] = shape; job.seg[job.n] = s; ++job.n; };
    if (D + 5 > DW16_MAXSEG) return hipErrorInvalidValue;
    const bool merged = (W == 256 || W == 128);       // (Dw16SkipMerged)
    for (int l = 0; l < D; ++l) {
        const int ldw = mcn_layer_in(L, l);
        float* dWl = a.grads + L.pW[l];
        float* dbl = a.grads + L.pB[l];
        if (l == 0)                       // encoded-input columns only
            add(1, Dw16Seg{dy(l), KS, enc, MCN16_ENCKS, nullptr, 0, 0, W, 0, MCN_ENC, 0, 0, dWl, ldw, dbl, ME, 0, 0});
        else if (l == L.skip && merged)   // [hidden | encoded] in one pass: hidden k -> column 63 + k, encoded k -> column k
            add(3, Dw16Seg{dy(l), KS, act(l - 1), KS, enc, MCN16_ENCKS, 0, W, L.nenc, W, 0, MCN_ENC, dWl, ldw, dbl, 0, ME, 0});
        else if (l == L.skip) {
            add(1, Dw16Seg{dy(l), KS, enc, MCN16_ENCKS, nullptr, 0, 0, W, 0, MCN_ENC, 0, 0, dWl, ldw, dbl, ME, 0, 0});
            add(0, Dw16Seg{dy(l), KS, act(l - 1), KS, nullptr, 0, 0, W, L.nenc, W, 0, 0, dWl, ldw, nullptr, 0, 0, 0});
        } else
            add(0, Dw16Seg{dy(l), KS, act(l - 1), KS, nullptr, 0, 0, W, 0, W, 0, 0, dWl, ldw, dbl, 0, 0, 0});
    }
    add(0, Dw16Seg{dy(D), KS, act(D - 1), KS, nullptr, 0, 0, W, 0, W, 0, 0, a.grads + L.pWs1, W, a.grads + L.pBs1, 0, 0, 0});
    add(0, Dw16Seg{dy(D + 1), KS, act(D - 1), KS, nullptr, 0, 0, W, 0, W, 0, 0, a.grads + L.pWc1, W, a.grads + L.pBc1, 0, 0, 0});
    add(2, Dw16Seg{dsh, 2, act(D + 1), KS, nullptr, 0, 0, MCN_NSH, 0, W, 0, 0, a.grads + L.pWc2, W, a.grads + L.pBc2, 0, 0, MS});
    // sigma.2 (1 x W): d sigma sits in column 27 of dsh, its input is the sigma hidden layer
    add(2, Dw16Seg{dsh, 2, act(D), KS, nullptr, 0, MCN_NSH, MCN_NSH + 1, 0, W, 0, 0, a.grads + L.pWs2, W, a.grads + L.pBs2, 0, 0, 0});
    switch (W) {
        case 256: return dw16_launch_job<256>(job, a.count, a.rows, a.bf16, a.gmax_bits, st);
        case 128: return dw16_launch_job<128>(job, a.count, a.rows, a.bf16, a.gmax_bits, st);
        case 64:  return dw16_launch_job<64>(job, a.count, a.rows, a.bf16, a.gmax_bits, st);
        case 32:  return dw16_launch_job<32>(job, a.count, a.rows, a.bf16, a.gmax_bits, st);
        default:  return hipErrorInvalidValue;
    }
}
