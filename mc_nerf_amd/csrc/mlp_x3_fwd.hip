// Fused NeRF MLP forward, split-f16 ("f16x3") mode on the register-chain architecture (mcnerf_x3.h): sample generation
// -> sinusoidal encoding -> trunk (+skip) -> sigma / SH heads -> SH colour -> sigmoid.  One wave = 32 samples carried
// through the whole network in registers as (hi, lo) f16 fragment pairs, three MFMAs per k-step into one fp32
// accumulator; the waves of a workgroup share the LDS ring that streams the packed (hi, lo) weight pieces.
// Persistent: one workgroup per CU walks the passes.
//
// MFMA shape: v_mfma_f32_16x16x32_f16, the wave's 32 samples as two groups of 16 (mcnerf_x3.h, "the 16 x 16 x 32 form"): lane =
// 16 rg + n holds sample 16 gi + n of group gi in row group rg.  The per-sample work (inputs, encoding, sigma / colour epilogue)
// is done ONCE per sample pair of lanes, as before: lane L "owns" sample m = 16 (L >> 5) + (L & 15) (copy r1 = (L >> 4) & 1), and
// what the other group's fragments need crosses the wave halves by v_permlane32_swap.
// Replaces (fp32-grade arithmetic: 22-bit operands, fp32 accumulate / bias / ReLU / epilogues) SinCosEmbedding.forward
// (model/net_block.py:20-35), CorseFine_NeRF.forward (model/net_block.py:67-78), eval_sh (model/net_utils.py:103-191)
// and the gather / scatter of NeRF_Model.inference (model/mc_nerf.py:688-701).
#include "mcnerf_x3.h"

template <int W>
struct FwdX3Smem {
    static constexpr int oBias = MCN16_RING * MCN16_SLAB * 1024;   // fp32 [MAXD + 2][W]: trunk, sigma.0, sh.0 biases x (SX SW)
    static constexpr int oW2 = oBias + (MCN_MAXD + 2) * W * 4;     // sigma.2 weight row [W] / SW
    static constexpr int oBc2 = oW2 + W * 4;                        // sh.2 bias [32] (27 + zero pad) x (SX SW)
    static constexpr int oIdx = oBc2 + 32 * 4;                      // per wave: the NEXT pass's (ray, sample) pairs [32][2] (LDS-DMA)
    static constexpr int oIn = oIdx + mcnx3_waves(W) * 256;         // per wave: the next pass's per-sample inputs, 8 x [64 lanes] dwords (LDS-DMA gathers)
    static constexpr int total = oIn + mcnx3_waves(W) * 8 * 256;
};

// float offset of the first of the four consecutive channels that rows 4 rg .. 4 rg + 3 of tile t2 are, less the lane's part
// 16 (rg >> 1) + 4 (rg & 1) (mcnx3_row16)
__host__ __device__ constexpr int mcnx3_tile_off(int t2) { return 32 * (t2 >> 1) + 8 * (t2 & 1); }
// Placement of a tile's NIT epilogue items in the G MFMA gaps of the tile that follows: one item every STRIDE gaps from gap START
// where the tile has that many gaps (START = 3, or as late as fits: the drained tile's last MFMAs have landed by then, no wait state
// in front of the first ReLU item), IPG per gap from gap 1 otherwise.
__host__ __device__ constexpr int mcnx3_item_stride(int G, int NIT) {
    const int stride = G / NIT;
    return (stride >= 1 && 1 + stride * (NIT - 1) <= G - 1) ? stride : 0;
}
__host__ __device__ constexpr int mcnx3_item_start(int G, int NIT) {
    const int stride = mcnx3_item_stride(G, NIT);
    if (stride == 0) return 1;
    const int room = G - 1 - stride * (NIT - 1);           // the latest start that still fits
    return room < 3 ? room : 3;
}
__host__ __device__ constexpr int mcnx3_items_per_gap(int G, int NIT) { return (NIT + (G - 1) - 1) / (G - 1); }

// One layer: NT output tiles of 16 channels, each the chain of KENC encoded-input k-steps and KHID hidden-input k-steps (of 32
// channels) over the (hi, lo) A pieces taken from the weight ring in stream order; per k-step and group three MFMAs (lo*hi, hi*lo,
// hi*hi), the two groups alternating.
//   EPI 0: out = split(relu(acc / SW)) as the next layer's fragments (saved with their ReLU bits when SAVE)
//   EPI 1: the sigma head's hidden layer: additionally dot[gi] += sum_n relu(acc)[n] * (w2[n] / SW)
// Software pipeline pinned with sched_barriers: A pieces are read MCNX3_PF k-steps ahead; the epilogue of tile t is cut into items of
// TWO independent vector instructions -- the same step for the two groups' words, so that no item reads what its first instruction
// wrote (v_fma_mix{lo,hi} are partial writes: a dependent neighbour costs a wait state) -- issued one item every second MFMA gap
// of tile t + 1 (a gap hides two vector instructions beside a 16-cycle MFMA).  The ReLU items come first and copy: tile t's accumulator
// set is free after four items and takes tile t + 2's (scaled) bias straight from LDS.  A lane's four words of tiles 2 so, 2 so + 1 are
// one 16-byte chunk of the saved plane: stored two tiles later.
//   SV 0: nothing saved; 1: the (hi, lo) fragment planes + ReLU bits; 2: the hi plane only, in the 16-bit modes' workspace layout (dtype 3:
//   the weight-gradient kernel of that mode is the single-pass f16 one on the hi planes)
template <int W, int SV, int KENC, int KHID, int EPI, int PPW>
__device__ __forceinline__ void mcnx3_layer(Mcn16Ring& ring, char* smem, int lane, float isw,
                                            const u32x4_t (&ench)[2][MCNX3_ENCKS2], const u32x4_t (&encl)[2][MCNX3_ENCKS2],
                                            const u32x4_t (&inh)[2][W / 32], const u32x4_t (&inl)[2][W / 32],
                                            u32x4_t (&outh)[2][W / 32], u32x4_t (&outl)[2][W / 32], const float* bias_l,
                                            const float* w2_l, float (&dot)[2], char* save_lane, unsigned* mask_lane, unsigned msh) {
    constexpr bool SAVE = SV != 0, HI = SV == 2;
    constexpr int NT = W / 16, KS2 = W / 32, KS = W / 16, KTOT = KENC + KHID, F = NT * KTOT, MW = W >= 64 ? W / 64 : 1;
    constexpr int G = 6 * KTOT;                                   // MFMA gaps per tile
    // items of one tile's epilogue: 4 ReLU items (group x word), then per word p (both groups in every item):
    //   hi.lo-half, hi.hi-half, lo.lo-half, lo.hi-half [, ReLU bit, bit word][, sigma dot x 2]
    constexpr int PER_P = 4 + (SAVE ? 2 : 0) + (EPI == 1 ? 2 : 0), NIT = 4 + 2 * PER_P;
    constexpr int STRIDE = mcnx3_item_stride(G, NIT), START = mcnx3_item_start(G, NIT), IPG = mcnx3_items_per_gap(G, NIT);
    constexpr int RELU_DONE = STRIDE > 0 ? START + 3 * STRIDE : 1 + 3 / IPG;    // gap of the last item that reads the drained accumulators
    constexpr int BG = (G - 6 > RELU_DONE + 1) ? G - 6 : RELU_DONE + 1;         // the next tile's accumulators (that set) take the bias here
    constexpr int SG_H = G >= 18 ? 9 : G - 2, SG_L = G >= 18 ? 15 : G - 1;     // gaps of the fragment stores
    static_assert(BG <= G - 1, "bias gap");
    Mcn16Cursor cur;
    unsigned mwd[2][MW], mb[2] = {0u, 0u}, macc[2] = {0u, 0u};
#pragma unroll
    for (int i = 0; i < MW; ++i) { mwd[0][i] = 0u; mwd[1][i] = 0u; }
    u32x4_t afh[MCNX3_PF], afl[MCNX3_PF];
    f32x4 acc[2][2];                              // [register set][group]
    f32x4 w2q = {0.f, 0.f, 0.f, 0.f};
    float rv[2][2][2];                            // [group][word][element]: relu(acc) of the tile being drained
    unsigned hw[2] = {0u, 0u}, lw[2] = {0u, 0u}, nzb[2] = {0u, 0u};      // per group: the words in flight
#pragma unroll
    for (int gi = 0; gi < 2; ++gi)
#pragma unroll
        for (int p = 0; p < 2; ++p) { rv[gi][p][0] = 0.f; rv[gi][p][1] = 0.f; }
    // work item k of the epilogue of tile t (accumulators a[group])
    auto item = [&](const f32x4 (&a)[2], int t, int k) {
        const int so = t >> 1, u = t & 1;
        if (k < 4) {
            const int gi = k & 1, p = k >> 1;
            rv[gi][p][0] = mcnx3_relu(a[gi][2 * p]);
            rv[gi][p][1] = mcnx3_relu(a[gi][2 * p + 1]);
            return;
        }
        const int p = (k - 4) / PER_P, j = (k - 4) % PER_P;
        if (j == 0) {
#pragma unroll
            for (int gi = 0; gi < 2; ++gi) asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hw[gi]) : "v"(rv[gi][p][0]), "s"(isw));
        } else if (j == 1) {
#pragma unroll
            for (int gi = 0; gi < 2; ++gi) {
                asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(hw[gi]) : "v"(rv[gi][p][1]), "s"(isw));
                outh[gi][so][2 * u + p] = hw[gi];
            }
        } else if (j == 2) {          // the lo halves of the (hi, lo) split: f16(r / SW - hi), r / SW and the difference exact in fp32
#pragma unroll
            for (int gi = 0; gi < 2; ++gi)
                asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(lw[gi]) : "v"(rv[gi][p][0]), "s"(isw), "v"(hw[gi]));
        } else if (j == 3) {
#pragma unroll
            for (int gi = 0; gi < 2; ++gi) {
                asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lw[gi]) : "v"(rv[gi][p][1]), "s"(isw), "v"(hw[gi]));
                outl[gi][so][2 * u + p] = lw[gi];
            }
        } else if (SAVE && j == 4) {
            nzb[0] = mcn16_nz(hw[0]);
            nzb[1] = mcn16_nz(hw[1]);
        } else if (SAVE && j == 5) {      // this lane's four bits of k-step-of-16 2 so + (rg >> 1): words 0 .. 3, the first one highest
#pragma unroll
            for (int gi = 0; gi < 2; ++gi) {
                mb[gi] = (u == 0 && p == 0) ? nzb[gi] : ((mb[gi] << 1) | nzb[gi]);
                if (u == 1 && p == 1) {
                    macc[gi] = (so & 1) ? (macc[gi] | (mb[gi] << 8)) : mb[gi];
                    if ((so & 1) || so == KS2 - 1) mwd[gi][so >> 1] = macc[gi] << msh;
                }
            }
        } else {                          // EPI 1: the sigma output layer's dot, one element of the word per item
            const int e = j - (PER_P - 2);
            dot[0] = fmaf(rv[0][p][e], w2q[2 * p + e], dot[0]);
            dot[1] = fmaf(rv[1][p][e], w2q[2 * p + e], dot[1]);
        }
    };
    // the plane chunks of (group gi, k-step-of-32 so): complete once the items of tile 2 so + 1 are done
    auto store_hi = [&](int gi, int so) {
        mcn16_ws_store(outh[gi][so], reinterpret_cast<u32x4_t*>(save_lane + so * 2048 + gi * 256));
    };
    auto store_lo = [&](int gi, int so) {
        mcn16_ws_store(outl[gi][so], reinterpret_cast<u32x4_t*>(save_lane + KS * 1024 + so * 2048 + gi * 256));
    };
    // (two reads of the same quad, through two pointers the compiler cannot tell equal: each group's accumulator is loaded in place and
    //  every MFMA of the tile is acc = A B + acc on one register quad -- with one shared read the first MFMAs write other registers than
    //  they read and the register rotation costs wait states in front of whatever re-uses them)
    int off1 = 0;
    asm volatile("" : "+v"(off1));                  // (an opaque zero OFFSET: a laundered pointer would lose its LDS address space)
    const float* bias_l1 = bias_l + off1;
    auto bias_init = [&](f32x4 (&a)[2], int t) {
        a[0] = *reinterpret_cast<const f32x4*>(bias_l + mcnx3_tile_off(t));
        a[1] = *reinterpret_cast<const f32x4*>(bias_l1 + mcnx3_tile_off(t));
    };
    cur.cur = ring.next_off;
#pragma unroll
    for (int i = 0; i < MCNX3_PF; ++i)
        if (i < F) {
            afh[i] = *reinterpret_cast<const u32x4_t*>(smem + ring.next_off + i * 2048 + lane * 16);
            afl[i] = *reinterpret_cast<const u32x4_t*>(smem + ring.next_off + i * 2048 + 1024 + lane * 16);
        }
    bias_init(acc[0], 0);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int s = 0; s < KTOT; ++s) {
            const int f = t * KTOT + s;
            mcnx3_before_mfma_spread<F, PPW>(ring, cur, f);
            const u32x4_t a_h = afh[f % MCNX3_PF], a_l = afl[f % MCNX3_PF];
            if (f + MCNX3_PF < F) {
                const unsigned o = mcnx3_frag_off(ring, cur, f, f + MCNX3_PF) + lane * 16;
                afh[f % MCNX3_PF] = *reinterpret_cast<const u32x4_t*>(smem + o);
                afl[f % MCNX3_PF] = *reinterpret_cast<const u32x4_t*>(smem + o + 1024);
            }
#pragma unroll
            for (int g6 = 0; g6 < 6; ++g6) {
                const int gi = g6 & 1, term = g6 >> 1, gap = 6 * s + g6;
                if (t > 0) {
                    if (EPI == 1 && gap == 0) w2q = *reinterpret_cast<const f32x4*>(w2_l + mcnx3_tile_off(t - 1));
                    if constexpr (STRIDE > 0) {
                        if (gap >= START && (gap - START) % STRIDE == 0 && (gap - START) / STRIDE < NIT) item(acc[(t - 1) & 1], t - 1, (gap - START) / STRIDE);
                    } else if (gap >= 1) {
#pragma unroll
                        for (int k = (gap - 1) * IPG; k < gap * IPG; ++k)
                            if (k < NIT) item(acc[(t - 1) & 1], t - 1, k);
                    }
                }
                if (SAVE && gap == SG_H) {
                    if (t >= 3 && (t & 1)) store_hi(0, (t - 3) / 2);
                    if (t >= 4 && !(t & 1)) store_hi(1, (t - 4) / 2);
                }
                if (SV == 1 && gap == SG_L) {
                    if (t >= 3 && (t & 1)) store_lo(0, (t - 3) / 2);
                    if (t >= 4 && !(t & 1)) store_lo(1, (t - 4) / 2);
                }
                if (gap == BG && t + 1 < NT) bias_init(acc[(t + 1) & 1], t + 1);
                mcnx3_gap_dma6<F, PPW>(ring, 6 * f + g6);
                const u32x4_t b = s < KENC ? (term == 1 ? encl[gi][s < KENC ? s : 0] : ench[gi][s < KENC ? s : 0])
                                           : (term == 1 ? inl[gi][s >= KENC ? s - KENC : 0] : inh[gi][s >= KENC ? s - KENC : 0]);
                __builtin_amdgcn_sched_barrier(0);
                acc[t & 1][gi] = mcnx3_mfma16(term == 0 ? a_l : a_h, b, acc[t & 1][gi]);     // lo*hi, hi*lo, hi*hi
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    mcnx3_layer_end6<F, PPW>(ring);
    if (EPI == 1) w2q = *reinterpret_cast<const f32x4*>(w2_l + mcnx3_tile_off(NT - 1));
#pragma unroll
    for (int k = 0; k < NIT; ++k) item(acc[(NT - 1) & 1], NT - 1, k);
    if (SAVE) {
#pragma unroll
        for (int so = 0; so < KS2; ++so)
#pragma unroll
            for (int gi = 0; gi < 2; ++gi)
                if (2 * so + 3 + gi > NT - 1) {              // (not stored inside the loop)
                    store_hi(gi, so);
                    if (!HI) store_lo(gi, so);
                }
        // ReLU bit words in the 32-row format: the word of (sample m, half h) is the OR of the nibbles of lanes (n, rg = h) and
        // (n, rg = h + 2) of the sample's group; after the swap lanes 0-31 hold both nibbles of group 0, lanes 32-63 of group 1
#pragma unroll
        for (int i = 0; i < MW; ++i) {
            mcnx3_swap32(mwd[0][i], mwd[1][i]);
            mcn16_ws_store(mwd[0][i] | mwd[1][i], mask_lane + i);
        }
    }
}

template <int W, int SV>
__global__ __launch_bounds__(64 * mcnx3_waves(W), mcnx3_waves(W) / 4) void mlp_x3_fwd_kernel(Mcn16FwdArgs a) {
    using SM = FwdX3Smem<W>;
    constexpr bool SAVE = SV != 0, HI = SV == 2;
    constexpr int PL = HI ? 1 : 2;                 // fragment planes per saved tile
    constexpr int WAVES = mcnx3_waves(W), ROWS = 32 * WAVES, PPW = 16 / WAVES;
    constexpr int KS = W / 16, KS2 = W / 32, MW = W >= 64 ? W / 64 : 1;
    constexpr float SXW = MCNX3_SX * MCNX3_SW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, rg = lane >> 4, r1 = rg & 1, gown = lane >> 5;
    const int m = 16 * gown + n;                   // the sample of the wave's 32-row tile this lane owns (twice: r1 = 0, 1)
    const int lane32 = 32 * r1 + 16 * gown + n;    // position (32 h + m) of (m, h = r1) in the 32-row formats
    const int D = a.lay.depth, skip = a.lay.skip;
    const long long total = a.count ? (long long)min(*a.count, a.max_rows) : (long long)a.n_rays * a.S;
    if ((long long)blockIdx.x * ROWS >= total) return;

    // ---- once per workgroup: scaled biases, sigma.2 row, scaled sh.2 bias -> LDS (before the ring starts: plain loads drain vmcnt)
    float* sbias = reinterpret_cast<float*>(smem + SM::oBias);
    float* sw2 = reinterpret_cast<float*>(smem + SM::oW2);
    float* sbc2 = reinterpret_cast<float*>(smem + SM::oBc2);
    for (int l = 0; l < D; ++l)
        for (int i = tid; i < W; i += 64 * WAVES) sbias[l * W + i] = a.params[a.lay.pB[l] + i] * SXW;
    for (int i = tid; i < W; i += 64 * WAVES) {
        sbias[D * W + i] = a.params[a.lay.pBs1 + i] * SXW;
        sbias[(D + 1) * W + i] = a.params[a.lay.pBc1 + i] * SXW;
        sw2[i] = a.params[a.lay.pWs2 + i] * (1.0f / MCNX3_SW);
    }
    if (tid < 32) {      // sh.2 bias in the kernel's 27-row geometry (a degree below 2: the rows the net has, zero elsewhere)
        const int row = tid < MCN_NSH ? mcn_sh_row(tid, a.lay.sh_deg) : -1;
        sbc2[tid] = row >= 0 ? a.params[a.lay.pBc2 + row] * SXW : 0.f;
    }
    const float bs2 = a.params[a.lay.pBs2];
    float bw[MCN_NFREQ];
#pragma unroll
    for (int f = 0; f < MCN_NFREQ; ++f) bw[f] = a.barf_w[f];
    const unsigned long long r1_mask = __builtin_amdgcn_ballot_w64(r1 != 0);
    const float isw = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, 1.0f / MCNX3_SW)));
    __syncthreads();

    // ---- per-sample inputs.  On the wide net (one wave per SIMD: nothing else covers a memory round trip) they are
    // fetched ONE PASS AHEAD by LDS-DMA, invisible to the compiler's vmcnt bookkeeping (an ordinary load would make hipcc
    // drain the weight ring at its first use): the next pass's (ray, sample) pairs right after this pass's prologue, the
    // gathers through them in front of the SH head; explicit counted waits (at most PPW * AHEAD operations are ever younger
    // than something issued a whole layer earlier) guarantee the landing.
    constexpr bool PREF = WAVES == 4;
    // (counted waits: in-order completion, at least N younger ring pieces issued since the operation waited for -- the index
    //  pairs: layer 0 + the sigma head; the gathers: the SH head + sh.2 -- and as many as vmcnt can express otherwise, so that the
    //  workspace stores in flight are not waited for)
    constexpr int SLABS_FULL = (W / 32) * (W / 16) / MCNX3_SLABF, SLABS_L0 = ((W / 32) * MCN16_ENCKS + MCNX3_SLABF - 1) / MCNX3_SLABF;
    constexpr int SLABS_SH2 = (W / 16 + MCNX3_SLABF - 1) / MCNX3_SLABF;
    constexpr int N_IDX = PPW * (SLABS_L0 + SLABS_FULL - 1) < 63 ? PPW * (SLABS_L0 + SLABS_FULL - 1) : 63;
    constexpr int N_TOP = PPW * (SLABS_FULL + SLABS_SH2 - 1) < 63 ? PPW * (SLABS_FULL + SLABS_SH2 - 1) : 63;
    const unsigned idx_lds = (unsigned)reinterpret_cast<size_t>((mcn16_lds_ptr_t)smem) + SM::oIdx + wave * 256;
    const unsigned in_lds = (unsigned)reinterpret_cast<size_t>((mcn16_lds_ptr_t)smem) + SM::oIn + wave * (8 * 256);
    const float* in_rd = reinterpret_cast<const float*>(smem + SM::oIn + wave * (8 * 256)) + lane;
    auto row_of = [&](long long pass_) -> long long {           // this lane's row of a pass, clamped into the list
        const long long g_ = (pass_ * WAVES + wave) * 32 + m;
        return g_ < total ? g_ : total - 1;
    };
    auto gather_dma = [&](int ray_, int j_) {
        mcn16_dma4(a.zgrid + j_, in_lds);
        if (a.jitter) mcn16_dma4(a.jitter + ray_, in_lds + 256);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            mcn16_dma4(a.rays_d + ray_ * 3 + c, in_lds + (2 + c) * 256);
            mcn16_dma4(a.rays_o + ray_ * 3 + c, in_lds + (5 + c) * 256);
        }
    };
    int ray_n = 0, j_n = 0;                                     // (ray, sample) of this lane's row of the coming pass
    {
        const long long gc0 = row_of(blockIdx.x);
        if (a.idx) { const int2 rj = a.idx[gc0]; ray_n = rj.x; j_n = rj.y; }
        else { ray_n = (int)(gc0 / a.S); j_n = (int)(gc0 - (long long)ray_n * a.S); }
        if (PREF) {
            gather_dma(ray_n, j_n);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }

    Mcn16Ring ring;
    mcnx3_ring_start<PPW>(ring, smem, a.packed, a.stream_slabs, wave, lane);

    // lane parts of the tile geometry: rows 4 rg .. 4 rg + 3 of a tile are channels mcnx3_tile_off(t2) + 16 (rg >> 1) + 4 (rg & 1) + i
    const int lane_ch = 16 * gown + 4 * r1;
    const float* bias_l = sbias + lane_ch;
    const float* w2_l = sw2 + lane_ch;
    const unsigned msh = 4u * (1u - (unsigned)gown);            // the nibble of a ReLU bit byte this lane's row group fills
    const size_t chunk_off = (size_t)gown * 1024 + (32 * r1 + n) * 16;      // of group 0's chunk inside a k-step-of-32 pair of fragments

    for (long long pass = blockIdx.x; pass * ROWS < total; pass += gridDim.x) {
        const long long tile = pass * WAVES + wave;             // global 32-row tile of this wave
        const long long g = tile * 32 + m;
        const bool valid = g < total;
        const long long gc = valid ? g : total - 1;
        const long long pass_n = pass + gridDim.x;
        // ---- per-sample setup (the lane's own sample; both copies of a sample compute the same values)
        int ray, j;
        float zv, dx, dy, dz, ox, oy, oz;
        if (PREF) {
            ray = ray_n; j = j_n;
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_TOP) : "memory");
            zv = in_rd[0];
            if (a.jitter) zv = __fadd_rn(zv, in_rd[64]);
            dx = in_rd[2 * 64]; dy = in_rd[3 * 64]; dz = in_rd[4 * 64];
            ox = in_rd[5 * 64]; oy = in_rd[6 * 64]; oz = in_rd[7 * 64];
        } else {
            if (a.idx) { const int2 rj = a.idx[gc]; ray = rj.x; j = rj.y; }
            else { ray = (int)(gc / a.S); j = (int)(gc - (long long)ray * a.S); }
            zv = a.zgrid[j];
            if (a.jitter) zv = __fadd_rn(zv, a.jitter[ray]);
            dx = a.rays_d[ray * 3 + 0]; dy = a.rays_d[ray * 3 + 1]; dz = a.rays_d[ray * 3 + 2];
            ox = a.rays_o[ray * 3 + 0]; oy = a.rays_o[ray * 3 + 1]; oz = a.rays_o[ray * 3 + 2];
        }
        float p[3];
        p[0] = __fadd_rn(ox, __fmul_rn(dx, zv));   // o + d z, two roundings (model/mc_nerf.py:602)
        p[1] = __fadd_rn(oy, __fmul_rn(dy, zv));
        p[2] = __fadd_rn(oz, __fmul_rn(dz, zv));
        if (PREF && a.idx) {       // lane L fetches dword L of the next pass's 32 (ray, sample) pairs
            const long long gn = (pass_n * WAVES + wave) * 32 + (lane >> 1);
            mcn16_dma4(reinterpret_cast<const int*>(a.idx) + 2 * (gn < total ? gn : total - 1) + (lane & 1), idx_lds);
        }
        const int addr = ray * a.S + j;
        // ---- encoded input fragments.  From its own sample a lane packs the chunks (k-step-of-16 2 s2, h = r1) and (2 s2 + 1, h = r1):
        // the first is the fragment of row group r1 (lanes 0-31's own group-0 fragment, and what lanes 0-31 need of group 1 from their
        // partner lane L + 32), the second that of row group r1 + 2 -- one swap of the wave halves per word puts each where it is used.
        u32x4_t ench[2][MCNX3_ENCKS2], encl[2][MCNX3_ENCKS2];
        {
            float E[64];
            mcnx3_encode_values(p, bw, E);
#pragma unroll
            for (int s2 = 0; s2 < MCNX3_ENCKS2; ++s2)
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const int c0 = mcn16_chan(2 * s2 + u, 0, 2 * d), c1 = mcn16_chan(2 * s2 + u, 0, 2 * d + 1);
                        unsigned wh, wl;
                        mcnx3_split2(mcnx3_sel(E[c0], E[c0 + 4], r1_mask) * MCNX3_SX, mcnx3_sel(E[c1], E[c1 + 4], r1_mask) * MCNX3_SX, wh, wl);
                        ench[u][s2][d] = wh; encl[u][s2][d] = wl;
                    }
#pragma unroll
            for (int s2 = 0; s2 < MCNX3_ENCKS2; ++s2)
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    unsigned x0 = ench[0][s2][d], x1 = ench[1][s2][d], y0 = encl[0][s2][d], y1 = encl[1][s2][d];
                    mcnx3_swap32(x0, x1);
                    mcnx3_swap32(y0, y1);
                    ench[0][s2][d] = x0; ench[1][s2][d] = x1; encl[0][s2][d] = y0; encl[1][s2][d] = y1;
                }
        }
        char* act_lane = SAVE ? reinterpret_cast<char*>(a.act_ws) + (size_t)tile * (PL * KS) * 1024 + chunk_off : nullptr;
        unsigned* mask_lane = SAVE ? a.mask_ws + ((size_t)tile * 64 + lane32) * MW : nullptr;
        if (SAVE) {
            char* e = reinterpret_cast<char*>(a.enc_ws) + (size_t)tile * (PL * MCN16_ENCKS) * 1024 + chunk_off;
#pragma unroll
            for (int s2 = 0; s2 < MCNX3_ENCKS2; ++s2)
#pragma unroll
                for (int gi = 0; gi < 2; ++gi) {
                    mcn16_ws_store(ench[gi][s2], reinterpret_cast<u32x4_t*>(e + s2 * 2048 + gi * 256));
                    if (!HI) mcn16_ws_store(encl[gi][s2], reinterpret_cast<u32x4_t*>(e + MCN16_ENCKS * 1024 + s2 * 2048 + gi * 256));
                }
        }

        u32x4_t xah[2][KS2], xal[2][KS2], xbh[2][KS2], xbl[2][KS2];
        float dot[2] = {0.f, 0.f};
        // ---- layer 0 (encoded input only), then the trunk two layers per trip (xb -> xa -> xb: no copies between layers);
        //      the skip layer takes [encoding, hidden]
        mcnx3_layer<W, SV, MCNX3_ENCKS2, 0, 0, PPW>(ring, smem, lane, isw, ench, encl, xah, xal, xbh, xbl, bias_l, nullptr, dot, act_lane, mask_lane, msh);
        for (int l = 1; l < D; l += 2) {
            char* sl = SAVE ? act_lane + (size_t)l * a.slot_bytes : nullptr;
            unsigned* ml = SAVE ? mask_lane + (size_t)l * a.mask_slot_words : nullptr;
            if (l == skip) mcnx3_layer<W, SV, MCNX3_ENCKS2, KS2, 0, PPW>(ring, smem, lane, isw, ench, encl, xbh, xbl, xah, xal, bias_l + l * W, nullptr, dot, sl, ml, msh);
            else mcnx3_layer<W, SV, 0, KS2, 0, PPW>(ring, smem, lane, isw, ench, encl, xbh, xbl, xah, xal, bias_l + l * W, nullptr, dot, sl, ml, msh);
            if (l + 1 < D) {
                sl = SAVE ? act_lane + (size_t)(l + 1) * a.slot_bytes : nullptr;
                ml = SAVE ? mask_lane + (size_t)(l + 1) * a.mask_slot_words : nullptr;
                if (l + 1 == skip) mcnx3_layer<W, SV, MCNX3_ENCKS2, KS2, 0, PPW>(ring, smem, lane, isw, ench, encl, xah, xal, xbh, xbl, bias_l + (l + 1) * W, nullptr, dot, sl, ml, msh);
                else mcnx3_layer<W, SV, 0, KS2, 0, PPW>(ring, smem, lane, isw, ench, encl, xah, xal, xbh, xbl, bias_l + (l + 1) * W, nullptr, dot, sl, ml, msh);
            } else {               // an even trunk depth ends in xa: one copy per pass
#pragma unroll
                for (int gi = 0; gi < 2; ++gi)
#pragma unroll
                    for (int s = 0; s < KS2; ++s) { xbh[gi][s] = xah[gi][s]; xbl[gi][s] = xal[gi][s]; }
            }
        }
        // ---- sigma head: hidden layer on the matrix pipe, the 1-wide output layer lane-local (on the fp32 activations)
        mcnx3_layer<W, SV, 0, KS2, 1, PPW>(ring, smem, lane, isw, ench, encl, xbh, xbl, xah, xal, bias_l + D * W, w2_l, dot,
                                             SAVE ? act_lane + (size_t)D * a.slot_bytes : nullptr, SAVE ? mask_lane + (size_t)D * a.mask_slot_words : nullptr, msh);
        if (PREF) {                // the coming pass's rows: index pair from LDS, gathers by LDS-DMA (landed long before the pass ends)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_IDX) : "memory");
            if (a.idx) {
                const int2 rj = *reinterpret_cast<const int2*>(smem + SM::oIdx + wave * 256 + m * 8);
                ray_n = rj.x; j_n = rj.y;
            } else {
                const long long gn = row_of(pass_n);
                ray_n = (int)(gn / a.S); j_n = (int)(gn - (long long)ray_n * a.S);
            }
            gather_dma(ray_n, j_n);
        }
        // ---- SH head: hidden layer (reads the same trunk output), then the 27 (32) coefficient rows as two tiles of 16
        mcnx3_layer<W, SV, 0, KS2, 0, PPW>(ring, smem, lane, isw, ench, encl, xbh, xbl, xah, xal, bias_l + (D + 1) * W, nullptr, dot,
                                             SAVE ? act_lane + (size_t)(D + 1) * a.slot_bytes : nullptr, SAVE ? mask_lane + (size_t)(D + 1) * a.mask_slot_words : nullptr, msh);
        f32x4 sacc[2][2];                          // [tile u][group]: rows 16 (rg >> 1) + 8 u + 4 (rg & 1) + i of the sh.2 outputs
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(sbc2 + lane_ch + 8 * u);
            sacc[u][0] = b; sacc[u][1] = b;
        }
        {
            Mcn16Cursor cur;
#pragma unroll
            for (int f = 0; f < 2 * KS2; ++f) {
                const int u = f / KS2, s = f % KS2;
                mcnx3_before_mfma<2 * KS2, PPW>(ring, cur, f);
                const unsigned o = cur.cur + (f & (MCNX3_SLABF - 1)) * 2048 + lane * 16;
                const u32x4_t a_h = *reinterpret_cast<const u32x4_t*>(smem + o);
                const u32x4_t a_l = *reinterpret_cast<const u32x4_t*>(smem + o + 1024);
#pragma unroll
                for (int g6 = 0; g6 < 6; ++g6) {
                    const int gi = g6 & 1, term = g6 >> 1;
                    sacc[u][gi] = mcnx3_mfma16(term == 0 ? a_l : a_h, term == 1 ? xal[gi][s] : xah[gi][s], sacc[u][gi]);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int gi = 0; gi < 2; ++gi) sacc[u][gi] *= 1.0f / SXW;
        if (SAVE) {          // the SH coefficients (bias included) for the backward's view-direction term: the fp32 accumulator tile of the
            //                   32-row form, whose quad q = 2 (rg >> 1) + u of lane (m, h = rg & 1) this is
            char* e = reinterpret_cast<char*>(a.sh_ws) + (size_t)tile * 4096 + (size_t)gown * 2048 + (32 * r1 + n) * 16;
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int gi = 0; gi < 2; ++gi)
                    mcn16_ws_store(sacc[u][gi], reinterpret_cast<f32x4*>(e + u * 1024 + gi * 256));
        }
        // ---- per-sample epilogue: sigma, SH colour (model/net_utils.py:154-169), sigmoid.  Accumulator i of tile u is SH row
        //      16 (rg >> 1) + 8 u + 4 (rg & 1) + i = 9 c + b (colour c, basis b) of the group's sample; group gi's direction is the
        //      lane's own where gi == L >> 5, its partner's (L ^ 32) otherwise.
        float dox = dx, doy = dy, doz = dz, dpx = dx, dpy = dy, dpz = dz;
        mcnx3_swap32(dox, dpx); mcnx3_swap32(doy, dpy); mcnx3_swap32(doz, dpz);
        // (after the swap: lanes 0-31: do* = own (group 0), dp* = partner's (group 1); lanes 32-63: do* = partner's (group 0), dp* = own (group 1))
        float pre[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {
            float bas[9];
            if (gi == 0) mcn_sh_basis(dox, doy, doz, bas);
            else mcn_sh_basis(dpx, dpy, dpz, bas);
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int v = 0; v < 4; ++v) {          // the four row groups hold different rows in this register
                        const int row = 16 * (v >> 1) + 8 * u + 4 * (v & 1) + i;
                        if (row < MCN_NSH) pre[gi][row / 9] = fmaf(sacc[u][gi][i], rg == v ? bas[row % 9] : 0.f, pre[gi][row / 9]);
                    }
        }
        // sums over the four row groups of a sample; a lane ends with its OWN sample's values (lanes 0-31: group 0, 32-63: group 1)
        float own[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float x0 = c < 3 ? pre[0][c] : dot[0], x1 = c < 3 ? pre[1][c] : dot[1];
            mcnx3_swap32(x0, x1);
            const float sown = x0 + x1;
            own[c] = sown + __shfl_xor(sown, 16);
        }
        const float sigma = own[3] * (1.0f / MCNX3_SX) + bs2;
        f32x4 o;
        o[0] = sigma;
#pragma unroll
        for (int c = 0; c < 3; ++c) o[1 + c] = 1.0f / (1.0f + expf(-own[c]));
        if (valid && r1 == 0) *reinterpret_cast<f32x4*>(a.out + (size_t)addr * 4) = o;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // ring pieces still in flight must land before the LDS is released
}

template <int W>
static hipError_t launch_fwd_x3(const Mcn16FwdArgs& a, long long max_rows, hipStream_t st) {
    using SM = FwdX3Smem<W>;
    constexpr int WAVES = mcnx3_waves(W), ROWS = 32 * WAVES;
    int dev = 0, cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
    long long passes = (max_rows + ROWS - 1) / ROWS;
    if (passes <= 0) return hipSuccess;
    const int grid = (int)(passes < cus ? passes : cus);
    const bool save = a.act_ws != nullptr;
    void (*kern)(Mcn16FwdArgs) = save ? (a.bf16 == 3 ? mlp_x3_fwd_kernel<W, 2> : mlp_x3_fwd_kernel<W, 1>) : mlp_x3_fwd_kernel<W, 0>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, SM::total);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WAVES), SM::total, st, a);
    return hipGetLastError();
}

hipError_t mcnx3_launch_fwd(const Mcn16FwdArgs& a, hipStream_t st) {
    const long long max_rows = a.count ? (long long)a.max_rows : (long long)a.n_rays * a.S;
    switch (a.lay.width) {
        case 256: return launch_fwd_x3<256>(a, max_rows, st);
        case 128: return launch_fwd_x3<128>(a, max_rows, st);
        case 64:  return launch_fwd_x3<64>(a, max_rows, st);
        case 32:  return launch_fwd_x3<32>(a, max_rows, st);
    }
    return hipErrorInvalidValue;
}
