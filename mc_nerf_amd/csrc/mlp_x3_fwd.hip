// Fused NeRF MLP forward, split-f16 ("f16x3") mode on the register-chain architecture (mcnerf_x3.h): sample generation
// -> sinusoidal encoding -> trunk (+skip) -> sigma / SH heads -> SH colour -> sigmoid.  One wave = 32 samples carried
// through the whole network in registers as (hi, lo) f16 fragment pairs, three MFMAs per k-step into one fp32
// accumulator; the waves of a workgroup share the LDS ring that streams the packed (hi, lo) weight pieces.
// Persistent: one workgroup per CU walks the passes.
// Replaces (fp32-grade arithmetic: 22-bit operands, fp32 accumulate / bias / ReLU / epilogues) SinCosEmbedding.forward
// (model/net_block.py:20-35), CorseFine_NeRF.forward (model/net_block.py:67-78), eval_sh (model/net_utils.py:103-191)
// and the gather / scatter of NeRF_Model.inference (model/mc_nerf.py:688-701).
#include "mcnerf_x3.h"

template <int W>
struct FwdX3Smem {
    static constexpr int oBias = MCN16_RING * MCN16_SLAB * 1024;   // fp32 [MAXD + 2][W]: trunk, sigma.0, sh.0 biases x (SX SW)
    static constexpr int oW2 = oBias + (MCN_MAXD + 2) * W * 4;     // sigma.2 weight row [W]
    static constexpr int oBc2 = oW2 + W * 4;                        // sh.2 bias [32] (27 + zero pad) x (SX SW)
    static constexpr int oIdx = oBc2 + 32 * 4;                      // per wave: the NEXT pass's (ray, sample) pairs [32][2] (LDS-DMA)
    static constexpr int oIn = oIdx + mcnx3_waves(W) * 256;         // per wave: the next pass's per-sample inputs, 8 x [64 lanes] dwords (LDS-DMA gathers)
    static constexpr int total = oIn + mcnx3_waves(W) * 8 * 256;
};

// One layer: NT output tiles, each the chain of KENC encoded-input k-steps and KHID hidden-input k-steps over the (hi, lo)
// A pieces taken from the weight ring in stream order, three MFMAs per k-step.
//   EPI 0: out = split(relu(acc / SW)) as the next layer's fragments (saved with their ReLU bits when SAVE)
//   EPI 1: the sigma head's hidden layer: additionally dot += sum_n relu(acc / SW)[n] * w2[n]
// Software pipeline pinned with sched_barriers: A pieces are read MCNX3_PF k-steps ahead; the epilogue of tile t (16 work
// items of <= 9 vector instructions + 4 stores) is issued one item per MFMA gap of tile t + 1, whose accumulator is the
// other of two register sets and starts at the (scaled) bias.
//   SV 0: nothing saved; 1: the (hi, lo) fragment planes + ReLU bits; 2: the hi plane only, in the 16-bit modes' workspace layout (dtype 3:
//   the weight-gradient kernel of that mode is the single-pass f16 one on the hi planes)
template <int W, int SV, int KENC, int KHID, int EPI, int PPW>
__device__ __forceinline__ void mcnx3_layer(Mcn16Ring& ring, char* smem, int lane,
                                            const u32x4_t (&ench)[MCN16_ENCKS], const u32x4_t (&encl)[MCN16_ENCKS],
                                            const u32x4_t (&inh)[W / 16], const u32x4_t (&inl)[W / 16],
                                            u32x4_t (&outh)[W / 16], u32x4_t (&outl)[W / 16], const float* bias_h,
                                            const float* w2_h, float& dot, char* save_lane, unsigned* mask_lane) {
    constexpr bool SAVE = SV != 0, HI = SV == 2;
    constexpr int NT = W / 32, KS = W / 16, KTOT = KENC + KHID, F = NT * KTOT, MW = W >= 64 ? W / 64 : 1;
    constexpr int G = 3 * KTOT;                                   // MFMA gaps per tile
    constexpr bool HAS2 = SAVE || EPI == 1;                       // a word has a second item (ReLU bit, sigma dot)
    constexpr int IPW = HAS2 ? 3 : 2;                             // items per packed word
    constexpr int NST = SAVE ? (HI ? 2 : 4) : 0;                  // fragment stores of a tile
    constexpr int NIT = 8 * IPW + NST;                            // work items of one tile's epilogue (each <= 6 vector instructions)
    // Placement of a tile's epilogue in the gaps of the next tile.  With fragment stores (SAVE) and room for it: the word items from
    // gap 3 (the previous tile's last MFMA must have landed), IPG per gap, then the four stores one every SSTR gaps over the rest of
    // the tile instead of back to back (same-box A/B, 3.28 M rows of the 256-wide net: saving forward 12.12 -> 11.72 ms, backward
    // 11.46 -> 11.33 ms; a gap of its own per (store, wave) pair needs a scalar branch per store and costs + 15 %).  Otherwise all
    // items in order, as many per gap as it takes.
    constexpr int NWI = 8 * IPW;                                  // word items
    constexpr int ROOM = G - 3 - 2 * NST;                         // gaps for the word items when every store gets two
    constexpr int IPGA = ROOM > 0 ? (NWI + ROOM - 1) / ROOM : 99;
    constexpr bool STAG = SAVE && IPGA <= 2;
    constexpr int START = STAG ? 3 : (G >= NIT + 4 ? 3 : 0);      // first gap that carries an item
    constexpr int IPG = STAG ? IPGA : (NIT + (G - START) - 1) / (G - START);    // items per gap
    constexpr int NITG = STAG ? NWI : NIT;                        // items placed by the items-per-gap rule
    constexpr int LASTG = START + (NITG + IPG - 1) / IPG - 1;     // gap of the last of them
    constexpr int SBASE = LASTG + 1, SSTR = STAG ? (G - SBASE) / NST : 1;
    constexpr int BIAS_G = (G - 6) > LASTG ? (G - 6) : LASTG;     // the next tile's accumulator (= the set just drained) is loaded here
    Mcn16Cursor cur;
    unsigned mw[MW];
#pragma unroll
    for (int i = 0; i < MW; ++i) mw[i] = 0u;
    u32x4_t afh[MCNX3_PF], afl[MCNX3_PF];
    f32x16 acc[2];
    unsigned mb = 0u;
    float v0 = 0.f, v1 = 0.f;
    unsigned wkeep = 0u;                          // the hi word of the pair in flight between its two items (hipcc 7.2 reads
                                                  // element 0 when an element of a u32x4 is bit-cast to f16x2: never re-read it from the vector)
    auto bias_init = [&](f32x16& a, int t) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {            // accumulator starts at the bias: rows 32t + 8q + 4h + e
            const f32x4 b = *reinterpret_cast<const f32x4*>(bias_h + 32 * t + 8 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) a[4 * q + e] = b[e];
        }
    };
    // work item i of the epilogue of tile t (accumulator a)
    auto item = [&](const f32x16& a, int t, int i) {
        if (i < 8 * IPW) {
            const int p = i / IPW, ph = i % IPW;  // packed word p of the tile: registers 2p, 2p + 1
            if (ph == 0) {
                v0 = mcnx3_relu(a[2 * p] * (1.0f / MCNX3_SW));
                v1 = mcnx3_relu(a[2 * p + 1] * (1.0f / MCNX3_SW));
                const unsigned w = Mcn16T<false>::pack(v0, v1);
                wkeep = w;
                outh[2 * t + (p >> 2)][p & 3] = w;
            } else if (ph == IPW - 1) {
                outl[2 * t + (p >> 2)][p & 3] = Mcn16T<false>::pack(mcnx3_residual<0>(v0, wkeep), mcnx3_residual<1>(v1, wkeep));
            } else {
                if (SAVE) {
                    mb = (p == 0) ? mcn16_nz(wkeep) : ((mb << 1) | mcn16_nz(wkeep));
                    if (p == 7) mw[t >> 1] |= mb << (8 * (t & 1));
                }
                if (EPI == 1) {
                    const f32x2_t ww = *reinterpret_cast<const f32x2_t*>(w2_h + 32 * t + 8 * (p >> 1) + 2 * (p & 1));
                    dot = fmaf(v0, ww[0], dot);
                    dot = fmaf(v1, ww[1], dot);
                }
            }
        } else if (SAVE) {
            const int k = i - 8 * IPW;           // 0, 1: hi plane k-steps 2t, 2t + 1; 2, 3: lo plane
            const int s = 2 * t + (k & 1);
            if (k < 2) mcn16_ws_store(outh[s], reinterpret_cast<u32x4_t*>(save_lane + s * 1024));
            else mcn16_ws_store(outl[s], reinterpret_cast<u32x4_t*>(save_lane + (KS + s) * 1024));      // (items k = 2, 3 do not exist with HI)
        }
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
            const u32x4_t b_h = s < KENC ? ench[s < KENC ? s : 0] : inh[s >= KENC ? s - KENC : 0];
            const u32x4_t b_l = s < KENC ? encl[s < KENC ? s : 0] : inl[s >= KENC ? s - KENC : 0];
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                const int gap = 3 * s + g;
                if (t > 0 && gap >= START) {
#pragma unroll
                    for (int i = (gap - START) * IPG; i < (gap - START + 1) * IPG; ++i)
                        if (i < NITG) item(acc[(t - 1) & 1], t - 1, i);
                    if (STAG && gap >= SBASE && (gap - SBASE) / SSTR < NST && (gap - SBASE) % SSTR == 0)
                        item(acc[(t - 1) & 1], t - 1, 8 * IPW + (gap - SBASE) / SSTR);
                }
                if (gap == BIAS_G && t + 1 < NT) bias_init(acc[(t + 1) & 1], t + 1);
                mcnx3_gap_dma<F, PPW>(ring, 3 * f + g);
                __builtin_amdgcn_sched_barrier(0);
                acc[t & 1] = mcnx3_mfma(g == 0 ? a_l : a_h, g == 1 ? b_l : b_h, acc[t & 1]);     // lo*hi, hi*lo, hi*hi
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    mcnx3_layer_end<F, PPW>(ring);
#pragma unroll
    for (int i = 0; i < NIT; ++i) item(acc[(NT - 1) & 1], NT - 1, i);
    if (SAVE) {
#pragma unroll
        for (int i = 0; i < MW; ++i) mcn16_ws_store(mw[i], mask_lane + i);
    }
}


template <int W, int SV>
__global__ __launch_bounds__(64 * mcnx3_waves(W), mcnx3_waves(W) / 4) void mlp_x3_fwd_kernel(Mcn16FwdArgs a) {
    using SM = FwdX3Smem<W>;
    constexpr bool SAVE = SV != 0, HI = SV == 2;
    constexpr int PL = HI ? 1 : 2;                 // fragment planes per saved tile
    constexpr int WAVES = mcnx3_waves(W), ROWS = 32 * WAVES, PPW = 16 / WAVES;
    constexpr int KS = W / 16, MW = W >= 64 ? W / 64 : 1;
    constexpr float SXW = MCNX3_SX * MCNX3_SW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 31, h = lane >> 5;
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
        sw2[i] = a.params[a.lay.pWs2 + i];
    }
    if (tid < 32) {      // sh.2 bias in the kernel's 27-row geometry (a degree below 2: the rows the net has, zero elsewhere)
        const int row = tid < MCN_NSH ? mcn_sh_row(tid, a.lay.sh_deg) : -1;
        sbc2[tid] = row >= 0 ? a.params[a.lay.pBc2 + row] * SXW : 0.f;
    }
    const float bs2 = a.params[a.lay.pBs2];
    float bw[MCN_NFREQ];
#pragma unroll
    for (int f = 0; f < MCN_NFREQ; ++f) bw[f] = a.barf_w[f];
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

    const float* bias_h = sbias + 4 * h;
    const float* w2_h = sw2 + 4 * h;

    for (long long pass = blockIdx.x; pass * ROWS < total; pass += gridDim.x) {
        const long long tile = pass * WAVES + wave;             // global 32-row tile of this wave
        const long long g = tile * 32 + m;
        const bool valid = g < total;
        const long long gc = valid ? g : total - 1;
        const long long pass_n = pass + gridDim.x;
        // ---- per-sample setup (lane-local; both lane halves of a sample compute the same values)
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
        u32x4_t ench[MCN16_ENCKS], encl[MCN16_ENCKS];
        {
            float E[64];
            mcnx3_encode_values(p, bw, E);
#pragma unroll
            for (int s = 0; s < MCN16_ENCKS; ++s)
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const int c0 = mcn16_chan(s, 0, 2 * d), c1 = mcn16_chan(s, 0, 2 * d + 1);
                    unsigned wh, wl;
                    mcnx3_split2((h ? E[c0 + 4] : E[c0]) * MCNX3_SX, (h ? E[c1 + 4] : E[c1]) * MCNX3_SX, wh, wl);
                    ench[s][d] = wh; encl[s][d] = wl;
                }
        }
        char* act_lane = SAVE ? reinterpret_cast<char*>(a.act_ws) + (size_t)tile * (PL * KS) * 1024 + lane * 16 : nullptr;
        unsigned* mask_lane = SAVE ? a.mask_ws + ((size_t)tile * 64 + lane) * MW : nullptr;
        if (SAVE) {
            char* e = reinterpret_cast<char*>(a.enc_ws) + (size_t)tile * (PL * MCN16_ENCKS) * 1024 + lane * 16;
#pragma unroll
            for (int s = 0; s < MCN16_ENCKS; ++s) {
                mcn16_ws_store(ench[s], reinterpret_cast<u32x4_t*>(e + s * 1024));
                if (!HI) mcn16_ws_store(encl[s], reinterpret_cast<u32x4_t*>(e + (MCN16_ENCKS + s) * 1024));
            }
        }

        u32x4_t xah[KS], xal[KS], xbh[KS], xbl[KS];
        float dot = 0.f;
        // ---- layer 0 (encoded input only), then the trunk two layers per trip (xb -> xa -> xb: no copies between layers);
        //      the skip layer takes [encoding, hidden]
        mcnx3_layer<W, SV, MCN16_ENCKS, 0, 0, PPW>(ring, smem, lane, ench, encl, xah, xal, xbh, xbl, bias_h, nullptr, dot, act_lane, mask_lane);
        for (int l = 1; l < D; l += 2) {
            char* sl = SAVE ? act_lane + (size_t)l * a.slot_bytes : nullptr;
            unsigned* ml = SAVE ? mask_lane + (size_t)l * a.mask_slot_words : nullptr;
            if (l == skip) mcnx3_layer<W, SV, MCN16_ENCKS, KS, 0, PPW>(ring, smem, lane, ench, encl, xbh, xbl, xah, xal, bias_h + l * W, nullptr, dot, sl, ml);
            else mcnx3_layer<W, SV, 0, KS, 0, PPW>(ring, smem, lane, ench, encl, xbh, xbl, xah, xal, bias_h + l * W, nullptr, dot, sl, ml);
            if (l + 1 < D) {
                sl = SAVE ? act_lane + (size_t)(l + 1) * a.slot_bytes : nullptr;
                ml = SAVE ? mask_lane + (size_t)(l + 1) * a.mask_slot_words : nullptr;
                if (l + 1 == skip) mcnx3_layer<W, SV, MCN16_ENCKS, KS, 0, PPW>(ring, smem, lane, ench, encl, xah, xal, xbh, xbl, bias_h + (l + 1) * W, nullptr, dot, sl, ml);
                else mcnx3_layer<W, SV, 0, KS, 0, PPW>(ring, smem, lane, ench, encl, xah, xal, xbh, xbl, bias_h + (l + 1) * W, nullptr, dot, sl, ml);
            } else {               // an even trunk depth ends in xa: one copy per pass
#pragma unroll
                for (int s = 0; s < KS; ++s) { xbh[s] = xah[s]; xbl[s] = xal[s]; }
            }
        }
        // ---- sigma head: hidden layer on the matrix pipe, the 1-wide output layer lane-local (on the fp32 activations)
        mcnx3_layer<W, SV, 0, KS, 1, PPW>(ring, smem, lane, ench, encl, xbh, xbl, xah, xal, bias_h + D * W, w2_h, dot,
                                            SAVE ? act_lane + (size_t)D * a.slot_bytes : nullptr, SAVE ? mask_lane + (size_t)D * a.mask_slot_words : nullptr);
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
        // ---- SH head: hidden layer (reads the same trunk output), then the 27 (32) coefficient rows
        mcnx3_layer<W, SV, 0, KS, 0, PPW>(ring, smem, lane, ench, encl, xbh, xbl, xah, xal, bias_h + (D + 1) * W, nullptr, dot,
                                            SAVE ? act_lane + (size_t)(D + 1) * a.slot_bytes : nullptr, SAVE ? mask_lane + (size_t)(D + 1) * a.mask_slot_words : nullptr);
        f32x16 acc;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(sbc2 + 8 * q + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[4 * q + e] = b[e];
        }
        {
            Mcn16Cursor cur;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                mcnx3_before_mfma<KS, PPW>(ring, cur, s);
                const unsigned o = cur.cur + (s & (MCNX3_SLABF - 1)) * 2048 + lane * 16;
                const u32x4_t a_h = *reinterpret_cast<const u32x4_t*>(smem + o);
                const u32x4_t a_l = *reinterpret_cast<const u32x4_t*>(smem + o + 1024);
                mcnx3_mfma3(acc, a_h, a_l, xah[s], xal[s]);
            }
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] *= 1.0f / SXW;
        if (SAVE) {          // the SH coefficients (bias included) for the backward's view-direction term: the fp32 accumulator tile
            char* e = reinterpret_cast<char*>(a.sh_ws) + (size_t)tile * 4096 + lane * 16;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                mcn16_ws_store(f32x4{acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]}, reinterpret_cast<f32x4*>(e + q * 1024));
        }
        // ---- per-sample epilogue: sigma, SH colour (model/net_utils.py:154-169), sigmoid.  Register 4q + e of this lane
        //      is SH row n = 8q + 4h + e = 9 c + i (colour c, basis i); the two lane halves hold complementary rows.
        float bas[9];
        mcn_sh_basis(dx, dy, dz, bas);
        float pre[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n0 = 8 * (r >> 2) + (r & 3), n1 = n0 + 4;
            const float b0 = n0 < MCN_NSH ? bas[n0 % 9] : 0.f, b1 = n1 < MCN_NSH ? bas[n1 % 9] : 0.f;
            const float contrib = acc[r] * (h ? b1 : b0);
            const int c0 = n0 / 9, c1 = n1 / 9;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const bool in0 = n0 < MCN_NSH && c0 == c, in1 = n1 < MCN_NSH && c1 == c;
                if (in0 && in1) pre[c] += contrib;
                else if (in0) pre[c] += h ? 0.f : contrib;
                else if (in1) pre[c] += h ? contrib : 0.f;
            }
        }
        const float sigma = (dot + __shfl_xor(dot, 32)) * (1.0f / MCNX3_SX) + bs2;
        f32x4 o;
        o[0] = sigma;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float pc = pre[c] + __shfl_xor(pre[c], 32);
            o[1 + c] = 1.0f / (1.0f + expf(-pc));
        }
        if (valid && h == 0) *reinterpret_cast<f32x4*>(a.out + (size_t)addr * 4) = o;
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
