// Fused NeRF MLP backward (activation-gradient chain), split-f16 ("f16x3") mode on the register-chain architecture
// (mcnerf_x3.h).  Same structure as mlp16_bwd.hip: a wave carries the gradients of its 32 samples through the transposed
// network in registers as (hi, lo) f16 fragment pairs (dX = W^T dY in the "sample on the lane" orientation, three MFMAs per
// k-step into one fp32 accumulator), the waves of a workgroup share the LDS ring that streams the transposed packed
// (hi, lo) weight pieces in consumption order.
//   sigmoid / SH backward (lane-local) -> sigma.0^T (kept as a (hi, lo) partial) -> sh.2^T -> sh.0^T (+ partial) ->
//   trunk D-1 .. 1 (the skip layer's and layer 0's encoded columns accumulate the encoded-input gradient in fp32) ->
//   encoding backward -> per-ray d o / d d (segmented wave reduction + fp32 atomics).
// Every pre-activation gradient is written fragment-major, hi plane then lo plane, to dy_ws / dsh_ws (operands of
// mlp_x3_dw.hip), scaled by the per-launch power of two SG (f16 range; derived from max|d_out|).
// Replaces autograd through model/net_block.py:22-33, 67-78 and model/mc_nerf.py:602, 635, 690-691.
#include "mcnerf_x3.h"

#define MCNX3_BWD_PREF_MINW 256
#include <cstdlib>

template <int W>
struct BwdX3Smem {
    static constexpr int WAVES = mcnx3_waves(W);
    static constexpr int oW2 = MCN16_RING * MCN16_SLAB * 1024;     // sigma.2 weight row [W] fp32
    static constexpr int oBarf = oW2 + W * 4;                      // BARF weights [10] (+ pad)
    static constexpr int MW = W >= 64 ? W / 64 : 1;                 // mask dwords per lane and slot
    static constexpr int oMask = oBarf + 16 * 4;                    // per wave: 3 buffers of [MW][64] dwords (ReLU bits, fetched ahead by LDS-DMA)
    static constexpr int oIdx = oMask + WAVES * 3 * MW * 256;       // per wave: the NEXT pass's (ray, sample) pairs [32][2] (LDS-DMA, one pass ahead)
    static constexpr int oIn = oIdx + WAVES * 256;                  // per wave (wide net): the next pass's per-sample inputs, 16 x [64 lanes] dwords (LDS-DMA gathers)
    static constexpr int total = oIn + (W >= 256 ? WAVES * 16 * 256 : 0);
};

// GEMM over one segment of NTILES output tiles x KSTEPS contraction steps (B fragment pairs `in`), software-pipelined like
// the forward (A pieces MCNX3_PF steps ahead, the epilogue of tile t one work item per MFMA gap of tile t + 1).
//   MODE 0: out = split(acc / SW)                               (a partial sum kept as a (hi, lo) pair)
//   MODE 1: out = mask(split(acc / SW)), saved                   (masked by the forward's ReLU bits `mk`)
//   MODE 2: as 1, with the accumulator started at the partial sum already in out
//   MODE 3: NTILES == 2: acc2[t] += ...                          (fp32 accumulators owned by the caller, no epilogue)
//   HI: only the hi plane of a saved dY is written, in the 16-bit modes' workspace layout (dtype 3)
template <int W, int KSTEPS, int NTILES, int MODE, int PPW, bool HI>
__device__ __forceinline__ void mcnx3_bwd_seg(Mcn16Ring& ring, char* smem, int lane,
                                              const u32x4_t (&inh)[W / 16], const u32x4_t (&inl)[W / 16],
                                              u32x4_t (&outh)[W / 16], u32x4_t (&outl)[W / 16],
                                              const unsigned (&mk)[W >= 64 ? W / 64 : 1], f32x16 (&acc2)[2], char* save_lane) {
    constexpr int KS = W / 16, F = NTILES * KSTEPS, G = 3 * KSTEPS;
    constexpr bool EPI = MODE != 3;
    constexpr int NIT = EPI ? (MODE == 0 ? 16 : 20) : 0;
    constexpr int START = (G >= NIT + 8) ? 3 : 0;
    constexpr int IPG = EPI ? (NIT + (G - START) - 1) / (G - START) : 1;
    constexpr int LASTA = EPI ? START + 14 / IPG : 0;             // gap of the last item that reads the accumulator (item 14)
    // (the four fragment stores of a tile one every SSTR gaps behind the word items where the tile has room, not back to back:
    //  mlp_x3_fwd.hip; the two-items-per-gap form of the forward measured 2 % slower here on the 128-wide net)
    constexpr int SBASE = START + 16, SSTR = (G - SBASE) / 4 > 0 ? (G - SBASE) / 4 : 1;
    constexpr bool STAG = (MODE == 1 || MODE == 2) && IPG == 1 && (G - SBASE) / 4 >= 2;
    constexpr int NITG = STAG ? 16 : NIT;
    constexpr int INIT_G = (G - 8) > LASTA ? (G - 8) : LASTA;     // the next tile's accumulator (= the set just drained) is initialised from here,
    constexpr int INIT_N = (INIT_G + 3 <= G - 1) ? 4 : 1;         // a quarter per gap where there is room
    Mcn16Cursor cur;
    u32x4_t afh[MCNX3_PF], afl[MCNX3_PF];
    f32x16 acc[2];
    float v0 = 0.f, v1 = 0.f;
    unsigned wkeep = 0u;
    // quarter q of the next tile's accumulator: the scaled partial sum (MODE 2) or zero
    auto acc_init = [&](f32x16& a, int t, int q) {
        if (MODE == 2) {
            const f16x8_t h8 = __builtin_bit_cast(f16x8_t, outh[2 * t + (q >> 1)]), l8 = __builtin_bit_cast(f16x8_t, outl[2 * t + (q >> 1)]);
#pragma unroll
            for (int e = 0; e < 4; ++e) a[4 * q + e] = ((float)h8[4 * (q & 1) + e] + (float)l8[4 * (q & 1) + e]) * MCNX3_SW;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) a[4 * q + e] = 0.f;
        }
    };
    auto item = [&](const f32x16& a, int t, int i) {
        if (i < 16) {
            const int p = i >> 1;
            const unsigned bits = MODE != 0 ? ((mk[t >> 1] >> (8 * (t & 1) + 7 - p)) & 0x00010001u) : 0u;
            if ((i & 1) == 0) {
                v0 = a[2 * p] * (1.0f / MCNX3_SW);
                v1 = a[2 * p + 1] * (1.0f / MCNX3_SW);
                const unsigned w = Mcn16T<false>::pack(v0, v1);
                wkeep = w;
                outh[2 * t + (p >> 2)][p & 3] = MODE != 0 ? mcn16_pkmul(w, bits) : w;
            } else {
                const unsigned w = Mcn16T<false>::pack(mcnx3_residual<0>(v0, wkeep), mcnx3_residual<1>(v1, wkeep));
                outl[2 * t + (p >> 2)][p & 3] = MODE != 0 ? mcn16_pkmul(w, bits) : w;
            }
        } else {
            const int k = i - 16, s = 2 * t + (k & 1);
            if (k < 2) mcn16_ws_store(outh[s], reinterpret_cast<u32x4_t*>(save_lane + s * 1024));
            else if (!HI) mcn16_ws_store(outl[s], reinterpret_cast<u32x4_t*>(save_lane + (KS + s) * 1024));      // (HI: the item keeps its gap, without the store)
        }
    };
    cur.cur = ring.next_off;
#pragma unroll
    for (int i = 0; i < MCNX3_PF; ++i)
        if (i < F) {
            afh[i] = *reinterpret_cast<const u32x4_t*>(smem + ring.next_off + i * 2048 + lane * 16);
            afl[i] = *reinterpret_cast<const u32x4_t*>(smem + ring.next_off + i * 2048 + 1024 + lane * 16);
        }
    if (EPI) {
#pragma unroll
        for (int q = 0; q < 4; ++q) acc_init(acc[0], 0, q);
    }
#pragma unroll
    for (int t = 0; t < NTILES; ++t) {
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            const int f = t * KSTEPS + s;
            mcnx3_before_mfma_spread<F, PPW>(ring, cur, f);
            const u32x4_t a_h = afh[f % MCNX3_PF], a_l = afl[f % MCNX3_PF];
            if (f + MCNX3_PF < F) {
                const unsigned o = mcnx3_frag_off(ring, cur, f, f + MCNX3_PF) + lane * 16;
                afh[f % MCNX3_PF] = *reinterpret_cast<const u32x4_t*>(smem + o);
                afl[f % MCNX3_PF] = *reinterpret_cast<const u32x4_t*>(smem + o + 1024);
            }
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                const int gap = 3 * s + g;
                if (EPI && t > 0 && gap >= START) {
#pragma unroll
                    for (int i = (gap - START) * IPG; i < (gap - START + 1) * IPG; ++i)
                        if (i < NITG) item(acc[(t - 1) & 1], t - 1, i);
                    if (STAG && gap >= SBASE && (gap - SBASE) / SSTR < 4 && (gap - SBASE) % SSTR == 0)
                        item(acc[(t - 1) & 1], t - 1, 16 + (gap - SBASE) / SSTR);
                }
                if (EPI && t + 1 < NTILES && gap >= INIT_G && gap < INIT_G + INIT_N) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (INIT_N == 1 || q == gap - INIT_G) acc_init(acc[(t + 1) & 1], t + 1, q);
                }
                mcnx3_gap_dma<F, PPW>(ring, 3 * f + g);
                __builtin_amdgcn_sched_barrier(0);
                if (EPI) acc[t & 1] = mcnx3_mfma(g == 0 ? a_l : a_h, g == 1 ? inl[s] : inh[s], acc[t & 1]);
                else acc2[t & 1] = mcnx3_mfma(g == 0 ? a_l : a_h, g == 1 ? inl[s] : inh[s], acc2[t & 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    mcnx3_layer_end<F, PPW>(ring);
    if (EPI) {
#pragma unroll
        for (int i = 0; i < NIT; ++i) item(acc[(NTILES - 1) & 1], NTILES - 1, i);
    }
}

template <int W, bool HI>
__global__ __launch_bounds__(64 * mcnx3_waves(W), mcnx3_waves(W) / 4) void mlp_x3_bwd_kernel(Mcn16BwdArgs a) {
    using SM = BwdX3Smem<W>;
    constexpr int PL = HI ? 1 : 2;                 // fragment planes per saved tile
    constexpr int WAVES = mcnx3_waves(W), ROWS = 32 * WAVES, PPW = 16 / WAVES;
    constexpr int NT = W / 32, KS = W / 16, MW = W >= 64 ? W / 64 : 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 31, h = lane >> 5;
    const int D = a.lay.depth, skip = a.lay.skip;
    const long long total = a.count ? (long long)min(*a.count, a.max_rows) : (long long)a.n_rays * a.S;
    if ((long long)blockIdx.x * ROWS >= total) return;

    float* sw2 = reinterpret_cast<float*>(smem + SM::oW2);
    for (int i = tid; i < W; i += 64 * WAVES) sw2[i] = a.params[a.lay.pWs2 + i];
    float* sbarf = reinterpret_cast<float*>(smem + SM::oBarf);
    if (tid < MCN_NFREQ) sbarf[tid] = a.barf_w[tid];
    // gradient scale: a power of two that puts max|d_out| of the launch near 2^4 (4096x headroom below the f16 maximum for
    // growth through the layers)
    const float gmax = __uint_as_float(*a.gmax_bits);
    const float sg = mcn16_grad_scale(gmax);
    const float inv_sg = 1.0f / sg;
    __syncthreads();

    // ---- the per-sample inputs of a pass are fetched ONE PASS AHEAD (mlp16_bwd.hip)
    struct In { int ray; float zg, jit; f32x4 o, go; float dx, dy, dz, ox, oy, oz; };
    auto gather = [&](int ray, int j) -> In {
        In r;
        r.ray = ray;
        r.zg = a.zgrid[j];
        r.jit = a.jitter ? a.jitter[ray] : 0.f;
        const size_t addr = (size_t)ray * a.S + j;
        r.o = *reinterpret_cast<const f32x4*>(a.out + addr * 4);
        r.go = *reinterpret_cast<const f32x4*>(a.d_out + addr * 4);
        r.dx = a.rays_d[ray * 3]; r.dy = a.rays_d[ray * 3 + 1]; r.dz = a.rays_d[ray * 3 + 2];
        r.ox = a.rays_o[ray * 3]; r.oy = a.rays_o[ray * 3 + 1]; r.oz = a.rays_o[ray * 3 + 2];
        return r;
    };
    auto row_of = [&](long long pass_) -> long long {           // this lane's row of a pass, clamped into the list
        const long long g_ = (pass_ * WAVES + wave) * 32 + m;
        return g_ < total ? g_ : total - 1;
    };
    // Wide net (one wave per SIMD, ReLU bits by LDS-DMA): the gathers go through LDS-DMA too, so that NOTHING in the pass loop is a
    // compiler-visible load -- a pass then never drains the wave's vector-memory queue (the workspace stores of a pass take
    // microseconds to retire; a vmcnt(0) per pass waits for all of them, with the matrix pipe idle).  Landing is guaranteed by
    // counted waits: in-order completion, and at least N younger ring pieces issued since (N_* below).
    constexpr bool PREFB = W >= MCNX3_BWD_PREF_MINW;
    constexpr int SLABS_FULL = (W / 32) * (W / 16) / MCNX3_SLABF;                       // ring slabs of one W x W segment
    constexpr int SLABS_ENC = (2 * (W / 16) + MCNX3_SLABF - 1) / MCNX3_SLABF;           // ... of a 2-tile encoded-column segment
    constexpr int N_IDX = PPW * (2 * SLABS_FULL - 1) < 63 ? PPW * (2 * SLABS_FULL - 1) : 63;      // index pairs: issued >= two W x W segments earlier
    constexpr int N_TOP = PPW * (SLABS_FULL + SLABS_ENC - 1) < 63 ? PPW * (SLABS_FULL + SLABS_ENC - 1) : 63;   // gathers: >= trunk layer 1 + layer 0's columns earlier
    const unsigned in_lds = (unsigned)reinterpret_cast<size_t>((mcn16_lds_ptr_t)smem) + SM::oIn + wave * (16 * 256);
    const float* in_rd = reinterpret_cast<const float*>(smem + SM::oIn + wave * (16 * 256)) + lane;
    auto gather_dma = [&](int ray, int j) {
        mcn16_dma4(a.zgrid + j, in_lds);
        if (a.jitter) mcn16_dma4(a.jitter + ray, in_lds + 256);
        const size_t addr = ((size_t)ray * a.S + j) * 4;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            mcn16_dma4(a.out + addr + c, in_lds + (2 + c) * 256);
            mcn16_dma4(a.d_out + addr + c, in_lds + (6 + c) * 256);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            mcn16_dma4(a.rays_d + ray * 3 + c, in_lds + (10 + c) * 256);
            mcn16_dma4(a.rays_o + ray * 3 + c, in_lds + (13 + c) * 256);
        }
    };
    auto read_in = [&](int ray) -> In {
        In r;
        r.ray = ray;
        r.zg = in_rd[0];
        r.jit = a.jitter ? in_rd[64] : 0.f;
        r.o = f32x4{in_rd[2 * 64], in_rd[3 * 64], in_rd[4 * 64], in_rd[5 * 64]};
        r.go = f32x4{in_rd[6 * 64], in_rd[7 * 64], in_rd[8 * 64], in_rd[9 * 64]};
        r.dx = in_rd[10 * 64]; r.dy = in_rd[11 * 64]; r.dz = in_rd[12 * 64];
        r.ox = in_rd[13 * 64]; r.oy = in_rd[14 * 64]; r.oz = in_rd[15 * 64];
        return r;
    };
    In cur;
    int ray_n = 0;                                              // (wide net) the coming pass's ray of this lane's row
    {
        const long long gc0 = row_of(blockIdx.x);
        int ray0, j0;
        if (a.idx) { const int2 rj = a.idx[gc0]; ray0 = rj.x; j0 = rj.y; }
        else { ray0 = (int)(gc0 / a.S); j0 = (int)(gc0 - (long long)ray0 * a.S); }
        if (PREFB) { gather_dma(ray0, j0); ray_n = ray0; }
        else cur = gather(ray0, j0);
    }

    Mcn16Ring ring;
    mcnx3_ring_start<PPW>(ring, smem, a.packed, a.stream_slabs, wave, lane);
    const float* w2_h = sw2 + 4 * h;
    const unsigned idx_lds = (unsigned)reinterpret_cast<size_t>((mcn16_lds_ptr_t)smem) + SM::oIdx + wave * 256;

    // ReLU bits: LDS-DMA into one of this wave's three buffers well ahead of their use on the wide net, ordinary loads one
    // layer ahead on the narrow ones (mlp16_bwd.hip).  Slot x <= D-1 lives in buffer (D + 1 - x) % 3; D in 0, D + 1 in 1.
    constexpr bool MASK_DMA = W >= MCNX3_BWD_PREF_MINW;
    const unsigned mlds = ring.lds_base + SM::oMask + wave * (3 * MW * 256);
    auto mask_issue = [&](const unsigned* mask_lane, int slot, int buf) {
        if (MASK_DMA) {
#pragma unroll
            for (int i = 0; i < MW; ++i) mcn16_dma4(mask_lane + (size_t)slot * a.mask_slot_words + i, mlds + (buf * MW + i) * 256);
        }
    };
    auto mask_read = [&](const unsigned* mask_lane, unsigned (&mk)[MW], int buf, int slot) {
#pragma unroll
        for (int i = 0; i < MW; ++i) {
            if (MASK_DMA) mk[i] = *reinterpret_cast<const unsigned*>(smem + SM::oMask + wave * (3 * MW * 256) + (buf * MW + i) * 256 + lane * 4);
            else mk[i] = mask_lane[(size_t)slot * a.mask_slot_words + i];
        }
    };
    auto buf_of = [&](int slot) { return (D + 1 - slot) % 3; };
    auto mask_lane_of = [&](long long pass_) { return a.mask_ws + ((size_t)(pass_ * WAVES + wave) * 64 + lane) * MW; };
    unsigned mk0_s[MW], mk0_c[MW], mk0_t[MW];                  // (narrow nets) the first three slots of the coming pass
    {
        const unsigned* ml0 = mask_lane_of(blockIdx.x);
        mask_issue(ml0, D, 0); mask_issue(ml0, D + 1, 1); mask_issue(ml0, D - 1, 2);
        if (!MASK_DMA) { mask_read(ml0, mk0_s, 0, D); mask_read(ml0, mk0_c, 1, D + 1); mask_read(ml0, mk0_t, 2, D - 1); }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }

    for (long long pass = blockIdx.x; pass * ROWS < total; pass += gridDim.x) {
        const long long tile = pass * WAVES + wave;
        const long long g = tile * 32 + m;
        const bool valid = g < total;
        const unsigned* mask_lane = mask_lane_of(pass);
        char* dy_lane = reinterpret_cast<char*>(a.dy_ws) + (size_t)tile * (PL * KS) * 1024 + lane * 16;
        // ---- per-sample prologue (lane-local): sigmoid and SH backward
        if (PREFB) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_TOP) : "memory");       // (the first pass: vmcnt(0) was waited for below the mask prefetch)
            cur = read_in(ray_n);
        }
        const int ray = cur.ray;
        unsigned mk_s[MW], mk_c[MW], mk_t[MW];
        float zv = cur.zg;
        if (a.jitter) zv = __fadd_rn(zv, cur.jit);
        const f32x4 o = cur.o;
        f32x4 go = cur.go;
        if (!valid) go = f32x4{0.f, 0.f, 0.f, 0.f};          // rows past the count contribute exactly zero everywhere
        const float x = cur.dx, y = cur.dy, z = cur.dz;
        float p[3];
        p[0] = __fadd_rn(cur.ox, __fmul_rn(x, zv));
        p[1] = __fadd_rn(cur.oy, __fmul_rn(y, zv));
        p[2] = __fadd_rn(cur.oz, __fmul_rn(z, zv));
        float bas[9];
        mcn_sh_basis(x, y, z, bas);
        float dpre[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) dpre[c] = go[1 + c] * o[1 + c] * (1.f - o[1 + c]) * sg;
        const float dsg = go[0] * sg;
        // dsh fragments (2 k-steps over the 32 padded sh.2 outputs): element (s, j) = column n = c(s,h,j); column 27 carries
        // d sigma for the sigma.2 weight gradient (the packed sh.2^T has zero rows there)
        u32x4_t dshh[2], dshl[2];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                float v[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int n0 = mcn16_chan(s, 0, 2 * d + u), n1 = n0 + 4;
                    const float va = n0 < MCN_NSH ? dpre[n0 / 9] * bas[n0 % 9] : (n0 == MCN_NSH ? dsg : 0.f);
                    const float vb = n1 < MCN_NSH ? dpre[n1 / 9] * bas[n1 % 9] : (n1 == MCN_NSH ? dsg : 0.f);
                    v[u] = h ? vb : va;
                }
                unsigned wh, wl;
                mcnx3_split2(v[0], v[1], wh, wl);
                dshh[s][d] = wh; dshl[s][d] = wl;
            }
        {
            char* e = reinterpret_cast<char*>(a.dsh_ws) + (size_t)tile * (PL * 2) * 1024 + lane * 16;
            mcn16_ws_store(dshh[0], reinterpret_cast<u32x4_t*>(e));
            mcn16_ws_store(dshh[1], reinterpret_cast<u32x4_t*>(e + 1024));
            if (!HI) {
                mcn16_ws_store(dshl[0], reinterpret_cast<u32x4_t*>(e + 2048));
                mcn16_ws_store(dshl[1], reinterpret_cast<u32x4_t*>(e + 3072));
            }
        }

        // (the first three mask slots and the per-sample inputs were fetched during the previous pass and waited for at its end)
        if (MASK_DMA) mask_read(mask_lane, mk_s, 0, D);
        else {
#pragma unroll
            for (int i = 0; i < MW; ++i) { mk_s[i] = mk0_s[i]; mk_c[i] = mk0_c[i]; mk_t[i] = mk0_t[i]; }
        }
        const long long pass_n = pass + gridDim.x;
        if (a.idx) {       // lane L fetches dword L of the next pass's 32 (ray, sample) pairs
            const long long gn = (pass_n * WAVES + wave) * 32 + (lane >> 1);
            mcn16_dma4(reinterpret_cast<const int*>(a.idx) + 2 * (gn < total ? gn : total - 1) + (lane & 1), idx_lds);
        }
        // (the forward's saved sh.2 outputs for the view-direction term: ordinary loads issued HERE and consumed at the end of the
        //  pass, behind hundreds of younger operations -- whatever count hipcc waits for there is long satisfied)
        const bool want_rays = a.d_rays_o || a.d_rays_d;
        f32x4 shs[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) shs[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (PREFB && want_rays) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                shs[q] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(a.sh_ws) + (size_t)tile * 4096 + q * 1024 + lane * 16);
        }
        u32x4_t xah[KS], xal[KS], xbh[KS], xbl[KS];
        f32x16 denc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) denc[t][e] = 0.f;
        // ---- dY of sigma.0 = d sigma * w_sigma2, masked by the sigma hidden layer's ReLU bits (outer product, no GEMM)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const f32x4 wa = *reinterpret_cast<const f32x4*>(w2_h + 16 * s), wb = *reinterpret_cast<const f32x4*>(w2_h + 16 * s + 8);
            const int t = s >> 1;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const float va = (d < 2 ? wa[2 * d] : wb[2 * d - 4]) * dsg, vb = (d < 2 ? wa[2 * d + 1] : wb[2 * d - 3]) * dsg;
                const int i = 4 * (s & 1) + d;
                const unsigned bits = (mk_s[t >> 1] >> (8 * (t & 1) + 7 - i)) & 0x00010001u;
                unsigned wh, wl;
                mcnx3_split2(va, vb, wh, wl);
                xah[s][d] = mcn16_pkmul(wh, bits);
                xal[s][d] = mcn16_pkmul(wl, bits);
            }
            mcn16_ws_store(xah[s], reinterpret_cast<u32x4_t*>(dy_lane + (size_t)D * a.slot_bytes + s * 1024));
            if (!HI) mcn16_ws_store(xal[s], reinterpret_cast<u32x4_t*>(dy_lane + (size_t)D * a.slot_bytes + (KS + s) * 1024));
        }
        // ---- sigma.0^T (partial, (hi, lo)) ; sh.2^T -> dY of sh.0 ; sh.0^T + partial -> dY_{D-1}
        if (D >= 2) mask_issue(mask_lane, D - 2, buf_of(D - 2));          // buffer 0 is free (the sigma path is done)
        unsigned mpend[MW];
        if (!MASK_DMA && D >= 2) mask_read(mask_lane, mpend, 0, D - 2);
        mcnx3_bwd_seg<W, KS, NT, 0, PPW, HI>(ring, smem, lane, xah, xal, xbh, xbl, mk_s, denc, nullptr);
        if (MASK_DMA) mask_read(mask_lane, mk_c, 1, D + 1);
        {
            u32x4_t dih[KS], dil[KS];
            dih[0] = dshh[0]; dih[1] = dshh[1]; dil[0] = dshl[0]; dil[1] = dshl[1];
            mcnx3_bwd_seg<W, 2, NT, 1, PPW, HI>(ring, smem, lane, dih, dil, xah, xal, mk_c, denc, dy_lane + (size_t)(D + 1) * a.slot_bytes);
        }
        if (D >= 3) mask_issue(mask_lane, D - 3, buf_of(D - 3));          // buffer 1 is free (sh.2^T is done)
        if (MASK_DMA) mask_read(mask_lane, mk_t, 2, D - 1);
        mcnx3_bwd_seg<W, KS, NT, 2, PPW, HI>(ring, smem, lane, xah, xal, xbh, xbl, mk_t, denc, dy_lane + (size_t)(D - 1) * a.slot_bytes);
        // ---- trunk, last layer to first, two layers per trip (dY_l in xb -> dY_{l-1} in xa -> dY_{l-2} in xb: no copies between layers)
        auto trunk_masks = [&](int l) {
            if (MASK_DMA) {
                if (l >= 3) mask_issue(mask_lane, l - 3, buf_of(l - 3));  // the buffer of slot l (previous segment) is free
                mask_read(mask_lane, mk_t, buf_of(l - 1), l - 1);
            } else {
#pragma unroll
                for (int i = 0; i < MW; ++i) mk_t[i] = mpend[i];
                if (l >= 2) mask_read(mask_lane, mpend, 0, l - 2);
            }
            if (l == 1 && pass_n * ROWS < total) {      // (workgroup-uniform) every buffer is free from here on: the next pass's first three slots
                const unsigned* mln = mask_lane_of(pass_n);
                mask_issue(mln, D, 0); mask_issue(mln, D + 1, 1); mask_issue(mln, D - 1, 2);
                if (!MASK_DMA) { mask_read(mln, mk0_s, 0, D); mask_read(mln, mk0_c, 1, D + 1); mask_read(mln, mk0_t, 2, D - 1); }
            }
        };
        auto next_rows = [&]() {     // (wide net) the coming pass's rows: index pair from LDS, gathers by LDS-DMA
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_IDX) : "memory");
            int rn, jn;
            if (a.idx) {
                const int2 rj = *reinterpret_cast<const int2*>(smem + SM::oIdx + wave * 256 + m * 8);
                rn = rj.x; jn = rj.y;
            } else {
                const long long gn = row_of(pass_n);
                rn = (int)(gn / a.S); jn = (int)(gn - (long long)rn * a.S);
            }
            gather_dma(rn, jn);
            ray_n = rn;
        };
        for (int l = D - 1; l >= 1; l -= 2) {
            if (PREFB && l == 1) next_rows();
            trunk_masks(l);
            if (l == skip) mcnx3_bwd_seg<W, KS, 2, 3, PPW, HI>(ring, smem, lane, xbh, xbl, xah, xal, mk_t, denc, nullptr);      // encoded columns of the skip layer
            mcnx3_bwd_seg<W, KS, NT, 1, PPW, HI>(ring, smem, lane, xbh, xbl, xah, xal, mk_t, denc, dy_lane + (size_t)(l - 1) * a.slot_bytes);
            if (l - 1 >= 1) {
                if (PREFB && l - 1 == 1) next_rows();
                trunk_masks(l - 1);
                if (l - 1 == skip) mcnx3_bwd_seg<W, KS, 2, 3, PPW, HI>(ring, smem, lane, xah, xal, xbh, xbl, mk_t, denc, nullptr);
                mcnx3_bwd_seg<W, KS, NT, 1, PPW, HI>(ring, smem, lane, xah, xal, xbh, xbl, mk_t, denc, dy_lane + (size_t)(l - 2) * a.slot_bytes);
            } else {               // an even trunk depth ends in xa: one copy per pass
#pragma unroll
                for (int s = 0; s < KS; ++s) { xbh[s] = xah[s]; xbl[s] = xal[s]; }
            }
        }
        // ---- the next pass's rows: index pair from LDS (DMA'd during this pass's prologue), gathers land under the last GEMM and the epilogue
        In nxt = cur;
        if (PREFB && D < 2) {        // (no trunk layer to hide the gathers behind: issue them here, drained by the settle below)
            next_rows();
        }
        if (!PREFB) {
            int rn, jn;
            if (a.idx) {
                const int2 rj = *reinterpret_cast<const int2*>(smem + SM::oIdx + wave * 256 + m * 8);
                rn = rj.x; jn = rj.y;
            } else {
                const long long gn = row_of(pass_n);
                rn = (int)(gn / a.S); jn = (int)(gn - (long long)rn * a.S);
            }
            nxt = gather(rn, jn);
        }
        // (narrow nets: the saved sh.2 outputs for the view-direction term are fetched under the last GEMM)
        if (!PREFB && want_rays) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                shs[q] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(a.sh_ws) + (size_t)tile * 4096 + q * 1024 + lane * 16);
        }
        mcnx3_bwd_seg<W, KS, 2, 3, PPW, HI>(ring, smem, lane, xbh, xbl, xah, xal, mk_t, denc, nullptr);       // layer 0: encoded columns
        auto settle = [&]() {
            if (PREFB && D >= 2) return;     // (wide net: nothing to wait for here -- the gathers are waited for at the top of the coming pass)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("" : "+v"(nxt.ray), "+v"(nxt.zg), "+v"(nxt.jit), "+v"(nxt.o), "+v"(nxt.go));
            asm volatile("" : "+v"(nxt.dx), "+v"(nxt.dy), "+v"(nxt.dz), "+v"(nxt.ox), "+v"(nxt.oy), "+v"(nxt.oz));
            if (!MASK_DMA) {
#pragma unroll
                for (int i = 0; i < MW; ++i) asm volatile("" : "+v"(mk0_s[i]), "+v"(mk0_c[i]), "+v"(mk0_t[i]));
            }
        };

        // ---- encoding backward -> d position; SH view-direction term; per-ray reduction
        if (want_rays) {
            // this lane holds d enc (x SW SG) of channels 32 te + 8 q + 4 h + e (register 4 q + e of denc[te]); channel 3 + 20 a + f is
            // w_f sin(2^f x_a), + 10: w_f cos(2^f x_a)  ->  d x_a += 2^f w_f (cos dsin - sin dcos), sin / cos fp32-accurate per octave
            auto dch = [&](int ch) -> float {              // d enc of channel ch if this lane half holds it, else 0
                const int hh = (ch >> 2) & 1;
                const float v = denc[ch >> 5][4 * ((ch >> 3) & 3) + (ch & 3)];
                return (h == hh) ? v : 0.f;
            };
            float dpos[3];
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
                float acc = dch(ax);
                float S[MCN_NFREQ], C[MCN_NFREQ];
                mcn_sincos_octaves<MCN_NFREQ>(p[ax], S, C);       // (the forward's values: mcnerf_x3.h)
#pragma unroll
                for (int f = 0; f < MCN_NFREQ; ++f) {
                    const float k = (float)(1 << f) * sbarf[f];
                    acc = fmaf(k * C[f], dch(3 + 20 * ax + f), acc);
                    acc = fmaf(-k * S[f], dch(3 + 20 * ax + 10 + f), acc);
                }
                dpos[ax] = acc * (1.0f / MCNX3_SW);
            }
            // SH term: d pre_c / d dir = sum_i sh[9c + i] d basis_i / d dir with the forward's saved sh.2 outputs (model/net_utils.py:154-169)
            float ddir[3] = {0.f, 0.f, 0.f};
            {
                const float C1 = 0.4886025119029199f, C20 = 1.0925484305920792f, C22 = 0.31539156525252005f, C24 = 0.5462742152960396f;
                // derivative of basis i wrt (x, y, z)
                const float gx[9] = {0.f, 0.f, 0.f, -C1, C20 * y, 0.f, -2.f * C22 * x, -C20 * z, 2.f * C24 * x};
                const float gy[9] = {0.f, -C1, 0.f, 0.f, C20 * x, -C20 * z, -2.f * C22 * y, 0.f, -2.f * C24 * y};
                const float gz[9] = {0.f, 0.f, C1, 0.f, 0.f, -C20 * y, 4.f * C22 * z, -C20 * x, 0.f};
#pragma unroll
                for (int r = 0; r < 16; ++r) {           // register r of the saved tile = SH row 8 (r >> 2) + 4 h + (r & 3)
                    const float v = shs[r >> 2][r & 3];
                    const int n0 = 8 * (r >> 2) + (r & 3), n1 = n0 + 4;
                    const float kx = h ? (n1 < MCN_NSH ? dpre[n1 / 9] * gx[n1 % 9] : 0.f) : (n0 < MCN_NSH ? dpre[n0 / 9] * gx[n0 % 9] : 0.f);
                    const float ky = h ? (n1 < MCN_NSH ? dpre[n1 / 9] * gy[n1 % 9] : 0.f) : (n0 < MCN_NSH ? dpre[n0 / 9] * gy[n0 % 9] : 0.f);
                    const float kz = h ? (n1 < MCN_NSH ? dpre[n1 / 9] * gz[n1 % 9] : 0.f) : (n0 < MCN_NSH ? dpre[n0 / 9] * gz[n0 % 9] : 0.f);
                    ddir[0] = fmaf(kx, v, ddir[0]); ddir[1] = fmaf(ky, v, ddir[1]); ddir[2] = fmaf(kz, v, ddir[2]);
                }
            }
            float red[6];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float dp = dpos[c] + __shfl_xor(dpos[c], 32);
                const float dd = ddir[c] + __shfl_xor(ddir[c], 32);
                red[c] = dp * inv_sg;                              // d origin
                red[3 + c] = (dp * zv + dd) * inv_sg;              // d direction: through x = o + d z, plus the SH term
            }
            settle();
            // segmented inclusive scan over the 32 samples: a run = consecutive rows of one ray (the rows of a ray are contiguous in
            // the selection order; after the random cap, model/mc_nerf.py:630-632, they need not be: equal keys that are not
            // adjacent are separate runs, so the scan carries head flags instead of comparing keys at a distance).  The last
            // row of each run adds the run's sum with 6 atomics instead of 6 per sample.
            const int rkey = valid ? ray : -1;
            const int rprev = __shfl_up(rkey, 1, 32);
            int head = (m == 0 || rprev != rkey) ? 1 : 0;
#pragma unroll
            for (int off = 1; off < 32; off <<= 1) {
                const int hup = __shfl_up(head, off, 32);
                const bool take = (m >= off) && !head;
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    const float up = __shfl_up(red[c], off, 32);
                    red[c] += take ? up : 0.f;
                }
                if (m >= off) head |= hup;
            }
            const int rnext = __shfl_down(rkey, 1, 32);
            if (h == 0 && valid && (m == 31 || rnext != rkey)) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    if (a.d_rays_o) atomicAdd(a.d_rays_o + ray * 3 + c, red[c]);
                    if (a.d_rays_d) atomicAdd(a.d_rays_d + ray * 3 + c, red[3 + c]);
                }
            }
        }
        if (!want_rays) settle();
        if (!PREFB) cur = nxt;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int W>
static hipError_t launch_bwd_x3(const Mcn16BwdArgs& a, long long max_rows, hipStream_t st) {
    using SM = BwdX3Smem<W>;
    constexpr int WAVES = mcnx3_waves(W), ROWS = 32 * WAVES;
    int dev = 0, cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
    long long passes = (max_rows + ROWS - 1) / ROWS;
    if (passes <= 0) return hipSuccess;
    int grid = (int)(passes < cus ? passes : cus);
    void (*kern)(Mcn16BwdArgs) = a.bf16 == 3 ? mlp_x3_bwd_kernel<W, true> : mlp_x3_bwd_kernel<W, false>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, SM::total);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WAVES), SM::total, st, a);
    return hipGetLastError();
}

hipError_t mcnx3_launch_bwd(const Mcn16BwdArgs& a, hipStream_t st) {
    const long long max_rows = a.count ? (long long)a.max_rows : (long long)a.n_rays * a.S;
    switch (a.lay.width) {
        case 256: return launch_bwd_x3<256>(a, max_rows, st);
        case 128: return launch_bwd_x3<128>(a, max_rows, st);
        case 64:  return launch_bwd_x3<64>(a, max_rows, st);
        case 32:  return launch_bwd_x3<32>(a, max_rows, st);
    }
    return hipErrorInvalidValue;
}
