// Fused NeRF MLP backward (activation-gradient chain), single-pass 16-bit MFMA mode (mcnerf_16.h).  Same structure
// as mlp16_fwd.hip: a wave carries the gradients of its 32 samples through the transposed network in registers
// (dX = W^T dY in the "sample on the lane" orientation, so a masked, converted accumulator tile IS the next GEMM's
// B fragment), the 8 waves share the LDS ring that streams the transposed packed weights in consumption order.
//   sigmoid / SH backward (lane-local) -> sigma.0^T (kept as a 16-bit partial) -> sh.2^T -> sh.0^T (+ partial) ->
//   trunk D-1 .. 1 (the skip layer's and layer 0's encoded columns accumulate the encoded-input gradient) ->
//   encoding backward -> per-ray d o / d d (segmented wave reduction + fp32 atomics).
// Every pre-activation gradient is written fragment-major to dy_ws / dsh_ws (operands of mlp16_dw.hip), scaled by the
// per-launch power of two SG (f16 range; derived from max|d_out| as in the split-f16 mode).
// Replaces autograd through model/net_block.py:22-33, 67-78 and model/mc_nerf.py:602, 635, 690-691.
#include "mcnerf_16.h"


template <int W>
struct Bwd16Smem {
    static constexpr int oW2 = MCN16_RING * MCN16_SLAB * 1024;     // sigma.2 weight row [W] fp32
    static constexpr int oBarf = oW2 + W * 4;                      // BARF weights [10] (+ pad)
    static constexpr int MW = W >= 64 ? W / 64 : 1;                 // mask dwords per lane and slot
    static constexpr int oMask = oBarf + 16 * 4;                    // per wave: 3 buffers of [MW][64] dwords (ReLU bits, fetched ahead by LDS-DMA)
    static constexpr int oIdx = oMask + MCN16_WAVES * 3 * MW * 256; // per wave: the NEXT pass's (ray, sample) pairs [32][2] (LDS-DMA, one pass ahead)
    static constexpr int total = oIdx + MCN16_WAVES * 256;
};

// GEMM over one segment of NTILES output tiles x KSTEPS contraction steps (B fragments `in`), software-pipelined like
// the forward (A fragments MCN16_PF steps ahead, epilogue of tile t between the MFMAs of tile t + 1).
//   MODE 0: out[2t], out[2t+1] = pack(acc)                      (a partial sum kept in 16 bit)
//   MODE 1: out = mask(pack(acc)), saved                         (masked by the forward's ReLU bits `mk`)
//   MODE 2: as 1, with the accumulator started at the partial sum already in out[2t], out[2t+1]
//   MODE 3: NTILES == 2: acc2[t] += ...                          (fp32 accumulators owned by the caller, no epilogue)
template <int W, bool BF, int KSTEPS, int NTILES, int MODE, int NOUT = W / 16>
__device__ __forceinline__ void mcn16_bwd_seg(Mcn16Ring& ring, char* smem, int lane, const u32x4_t (&in)[W / 16], u32x4_t (&out)[NOUT],
                                              const unsigned (&mk)[W >= 64 ? W / 64 : 1], f32x16 (&acc2)[2], char* save_lane) {
    using T = Mcn16T<BF>;
    constexpr int F = NTILES * KSTEPS;
    constexpr bool EPI = MODE != 3;
    constexpr int NSL = EPI ? (MODE == 0 ? 8 : 10) : 0;
    constexpr int START = (KSTEPS >= NSL + 3) ? 2 : 0;
    constexpr int SPS = EPI ? (NSL + KSTEPS - START - 1) / (KSTEPS - START) : 1;
    constexpr int LAST = EPI ? START + (NSL + SPS - 1) / SPS - 1 : 0;
    constexpr int INIT_AT = (KSTEPS - 4) > LAST ? (KSTEPS - 4) : LAST;
    Mcn16Cursor cur;
    u32x4_t af[MCN16_PF];
    f32x16 acc[2];
    u32x4_t o0, o1;
    auto acc_init = [&](f32x16& a, int t) {
        if (MODE == 2) {
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                a[2 * d] = T::lo(out[2 * t][d]); a[2 * d + 1] = T::hi(out[2 * t][d]);
                a[8 + 2 * d] = T::lo(out[2 * t + 1][d]); a[8 + 2 * d + 1] = T::hi(out[2 * t + 1][d]);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) a[e] = 0.f;
        }
    };
    auto epi_slice = [&](const f32x16& a, int t, int i) {
        if (i < 8) {
            unsigned w = T::pack(a[2 * i], a[2 * i + 1]);
            if (MODE != 0) w = mcn16_pkmul(w, (mk[t >> 1] >> (8 * (t & 1) + 7 - i)) & 0x00010001u);
            if (i < 4) o0[i] = w; else o1[i - 4] = w;
            if (i == 7) { out[2 * t] = o0; out[2 * t + 1] = o1; }
        } else if (MODE != 0) {
            if (i == 8) mcn16_ws_store(o0, reinterpret_cast<u32x4_t*>(save_lane + (2 * t) * 1024));
            else mcn16_ws_store(o1, reinterpret_cast<u32x4_t*>(save_lane + (2 * t + 1) * 1024));
        }
    };
    cur.cur = ring.next_off;
#pragma unroll
    for (int i = 0; i < MCN16_PF; ++i)
        if (i < F) af[i] = *reinterpret_cast<const u32x4_t*>(smem + ring.next_off + i * 1024 + lane * 16);
    if (EPI) acc_init(acc[0], 0);
#pragma unroll
    for (int t = 0; t < NTILES; ++t) {
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            const int f = t * KSTEPS + s;
            mcn16_before_mfma<F>(ring, smem, cur, f);
            const u32x4_t a_now = af[f % MCN16_PF];
            if (f + MCN16_PF < F)
                af[f % MCN16_PF] = *reinterpret_cast<const u32x4_t*>(smem + mcn16_frag_off(ring, cur, f, f + MCN16_PF) + lane * 16);
            if (EPI && t > 0 && s >= START) {
#pragma unroll
                for (int i = (s - START) * SPS; i < (s - START + 1) * SPS; ++i)
                    if (i < NSL) epi_slice(acc[(t - 1) & 1], t - 1, i);
            }
            if (EPI && s == INIT_AT && t + 1 < NTILES) acc_init(acc[(t + 1) & 1], t + 1);
            __builtin_amdgcn_sched_barrier(0);
            if (EPI) acc[t & 1] = T::mfma(a_now, in[s], acc[t & 1]);
            else acc2[t & 1] = T::mfma(a_now, in[s], acc2[t & 1]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (EPI) {
#pragma unroll
        for (int i = 0; i < NSL; ++i) epi_slice(acc[(NTILES - 1) & 1], NTILES - 1, i);
    }
}

template <int W, bool BF>
__global__ __launch_bounds__(64 * MCN16_WAVES, 2) void mlp16_bwd_kernel(Mcn16BwdArgs a) {
    using T = Mcn16T<BF>;
    using SM = Bwd16Smem<W>;
    constexpr int NT = W / 32, KS = W / 16, MW = W >= 64 ? W / 64 : 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 31, h = lane >> 5;
    const int D = a.lay.depth, skip = a.lay.skip;
    const long long total = a.count ? (long long)min(*a.count, a.max_rows) : (long long)a.n_rays * a.S;
    if ((long long)blockIdx.x * MCN16_ROWS >= total) return;

    float* sw2 = reinterpret_cast<float*>(smem + SM::oW2);
    for (int i = tid; i < W; i += 64 * MCN16_WAVES) sw2[i] = a.params[a.lay.pWs2 + i];
    float* sbarf = reinterpret_cast<float*>(smem + SM::oBarf);
    if (tid < MCN_NFREQ) sbarf[tid] = a.barf_w[tid];
    // gradient scale: a power of two that puts max|d_out| of the launch near 2^4 (4096x headroom below the f16 maximum
    // for growth through the layers); bf16 has fp32's range but shares the arithmetic
    const float gmax = __uint_as_float(*a.gmax_bits);
    const float sg = mcn16_grad_scale(gmax);
    const float inv_sg = 1.0f / sg;
    __syncthreads();

    // ---- the per-sample inputs of a pass are fetched ONE PASS AHEAD (software pipeline over the persistent loop): a pass
    // starts with two dependent gathers (index pair -> ray / output rows) behind every store of the previous pass in the
    // in-order vmcnt queue, 15.8 k of the 150 k cycles of a pass when taken at the top (in-kernel stamps, NOTES.md §3.1).
    // The index pairs of pass p + 1 go to LDS by DMA right after the prologue of pass p; the row gathers are issued
    // when the GEMMs of pass p are done and land under its encoding backward.
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
        const long long g_ = (pass_ * MCN16_WAVES + wave) * 32 + m;
        return g_ < total ? g_ : total - 1;
    };
    In cur;
    {
        const long long gc0 = row_of(blockIdx.x);
        int ray0, j0;
        if (a.idx) { const int2 rj = a.idx[gc0]; ray0 = rj.x; j0 = rj.y; }
        else { ray0 = (int)(gc0 / a.S); j0 = (int)(gc0 - (long long)ray0 * a.S); }
        cur = gather(ray0, j0);
    }

    Mcn16Ring ring;
    mcn16_ring_start(ring, smem, a.packed, a.stream_slabs, wave, lane);
    const float* w2_h = sw2 + 4 * h;
    const unsigned idx_lds = (unsigned)reinterpret_cast<size_t>((mcn16_lds_ptr_t)smem) + SM::oIdx + wave * 256;

    // ReLU bits of a slot are fetched by LDS-DMA into one of this wave's three buffers well ahead of their use (an
    // ordinary load in the layer loop makes hipcc wait vmcnt(0) at its use: a full drain of the weight ring and of
    // the workspace stores once per layer).  Slot x <= D-1 lives in buffer (D + 1 - x) % 3; D in 0, D + 1 in 1.
    // Issue points are at least two segments ahead, so the counted ring waits in between cover their landing; the
    // first three slots of a pass (D, D + 1, D - 1) are issued during the PREVIOUS pass's epilogue.
    // Narrower nets (short layers: the drain is cheap, the extra DMA instructions are not: measured 0.75 vs 0.83 ms on
    // the 4x128 net) use ordinary loads into a register set (`mpend`) one layer ahead of their use, the first three slots
    // one pass ahead (mk0_*).
    constexpr bool MASK_DMA = W >= 256;
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
    auto mask_lane_of = [&](long long pass_) { return a.mask_ws + ((size_t)(pass_ * MCN16_WAVES + wave) * 64 + lane) * MW; };
    unsigned mk0_s[MW], mk0_c[MW], mk0_t[MW];                  // (narrow nets) the first three slots of the coming pass
    {
        const unsigned* ml0 = mask_lane_of(blockIdx.x);
        mask_issue(ml0, D, 0); mask_issue(ml0, D + 1, 1); mask_issue(ml0, D - 1, 2);
        if (!MASK_DMA) { mask_read(ml0, mk0_s, 0, D); mask_read(ml0, mk0_c, 1, D + 1); mask_read(ml0, mk0_t, 2, D - 1); }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }

    for (long long pass = blockIdx.x; pass * MCN16_ROWS < total; pass += gridDim.x) {
        const long long tile = pass * MCN16_WAVES + wave;
        const long long g = tile * 32 + m;
        const bool valid = g < total;
        const unsigned* mask_lane = a.mask_ws + ((size_t)tile * 64 + lane) * MW;
        char* dy_lane = reinterpret_cast<char*>(a.dy_ws) + (size_t)tile * KS * 1024 + lane * 16;
        // ---- per-sample prologue (lane-local): sigmoid and SH backward
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
        u32x4_t dshf[2];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                float v[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int n0 = mcn16_chan(s, 0, 2 * d + u), n1 = n0 + 4;
                    const float v0 = n0 < MCN_NSH ? dpre[n0 / 9] * bas[n0 % 9] : (n0 == MCN_NSH ? dsg : 0.f);
                    const float v1 = n1 < MCN_NSH ? dpre[n1 / 9] * bas[n1 % 9] : (n1 == MCN_NSH ? dsg : 0.f);
                    v[u] = h ? v1 : v0;
                }
                dshf[s][d] = T::pack(v[0], v[1]);
            }
        {
            char* e = reinterpret_cast<char*>(a.dsh_ws) + (size_t)tile * 2 * 1024 + lane * 16;
            mcn16_ws_store(dshf[0], reinterpret_cast<u32x4_t*>(e));
            mcn16_ws_store(dshf[1], reinterpret_cast<u32x4_t*>(e + 1024));
        }

        // (the first three mask slots and the per-sample inputs were fetched during the previous pass and waited for at its end)
        if (MASK_DMA) mask_read(mask_lane, mk_s, 0, D);
        else {
#pragma unroll
            for (int i = 0; i < MW; ++i) { mk_s[i] = mk0_s[i]; mk_c[i] = mk0_c[i]; mk_t[i] = mk0_t[i]; }
        }
        const long long pass_n = pass + gridDim.x;
        if (a.idx) {       // lane L fetches dword L of the next pass's 32 (ray, sample) pairs
            const long long gn = (pass_n * MCN16_WAVES + wave) * 32 + (lane >> 1);
            mcn16_dma4(reinterpret_cast<const int*>(a.idx) + 2 * (gn < total ? gn : total - 1) + (lane & 1), idx_lds);
        }
        u32x4_t xa[KS], xb[KS];
        f32x16 denc[2];
        u32x4_t dencp[4];            // the skip layer's share of the encoded-input gradient, parked in 16 bit until layer 0
#pragma unroll
        for (int i = 0; i < 4; ++i) dencp[i] = u32x4_t{0u, 0u, 0u, 0u};
        // ---- dY of sigma.0 = d sigma * w_sigma2, masked by the sigma hidden layer's ReLU bits (outer product, no GEMM)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const f32x4 wa = *reinterpret_cast<const f32x4*>(w2_h + 16 * s), wb = *reinterpret_cast<const f32x4*>(w2_h + 16 * s + 8);
            const int t = s >> 1;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const float v0 = (d < 2 ? wa[2 * d] : wb[2 * d - 4]) * dsg, v1 = (d < 2 ? wa[2 * d + 1] : wb[2 * d - 3]) * dsg;
                const int i = 4 * (s & 1) + d;
                xa[s][d] = mcn16_pkmul(T::pack(v0, v1), (mk_s[t >> 1] >> (8 * (t & 1) + 7 - i)) & 0x00010001u);
            }
            mcn16_ws_store(xa[s], reinterpret_cast<u32x4_t*>(dy_lane + (size_t)D * a.slot_bytes + s * 1024));
        }
        // ---- sigma.0^T (partial, 16 bit) ; sh.2^T -> dY of sh.0 ; sh.0^T + partial -> dY_{D-1}
        if (D >= 2) mask_issue(mask_lane, D - 2, buf_of(D - 2));          // buffer 0 is free (the sigma path is done)
        unsigned mpend[MW];
        if (!MASK_DMA && D >= 2) mask_read(mask_lane, mpend, 0, D - 2);
        mcn16_bwd_seg<W, BF, KS, NT, 0>(ring, smem, lane, xa, xb, mk_s, denc, nullptr);
        if (MASK_DMA) mask_read(mask_lane, mk_c, 1, D + 1);
        {
            u32x4_t dsh_in[KS];
            dsh_in[0] = dshf[0]; dsh_in[1] = dshf[1];
            mcn16_bwd_seg<W, BF, 2, NT, 1>(ring, smem, lane, dsh_in, xa, mk_c, denc, dy_lane + (size_t)(D + 1) * a.slot_bytes);
        }
        if (D >= 3) mask_issue(mask_lane, D - 3, buf_of(D - 3));          // buffer 1 is free (sh.2^T is done)
        if (MASK_DMA) mask_read(mask_lane, mk_t, 2, D - 1);
        mcn16_bwd_seg<W, BF, KS, NT, 2>(ring, smem, lane, xa, xb, mk_t, denc, dy_lane + (size_t)(D - 1) * a.slot_bytes);
        // ---- trunk, last layer to first: xb = dY_l
        for (int l = D - 1; l >= 1; --l) {
#pragma unroll
            for (int s = 0; s < KS; ++s) xa[s] = xb[s];
            if (MASK_DMA) {
                if (l >= 3) mask_issue(mask_lane, l - 3, buf_of(l - 3));  // the buffer of slot l (previous segment) is free
                mask_read(mask_lane, mk_t, buf_of(l - 1), l - 1);
            } else {
#pragma unroll
                for (int i = 0; i < MW; ++i) mk_t[i] = mpend[i];
                if (l >= 2) mask_read(mask_lane, mpend, 0, l - 2);
            }
            if (l == 1 && pass_n * MCN16_ROWS < total) {      // (workgroup-uniform) every buffer is free from here on: the next pass's first three slots
                const unsigned* mln = mask_lane_of(pass_n);
                mask_issue(mln, D, 0); mask_issue(mln, D + 1, 1); mask_issue(mln, D - 1, 2);
                if (!MASK_DMA) { mask_read(mln, mk0_s, 0, D); mask_read(mln, mk0_c, 1, D + 1); mask_read(mln, mk0_t, 2, D - 1); }
            }
            if (l == skip) mcn16_bwd_seg<W, BF, KS, 2, 0, 4>(ring, smem, lane, xa, dencp, mk_t, denc, nullptr);
            mcn16_bwd_seg<W, BF, KS, NT, 1>(ring, smem, lane, xa, xb, mk_t, denc, dy_lane + (size_t)(l - 1) * a.slot_bytes);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                denc[t][2 * d] = T::lo(dencp[2 * t][d]); denc[t][2 * d + 1] = T::hi(dencp[2 * t][d]);
                denc[t][8 + 2 * d] = T::lo(dencp[2 * t + 1][d]); denc[t][8 + 2 * d + 1] = T::hi(dencp[2 * t + 1][d]);
            }
        // ---- the next pass's rows: index pair from LDS (DMA'd during this pass's prologue), gathers land under the last GEMM and the epilogue (memory latency under this load is several us)
        In nxt;
        {
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
        // (the saved sh.2 outputs for the view-direction term are fetched under the last GEMM)
        const bool want_rays = a.d_rays_o || a.d_rays_d;
        u32x4_t shs0 = {0u, 0u, 0u, 0u}, shs1 = {0u, 0u, 0u, 0u};
        if (want_rays) {
            shs0 = *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const char*>(a.sh_ws) + (size_t)tile * 2 * 1024 + lane * 16);
            shs1 = *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const char*>(a.sh_ws) + (size_t)tile * 2 * 1024 + 1024 + lane * 16);
        }
        mcn16_bwd_seg<W, BF, KS, 2, 3>(ring, smem, lane, xb, xa, mk_t, denc, nullptr);       // layer 0: encoded columns
        // everything fetched for the next pass is waited for HERE, in front of the ray atomics (long landed: issued a whole
        // encoding backward earlier), so that the top of the next pass waits for nothing -- least of all for those atomics
        auto settle = [&]() {
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
            // this lane holds d enc of channels 32 te + 8 q + 4 h + e (register 4 q + e of denc[te]); channel 3 + 20 a + f is
            // w_f sin(2^f x_a), + 10: w_f cos(2^f x_a)  ->  d x_a += 2^f w_f (cos dsin - sin dcos)
            auto dch = [&](int ch) -> float {              // d enc of channel ch if this lane half holds it, else 0
                const int hh = (ch >> 2) & 1;
                const float v = denc[ch >> 5][4 * ((ch >> 3) & 3) + (ch & 3)];
                return (h == hh) ? v : 0.f;
            };
            float dpos[3];
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
                float s, c;
                mcn_sincos(p[ax], s, c);
                float acc = dch(ax);
#pragma unroll
                for (int f = 0; f < MCN_NFREQ; ++f) {
                    const float k = (float)(1 << f) * sbarf[f];
                    acc = fmaf(k * c, dch(3 + 20 * ax + f), acc);
                    acc = fmaf(-k * s, dch(3 + 20 * ax + 10 + f), acc);
                    const float s2 = 2.f * s * c, c2 = (c - s) * (c + s);
                    s = s2; c = c2;
                }
                dpos[ax] = acc;
            }
            // SH term: d pre_c / d dir = sum_i sh[9c + i] d basis_i / d dir with the forward's saved sh.2 outputs (model/net_utils.py:154-169)
            float ddir[3] = {0.f, 0.f, 0.f};
            {
                const u32x4_t s0 = shs0, s1 = shs1;
                const float C1 = 0.4886025119029199f, C20 = 1.0925484305920792f, C22 = 0.31539156525252005f, C24 = 0.5462742152960396f;
                // derivative of basis i wrt (x, y, z)
                const float gx[9] = {0.f, 0.f, 0.f, -C1, C20 * y, 0.f, -2.f * C22 * x, -C20 * z, 2.f * C24 * x};
                const float gy[9] = {0.f, -C1, 0.f, 0.f, C20 * x, -C20 * z, -2.f * C22 * y, 0.f, -2.f * C24 * y};
                const float gz[9] = {0.f, 0.f, C1, 0.f, 0.f, -C20 * y, 4.f * C22 * z, -C20 * x, 0.f};
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) {
                        const unsigned w = s ? s1[jj >> 1] : s0[jj >> 1];
                        const float v = (jj & 1) ? T::hi(w) : T::lo(w);
                        const int n0 = mcn16_chan(s, 0, jj), n1 = n0 + 4;
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
        cur = nxt;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int W>
static hipError_t launch_bwd16(const Mcn16BwdArgs& a, long long max_rows, hipStream_t st) {
    using SM = Bwd16Smem<W>;
    int dev = 0, cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
    long long passes = (max_rows + MCN16_ROWS - 1) / MCN16_ROWS;
    if (passes <= 0) return hipSuccess;
    const int grid = (int)(passes < cus ? passes : cus);
    void (*kern)(Mcn16BwdArgs) = a.bf16 ? mlp16_bwd_kernel<W, true> : mlp16_bwd_kernel<W, false>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, SM::total);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * MCN16_WAVES), SM::total, st, a);
    return hipGetLastError();
}

hipError_t mcn16_launch_bwd(const Mcn16BwdArgs& a, hipStream_t st) {
    const long long max_rows = a.count ? (long long)a.max_rows : (long long)a.n_rays * a.S;
    switch (a.lay.width) {
        case 256: return launch_bwd16<256>(a, max_rows, st);
        case 128: return launch_bwd16<128>(a, max_rows, st);
        case 64:  return launch_bwd16<64>(a, max_rows, st);
        case 32:  return launch_bwd16<32>(a, max_rows, st);
    }
    return hipErrorInvalidValue;
}
