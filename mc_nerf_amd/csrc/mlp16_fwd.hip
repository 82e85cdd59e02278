// Fused NeRF MLP forward, single-pass 16-bit MFMA mode (mcnerf_16.h): sample generation -> sinusoidal encoding ->
// trunk (+skip) -> sigma / SH heads -> SH colour -> sigmoid, one wave = 32 samples carried through the whole network in
// registers, 8 waves share the LDS weight ring.  Persistent: one workgroup per CU walks the 256-row passes.
// Replaces (in 16-bit arithmetic, fp32 accumulate) SinCosEmbedding.forward (model/net_block.py:20-35),
// CorseFine_NeRF.forward (model/net_block.py:67-78), eval_sh (model/net_utils.py:103-191) and the gather / scatter of
// NeRF_Model.inference (model/mc_nerf.py:688-701).
#include "mcnerf_16.h"

template <int W>
struct Fwd16Smem {
    static constexpr int oBias = MCN16_RING * MCN16_SLAB * 1024;   // fp32 [MAXD + 2][W]: trunk, sigma.0, sh.0 biases
    static constexpr int oW2 = oBias + (MCN_MAXD + 2) * W * 4;     // sigma.2 weight row [W]
    static constexpr int oBc2 = oW2 + W * 4;                        // sh.2 bias [32] (27 + zero pad)
    static constexpr int total = oBc2 + 32 * 4;
};

__device__ __forceinline__ float mcn16_relu(float x) {       // integer max: no canonicalising v_max in front
    const int i = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, i > 0 ? i : 0);
}

// The 63 (+1 pad) encoded channels of one sample in this lane's fragment arrangement (model/net_block.py:22-33 order:
// x, y, z, then per axis sin(2^f x) f = 0..9, cos(2^f x) f = 0..9, each times the BARF weight of f).  The octaves come
// from one accurate sin/cos per axis and the double-angle recurrence (absolute error ~3e-5 at 2^9: far inside the
// 16-bit operand rounding).
template <bool BF>
__device__ __forceinline__ void mcn16_encode(const float (&p)[3], const float (&bw)[MCN_NFREQ], int h, u32x4_t (&encf)[MCN16_ENCKS]) {
    float E[64];
    E[0] = p[0]; E[1] = p[1]; E[2] = p[2]; E[63] = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float s, c;
        mcn_sincos(p[a], s, c);
#pragma unroll
        for (int f = 0; f < MCN_NFREQ; ++f) {
            E[3 + 20 * a + f] = s * bw[f];
            E[3 + 20 * a + 10 + f] = c * bw[f];
            const float s2 = 2.f * s * c, c2 = (c - s) * (c + s);
            s = s2; c = c2;
        }
    }
#pragma unroll
    for (int s = 0; s < MCN16_ENCKS; ++s)
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int c0 = mcn16_chan(s, 0, 2 * d), c1 = mcn16_chan(s, 0, 2 * d + 1);
            encf[s][d] = Mcn16T<BF>::pack(h ? E[c0 + 4] : E[c0], h ? E[c1 + 4] : E[c1]);
        }
}

// 16 mask bits of one output tile (the two packed fragments o0, o1 = registers 0..7, 8..15 after the ReLU): word i of
// the 8 contributes bit (7 - i) (element 2i) and bit (23 - i) (element 2i + 1).
__device__ __forceinline__ unsigned mcn16_tile_bits(const u32x4_t& o0, const u32x4_t& o1) {
    unsigned mb = mcn16_nz(o0[0]);
#pragma unroll
    for (int d = 1; d < 4; ++d) mb = (mb << 1) | mcn16_nz(o0[d]);
#pragma unroll
    for (int d = 0; d < 4; ++d) mb = (mb << 1) | mcn16_nz(o1[d]);
    return mb;
}

// One layer: NT output tiles, each the chain of KENC encoded-input k-steps (fragments encf) and KHID hidden-input k-steps
// (fragments in) over A fragments taken from the weight ring in stream order.
//   EPI 0: out[2t], out[2t+1] = relu(acc) as the next layer's fragments (saved with their ReLU bits when SAVE)
//   EPI 1: the sigma head's hidden layer: additionally dot += sum_n relu(acc)[n] * w2[n]
// Software pipeline, pinned with sched_barriers: A fragments are read MCN16_PF k-steps ahead of their MFMA; the
// epilogue of tile t (convert, ReLU, mask bits, stores) is issued in slices between the MFMAs of tile t + 1, whose
// accumulator is the other of two register sets and starts at the bias.
template <int W, bool BF, bool SAVE, int KENC, int KHID, int EPI>
__device__ __forceinline__ void mcn16_layer(Mcn16Ring& ring, char* smem, int lane, const u32x4_t (&encf)[MCN16_ENCKS],
                                            const u32x4_t (&in)[W / 16], u32x4_t (&out)[W / 16], const float* bias_h,
                                            const float* w2_h, float& dot, char* save_lane, unsigned* mask_lane) {
    using T = Mcn16T<BF>;
    constexpr int NT = W / 32, KTOT = KENC + KHID, F = NT * KTOT, MW = W >= 64 ? W / 64 : 1;
    constexpr int NSL = 8 + (SAVE ? 2 : 0);                    // epilogue slices of one tile: 8 packs (+ 2 stores)
    constexpr int START = KTOT >= NSL + 3 ? 2 : 0;             // first k-step that carries a slice (the previous tile's last MFMA needs ~2 MFMA times to land)
    constexpr int SPS = (NSL + KTOT - START - 1) / (KTOT - START);   // slices per k-step
    constexpr int LAST = START + (NSL + SPS - 1) / SPS - 1;    // k-step of the last slice
    constexpr int BIAS_AT = (KTOT - 4) > LAST ? (KTOT - 4) : LAST;   // the next tile's accumulator (= the set just drained) is loaded here
    Mcn16Cursor cur;
    unsigned mw[MW];
#pragma unroll
    for (int i = 0; i < MW; ++i) mw[i] = 0u;
    u32x4_t af[MCN16_PF];
    f32x16 acc[2];
    u32x4_t o0, o1;
    unsigned mb = 0u;
    auto bias_init = [&](f32x16& a, int t) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {            // accumulator starts at the bias: rows 32t + 8q + 4h + e
            const f32x4 b = *reinterpret_cast<const f32x4*>(bias_h + 32 * t + 8 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) a[4 * q + e] = b[e];
        }
    };
    // slice i of the epilogue of tile t (accumulator a)
    auto epi_slice = [&](const f32x16& a, int t, int i) {
        if (i < 8) {
            const unsigned w = T::relu_pack(a[2 * i], a[2 * i + 1]);
            if (i < 4) o0[i] = w; else o1[i - 4] = w;
            if (SAVE) mb = (i == 0) ? mcn16_nz(w) : ((mb << 1) | mcn16_nz(w));
            if (EPI == 1) {
                // (a float vector, not bit-cast integers: hipcc 7.2 folds fmaf((float)half, bit_cast<float>(u32x2[1]), acc) into a
                //  v_fma_mix_f32 that reads element 0 again)
                typedef float f32x2_t __attribute__((ext_vector_type(2)));
                const f32x2_t ww = *reinterpret_cast<const f32x2_t*>(w2_h + 32 * t + 8 * (i >> 1) + 2 * (i & 1));
                dot = fmaf(T::lo(w), ww[0], dot);
                dot = fmaf(T::hi(w), ww[1], dot);
            }
            if (i == 7) {
                if (EPI == 0) { out[2 * t] = o0; out[2 * t + 1] = o1; }
                if (SAVE) mw[t >> 1] |= mb << (8 * (t & 1));
            }
        } else if (SAVE) {
            if (i == 8) mcn16_ws_store(o0, reinterpret_cast<u32x4_t*>(save_lane + (2 * t) * 1024));
            else mcn16_ws_store(o1, reinterpret_cast<u32x4_t*>(save_lane + (2 * t + 1) * 1024));
        }
    };
    cur.cur = ring.next_off;
#pragma unroll
    for (int i = 0; i < MCN16_PF; ++i)
        if (i < F) af[i] = *reinterpret_cast<const u32x4_t*>(smem + ring.next_off + i * 1024 + lane * 16);
    bias_init(acc[0], 0);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int s = 0; s < KTOT; ++s) {
            const int f = t * KTOT + s;
            mcn16_before_mfma<F>(ring, smem, cur, f);
            const u32x4_t a_now = af[f % MCN16_PF];
            if (f + MCN16_PF < F)
                af[f % MCN16_PF] = *reinterpret_cast<const u32x4_t*>(smem + mcn16_frag_off(ring, cur, f, f + MCN16_PF) + lane * 16);
            if (t > 0 && s >= START) {
#pragma unroll
                for (int i = (s - START) * SPS; i < (s - START + 1) * SPS; ++i)
                    if (i < NSL) epi_slice(acc[(t - 1) & 1], t - 1, i);
            }
            if (s == BIAS_AT && t + 1 < NT) bias_init(acc[(t + 1) & 1], t + 1);
            __builtin_amdgcn_sched_barrier(0);
            acc[t & 1] = T::mfma(a_now, s < KENC ? encf[s < KENC ? s : 0] : in[s >= KENC ? s - KENC : 0], acc[t & 1]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int i = 0; i < NSL; ++i) epi_slice(acc[(NT - 1) & 1], NT - 1, i);
    if (SAVE) {
#pragma unroll
        for (int i = 0; i < MW; ++i) mcn16_ws_store(mw[i], mask_lane + i);
    }
}


template <int W, bool SAVE, bool BF>
__global__ __launch_bounds__(64 * MCN16_WAVES, 2) void mlp16_fwd_kernel(Mcn16FwdArgs a) {
    using T = Mcn16T<BF>;
    using SM = Fwd16Smem<W>;
    constexpr int KS = W / 16, MW = W >= 64 ? W / 64 : 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 31, h = lane >> 5;
    const int D = a.lay.depth, skip = a.lay.skip;
    const long long total = a.count ? (long long)min(*a.count, a.max_rows) : (long long)a.n_rays * a.S;
    if ((long long)blockIdx.x * MCN16_ROWS >= total) return;

    // ---- once per workgroup: biases, sigma.2 row, sh.2 bias -> LDS (before the ring starts: plain loads drain vmcnt)
    float* sbias = reinterpret_cast<float*>(smem + SM::oBias);
    float* sw2 = reinterpret_cast<float*>(smem + SM::oW2);
    float* sbc2 = reinterpret_cast<float*>(smem + SM::oBc2);
    for (int l = 0; l < D; ++l)
        for (int i = tid; i < W; i += 64 * MCN16_WAVES) sbias[l * W + i] = a.params[a.lay.pB[l] + i];
    for (int i = tid; i < W; i += 64 * MCN16_WAVES) {
        sbias[D * W + i] = a.params[a.lay.pBs1 + i];
        sbias[(D + 1) * W + i] = a.params[a.lay.pBc1 + i];
        sw2[i] = a.params[a.lay.pWs2 + i];
    }
    if (tid < 32) {      // sh.2 bias in the kernel's 27-row geometry (a degree below 2: the rows the net has, zero elsewhere)
        const int row = tid < MCN_NSH ? mcn_sh_row(tid, a.lay.sh_deg) : -1;
        sbc2[tid] = row >= 0 ? a.params[a.lay.pBc2 + row] : 0.f;
    }
    const float bs2 = a.params[a.lay.pBs2];
    float bw[MCN_NFREQ];
#pragma unroll
    for (int f = 0; f < MCN_NFREQ; ++f) bw[f] = a.barf_w[f];
    __syncthreads();

    Mcn16Ring ring;
    mcn16_ring_start(ring, smem, a.packed, a.stream_slabs, wave, lane);

    const float* bias_h = sbias + 4 * h;
    const float* w2_h = sw2 + 4 * h;

    for (long long pass = blockIdx.x; pass * MCN16_ROWS < total; pass += gridDim.x) {
        const long long tile = pass * MCN16_WAVES + wave;       // global 32-row tile of this wave
        const long long g = tile * 32 + m;
        const bool valid = g < total;
        const long long gc = valid ? g : total - 1;
        // ---- per-sample setup (lane-local; both lane halves of a sample compute the same values)
        int ray, j;
        if (a.idx) { const int2 rj = a.idx[gc]; ray = rj.x; j = rj.y; }
        else { ray = (int)(gc / a.S); j = (int)(gc - (long long)ray * a.S); }
        float zv = a.zgrid[j];
        if (a.jitter) zv = __fadd_rn(zv, a.jitter[ray]);
        const float dx = a.rays_d[ray * 3 + 0], dy = a.rays_d[ray * 3 + 1], dz = a.rays_d[ray * 3 + 2];
        float p[3];
        p[0] = __fadd_rn(a.rays_o[ray * 3 + 0], __fmul_rn(dx, zv));   // o + d z, two roundings (model/mc_nerf.py:602)
        p[1] = __fadd_rn(a.rays_o[ray * 3 + 1], __fmul_rn(dy, zv));
        p[2] = __fadd_rn(a.rays_o[ray * 3 + 2], __fmul_rn(dz, zv));
        const int addr = ray * a.S + j;
        u32x4_t encf[MCN16_ENCKS];
        mcn16_encode<BF>(p, bw, h, encf);
        const long long wtile = tile;
        char* act_lane = SAVE ? reinterpret_cast<char*>(a.act_ws) + (size_t)wtile * KS * 1024 + lane * 16 : nullptr;
        unsigned* mask_lane = SAVE ? a.mask_ws + ((size_t)wtile * 64 + lane) * MW : nullptr;
        if (SAVE) {
            char* e = reinterpret_cast<char*>(a.enc_ws) + (size_t)wtile * MCN16_ENCKS * 1024 + lane * 16;
#pragma unroll
            for (int s = 0; s < MCN16_ENCKS; ++s) mcn16_ws_store(encf[s], reinterpret_cast<u32x4_t*>(e + s * 1024));
        }

        u32x4_t xa[KS], xb[KS];
        float dot = 0.f;
        // ---- layer 0 (encoded input only), then the trunk; the skip layer takes [encoding, hidden]
        mcn16_layer<W, BF, SAVE, MCN16_ENCKS, 0, 0>(ring, smem, lane, encf, xa, xb, bias_h, nullptr, dot, act_lane, mask_lane);
        for (int l = 1; l < D; ++l) {
#pragma unroll
            for (int s = 0; s < KS; ++s) xa[s] = xb[s];
            char* sl = SAVE ? act_lane + (size_t)l * a.slot_bytes : nullptr;
            unsigned* ml = SAVE ? mask_lane + (size_t)l * a.mask_slot_words : nullptr;
            if (l == skip) mcn16_layer<W, BF, SAVE, MCN16_ENCKS, KS, 0>(ring, smem, lane, encf, xa, xb, bias_h + l * W, nullptr, dot, sl, ml);
            else mcn16_layer<W, BF, SAVE, 0, KS, 0>(ring, smem, lane, encf, xa, xb, bias_h + l * W, nullptr, dot, sl, ml);
        }
        // ---- sigma head: hidden layer on the matrix pipe, the 1-wide output layer lane-local
        mcn16_layer<W, BF, SAVE, 0, KS, 1>(ring, smem, lane, encf, xb, xa, bias_h + D * W, w2_h, dot,
                                           SAVE ? act_lane + (size_t)D * a.slot_bytes : nullptr, SAVE ? mask_lane + (size_t)D * a.mask_slot_words : nullptr);
        // ---- SH head: hidden layer (reads the same trunk output), then the 27 (32) coefficient rows
        mcn16_layer<W, BF, SAVE, 0, KS, 0>(ring, smem, lane, encf, xb, xa, bias_h + (D + 1) * W, nullptr, dot,
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
                mcn16_before_mfma<KS>(ring, smem, cur, s);
                const u32x4_t af = *reinterpret_cast<const u32x4_t*>(smem + cur.cur + (s & (MCN16_SLAB - 1)) * 1024 + lane * 16);
                acc = T::mfma(af, xa[s], acc);
            }
        }
        if (SAVE) {          // the SH coefficients (bias included) for the backward's view-direction term: 2 fragments of 16 bit
            char* e = reinterpret_cast<char*>(a.sh_ws) + (size_t)tile * 2 * 1024 + lane * 16;
            u32x4_t s0, s1;
#pragma unroll
            for (int d = 0; d < 4; ++d) { s0[d] = T::pack(acc[2 * d], acc[2 * d + 1]); s1[d] = T::pack(acc[8 + 2 * d], acc[8 + 2 * d + 1]); }
            mcn16_ws_store(s0, reinterpret_cast<u32x4_t*>(e));
            mcn16_ws_store(s1, reinterpret_cast<u32x4_t*>(e + 1024));
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
        float sigma = dot + __shfl_xor(dot, 32) + bs2;
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
static hipError_t launch_fwd16(const Mcn16FwdArgs& a, long long max_rows, hipStream_t st) {
    using SM = Fwd16Smem<W>;
    int dev = 0, cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
    long long passes = (max_rows + MCN16_ROWS - 1) / MCN16_ROWS;
    if (passes <= 0) return hipSuccess;
    const int grid = (int)(passes < cus ? passes : cus);
    const bool save = a.act_ws != nullptr;
    void (*kern)(Mcn16FwdArgs) = a.bf16 ? (save ? mlp16_fwd_kernel<W, true, true> : mlp16_fwd_kernel<W, false, true>)
                                        : (save ? mlp16_fwd_kernel<W, true, false> : mlp16_fwd_kernel<W, false, false>);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, SM::total);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * MCN16_WAVES), SM::total, st, a);
    return hipGetLastError();
}

hipError_t mcn16_launch_fwd(const Mcn16FwdArgs& a, hipStream_t st) {
    const long long max_rows = a.count ? (long long)a.max_rows : (long long)a.n_rays * a.S;
    switch (a.lay.width) {
        case 256: return launch_fwd16<256>(a, max_rows, st);
        case 128: return launch_fwd16<128>(a, max_rows, st);
        case 64:  return launch_fwd16<64>(a, max_rows, st);
        case 32:  return launch_fwd16<32>(a, max_rows, st);
    }
    return hipErrorInvalidValue;
}
