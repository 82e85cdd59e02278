// Fused NeRF MLP forward, split-f16 precision mode ("f16x3", see mcnerf_h.h): same structure, tile geometry
// and outputs as mlp_fwd.hip, but every GEMM runs as three v_mfma_f32_32x32x16_f16 per product on split
// operands (packed split weights from L2, split activations in LDS) with fp32 accumulation, bias, ReLU and
// epilogues.  Saved activations and encodings are written as SPLIT WORDS (hi | lo << 16, scaled by MCN_SX,
// mcnerf_h.h) in the fp32 workspaces' layout: the split-f16 weight-gradient kernel consumes them directly.
#include "mcnerf_h.h"
#include "mcnerf_kernels.h"

#ifdef MCN_STAMPS      // (diagnostic build: in-kernel cycle stamps of one trunk layer, read back by scripts/stamps.py)
__device__ unsigned long long g_mcn_stamps[64 * 8 * 8];
extern "C" int mcnerf_debug_stamps(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_mcn_stamps), sizeof(g_mcn_stamps));
}
#define MCN_STAMP_T(i) do { if (WIDTH == 256 && blockIdx.x >= 2048 && blockIdx.x < 2048 + 64 && lane == 0) \
        g_mcn_stamps[((blockIdx.x - 2048) * 8 + wave) * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
#if MCN_STAMPS == 2     // variant 2: slots 0..4 = end of layers 0..4 (after barrier 2), 5 = start, 6 = setup done, 7 = encoding done
#define MCN_STAMP(i) do { if ((i) == 4 && l <= 4) MCN_STAMP_T(l); } while (0)
#else
#define MCN_STAMP(i) do { if (WIDTH == 256 && l == 3 && blockIdx.x >= 2048 && blockIdx.x < 2048 + 64 && lane == 0) \
        g_mcn_stamps[((blockIdx.x - 2048) * 8 + wave) * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
#endif
#else
#define MCN_STAMP(i) do { } while (0)
#define MCN_STAMP_T(i) do { } while (0)
#endif

template <int WIDTH>
struct FwdSmemH {
    using G = McnGeomH<WIDTH>;
    static constexpr int MT = G::WM * G::MI * 32;
    static constexpr int NT = G::WN * G::WM * 64;       // threads per workgroup
    static constexpr int XW = WIDTH > 64 ? WIDTH : 64;
    static constexpr int oX = 0;
    static constexpr int oXyz = oX + MT * XW;          // (X region: Xh [MT][XW] f16 then Xl [MT][XW] f16 = MT*XW floats)
    static constexpr int oDir = oXyz + MT * 4;         // [MT][4]  dx,dy,dz,-
    static constexpr int oSig = oDir + MT * 4;         // [WN][MT] partial sigma
    static constexpr int oSh = oSig + G::WN * MT;      // [MT][33] sh coefficients
    static constexpr int oAddr = oSh + MT * 33;        // [MT] int: ray*S + j  (or -1)
    static constexpr int oMask = oAddr + MT;           // [MT][WIDTH / 32] ReLU bit masks of the layer just finished (store-wave builds)
    static constexpr int total = oMask + MT * (WIDTH / 32);
    static constexpr size_t bytes = (size_t)total * 4;
};

// Writes the 63(+1 pad) encoded channels of every tile row into the split LDS tile (a training forward then
// copies the tile rows to enc_save).  Channel order as in mlp_fwd.hip / model/net_block.py:22-33.
template <int MT, int XW>
__device__ __forceinline__ void write_encoding_h(_Float16* Xh, _Float16* Xl, const float* sxyz, const float* barf_w, int tid, int nthreads) {
    auto put = [&](int m, int ch, float v) {
        _Float16 hi, lo;
        mcn_split(v * MCN_SX, hi, lo);
        const int o = mcn_hoff<XW>(m, ch >> 3) + (ch & 7);
        Xh[o] = hi; Xl[o] = lo;
    };
    for (int it = tid; it < MT * 30; it += nthreads) {
        const int m = it / 30, cf = it - m * 30;
        const int c = cf / 10, f = cf - c * 10;
        const float v = sxyz[m * 4 + c] * (float)(1 << f);     // exact: power-of-two scale
        float s, co;
        mcn_sincos(v, s, co);
        const float w = barf_w[f];
        put(m, 3 + c * 20 + f, s * w);
        put(m, 3 + c * 20 + 10 + f, co * w);
    }
    for (int it = tid; it < MT * 4; it += nthreads) {
        const int m = it >> 2, c = it & 3;
        put(m, c == 3 ? 63 : c, (c == 3) ? 0.f : sxyz[m * 4 + c]);
    }
}

// Layer epilogue shared by the trunk layers and the two head hidden layers: v = relu(acc + bias);
//   TO_LDS : write v into the LDS tile (next layer's input)
//   SAVE   : store the 1-bit ReLU mask (backward chain) and, unless TO_LDS, v as split words (dW operand)
//   DOT    : accumulate sum_n v[n] * w2[n] per sample (the 1-wide sigma output layer, lane-local)
// A lane holds 16 of the 32 columns of its row per tile (the other 16 sit in lane ^ 32), so the mask halves
// are combined with one cross-lane move.
template <int WIDTH, int NI, int MI, bool TO_LDS, bool SAVE, bool DOT, bool MASKS = SAVE, bool LMASK = false>
__device__ __forceinline__ void layer_epilogue_h(f32x16 (&acc)[NI][MI], const float* __restrict__ bias, const float* __restrict__ w2,
                                               _Float16* Xh, _Float16* Xl, float* __restrict__ save, unsigned int* __restrict__ msave,
                                               float (&dot)[MI], int mrow0, int ncol0, long long row0, long long total, int lane) {
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) dot[mi] = 0.f;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int m = mrow0 + mi * 32 + r;
            const bool ok = row0 + m < total;
            unsigned bits = 0u;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n4 = ncol0 + ni * 32 + 8 * q + 4 * h;
                const f32x4 bb = *reinterpret_cast<const f32x4*>(bias + n4);
                f32x4 v;
                u32x4 w;
                if (TO_LDS && !DOT) {
                    // straight into the activation scale: relu(acc / SW + SX * b) == SX * relu(acc / (SW * SX) + b) bit for
                    // bit (power-of-two scales commute with rounding); one fma per value instead of mul, add and mul
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = fmaxf(fmaf(acc[ni][mi][4 * q + e], 1.0f / MCN_SW, bb[e] * MCN_SX), 0.f);
                        if (MASKS) bits |= (v[e] > 0.f ? 1u : 0u) << (8 * q + 4 * h + e);
                    }
                    w = mcn_store_split4<(WIDTH > 64 ? WIDTH : 64)>(Xh, Xl, m, n4, v, 1.0f);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = fmaxf(acc[ni][mi][4 * q + e] * (1.0f / (MCN_SW * MCN_SX)) + bb[e], 0.f);
                        if (MASKS) bits |= (v[e] > 0.f ? 1u : 0u) << (8 * q + 4 * h + e);
                    }
                    if (DOT) {
                        const f32x4 ww = *reinterpret_cast<const f32x4*>(w2 + n4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) dot[mi] = fmaf(v[e], ww[e], dot[mi]);
                    }
                    if (TO_LDS) w = mcn_store_split4<(WIDTH > 64 ? WIDTH : 64)>(Xh, Xl, m, n4, v);
                    else if (SAVE) w = mcn_words4(v, MCN_SX);
                }
                // (layers that go to the LDS tile are saved from there, row-coalesced: mcn_copy_tile_words)
                if (SAVE && !TO_LDS && ok) *reinterpret_cast<u32x4*>(save + (size_t)(row0 + m) * WIDTH + n4) = w;     // split words
            }
            if (MASKS) {
                const unsigned w = bits | (unsigned)__shfl_xor((int)bits, 32);
                if (LMASK) { if (h == 0) msave[m * (WIDTH / 32) + (ncol0 >> 5) + ni] = w; }          // msave = LDS mask tile
                else if (h == 0 && ok) msave[(size_t)(row0 + m) * (WIDTH / 32) + (ncol0 >> 5) + ni] = w;
            }
        }
}

// HELP: the workgroup carries MCN_HELP_THREADS extra threads ("store waves", mcnerf_h.h) that mirror the barrier
// sequence of the MFMA waves and, between barriers, copy the tile that the MFMA waves are reading (the encoding, then
// each layer's output) and its ReLU masks to the workspaces.
template <int WIDTH, bool SAVE, bool HELP>
__device__ __forceinline__ void mlp_fwd_h_body(const McnMlpFwdArgs& a) {
    using G = McnGeomH<WIDTH>;
    using SM = FwdSmemH<WIDTH>;
    constexpr int MT = SM::MT, XW = SM::XW, NI = G::NI, MI = G::MI, WN = G::WN, NT = SM::NT, WAVES = NT / 64;
    constexpr int KSH = WIDTH / 16;     // k-steps (of 16) of a hidden segment
    constexpr int KSE = MCN_ENCP / 16;  // ... of the encoded segment
    constexpr bool COPY = SAVE && !HELP;   // the MFMA waves save the tiles themselves
    static_assert(!HELP || SAVE, "store waves only exist in the saving instantiation");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    _Float16* Xh = reinterpret_cast<_Float16*>(smem + SM::oX);
    _Float16* Xl = Xh + MT * XW;
    float* sxyz = smem + SM::oXyz;
    float* sdir = smem + SM::oDir;
    float* ssig = smem + SM::oSig;
    float* ssh = smem + SM::oSh;
    int* saddr = reinterpret_cast<int*>(smem + SM::oAddr);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave % WN, wm = wave / WN;
    const int r = lane & 31, h = lane >> 5;
    const int mrow0 = wm * MI * 32;
    const int ncol0 = wn * NI * 32;
    const long long total = a.count ? (long long)min(*a.count, a.max_rows) : (long long)a.n_rays * a.S;
    const long long row0 = (long long)blockIdx.x * MT;
    if (row0 >= total) return;
    const McnLayout& L = a.lay;
    const float* __restrict__ prm = a.params;
    const h8* __restrict__ pk = reinterpret_cast<const h8*>(a.packed);     // split-f16 packed weights (16-byte units)

    if (HELP && __builtin_amdgcn_readfirstlane(tid) >= NT) {
        // ---- store waves: one __syncthreads() for every one of the MFMA waves below, in the same order
        constexpr int HT = MCN_HELP_THREADS;
        const int ht = tid - NT;
        __builtin_amdgcn_s_setprio(MCN_HELP_PRIO);        // the store waves are light: let them issue ahead of the MFMA wave on their SIMD
        const size_t AS = a.act_stride;
        unsigned* const msk = a.mask_save;
        const unsigned* const smask = reinterpret_cast<const unsigned*>(smem + SM::oMask);
        // the finished layer's mask tile [MT][WIDTH/32] is one contiguous block of the workspace rows row0 .. row0+MT-1
        auto copy_masks = [&](int slot) {
            constexpr int W32 = WIDTH / 32;
            for (int i = ht; i < MT * W32; i += HT)
                if (row0 + i / W32 < total) msk[(size_t)slot * (AS / 32) + (size_t)row0 * W32 + i] = smask[i];
        };
        __syncthreads();                                   // per-sample setup done
        __syncthreads();                                   // encoding in X
        for (int l = 0; l < L.depth; ++l) {
            if (l == 0) {
                mcn_copy_tile_words<MT, XW, MCN_ENCP, HT>(Xh, Xl, a.enc_save, row0, total, ht);
            } else {                                       // X = output of layer l-1, stable until this layer's epilogue
                MCN_STAMP(0);
                mcn_copy_tile_words<MT, XW, WIDTH, HT, 8>(Xh, Xl, a.act_save + (size_t)(l - 1) * AS, row0, total, ht);
                MCN_STAMP(1);
                copy_masks(l - 1);
                MCN_STAMP(2);
            }
            if (l == L.skip) { __syncthreads(); __syncthreads(); }     // re-encoding of the skip layer
            __syncthreads();                               // GEMM done
            MCN_STAMP(3);
            __syncthreads();                               // epilogue done
            MCN_STAMP(4);
        }
        mcn_copy_tile_words<MT, XW, WIDTH, HT, 8>(Xh, Xl, a.act_save + (size_t)(L.depth - 1) * AS, row0, total, ht);
        copy_masks(L.depth - 1);
        __syncthreads();                                   // sigma head + SH hidden GEMM done
        __syncthreads();                                   // SH hidden layer in X
        mcn_copy_tile_words<MT, XW, WIDTH, HT, 8>(Xh, Xl, a.act_save + (size_t)(L.depth + 1) * AS, row0, total, ht);
        copy_masks(L.depth + 1);
        __syncthreads();                                   // SH output layer done
        return;
    }

    MCN_STAMP_T(5);
    // ---- per-sample setup: position, direction, output address
    for (int m = tid; m < MT; m += NT) {
        const long long g = row0 + m;
        float x = 0.f, y = 0.f, z = 0.f, dx = 0.f, dy = 0.f, dz = 1.f, zv = 0.f;
        int addr = -1;
        if (g < total) {
            int ray, j;
            if (a.idx) { const int2 rj = a.idx[g]; ray = rj.x; j = rj.y; }
            else { ray = (int)(g / a.S); j = (int)(g - (long long)ray * a.S); }
            zv = a.zgrid[j];
            if (a.jitter) zv = __fadd_rn(zv, a.jitter[ray]);
            dx = a.rays_d[ray * 3 + 0]; dy = a.rays_d[ray * 3 + 1]; dz = a.rays_d[ray * 3 + 2];
            // o + d*z with separate roundings, as the reference's broadcasted mul then add (mc_nerf.py:602)
            x = __fadd_rn(a.rays_o[ray * 3 + 0], __fmul_rn(dx, zv));
            y = __fadd_rn(a.rays_o[ray * 3 + 1], __fmul_rn(dy, zv));
            z = __fadd_rn(a.rays_o[ray * 3 + 2], __fmul_rn(dz, zv));
            addr = ray * a.S + j;
        }
        sxyz[m * 4 + 0] = x; sxyz[m * 4 + 1] = y; sxyz[m * 4 + 2] = z; sxyz[m * 4 + 3] = zv;
        sdir[m * 4 + 0] = dx; sdir[m * 4 + 1] = dy; sdir[m * 4 + 2] = dz; sdir[m * 4 + 3] = 0.f;
        saddr[m] = addr;
    }
    __syncthreads();
#if defined(MCN_STAMPS) && MCN_STAMPS == 2
    MCN_STAMP_T(6);
#endif
    write_encoding_h<MT, XW>(Xh, Xl, sxyz, a.barf_w, tid, NT);
    __syncthreads();
#if defined(MCN_STAMPS) && MCN_STAMPS == 2
    MCN_STAMP_T(7);
#endif

    f32x16 acc[NI][MI];
    // ---- trunk
    for (int l = 0; l < L.depth; ++l) {
        mcn_zero<NI, MI>(acc);
        MCN_STAMP(0);
        // a training forward trickles the tile it is reading (encoding / previous layer's output) out to the
        // workspaces during the GEMM, one row group per k-step
        float* const prev = COPY && l > 0 ? a.act_save + (size_t)(l - 1) * a.act_stride : nullptr;
        if (l == 0) {
            float* const encs = a.enc_save;
            mcn_gemm_seg_h<XW, NI, MI>(acc, Xh, Xl, mrow0, 0, KSE, pk + (L.fEnc0 >> 2) + (wn * NI) * KSE * 128, lane,
                [=](int ks) { if (COPY) mcn_copy_tile_step<MT, XW, MCN_ENCP, NT, KSE>(Xh, Xl, encs, row0, total, tid, ks); },
                [=]() { if (COPY && MCN_COPY_MODE == 2) mcn_copy_tile_words<MT, XW, MCN_ENCP, NT>(Xh, Xl, encs, row0, total, tid); });
        } else {
            mcn_gemm_seg_h<XW, NI, MI>(acc, Xh, Xl, mrow0, 0, KSH, pk + (L.fH[l] >> 2) + (wn * NI) * KSH * 128, lane,
                [=](int ks) { if (COPY) mcn_copy_tile_step<MT, XW, WIDTH, NT, KSH>(Xh, Xl, prev, row0, total, tid, ks); },
                [=]() { if (COPY && MCN_COPY_MODE == 2) mcn_copy_tile_words<MT, XW, WIDTH, NT>(Xh, Xl, prev, row0, total, tid); });
            if (l == L.skip) {
                __syncthreads();                       // everyone finished reading h from X
                write_encoding_h<MT, XW>(Xh, Xl, sxyz, a.barf_w, tid, NT);
                __syncthreads();
                mcn_gemm_seg_h<XW, NI, MI>(acc, Xh, Xl, mrow0, 0, KSE, pk + (L.fEncS >> 2) + (wn * NI) * KSE * 128, lane);
            }
        }
        MCN_STAMP(1);
        __syncthreads();
        MCN_STAMP(2);
        float unused[MI];
        layer_epilogue_h<WIDTH, NI, MI, true, SAVE, false, SAVE, HELP>(acc, prm + L.pB[l], nullptr, Xh, Xl,
            SAVE ? a.act_save + (size_t)l * a.act_stride : nullptr,
            HELP ? reinterpret_cast<unsigned*>(smem + SM::oMask) : SAVE ? a.mask_save + (size_t)l * (a.act_stride / 32) : nullptr,
            unused, mrow0, ncol0, row0, total, lane);
        MCN_STAMP(3);
        __syncthreads();
        MCN_STAMP(4);
    }

#if !(defined(MCN_STAMPS) && MCN_STAMPS == 2)
    MCN_STAMP_T(6);
#endif
    // ---- sigma head: hidden layer on MFMA, the 1-wide output layer lane-local on the VALU
    {
        mcn_zero<NI, MI>(acc);
        float* const prev = COPY ? a.act_save + (size_t)(L.depth - 1) * a.act_stride : nullptr;
        mcn_gemm_seg_h<XW, NI, MI>(acc, Xh, Xl, mrow0, 0, KSH, pk + (L.fS1 >> 2) + (wn * NI) * KSH * 128, lane,
            [=](int ks) { if (COPY) mcn_copy_tile_step<MT, XW, WIDTH, NT, KSH>(Xh, Xl, prev, row0, total, tid, ks); },
            [=]() { if (COPY && MCN_COPY_MODE == 2) mcn_copy_tile_words<MT, XW, WIDTH, NT>(Xh, Xl, prev, row0, total, tid); });
        float s[MI];
        layer_epilogue_h<WIDTH, NI, MI, false, SAVE, true>(acc, prm + L.pBs1, prm + L.pWs2, Xh, Xl,
            SAVE ? a.act_save + (size_t)L.depth * a.act_stride : nullptr,
            SAVE ? a.mask_save + (size_t)L.depth * (a.act_stride / 32) : nullptr, s, mrow0, ncol0, row0, total, lane);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            s[mi] += __shfl_xor(s[mi], 32);
            if (h == 0) ssig[wn * MT + mrow0 + mi * 32 + r] = s[mi];
        }
    }
    // ---- SH head hidden layer (reads the same trunk output still resident in X)
    {
        mcn_zero<NI, MI>(acc);
        mcn_gemm_seg_h<XW, NI, MI>(acc, Xh, Xl, mrow0, 0, KSH, pk + (L.fC1 >> 2) + (wn * NI) * KSH * 128, lane);
        __syncthreads();
        float unused[MI];
        layer_epilogue_h<WIDTH, NI, MI, true, SAVE, false, SAVE, HELP>(acc, prm + L.pBc1, nullptr, Xh, Xl,
            SAVE ? a.act_save + (size_t)(L.depth + 1) * a.act_stride : nullptr,
            HELP ? reinterpret_cast<unsigned*>(smem + SM::oMask) : SAVE ? a.mask_save + (size_t)(L.depth + 1) * (a.act_stride / 32) : nullptr,
            unused, mrow0, ncol0, row0, total, lane);
        __syncthreads();
    }
    if (COPY) mcn_copy_tile_words<MT, XW, WIDTH, NT>(Xh, Xl, a.act_save + (size_t)(L.depth + 1) * a.act_stride, row0, total, tid);
    // ---- SH output layer (27 -> 32 padded outputs): one 32-row m-tile per wave
    for (int mt = wave; mt < MT / 32; mt += WAVES) {
        f32x16 a1[1][1];
        mcn_zero<1, 1>(a1);
        mcn_gemm_seg_h<XW, 1, 1>(a1, Xh, Xl, mt * 32, 0, KSH, pk + (L.fC2 >> 2), lane);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int n = (e & 3) + 8 * (e >> 2) + 4 * h;
            ssh[(mt * 32 + r) * 33 + n] = a1[0][0][e] * (1.0f / (MCN_SW * MCN_SX));
        }
    }
    __syncthreads();
    // ---- per-sample epilogue: sigma, SH colour, sigmoid
    for (int m = tid; m < MT; m += NT) {
        const long long g = row0 + m;
        if (g >= total) continue;
        float sigma = prm[L.pBs2];
#pragma unroll
        for (int w = 0; w < WN; ++w) sigma += ssig[w * MT + m];
        float sh[MCN_NSH];
#pragma unroll
        for (int i = 0; i < MCN_NSH; ++i) sh[i] = ssh[m * 33 + i] + prm[L.pBc2 + i];
        float b[9];
        mcn_sh_basis(sdir[m * 4 + 0], sdir[m * 4 + 1], sdir[m * 4 + 2], b);
        f32x4 o;
        o[0] = sigma;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float pre = b[0] * sh[9 * c];
#pragma unroll
            for (int i = 1; i < 9; ++i) pre += b[i] * sh[9 * c + i];
            o[1 + c] = 1.0f / (1.0f + expf(-pre));
        }
        *reinterpret_cast<f32x4*>(a.out + (size_t)saddr[m] * 4) = o;
        if (SAVE) {
            float* dst = a.sh_save + (size_t)g * MCN_NSHP;             // 128-byte row: 7 vector stores (column 27 is padding)
#pragma unroll
            for (int i = 0; i < MCN_NSH + 1; i += 4) {
                f32x4 v4;
#pragma unroll
                for (int e = 0; e < 4; ++e) v4[e] = i + e < MCN_NSH ? sh[i + e] : 0.f;
                *reinterpret_cast<f32x4*>(dst + i) = v4;
            }
        }
    }
#if !(defined(MCN_STAMPS) && MCN_STAMPS == 2)
    MCN_STAMP_T(7);
#endif
}

template <int WIDTH, bool SAVE>
__global__ __launch_bounds__(McnGeomH<WIDTH>::WN * McnGeomH<WIDTH>::WM * 64, McnGeomH<WIDTH>::WGS)
void mlp_fwd_h_kernel(McnMlpFwdArgs a) { mlp_fwd_h_body<WIDTH, SAVE, false>(a); }

// store-wave build of the saving forward: 4 MFMA waves + MCN_HELP_THREADS / 64 store waves per workgroup
#if MCN_HELP_WGS == 2
#define MCN_HELP_KERNEL_ATTR __attribute__((amdgpu_flat_work_group_size(64, 384), amdgpu_waves_per_eu(3, 3)))
#else
#define MCN_HELP_KERNEL_ATTR __launch_bounds__(256 + MCN_HELP_THREADS, 1)
#endif
template <int WIDTH>
__global__ MCN_HELP_KERNEL_ATTR void mlp_fwd_h_help_kernel(McnMlpFwdArgs a) { mlp_fwd_h_body<WIDTH, true, true>(a); }

template <int WIDTH>
static hipError_t launch_fwd_h(const McnMlpFwdArgs& a, long long max_rows, hipStream_t st) {
    using SM = FwdSmemH<WIDTH>;
    const int grid = (int)((max_rows + SM::MT - 1) / SM::MT);
    if (grid <= 0) return hipSuccess;
    const bool save = a.act_save != nullptr;
    constexpr bool HELP = MCN_HELP_MIN_WIDTH > 0 && WIDTH >= MCN_HELP_MIN_WIDTH;
    auto kern = !save ? mlp_fwd_h_kernel<WIDTH, false> : HELP ? mlp_fwd_h_help_kernel<WIDTH> : mlp_fwd_h_kernel<WIDTH, true>;
    const size_t lds = SM::bytes;       // (with store waves the 8 waves x 256 registers fill the CU: one workgroup per CU)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(SM::NT + ((save && HELP) ? MCN_HELP_THREADS : 0)), lds, st, a);
    return hipGetLastError();
}

hipError_t mcn_launch_mlp_fwd_h(const McnMlpFwdArgs& a, hipStream_t st) {
    const long long max_rows = a.count ? (long long)a.max_rows : (long long)a.n_rays * a.S;
    switch (a.lay.width) {
        case 256: return launch_fwd_h<256>(a, max_rows, st);
        case 128: return launch_fwd_h<128>(a, max_rows, st);
        case 64:  return launch_fwd_h<64>(a, max_rows, st);
        case 32:  return launch_fwd_h<32>(a, max_rows, st);
    }
    return hipErrorInvalidValue;
}

