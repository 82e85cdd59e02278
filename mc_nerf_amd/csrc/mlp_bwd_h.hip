// Fused NeRF MLP backward (activation-gradient chain), split-f16 precision mode ("f16x3", mcnerf_h.h): same
// structure as mlp_bwd.hip; the LDS tile holds dY as split f16 scaled by a per-launch power of two SG chosen
// from max|d_out| (gradients are far below the f16 normal range otherwise), the transposed packed weights are
// split f16, every GEMM is three v_mfma_f32_32x32x16_f16 per product with fp32 accumulation.  dy_save /
// dsh_save receive SPLIT WORDS scaled by SG (mcnerf_h.h) for the split-f16 weight-gradient kernel; enc_save
// (written by the split-f16 forward) is read back as split words.
#include "mcnerf_h.h"
#include "mcnerf_kernels.h"

#ifdef MCN_STAMPS      // (diagnostic build: in-kernel cycle stamps of the tile phases, read back by scripts/stamps_bwd.py)
__device__ unsigned long long g_mcn_bstamps[64 * 4 * 8];
extern "C" int mcnerf_debug_stamps_bwd(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_mcn_bstamps), sizeof(g_mcn_bstamps));
}
#define MCN_BSTAMP(i) do { if (WIDTH == 256 && blockIdx.x >= 2048 && blockIdx.x < 2048 + 64 && lane == 0 && wave < 4) \
        g_mcn_bstamps[((blockIdx.x - 2048) * 4 + wave) * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define MCN_BSTAMP(i) do { } while (0)
#endif

template <int WIDTH>
struct BwdSmemH {
    using G = McnGeomH<WIDTH>;
    static constexpr int MT = G::WM * G::MI * 32;
    static constexpr int NT = G::WN * G::WM * 64;
    static constexpr int XW = WIDTH > 64 ? WIDTH : 64;
    static constexpr int oX = 0;
    static constexpr int oDsig = oX + MT * XW;        // [MT] d sigma_raw
    static constexpr int oDdir = oDsig + MT;          // [MT][4] d view-direction from the SH colour
    static constexpr int oGo = oDdir + MT * 4;        // [MT][4] d sample position (= d ray origin contribution)
    static constexpr int oAddr = oGo + MT * 4;        // [MT] int ray id (or -1)
    static constexpr int oZ = oAddr + MT;             // [MT] z value
    static constexpr int total = oZ + MT;
    static constexpr size_t bytes = (size_t)total * 4;
};

// 4 consecutive gradients of row m (fp32, unscaled) -> split f16 scaled by sg in the LDS tile
template <int XW>
__device__ __forceinline__ u32x4 store_split4_s(_Float16* Xh, _Float16* Xl, int m, int n4, const f32x4& v, float sg) {
    return mcn_store_split4<XW>(Xh, Xl, m, n4, v, sg);
}

// acc (scaled gradient wrt a post-ReLU activation) * inv -> masked by the forward's ReLU bit mask -> LDS tile
// (split f16, scaled by sg); the caller then copies the tile rows to dy_save (mcn_copy_tile_words, row-coalesced).
// Mask words are read with unconditional loads.
template <int WIDTH, int NI, int MI>
__device__ __forceinline__ void mask_store_h(f32x16 (&acc)[NI][MI], const unsigned int* __restrict__ msave,
                                             _Float16* Xh, _Float16* Xl, float inv, float sg, int mrow0, int ncol0,
                                             long long row0, long long total, int lane) {
    constexpr int XW = WIDTH > 64 ? WIDTH : 64;
    const int r = lane & 31, h = lane >> 5;
    unsigned wm[NI][MI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const long long g = row0 + mrow0 + mi * 32 + r;
            const long long gc = g < total ? g : total - 1;
            wm[ni][mi] = msave[(size_t)gc * (WIDTH / 32) + (ncol0 >> 5) + ni];
        }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k4 = ncol0 + ni * 32 + 8 * q + 4 * h;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int m = mrow0 + mi * 32 + r;
                const bool ok = row0 + m < total;
                const unsigned w = ok ? wm[ni][mi] >> (8 * q + 4 * h) : 0u;
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = ((w >> e) & 1u) ? acc[ni][mi][4 * q + e] * inv : 0.f;
                store_split4_s<XW>(Xh, Xl, m, k4, v, sg);
            }
        }
}

// HELP: store waves (mcnerf_h.h) mirror the barrier sequence below and copy every dY tile to dy_save while the MFMA
// waves run the GEMM that reads it; without them the MFMA waves issue those copies themselves.
template <int WIDTH, bool HELP>
__device__ __forceinline__ void mlp_bwd_h_body(const McnMlpBwdArgs& a) {
    using G = McnGeomH<WIDTH>;
    using SM = BwdSmemH<WIDTH>;
    constexpr int MT = SM::MT, XW = SM::XW, NI = G::NI, MI = G::MI, WN = G::WN, NT = SM::NT, WAVES = NT / 64;
    constexpr int NSH = WIDTH / 16;           // reduction steps (of 16) over a hidden-wide dY
    constexpr int W4 = WIDTH / 4;
    constexpr bool COPY = !HELP;              // the MFMA waves save the dY tiles themselves
    extern __shared__ __attribute__((aligned(16))) float smem[];
    _Float16* Xh = reinterpret_cast<_Float16*>(smem + SM::oX);
    _Float16* Xl = Xh + MT * XW;
    float* Xf = smem + SM::oX;                // fp32 view of the same region for the final encoding-gradient stage
    float* sdsig = smem + SM::oDsig;
    float* sddir = smem + SM::oDdir;
    float* sgo = smem + SM::oGo;
    int* sray = reinterpret_cast<int*>(smem + SM::oAddr);
    float* sz = smem + SM::oZ;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave % WN, wm = wave / WN;
    const int mrow0 = wm * MI * 32, ncol0 = wn * NI * 32;
    const long long total = a.count ? (long long)min(*a.count, a.max_rows) : (long long)a.n_rays * a.S;
    const long long row0 = (long long)blockIdx.x * MT;
    if (row0 >= total) return;
    const McnLayout& L = a.lay;
    const int D = L.depth;
    const float* __restrict__ prm = a.params;
    const h8* __restrict__ pk = reinterpret_cast<const h8*>(a.packed);     // split-f16 transposed packed weights
    const size_t AS = a.act_stride;

    if (HELP && __builtin_amdgcn_readfirstlane(tid) >= NT) {
        // ---- store waves: one __syncthreads() for every one of the MFMA waves below, in the same order
        constexpr int HT = MCN_HELP_THREADS;
        const int ht = tid - NT;
        __builtin_amdgcn_s_setprio(MCN_HELP_PRIO);        // the store waves are light: let them issue ahead of the MFMA wave on their SIMD
        __syncthreads();                                   // prologue done (X = dsh)
        __syncthreads();                                   // sh.2^T GEMM done
        __syncthreads();                                   // X = dY of sh.0
        mcn_copy_tile_words<MT, XW, WIDTH, HT, 8>(Xh, Xl, a.dy_save + (size_t)(D + 1) * AS, row0, total, ht);
        __syncthreads();                                   // sh.0^T GEMM done
        __syncthreads();                                   // X = dY of sigma.0
        mcn_copy_tile_words<MT, XW, WIDTH, HT, 8>(Xh, Xl, a.dy_save + (size_t)D * AS, row0, total, ht);
        __syncthreads();                                   // sigma.0^T GEMM done
        __syncthreads();                                   // X = dY_{D-1}
        for (int l = D - 1; l >= 0; --l) {
            mcn_copy_tile_words<MT, XW, WIDTH, HT, 8>(Xh, Xl, a.dy_save + (size_t)l * AS, row0, total, ht);   // X = dY_l
            if (l == 0) break;
            __syncthreads();                               // GEMM of layer l done
            __syncthreads();                               // X = dY_{l-1}
        }
        __syncthreads();                                   // encoded-gradient GEMMs done
        __syncthreads();                                   // encoded-input gradient in LDS
        if (a.d_rays_o || a.d_rays_d) __syncthreads();     // per-sample d o / d d in LDS
        return;
    }
    // gradient scale: a power of two that puts max|d_out| of this launch near 2^4 in f16 (4096x headroom below
    // the f16 maximum for growth through the layers, ~2^-18 of the maximum before f16 subnormals start)
    const float gmax = a.gmax_bits ? __uint_as_float(*a.gmax_bits) : 1.f;
    const float sg = (gmax > 0.f && gmax < 3e38f) ? exp2f(4.f - ceilf(log2f(gmax))) : 1.f;
    const float inv = 1.0f / (MCN_SW * sg);

    MCN_BSTAMP(0);
    // ---- per-sample prologue: sigmoid and SH backward -> dsh (the dY of sh.2), d sigma, d dir
    for (int m = tid; m < MT; m += NT) {
        const long long g = row0 + m;
        __attribute__((aligned(16))) float dsh[MCN_NSHP];
#pragma unroll
        for (int i = 0; i < MCN_NSHP; ++i) dsh[i] = 0.f;
        float dsg = 0.f, ddx = 0.f, ddy = 0.f, ddz = 0.f, zv = 0.f;
        int ray = -1;
        if (g < total) {
            int j;
            if (a.idx) { const int2 rj = a.idx[g]; ray = rj.x; j = rj.y; }
            else { ray = (int)(g / a.S); j = (int)(g - (long long)ray * a.S); }
            zv = a.zgrid[j];
            if (a.jitter) zv = __fadd_rn(zv, a.jitter[ray]);
            const size_t addr = (size_t)ray * a.S + j;
            const f32x4 o = *reinterpret_cast<const f32x4*>(a.out + addr * 4);
            const f32x4 go = *reinterpret_cast<const f32x4*>(a.d_out + addr * 4);
            dsg = go[0];
            const float x = a.rays_d[ray * 3], y = a.rays_d[ray * 3 + 1], z = a.rays_d[ray * 3 + 2];
            float b[9];
            mcn_sh_basis(x, y, z, b);
            __attribute__((aligned(16))) float sh[MCN_NSHP];                 // the saved SH row: 7 vector loads
#pragma unroll
            for (int i = 0; i < MCN_NSH + 1; i += 4)
                *reinterpret_cast<f32x4*>(&sh[i]) = *reinterpret_cast<const f32x4*>(a.sh_save + (size_t)g * MCN_NSHP + i);
            const float C1 = 0.4886025119029199f, C20 = 1.0925484305920792f, C22 = 0.31539156525252005f,
                        C24 = 0.5462742152960396f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float dpre = go[1 + c] * o[1 + c] * (1.f - o[1 + c]);
#pragma unroll
                for (int i = 0; i < 9; ++i) dsh[9 * c + i] = dpre * b[i];
                const float* s = sh + 9 * c;
                // d(pre)/d(dir): derivatives of the nine basis polynomials
                ddx += dpre * (-C1 * s[3] + C20 * y * s[4] - 2.f * C22 * x * s[6] - C20 * z * s[7] + 2.f * C24 * x * s[8]);
                ddy += dpre * (-C1 * s[1] + C20 * x * s[4] - C20 * z * s[5] - 2.f * C22 * y * s[6] - 2.f * C24 * y * s[8]);
                ddz += dpre * (C1 * s[2] - C20 * y * s[5] + 4.f * C22 * z * s[6] - C20 * x * s[7]);
            }
            dsh[MCN_NSH] = dsg;      // spare column 27 carries d sigma: the dW kernel reduces d sigma.2.{weight,bias} from it
            float* dst = a.dsh_save + (size_t)g * MCN_NSHP;
#pragma unroll
            for (int i = 0; i < MCN_NSHP; i += 4)            // split words scaled by sg (operand of the sh.2 / sigma.2 dW)
                *reinterpret_cast<u32x4*>(dst + i) = mcn_words4(*reinterpret_cast<f32x4*>(&dsh[i]), sg);
        }
#pragma unroll
        for (int c4 = 0; c4 < MCN_NSHP / 4; ++c4) {
            dsh[MCN_NSH] = 0.f;            // the sigma gradient rides in dsh_save only, not in the sh.2 GEMM operand
            store_split4_s<XW>(Xh, Xl, m, 4 * c4, *reinterpret_cast<f32x4*>(&dsh[4 * c4]), sg);
        }
        sdsig[m] = dsg; sray[m] = ray; sz[m] = zv;
        sddir[m * 4] = ddx; sddir[m * 4 + 1] = ddy; sddir[m * 4 + 2] = ddz; sddir[m * 4 + 3] = 0.f;
    }
    __syncthreads();

    MCN_BSTAMP(1);
    f32x16 acc[NI][MI];
    // ---- sh.2^T : dsh [MT][32] -> d hc ; mask with hc -> dY of sh.0
    mcn_zero<NI, MI>(acc);
    mcn_gemm_seg_h<XW, NI, MI>(acc, Xh, Xl, mrow0, 0, MCN_NSHP / 16, pk + (L.bC2 >> 2) + (wn * NI) * (MCN_NSHP / 16) * 128, lane);
    __syncthreads();
    mask_store_h<WIDTH, NI, MI>(acc, a.mask_save + (size_t)(D + 1) * (AS / 32), Xh, Xl, inv, sg, mrow0, ncol0, row0, total, lane);
    __syncthreads();
    // ---- sh.0^T and sigma.0^T both feed d h_{D-1}.  (Every dY tile is trickled out to dy_save, row-coalesced,
    //      by the GEMM that reads it: one row group per k-step.)
    mcn_zero<NI, MI>(acc);
    {
        float* const dst = a.dy_save + (size_t)(D + 1) * AS;
        mcn_gemm_seg_h<XW, NI, MI>(acc, Xh, Xl, mrow0, 0, NSH, pk + (L.bC1 >> 2) + (wn * NI) * NSH * 128, lane,
            [=](int ks) { if (COPY) mcn_copy_tile_step<MT, XW, WIDTH, NT, NSH>(Xh, Xl, dst, row0, total, tid, ks); },
            [=]() { if (COPY && MCN_COPY_MODE == 2) mcn_copy_tile_words<MT, XW, WIDTH, NT>(Xh, Xl, dst, row0, total, tid); });
    }
    __syncthreads();
    {   // dY of sigma.0 = d sigma * w_sigma2 masked by hs > 0 (outer product, no GEMM)
        const unsigned int* hm = a.mask_save + (size_t)D * (AS / 32);
        float* dys = a.dy_save + (size_t)D * AS;
        const float* w2 = prm + L.pWs2;
        constexpr int MG = NT / W4;             // sample groups (threads / chunks per row)
        const int c4 = tid % W4, mg = tid / W4;
        const f32x4 ww = *reinterpret_cast<const f32x4*>(w2 + 4 * c4);
        // all mask words of this thread's rows first (unconditional, clamped loads issued back to back): the row loop
        // below otherwise pays one exposed HBM latency per row
        unsigned wrow[MT / MG];
#pragma unroll
        for (int k = 0; k < MT / MG; ++k) {
            const long long g = row0 + mg + k * MG;
            wrow[k] = hm[(size_t)(g < total ? g : total - 1) * (WIDTH / 32) + (c4 >> 3)];
        }
#pragma unroll
        for (int k = 0; k < MT / MG; ++k) {
            const int m = mg + k * MG;
            const bool ok = row0 + m < total;
            const unsigned w = wrow[k] >> (4 * (c4 & 7));
            const float ds = ok ? sdsig[m] : 0.f;
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = ((w >> e) & 1u) ? ds * ww[e] : 0.f;
            const u32x4 w4 = store_split4_s<XW>(Xh, Xl, m, 4 * c4, v, sg);
            if (COPY && ok) *reinterpret_cast<u32x4*>(dys + (size_t)(row0 + m) * WIDTH + 4 * c4) = w4;
        }
    }
    __syncthreads();
    mcn_gemm_seg_h<XW, NI, MI>(acc, Xh, Xl, mrow0, 0, NSH, pk + (L.bS1 >> 2) + (wn * NI) * NSH * 128, lane);
    __syncthreads();
    mask_store_h<WIDTH, NI, MI>(acc, a.mask_save + (size_t)(D - 1) * (AS / 32), Xh, Xl, inv, sg, mrow0, ncol0, row0, total, lane);
    __syncthreads();

    MCN_BSTAMP(2);
    // ---- trunk, last layer to first.  X holds dY_l; the encoded-input gradient accumulates in denc.
    f32x16 denc[1][1];
    mcn_zero<1, 1>(denc);
    constexpr int ENC_TILES = 2 * (MT / 32);          // (2 k-tiles of the 64 encoded channels) x m-tiles
    static_assert(ENC_TILES <= 2 * WAVES, "at most two encoded-gradient tiles per wave");
    f32x16 denc2[1][1];                                // second tile when ENC_TILES > WAVES
    mcn_zero<1, 1>(denc2);
    for (int l = D - 1; l >= 0; --l) {
        if (l == 0 || l == L.skip) {
            const h8* pe = pk + ((l == 0 ? L.bEnc0 : L.bEncS) >> 2);
            {
                const int t = wave;
                if (t < ENC_TILES) mcn_gemm_seg_h<XW, 1, 1>(denc, Xh, Xl, (t >> 1) * 32, 0, NSH, pe + (t & 1) * NSH * 128, lane);
            }
            if (ENC_TILES > WAVES) {
                const int t = wave + WAVES;
                if (t < ENC_TILES) mcn_gemm_seg_h<XW, 1, 1>(denc2, Xh, Xl, (t >> 1) * 32, 0, NSH, pe + (t & 1) * NSH * 128, lane);
            }
        }
        float* const dst = a.dy_save + (size_t)l * AS;           // X holds dY_l
        if (l == 0) {
            if (COPY) mcn_copy_tile_words<MT, XW, WIDTH, NT>(Xh, Xl, dst, row0, total, tid);
            break;
        }
        mcn_zero<NI, MI>(acc);
        mcn_gemm_seg_h<XW, NI, MI>(acc, Xh, Xl, mrow0, 0, NSH, pk + (L.bH[l] >> 2) + (wn * NI) * NSH * 128, lane,
            [=](int ks) { if (COPY) mcn_copy_tile_step<MT, XW, WIDTH, NT, NSH>(Xh, Xl, dst, row0, total, tid, ks); },
            [=]() { if (COPY && MCN_COPY_MODE == 2) mcn_copy_tile_words<MT, XW, WIDTH, NT>(Xh, Xl, dst, row0, total, tid); });
        __syncthreads();
        mask_store_h<WIDTH, NI, MI>(acc, a.mask_save + (size_t)(l - 1) * (AS / 32), Xh, Xl, inv, sg, mrow0, ncol0, row0, total, lane);
        __syncthreads();
    }
    __syncthreads();
    MCN_BSTAMP(3);
    // ---- encoded-input gradient -> LDS [MT][64]
    {
        const int r = lane & 31, h = lane >> 5;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int t = wave + WAVES * pass;
            if (t < ENC_TILES) {
                const int mt = t >> 1, kt = t & 1;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (pass == 0 ? denc[0][0][4 * q + e] : denc2[0][0][4 * q + e]) * inv;
                    *reinterpret_cast<f32x4*>(&Xf[mcn_swz_chunk(mt * 32 + r, (kt * 32 + 8 * q + 4 * h) >> 2, 64)]) = v;
                }
            }
        }
    }
    __syncthreads();
    MCN_BSTAMP(4);
    // ---- encoding backward -> d xyz -> d rays_o / d rays_d
    //   enc channel 3+20c+f = w_f sin(2^f x_c), 3+20c+10+f = w_f cos(2^f x_c)  (w_f already inside enc_save)
    if (a.d_rays_o || a.d_rays_d) {
        for (int it = tid; it < MT * 3; it += NT) {
            const int m = it / 3, c = it - m * 3;
            const long long g = row0 + m;
            float dx = 0.f;
            if (g < total) {
                const unsigned* en = reinterpret_cast<const unsigned*>(a.enc_save) + (size_t)g * MCN_ENCP;   // split words
                dx = Xf[mcn_swz(m, c, 64)];
#pragma unroll
                for (int f = 0; f < MCN_NFREQ; ++f) {
                    const float s = mcn_unword(en[3 + 20 * c + f], 1.0f / MCN_SX), co = mcn_unword(en[3 + 20 * c + 10 + f], 1.0f / MCN_SX);
                    const float ds = Xf[mcn_swz(m, 3 + 20 * c + f, 64)], dc = Xf[mcn_swz(m, 3 + 20 * c + 10 + f, 64)];
                    dx += (float)(1 << f) * (co * ds - s * dc);
                }
            }
            sgo[m * 4 + c] = dx;                                       // d origin
            sddir[m * 4 + c] = dx * sz[m] + sddir[m * 4 + c];          // d direction: through x = o + d z, plus the SH term
        }
        __syncthreads();
        // samples of one ray are contiguous in the tile: the first sample of each run sums the run, so a
        // ray costs 6 atomics per tile instead of 6 per sample (64-way same-address contention otherwise)
        for (int m = tid; m < MT; m += NT) {
            const int ray = sray[m];
            if (ray < 0 || (m > 0 && sray[m - 1] == ray)) continue;
            float so[3] = {0.f, 0.f, 0.f}, sd[3] = {0.f, 0.f, 0.f};
            for (int mm = m; mm < MT && sray[mm] == ray; ++mm) {
#pragma unroll
                for (int c = 0; c < 3; ++c) { so[c] += sgo[mm * 4 + c]; sd[c] += sddir[mm * 4 + c]; }
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                if (a.d_rays_o) atomicAdd(a.d_rays_o + ray * 3 + c, so[c]);
                if (a.d_rays_d) atomicAdd(a.d_rays_d + ray * 3 + c, sd[c]);
            }
        }
    }
    MCN_BSTAMP(5);
}

template <int WIDTH>
__global__ __launch_bounds__(McnGeomH<WIDTH>::WN * McnGeomH<WIDTH>::WM * 64, McnGeomH<WIDTH>::WGS_BWD)
void mlp_bwd_h_kernel(McnMlpBwdArgs a) { mlp_bwd_h_body<WIDTH, false>(a); }

#if MCN_HELP_WGS == 2
#define MCN_HELP_KERNEL_ATTR __attribute__((amdgpu_flat_work_group_size(64, 384), amdgpu_waves_per_eu(3, 3)))
#else
#define MCN_HELP_KERNEL_ATTR __launch_bounds__(256 + MCN_HELP_THREADS, 1)
#endif
template <int WIDTH>
__global__ MCN_HELP_KERNEL_ATTR void mlp_bwd_h_help_kernel(McnMlpBwdArgs a) { mlp_bwd_h_body<WIDTH, true>(a); }

template <int WIDTH>
static hipError_t launch_bwd_h(const McnMlpBwdArgs& a, long long max_rows, hipStream_t st) {
    using SM = BwdSmemH<WIDTH>;
    const int grid = (int)((max_rows + SM::MT - 1) / SM::MT);
    if (grid <= 0) return hipSuccess;
    constexpr bool HELP = MCN_HELP_MIN_WIDTH > 0 && WIDTH >= MCN_HELP_MIN_WIDTH;
    auto kern = HELP ? mlp_bwd_h_help_kernel<WIDTH> : mlp_bwd_h_kernel<WIDTH>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SM::bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(SM::NT + (HELP ? MCN_HELP_THREADS : 0)), SM::bytes, st, a);
    return hipGetLastError();
}

hipError_t mcn_launch_mlp_bwd_h(const McnMlpBwdArgs& a, hipStream_t st) {
    const long long max_rows = a.count ? (long long)a.max_rows : (long long)a.n_rays * a.S;
    switch (a.lay.width) {
        case 256: return launch_bwd_h<256>(a, max_rows, st);
        case 128: return launch_bwd_h<128>(a, max_rows, st);
        case 64:  return launch_bwd_h<64>(a, max_rows, st);
        case 32:  return launch_bwd_h<32>(a, max_rows, st);
    }
    return hipErrorInvalidValue;
}
