// Packs the fp32 Linear weights of one net into the two 16-bit fragment STREAMS of the single-pass 16-bit mode
// (mcnerf_16.h): forward (mlp16_fwd.hip consumes it front to back) and backward (transposed weights in the order of
// mlp16_bwd.hip).  One thread = one lane's 8 elements of one 1 KiB fragment.  No reference counterpart (torch's addmm
// reads nn.Linear.weight directly, model/net_block.py:69-74).
#include "mcnerf_16.h"

// Range watch (both kernels): a weight whose 16-bit image is not finite -- |w| > 65504 in f16, |w| 2^8 > 65504 in the split-f16
// mode, a non-finite w in any mode -- raises word `segment index` of `flags` (forward stream only: segment s IS packed weight
// tensor s: trunk layers 0 .. D-1, sigma.0, sh.0, sh.2).  Sticky: the caller zeroes the words once and reads them when the
// optimiser's guard has refused a step, so that "skipped step" becomes "tensor X is out of the mode's range".
__device__ __forceinline__ void mcn16_range_watch(const float (&v)[8], float limit, unsigned* flags, int seg) {
    float m = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf(v[j])) + (v[j] != v[j] ? __builtin_inff() : 0.f);
    if (!(m <= limit)) atomicOr(flags + seg, 1u);
}

// element (output o, contraction index c) of a part in the kernels' geometry <- the net's own tensor (index maps: Mcn16Part)
__device__ __forceinline__ float mcn16_pack_src(const float* __restrict__ params, const Mcn16Part& p, int transposed, int o, int c) {
    if (p.map) {
        if (mcn16_map_enc(p.map)) { if (transposed) o = mcn_enc_col(o, p.map); else c = mcn_enc_col(c, p.map); }
        else { if (transposed) c = mcn_sh_row(c, p.map - 16); else o = mcn_sh_row(o, p.map - 16); }
        if (o < 0 || c < 0) return 0.f;
    }
    return transposed ? params[p.src + (size_t)c * p.ld + p.col0 + o] : params[p.src + (size_t)o * p.ld + p.col0 + c];
}

template <bool BF>
__global__ void pack16_kernel(Mcn16Stream sf, Mcn16Stream sb, const float* __restrict__ params, char* __restrict__ pf, char* __restrict__ pb, unsigned* flags) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int nf = sf.total_frags * 64, nb = sb.total_frags * 64;
    if (gid >= nf + nb) return;
    const bool bwd = gid >= nf;
    const Mcn16Stream& st = bwd ? sb : sf;
    const int id = bwd ? gid - nf : gid;
    const int frag = id >> 6, lane = id & 63;
    const int i = lane & 31, h = lane >> 5;
    int s = 0;
    while (s + 1 < st.nseg && frag >= st.seg[s + 1].first_frag) ++s;
    const Mcn16Seg sg = st.seg[s];
    const int loc = frag - sg.first_frag;
    const int per_tile = sg.a.ksteps + sg.b.ksteps;
    const int t = loc / per_tile, ks = loc - t * per_tile;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
    if (t < sg.tiles) {                                  // (fragments past the segment's tiles are slab padding: zeros)
        const bool inb = ks >= sg.a.ksteps;
        const Mcn16Part p = inb ? sg.b : sg.a;
        const int kk = inb ? ks - sg.a.ksteps : ks;
        const int o = 32 * t + i;                        // output index of this lane's row of the fragment tile
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = mcn16_chan(kk, h, j);          // contraction index
            if (o < p.out_real && c < p.con_real) v[j] = mcn16_pack_src(params, p, sg.transposed, o, c);
        }
    }
    if (flags && !bwd) mcn16_range_watch(v, BF ? 3.0e38f : 65504.f, flags, s);
    u32x4_t w;
#pragma unroll
    for (int d = 0; d < 4; ++d) w[d] = Mcn16T<BF>::pack(v[2 * d], v[2 * d + 1]);
    *reinterpret_cast<u32x4_t*>((bwd ? pb : pf) + (size_t)id * 16) = w;
}

hipError_t mcn16_launch_pack(const McnLayout& L, const float* params, void* packed_fwd, void* packed_bwd, int bf16, unsigned* flags, hipStream_t st) {
    const Mcn16Stream sf = mcn16_fwd_stream(L), sb = mcn16_bwd_stream(L);
    const int total = (sf.total_frags + sb.total_frags) * 64;
    const int threads = 256, grid = (total + threads - 1) / threads;
    if (bf16) hipLaunchKernelGGL(pack16_kernel<true>, dim3(grid), dim3(threads), 0, st, sf, sb, params, (char*)packed_fwd, (char*)packed_bwd, flags);
    else hipLaunchKernelGGL(pack16_kernel<false>, dim3(grid), dim3(threads), 0, st, sf, sb, params, (char*)packed_fwd, (char*)packed_bwd, flags);
    return hipGetLastError();
}

// ---- split-f16 ("f16x3") streams (mcnerf_x3.h): per logical fragment the hi piece (f16(w SW)) then the lo piece
//      (f16(w SW - hi)), segments padded to whole slabs of 8 logical fragments.
#include "mcnerf_x3.h"
__global__ void packx3_kernel(Mcn16Stream sf, Mcn16Stream sb, const float* __restrict__ params, char* __restrict__ pf, char* __restrict__ pb, unsigned* flags) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int nf = sf.total_frags * 64, nb = sb.total_frags * 64;
    if (gid >= nf + nb) return;
    const bool bwd = gid >= nf;
    const Mcn16Stream& st = bwd ? sb : sf;
    const int id = bwd ? gid - nf : gid;
    const int frag = id >> 6, lane = id & 63;
    const int i = lane & 31, h = lane >> 5;
    int s = 0;
    while (s + 1 < st.nseg && frag >= st.seg[s + 1].first_frag) ++s;
    const Mcn16Seg sg = st.seg[s];
    const int loc = frag - sg.first_frag;
    const int per_tile = sg.a.ksteps + sg.b.ksteps;
    const int t = loc / per_tile, ks = loc - t * per_tile;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
    if (t < sg.tiles) {
        const bool inb = ks >= sg.a.ksteps;
        const Mcn16Part p = inb ? sg.b : sg.a;
        const int kk = inb ? ks - sg.a.ksteps : ks;
        const int o = 32 * t + i;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = mcn16_chan(kk, h, j);
            if (o < p.out_real && c < p.con_real) v[j] = mcn16_pack_src(params, p, sg.transposed, o, c);
        }
    }
    if (flags && !bwd) mcn16_range_watch(v, 65504.f / MCNX3_SW, flags, s);
    u32x4_t wh, wl;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        unsigned a, b;
        mcnx3_split2(v[2 * d] * MCNX3_SW, v[2 * d + 1] * MCNX3_SW, a, b);
        wh[d] = a; wl[d] = b;
    }
    char* dst = (bwd ? pb : pf) + (size_t)frag * 2048 + lane * 16;
    *reinterpret_cast<u32x4_t*>(dst) = wh;
    *reinterpret_cast<u32x4_t*>(dst + 1024) = wl;
}

hipError_t mcnx3_launch_pack(const McnLayout& L, const float* params, void* packed_fwd, void* packed_bwd, unsigned* flags, hipStream_t st) {
    const Mcn16Stream sf = mcnx3_fwd_stream(L), sb = mcnx3_bwd_stream(L);
    const int total = (sf.total_frags + sb.total_frags) * 64;
    const int threads = 256, grid = (total + threads - 1) / threads;
    hipLaunchKernelGGL(packx3_kernel, dim3(grid), dim3(threads), 0, st, sf, sb, params, (char*)packed_fwd, (char*)packed_bwd, flags);
    return hipGetLastError();
}
