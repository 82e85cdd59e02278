// Re-lays the [out][in] Linear weights of one net into MFMA operand-fragment order (both the
// forward orientation and the transposed one used by dX = W^T dY).  One launch per optimiser step;
// 2 x 0.63 M floats for the 8x256 net, i.e. microseconds.
#include "mcnerf_kernels.h"

struct PackSeg {
    int src, ld, col0;       // W[n][k] = params[src + n*ld + col0 + k]
    int n_real, k_real;      // valid extent (zero padding beyond)
    int n_pad, k_pad;
    int dst_f, dst_b;        // float offsets in the packed buffer
    int first4;              // first float4 index (within one orientation) of this segment
};
struct PackTable {
    int nseg;
    int total4;              // float4 count of one orientation
    PackSeg seg[2 * MCN_MAXD + 4];
};

__global__ void pack_kernel(PackTable t, const float* __restrict__ params, float* __restrict__ packed) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= 2 * t.total4) return;
    const bool transposed = gid >= t.total4;
    const int id4 = transposed ? gid - t.total4 : gid;
    int s = 0;
    while (s + 1 < t.nseg && id4 >= t.seg[s + 1].first4) ++s;
    const PackSeg sg = t.seg[s];
    const int loc = id4 - sg.first4;          // float4 index inside the segment
    const int lane = loc & 63;
    const int blk = loc >> 6;                 // (tile, step) flattened
    const int r = lane & 31, h = lane >> 5;
    f32x4 v;
    if (!transposed) {
        const int KS = sg.k_pad / 8;
        const int ntile = blk / KS, ks = blk - ntile * KS;
        const int n = 32 * ntile + r;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = 8 * ks + 4 * h + i;
            v[i] = (n < sg.n_real && k < sg.k_real) ? params[sg.src + n * sg.ld + sg.col0 + k] : 0.f;
        }
        *reinterpret_cast<f32x4*>(packed + sg.dst_f + (size_t)loc * 4) = v;
    } else {
        const int NS = sg.n_pad / 8;
        const int ktile = blk / NS, ns = blk - ktile * NS;
        const int k = 32 * ktile + r;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = 8 * ns + 4 * h + i;
            v[i] = (n < sg.n_real && k < sg.k_real) ? params[sg.src + n * sg.ld + sg.col0 + k] : 0.f;
        }
        *reinterpret_cast<f32x4*>(packed + sg.dst_b + (size_t)loc * 4) = v;
    }
}


static void build_pack_table(const McnLayout& L, PackTable& t);

hipError_t mcn_launch_pack(const McnLayout& L, const float* params, float* packed, hipStream_t st) {
    PackTable t;
    build_pack_table(L, t);
    const int threads = 256;
    const int grid = (2 * t.total4 + threads - 1) / threads;
    hipLaunchKernelGGL(pack_kernel, dim3(grid), dim3(threads), 0, st, t, params, packed);
    return hipGetLastError();
}

static void build_pack_table(const McnLayout& L, PackTable& t) {
    int ns = 0, first4 = 0;
    auto add = [&](int src, int ld, int col0, int n_real, int k_real, int n_pad, int k_pad, int dst_f, int dst_b) {
        PackSeg& s = t.seg[ns++];
        s.src = src; s.ld = ld; s.col0 = col0; s.n_real = n_real; s.k_real = k_real;
        s.n_pad = n_pad; s.k_pad = k_pad; s.dst_f = dst_f; s.dst_b = dst_b; s.first4 = first4;
        first4 += n_pad * k_pad / 4;
    };
    const int W = L.width;
    add(L.pW[0], L.nenc, 0, W, L.nenc, W, MCN_ENCP, L.fEnc0, L.bEnc0);
    for (int i = 1; i < L.depth; ++i) {
        if ((L.skip_mask >> i) & 1u) {
            add(L.pW[i], W + L.nenc, 0, W, L.nenc, W, MCN_ENCP, L.fEncS[i], L.bEncS[i]);
            add(L.pW[i], W + L.nenc, L.nenc, W, W, W, W, L.fH[i], L.bH[i]);
        } else {
            add(L.pW[i], W, 0, W, W, W, W, L.fH[i], L.bH[i]);
        }
    }
    add(L.pWs1, W, 0, W, W, W, W, L.fS1, L.bS1);
    add(L.pWc1, W, 0, W, W, W, W, L.fC1, L.bC1);
    add(L.pWc2, W, 0, L.nsh, W, L.nshp, W, L.fC2, L.bC2);
    t.nseg = ns;
    t.total4 = first4;
}
