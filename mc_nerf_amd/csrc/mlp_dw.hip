// Weight / bias gradients of the NeRF MLPs for gfx950:  dW[n][k] = sum_m dY[m][n] X[m][k],
// db[n] = sum_m dY[m][n], over the (hundreds of thousands of) samples of one step.
//
// Replaces the dW half of autograd's addmm backward for every nn.Linear of CorseFine_NeRF
// (model/net_block.py:51-65).  The reduction runs over samples, so each workgroup takes a chunk of
// rows, keeps its full (wave-tiled) dW block in MFMA accumulators for the whole chunk and issues one
// float-atomic pass at the end (>= 512 FLOP per atomic byte, far above the atomic roofline).
// Operands stream straight from HBM/L2 into registers: per two sample rows a lane loads VN
// consecutive dY values (the n index is interleaved over the VN n-tiles so this is one vector load)
// and KT separate X values (k contiguous per tile, so the final atomics are 128-byte row segments).
#include "mcnerf_common.h"
#include "mcnerf_kernels.h"

struct DwSeg {
    const float* dY; int ldy;     // [rows][ldy], columns nbase.. are the outputs
    const float* X;  int ldx;     // [rows][ldx]
    int N, n_real;                // padded / real output count
    int K, k_real;                // padded / real input count
    float* dW; int ldw;           // destination (already offset to the segment's first column)
    float* db;                    // bias gradient or null
};

template <int V> struct VecT;
template <> struct VecT<1> { typedef float T; };
template <> struct VecT<2> { typedef float T __attribute__((ext_vector_type(2))); };
template <> struct VecT<4> { typedef float T __attribute__((ext_vector_type(4))); };
template <int V> __device__ __forceinline__ float vget(const typename VecT<V>::T& v, int i) { return v[i]; }
template <> __device__ __forceinline__ float vget<1>(const float& v, int) { return v; }

template <int VN, int KT>
__global__ __launch_bounds__(512) void dw_kernel(DwSeg s, const int* count, int rows_cap, int rows_per_wg) {
    typedef typename VecT<VN>::T AV;
    const int rows = count ? min(*count, rows_cap) : rows_cap;
    const int r0 = blockIdx.x * rows_per_wg;
    if (r0 >= rows) return;
    const int r1 = min(r0 + rows_per_wg, rows);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int NG = s.N / (32 * VN), KG = s.K / (32 * KT);
    const int G = NG * KG;                  // wave tiles per workgroup; the remaining factor splits rows
    const int MS = 8 / G;
    const int gi = wave % G, ms = wave / G;
    const int nbase = (gi % NG) * 32 * VN, kbase = (gi / NG) * 32 * KT;

    f32x16 acc[VN][KT];
    mcn_zero<VN, KT>(acc);
    float bsum[VN];
#pragma unroll
    for (int t = 0; t < VN; ++t) bsum[t] = 0.f;

    const float* pa = s.dY + nbase + VN * r;
    const float* pb = s.X + kbase + r;
    // Register pipeline: PF row pairs in flight per wave (HBM latency ~2 us vs 8 MFMAs = 0.2 us per pair).
    constexpr int PF = 8;
    AV av[PF];
    float bv[PF][KT];
    const int stride = 2 * MS;
    auto load = [&](AV& a_dst, float (&b_dst)[KT], int mrow) {
        const int row = mrow + h;
        const bool ok = row < r1;
        if (ok) a_dst = *reinterpret_cast<const AV*>(pa + (size_t)row * s.ldy); else a_dst = AV(0.f);
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) b_dst[kt] = ok ? pb[(size_t)row * s.ldx + 32 * kt] : 0.f;
    };
    int m = r0 + 2 * ms;
#pragma unroll
    for (int p = 0; p < PF; ++p) load(av[p], bv[p], m + p * stride);
    for (; m < r1; m += PF * stride) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            const AV a_c = av[p];
            float b_c[KT];
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) b_c[kt] = bv[p][kt];
            load(av[p], bv[p], m + (p + PF) * stride);          // refill the slot for PF pairs later
#pragma unroll
            for (int t = 0; t < VN; ++t) {
                const float a1 = vget<VN>(a_c, t);
                bsum[t] += a1;
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) acc[t][kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b_c[kt], acc[t][kt], 0, 0, 0);
            }
        }
    }
    // accumulators -> global (float atomics; one register = two 128-byte row segments)
#pragma unroll
    for (int t = 0; t < VN; ++t)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            const int k = kbase + 32 * kt + r;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = nbase + VN * ((e & 3) + 8 * (e >> 2) + 4 * h) + t;
                if (n < s.n_real && k < s.k_real) atomicAdd(s.dW + (size_t)n * s.ldw + k, acc[t][kt][e]);
            }
        }
    if (s.db && kbase == 0) {
#pragma unroll
        for (int t = 0; t < VN; ++t) {
            const float b = bsum[t] + __shfl_xor(bsum[t], 32);
            const int n = nbase + VN * r + t;
            if (h == 0 && n < s.n_real) atomicAdd(s.db + n, b);
        }
    }
}

static hipError_t launch_seg(const DwSeg& s, const int* count, int rows_cap, hipStream_t st) {
    const int rows_per_wg = 1024;
    const int grid = (rows_cap + rows_per_wg - 1) / rows_per_wg;
    if (grid <= 0) return hipSuccess;
    const int vn = s.N >= 128 ? 4 : s.N / 32;
    int kt;
    if (s.N == 32) kt = s.K >= 128 ? 4 : s.K / 32;
    else kt = s.K >= 64 ? 2 : 1;
#define DW_LAUNCH(VN, KT) hipLaunchKernelGGL((dw_kernel<VN, KT>), dim3(grid), dim3(512), 0, st, s, count, rows_cap, rows_per_wg)
    if (vn == 4 && kt == 2) DW_LAUNCH(4, 2);
    else if (vn == 2 && kt == 2) DW_LAUNCH(2, 2);
    else if (vn == 1 && kt == 1) DW_LAUNCH(1, 1);
    else if (vn == 1 && kt == 2) DW_LAUNCH(1, 2);
    else if (vn == 1 && kt == 4) DW_LAUNCH(1, 4);
    else return hipErrorInvalidValue;
#undef DW_LAUNCH
    return hipGetLastError();
}

hipError_t mcn_launch_dw(const McnDwArgs& a, hipStream_t st) {
    const McnLayout& L = a.lay;
    const int W = L.width, D = L.depth;
    const size_t AS = a.act_stride;
    auto act = [&](int slot) { return a.act_save + (size_t)slot * AS; };
    auto dy = [&](int slot) { return a.dy_save + (size_t)slot * AS; };
    hipError_t e;
    for (int l = 0; l < D; ++l) {
        const int ldw = mcn_in_features(D, W, L.skip, l);
        if (l == 0 || l == L.skip) {      // encoded-input columns
            DwSeg s = {dy(l), W, a.enc_save, MCN_ENCP, W, W, MCN_ENCP, MCN_ENC, a.grads + L.pW[l], ldw, a.grads + L.pB[l]};
            if ((e = launch_seg(s, a.count, a.rows, st)) != hipSuccess) return e;
        }
        if (l > 0) {                      // hidden-input columns (after the 63 encoded ones at the skip layer)
            DwSeg s = {dy(l), W, act(l - 1), W, W, W, W, W, a.grads + L.pW[l] + (l == L.skip ? MCN_ENC : 0), ldw,
                       l == L.skip ? nullptr : a.grads + L.pB[l]};
            if ((e = launch_seg(s, a.count, a.rows, st)) != hipSuccess) return e;
        }
    }
    {   // sigma.0 and sh.0 read the last trunk activation; sh.2 reads the sh hidden layer
        DwSeg s1 = {dy(D), W, act(D - 1), W, W, W, W, W, a.grads + L.pWs1, W, a.grads + L.pBs1};
        if ((e = launch_seg(s1, a.count, a.rows, st)) != hipSuccess) return e;
        DwSeg c1 = {dy(D + 1), W, act(D - 1), W, W, W, W, W, a.grads + L.pWc1, W, a.grads + L.pBc1};
        if ((e = launch_seg(c1, a.count, a.rows, st)) != hipSuccess) return e;
        DwSeg c2 = {a.dsh_save, MCN_NSHP, act(D + 1), W, MCN_NSHP, MCN_NSH, W, W, a.grads + L.pWc2, W, a.grads + L.pBc2};
        if ((e = launch_seg(c2, a.count, a.rows, st)) != hipSuccess) return e;
    }
    return hipSuccess;
}
