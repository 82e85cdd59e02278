// Weight / bias gradients of the NeRF MLPs for gfx950:  dW[n][k] = sum_m dY[m][n] X[m][k],
// db[n] = sum_m dY[m][n], over the (hundreds of thousands of) samples of one step.
//
// Replaces the dW half of autograd's addmm backward for every nn.Linear of CorseFine_NeRF
// (model/net_block.py:51-65).  The reduction runs over samples, so each workgroup takes a chunk of
// rows, keeps its full (wave-tiled) dW block in MFMA accumulators for the whole chunk and issues one
// float-atomic pass at the end (>= 512 FLOP per atomic byte, far above the atomic roofline).
// (Design notes at the kernel.)
#include "mcnerf_kernels.h"

struct DwSeg {
    const float* dY; int ldy;     // [rows][ldy], columns nbase.. are the outputs
    const float* X;  int ldx;     // [rows][ldx]
    int N, n_lo, n_real;          // padded output count; outputs n_lo <= n < n_real are real (row n - n_lo of dW)
    int K, k_real;                // padded / real input count
    float* dW; int ldw;           // destination (already offset to the segment's first column)
    float* db;                    // bias gradient or null
};

// cache policy of the operand stream (LDS-DMA aux bits on gfx950: 1 = sc0, 2 = nt, 16 = sc1): the rows are read once
#define MCN_DW_AUX 2
#define DW_SLAB_ROWS 16
template <int V> struct VecT;
template <> struct VecT<1> { typedef float T; };
template <> struct VecT<2> { typedef float T __attribute__((ext_vector_type(2))); };
template <> struct VecT<4> { typedef float T __attribute__((ext_vector_type(4))); };
template <int V> __device__ __forceinline__ float vget(const typename VecT<V>::T& v, int i) { return v[i]; }
template <> __device__ __forceinline__ float vget<1>(const float& v, int) { return v; }

// One workgroup = 8 waves, PERSISTENT: the grid is one workgroup per CU and each takes a contiguous
// chunk of ceil(rows / grid) rows, so the float-atomic epilogue (the full dW block, 256 KB at width 256)
// is paid once per CU instead of once per 1024 rows (the chip-wide atomic rate is only ~1.3 TB/s).
// Operands are staged through LDS in slabs of RS rows (LDS-DMA into a 3-deep ring, one raw barrier per
// slab, counted vmcnt): every dY / X row is fetched from HBM once per workgroup and shared by the 8 waves,
// instead of each wave re-fetching its fragments (4x / 2x redundant) through the L1.
//   wave tile = (32*VN) outputs x (32*KT) inputs; lane (r = lane&31, h = lane>>5) owns outputs
//   nbase + VN*r + t (interleaved, so its A fragment is ONE ds_read of VN floats) and inputs
//   kbase + 32*kt + r (contiguous per tile, so the final atomics are 128-byte row segments).
template <int N, int K>
__global__ __launch_bounds__(512) void dw_kernel(DwSeg s, const int* count, int rows_cap) {
    // N, K (padded output / input counts) are compile-time so that the slab copy is a fixed, branch-free
    // sequence (a runtime-predicated load is an exec-masked branch and serialises on vmcnt(0))
    constexpr int VN = N >= 128 ? 4 : N / 32;
    constexpr int KT = N == 32 ? (K >= 128 ? 4 : K / 32) : (K >= 64 ? 2 : 1);
    typedef typename VecT<VN>::T AV;
    constexpr int RS = DW_SLAB_ROWS;               // rows per slab (8 MFMA k-steps of 2 rows)
    // LDS ring of 3 slabs: slab s is consumed while slab s+2 is in flight
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int rows = count ? min(*count, rows_cap) : rows_cap;
    int chunk = (rows + (int)gridDim.x - 1) / (int)gridDim.x;
    chunk = (chunk + RS - 1) / RS * RS;
    const int r0 = blockIdx.x * chunk;
    if (r0 >= rows) return;
    const int r1 = min(r0 + chunk, rows);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    constexpr int NG = N / (32 * VN), KG = K / (32 * KT);
    constexpr int G = NG * KG;              // wave tiles per workgroup; the remaining factor splits the slab rows
    constexpr int MS = 8 / G;
    static_assert(G >= 1 && G <= 8 && 8 % G == 0, "wave tiling");
    constexpr int PP = RS / 2 / MS;         // row pairs per wave per slab
    static_assert(PP >= 1, "slab too small for the row split");
    const int gi = wave % G, ms = wave / G;
    const int nbase = (gi % NG) * 32 * VN, kbase = (gi / NG) * 32 * KT;
    constexpr int slab = RS * (N + K);      // floats per buffer: sY[RS][N] then sX[RS][K]
    constexpr int n4 = N / 4, k4 = K / 4;
    constexpr int tot4 = RS * (n4 + k4);    // float4 per slab
    constexpr int NP = (tot4 + 511) / 512;  // LDS-DMA pieces (wave-instructions) per wave per slab

    f32x16 acc[VN][KT];
    mcn_zero<VN, KT>(acc);
    float bsum[VN];
#pragma unroll
    for (int t = 0; t < VN; ++t) bsum[t] = 0.f;

    // Slab rows -> LDS by LDS-DMA (global_load_lds, 16 B per lane): no staging registers.  The destination of
    // one wave-instruction is wave-uniform base + lane*16, i.e. exactly the linear slab image (float4 index
    // q = tid + i*512).  Rows beyond the chunk are clamped to its last row (finite data) and neutralised by
    // zeroing the A fragment when it is read.  (Segment fields are copied to locals: a by-reference capture of
    // the kernel-argument struct makes hipcc re-read it with ordinary global loads inside the loop, and any
    // such load next to LDS-DMA drains vmcnt(0).)
    const float* const gY = s.dY;
    const float* const gX = s.X;
    const int ldy = s.ldy, ldx = s.ldx;
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
    auto piece = [=](int i, int base_row, float* buf) {
        const int q = tid + i * 512;
        if ((i + 1) * 512 <= tot4 || q < tot4) {
            const bool isY = q < RS * n4;
            const int qq = isY ? q : q - RS * n4;
            const int w4 = isY ? n4 : k4;
            const int row = qq / w4, c4 = qq - row * w4;
            const int grow = base_row + row;
            const int rc = grow < r1 ? grow : r1 - 1;
            const float* src = isY ? gY + (size_t)rc * ldy + 4 * c4 : gX + (size_t)rc * ldx + 4 * c4;
            float* dst = buf + 4 * (i * 512 + (tid & ~63));                // wave-uniform; the lane offset is implicit
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)dst, 16, 0, MCN_DW_AUX);
        }
    };
    // All VMEM operations of the main loop are these pieces, NP per slab per wave, in program order, so
    // "slab s+1 has landed" == at most NP (the pieces of slab s+2) still outstanding: counted vmcnt + raw
    // s_barrier (a __syncthreads() would drain vmcnt(0) and expose the HBM latency every slab).
    // (When tot4 is not a multiple of 512 the trailing waves issue only NP-1 pieces per slab, so the count
    // allowed in flight is per wave: `full` is wave-uniform.)
    const bool full = (tot4 % 512 == 0) || (__builtin_amdgcn_readfirstlane(tid >> 6) * 64 + (NP - 1) * 512 < tot4);
#define DW_WAIT_ASM(n) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(n) : "memory")
#define DW_WAIT_BARRIER(k) do { if (full) DW_WAIT_ASM((k) * NP); else DW_WAIT_ASM((k) * (NP - 1)); } while (0)
    float* b_cur = lds;
    float* b_nxt = lds + slab;
    float* b_fill = lds + 2 * slab;
#pragma unroll
    for (int i = 0; i < NP; ++i) piece(i, r0, b_cur);
    if (r0 + RS < r1) {
#pragma unroll
        for (int i = 0; i < NP; ++i) piece(i, r0 + RS, b_nxt);
        DW_WAIT_BARRIER(1);
    } else {
        DW_WAIT_BARRIER(0);
    }
    for (int base = r0; base < r1; base += RS) {
        const bool fill = base + 2 * RS < r1;          // workgroup-uniform
        const float* sY = b_cur;
        const float* sX = sY + RS * N;
        // operand fragments are read from LDS one row pair ahead of the MFMAs that use them
        AV a_n = *reinterpret_cast<const AV*>(sY + (2 * ms + h) * N + nbase + VN * r);
        float b_n[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) b_n[kt] = sX[(2 * ms + h) * K + kbase + 32 * kt + r];
#pragma unroll
        for (int pp = 0; pp < PP; ++pp) {              // row pair p: lanes h=0 take row 2p, h=1 row 2p+1
            const int row = 2 * (ms + pp * MS) + h;
            AV a_c = a_n;
            float b_c[KT];
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) b_c[kt] = b_n[kt];
            if (pp + 1 < PP) {
                const int rown = row + 2 * MS;
                a_n = *reinterpret_cast<const AV*>(sY + rown * N + nbase + VN * r);
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) b_n[kt] = sX[rown * K + kbase + 32 * kt + r];
            }
            if (fill) {                                // this slab's share of the DMA, spread between the MFMA groups
#pragma unroll
                for (int i = 0; i < NP; ++i)
                    if ((i * PP) / NP == pp) piece(i, base + 2 * RS, b_fill);
            }
            if (base + row >= r1) a_c = AV(0.f);
            __builtin_amdgcn_sched_barrier(0);         // keep the next pair's LDS reads ABOVE this pair's MFMAs
#pragma unroll
            for (int t = 0; t < VN; ++t) {
                const float a1 = vget<VN>(a_c, t);
                bsum[t] += a1;
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) acc[t][kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b_c[kt], acc[t][kt], 0, 0, 0);
            }
        }
        // everyone is done reading b_cur; slab base+RS (in b_nxt) has landed for every wave
        if (fill) DW_WAIT_BARRIER(1); else DW_WAIT_BARRIER(0);
        float* t = b_cur; b_cur = b_nxt; b_nxt = b_fill; b_fill = t;
    }
#undef DW_WAIT_BARRIER
#undef DW_WAIT_ASM
    // accumulators -> global (float atomics; one register = two 128-byte row segments)
#pragma unroll
    for (int t = 0; t < VN; ++t)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            const int k = kbase + 32 * kt + r;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = nbase + VN * ((e & 3) + 8 * (e >> 2) + 4 * h) + t;
                if (n >= s.n_lo && n < s.n_real && k < s.k_real) atomicAdd(s.dW + (size_t)(n - s.n_lo) * s.ldw + k, acc[t][kt][e]);
            }
        }
    if (s.db && kbase == 0) {
#pragma unroll
        for (int t = 0; t < VN; ++t) {
            const float b = bsum[t] + __shfl_xor(bsum[t], 32);
            const int n = nbase + VN * r + t;
            if (h == 0 && n >= s.n_lo && n < s.n_real) atomicAdd(s.db + (n - s.n_lo), b);
        }
    }
}

static int dw_num_cus() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}

template <class Kern, class... Args>
static hipError_t dw_launch(Kern kern, int grid, int threads, size_t lds, hipStream_t st, Args... args) {
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, st, args...);
    return hipGetLastError();
}

// one segment: exact-fp32 kernel, or the split-f16 kernel for its shape (256 x 256: first formulation, see above)
template <int NN, int KK>
static hipError_t launch_seg_t(const DwSeg& s, const int* count, int rows_cap, int grid, size_t lds, hipStream_t st,
                               const unsigned int* gmax_bits, bool split16) {
    (void)gmax_bits; (void)split16;
    return dw_launch(dw_kernel<NN, KK>, grid, 512, lds, st, s, count, rows_cap);
}

static hipError_t launch_seg(const DwSeg& s, const int* count, int rows_cap, hipStream_t st, const unsigned int* gmax_bits = nullptr, bool split16 = false) {
    if (rows_cap <= 0) return hipSuccess;
    int grid = dw_num_cus();                                   // persistent: one workgroup per CU
    const int max_wgs = (rows_cap + DW_SLAB_ROWS - 1) / DW_SLAB_ROWS;
    if (grid > max_wgs) grid = max_wgs;
    const size_t lds = (size_t)3 * DW_SLAB_ROWS * (s.N + s.K) * sizeof(float);
#define DW_LAUNCH(NN, KK) return launch_seg_t<NN, KK>(s, count, rows_cap, grid, lds, st, gmax_bits, split16)
    switch (s.N * 1000 + s.K) {
        case 256256: DW_LAUNCH(256, 256);
        case 256064: DW_LAUNCH(256, 64);
        case 32256:  DW_LAUNCH(32, 256);
        case 128128: DW_LAUNCH(128, 128);
        case 128064: DW_LAUNCH(128, 64);
        case 32128:  DW_LAUNCH(32, 128);
        case 64064:  DW_LAUNCH(64, 64);
        case 32064:  DW_LAUNCH(32, 64);
        case 32032:  DW_LAUNCH(32, 32);
        default: return hipErrorInvalidValue;
    }
#undef DW_LAUNCH
}

hipError_t mcn_launch_dw(const McnDwArgs& a, hipStream_t st) {
    const McnLayout& L = a.lay;
    const int W = L.width, D = L.depth;
    const size_t AS = a.act_stride;
    auto act = [&](int slot) { return a.act_save + (size_t)slot * AS; };
    auto dy = [&](int slot) { return a.dy_save + (size_t)slot * AS; };
    hipError_t e;
    for (int l = 0; l < D; ++l) {
        const int ldw = mcn_layer_in(L, l);
        const bool takes_enc = l == 0 || ((L.skip_mask >> l) & 1u);
        if (takes_enc) {                  // encoded-input columns
            DwSeg s = {dy(l), W, a.enc_save, MCN_ENCP, W, 0, W, MCN_ENCP, L.nenc, a.grads + L.pW[l], ldw, a.grads + L.pB[l]};
            if ((e = launch_seg(s, a.count, a.rows, st, a.gmax_bits, a.split16)) != hipSuccess) return e;
        }
        if (l > 0) {                      // hidden-input columns (after the 63 encoded ones at the skip layer)
            DwSeg s = {dy(l), W, act(l - 1), W, W, 0, W, W, W, a.grads + L.pW[l] + (takes_enc ? L.nenc : 0), ldw,
                       takes_enc ? nullptr : a.grads + L.pB[l]};
            if ((e = launch_seg(s, a.count, a.rows, st, a.gmax_bits, a.split16)) != hipSuccess) return e;
        }
    }
    {   // sigma.0 and sh.0 read the last trunk activation; sh.2 reads the sh hidden layer
        DwSeg s1 = {dy(D), W, act(D - 1), W, W, 0, W, W, W, a.grads + L.pWs1, W, a.grads + L.pBs1};
        if ((e = launch_seg(s1, a.count, a.rows, st, a.gmax_bits, a.split16)) != hipSuccess) return e;
        DwSeg c1 = {dy(D + 1), W, act(D - 1), W, W, 0, W, W, W, a.grads + L.pWc1, W, a.grads + L.pBc1};
        if ((e = launch_seg(c1, a.count, a.rows, st, a.gmax_bits, a.split16)) != hipSuccess) return e;
        // sh.2: nsh = 3 (deg + 1)^2 outputs in rows of nshp floats (32, or 64 at degree 3: two 32-column halves of the same rows)
        const int NP_ = L.nshp, NS_ = L.nsh;
        for (int half = 0; half * 32 < NP_; ++half) {
            const int real = NS_ - 32 * half < 32 ? NS_ - 32 * half : 32;
            if (real <= 0) break;
            DwSeg c2 = {a.dsh_save + 32 * half, NP_, act(D + 1), W, 32, 0, real, W, W, a.grads + L.pWc2 + (size_t)32 * half * W, W, a.grads + L.pBc2 + 32 * half};
            if ((e = launch_seg(c2, a.count, a.rows, st, a.gmax_bits, a.split16)) != hipSuccess) return e;
        }
        // sigma.2 (1 x W): d sigma sits in the spare column nsh of dsh_save (27 at degree 2), its input is the sigma hidden layer
        DwSeg s2 = {a.dsh_save + 32 * (NS_ / 32), NP_, act(D), W, 32, NS_ % 32, NS_ % 32 + 1, W, W, a.grads + L.pWs2, W, a.grads + L.pBs2};
        if ((e = launch_seg(s2, a.count, a.rows, st, a.gmax_bits, a.split16)) != hipSuccess) return e;
    }
    return hipSuccess;
}
