// Single-pass 16-bit MFMA mode ("f16" / "bf16") of the NeRF MLP kernels for gfx950: one v_mfma_f32_32x32x16_{f16|bf16}
// per product, fp32 accumulate, 2-byte saved operands.  This is a different kernel structure from the fp32 / split-f16
// chains (mlp_fwd*.hip), designed around three facts of the hardware:
//
//  * REGISTER-CHAINED LAYERS.  In the "sample on the lane" orientation D[n][m] = sum_k W[n][k] X[m][k] the 32x32
//    accumulator of an output tile holds, in lane (m, h), the outputs n = 32t + 8q + 4h + e (register 4q + e).  Converted
//    pairwise to 16 bit, registers 8u .. 8u+7 ARE the B-operand fragment of k-step 2t + u of the next layer, provided the
//    next layer's weights are packed with the contraction index permuted the same way:
//        position (k-step s, lane half h, element j)  <->  channel c(s,h,j) = 16 s + 8 (j >> 2) + 4 h + (j & 3).
//    So a wave carries its 32 samples through the whole network in registers: no activation tile in LDS, no barrier
//    between layers, no LDS write-back.  Only one output tile's accumulator (16 registers) is live at a time.
//  * SHARED WEIGHT STREAM.  Every wave needs every weight fragment, so the 8 waves (256 samples) of a workgroup share
//    one LDS ring of 16-fragment slabs that is filled by LDS-DMA (global_load_lds, 1 KiB per wave instruction) several
//    slabs ahead of its use, walking the packed buffer LINEARLY: the packed weights are stored in exactly the order the
//    kernel consumes them, so the loader needs no knowledge of layers.  One raw s_barrier + counted vmcnt per slab
//    (= per 16 MFMAs per wave); L2 -> CU weight traffic is 512 B per sample-layer instead of 4 KB (64-row split tiles).
//  * FRAGMENT-MAJOR WORKSPACES.  The operands saved for the weight-gradient kernel are written straight from
//    registers as whole fragments (one 1 KiB coalesced store per k-step per wave):
//        ws[slot][tile of 32 rows][k-step][lane = 32 h + m][8 x 16 bit]
//    and the weight-gradient kernel reads them back with ds_read_b64_tr_b16 (mlp16_dw.hip).  ReLU masks are 1 bit per
//    value in the same lane-local arrangement.
#pragma once
#include "mcnerf_common.h"

#define MCN16_SLAB 16                 // fragments (1 KiB each) per ring slab (a multiple of 8: every wave issues SLAB / 8 pieces)
#define MCN16_RING (128 / MCN16_SLAB) // slabs in the LDS ring (128 KiB; a power of two)
#define MCN16_AHEAD (MCN16_RING - 2)  // slabs in flight ahead of the one being consumed (early sync: RING >= AHEAD + 2)
#define MCN16_PPW (MCN16_SLAB / 8)    // LDS-DMA pieces per wave and slab
#define MCN16_WAVES 8
#define MCN16_ROWS (32 * MCN16_WAVES) // rows per workgroup pass
#define MCN16_ENCKS 4                 // k-steps of the 64 (63 + pad) encoded channels

// One "part" of a packed segment: a [rows][ld] block of the flat parameter buffer.
//   forward    : fragment (tile t, step s), lane (i, h), element j = src[(32 t + i) * ld + col0 + c(s,h,j)]
//   transposed : ............................................... = src[c(s,h,j) * ld + col0 + 32 t + i]
// out_real = valid outputs (rows of the fragment tile), con_real = valid extent of the contraction index.
// map: 0 = indices as they are; 1 .. 10 = the part's ENCODED-channel index (the contraction index, or the output index of a
// transposed part) goes through mcn_enc_col(., map); 16 + deg = its sh.2-row index (the output index, or the contraction index of a
// transposed part) through mcn_sh_row(., deg) -- nets with fewer than 10 frequencies / an SH degree below 2 (mcnerf_common.h).
struct Mcn16Part { int src, ld, col0, out_real, con_real, ksteps, map; };
static inline __host__ __device__ int mcn16_map_enc(int map) { return map >= 1 && map <= MCN_NFREQ; }
struct Mcn16Seg { int first_frag, tiles, transposed; Mcn16Part a, b; };   // per tile: a.ksteps fragments of part a, then b.ksteps of part b
#define MCN16_MAXSEG 24
struct Mcn16Stream { int nseg, total_frags; Mcn16Seg seg[MCN16_MAXSEG]; };

static inline int mcn16_pad(int frags) { return (frags + MCN16_SLAB - 1) / MCN16_SLAB * MCN16_SLAB; }

static inline void mcn16_add(Mcn16Stream& st, int tiles, int transposed, Mcn16Part a, Mcn16Part b) {
    Mcn16Seg& s = st.seg[st.nseg++];
    s.first_frag = st.total_frags; s.tiles = tiles; s.transposed = transposed; s.a = a; s.b = b;
    st.total_frags += mcn16_pad(tiles * (a.ksteps + b.ksteps));
}

// Forward stream: layer 0, trunk 1..D-1 (skip layer: encoded k-steps first, then the hidden ones), sigma.0, sh.0, sh.2.
static inline Mcn16Stream mcn16_fwd_stream(const McnLayout& L) {
    Mcn16Stream st; st.nseg = 0; st.total_frags = 0;
    const int W = L.width, NT = W / 32, KS = W / 16;
    const int ME = L.nfreq == MCN_NFREQ ? 0 : L.nfreq, MS = L.sh_deg == 2 ? 0 : 16 + L.sh_deg;      // (index maps: see Mcn16Part)
    const Mcn16Part none = {0, 0, 0, 0, 0, 0, 0};
    mcn16_add(st, NT, 0, Mcn16Part{L.pW[0], L.nenc, 0, W, MCN_ENC, MCN16_ENCKS, ME}, none);
    for (int l = 1; l < L.depth; ++l) {
        if (l == L.skip) mcn16_add(st, NT, 0, Mcn16Part{L.pW[l], W + L.nenc, 0, W, MCN_ENC, MCN16_ENCKS, ME},
                                   Mcn16Part{L.pW[l], W + L.nenc, L.nenc, W, W, KS, 0});
        else mcn16_add(st, NT, 0, Mcn16Part{L.pW[l], W, 0, W, W, KS, 0}, none);
    }
    mcn16_add(st, NT, 0, Mcn16Part{L.pWs1, W, 0, W, W, KS, 0}, none);
    mcn16_add(st, NT, 0, Mcn16Part{L.pWc1, W, 0, W, W, KS, 0}, none);
    mcn16_add(st, 1, 0, Mcn16Part{L.pWc2, W, 0, MCN_NSH, W, KS, MS}, none);
    return st;
}
// Backward stream (dX = W^T dY, consumption order of mlp16_bwd.hip): sigma.0^T, sh.2^T, sh.0^T, then for l = D-1 .. 1
// [the encoded columns of the skip layer (2 tiles), the hidden columns], and the encoded columns of layer 0.
static inline Mcn16Stream mcn16_bwd_stream(const McnLayout& L) {
    Mcn16Stream st; st.nseg = 0; st.total_frags = 0;
    const int W = L.width, NT = W / 32, KS = W / 16;
    const int ME = L.nfreq == MCN_NFREQ ? 0 : L.nfreq, MS = L.sh_deg == 2 ? 0 : 16 + L.sh_deg;
    const Mcn16Part none = {0, 0, 0, 0, 0, 0, 0};
    mcn16_add(st, NT, 1, Mcn16Part{L.pWs1, W, 0, W, W, KS, 0}, none);
    mcn16_add(st, NT, 1, Mcn16Part{L.pWc2, W, 0, W, MCN_NSH, 2, MS}, none);
    mcn16_add(st, NT, 1, Mcn16Part{L.pWc1, W, 0, W, W, KS, 0}, none);
    for (int l = L.depth - 1; l >= 1; --l) {
        const int ld = (l == L.skip) ? W + L.nenc : W;
        if (l == L.skip) mcn16_add(st, 2, 1, Mcn16Part{L.pW[l], ld, 0, MCN_ENC, W, KS, ME}, none);
        mcn16_add(st, NT, 1, Mcn16Part{L.pW[l], ld, (l == L.skip) ? L.nenc : 0, W, W, KS, 0}, none);
    }
    mcn16_add(st, 2, 1, Mcn16Part{L.pW[0], L.nenc, 0, MCN_ENC, W, KS, ME}, none);
    return st;
}

// Workspace geometry (bytes), capacity rounded up to whole workgroup passes so that every fragment store is in bounds
// and unpredicated.  `slots` = depth + 2 (trunk outputs, sigma hidden, sh hidden).
static inline long long mcn16_cap_tiles(long long capacity) { return (capacity + MCN16_ROWS - 1) / MCN16_ROWS * MCN16_WAVES; }
static inline size_t mcn16_slot_bytes(long long capacity, int width) { return (size_t)mcn16_cap_tiles(capacity) * (width / 16) * 1024; }
static inline int mcn16_mask_words(int width) { return width >= 64 ? width / 64 : 1; }        // dwords per lane per slot
static inline size_t mcn16_mask_slot_bytes(long long capacity, int width) { return (size_t)mcn16_cap_tiles(capacity) * 64 * 4 * mcn16_mask_words(width); }
static inline size_t mcn16_enc_bytes(long long capacity) { return (size_t)mcn16_cap_tiles(capacity) * MCN16_ENCKS * 1024; }
static inline size_t mcn16_dsh_bytes(long long capacity) { return (size_t)mcn16_cap_tiles(capacity) * 2 * 1024; }

struct Mcn16FwdArgs {
    McnLayout lay;
    const float* params;        // flat fp32 parameters (biases, sigma.2)
    const void* packed;         // forward stream (mcn16_fwd_stream order), 16-bit
    int stream_slabs;           // total_frags / 16
    int bf16;                   // 0 = f16, 1 = bf16; the split-f16 launchers: 2 = (hi, lo) workspaces, 3 = hi planes only (16-bit layout)
    const float* rays_o; const float* rays_d; const float* zgrid; const float* jitter; const float* barf_w;
    const int2* idx; const int* count; int max_rows; int n_rays, S;
    float* out;                 // [n_rays,S,4]
    // training workspaces (null = inference)
    void* act_ws; size_t slot_bytes;          // [depth+2] slots of fragment-major activations
    void* enc_ws;                             // fragment-major encodings
    unsigned* mask_ws; size_t mask_slot_words;  // [depth+2] slots of lane-local ReLU bit masks
    void* sh_ws;                              // fragment-major sh.2 outputs (the SH coefficients; the backward's view-direction term)
};
hipError_t mcn16_launch_fwd(const Mcn16FwdArgs& a, hipStream_t st);

struct Mcn16BwdArgs {
    McnLayout lay;
    const float* params;
    const void* packed;         // backward stream (mcn16_bwd_stream order)
    int stream_slabs;
    int bf16;
    const float* rays_o; const float* rays_d; const float* zgrid; const float* jitter; const float* barf_w;
    const int2* idx; const int* count; int max_rows; int n_rays, S;
    const float* out; const float* d_out;
    const unsigned* mask_ws; size_t mask_slot_words;
    const void* enc_ws;
    const void* sh_ws;
    void* dy_ws; size_t slot_bytes;           // [depth+2] slots of fragment-major pre-activation gradients (x SG)
    void* dsh_ws;                             // fragment-major d(sh.2 outputs) (columns 0..26) and d sigma_raw (column 27) (x SG)
    float* d_rays_o; float* d_rays_d;
    const unsigned* gmax_bits;
};
hipError_t mcn16_launch_bwd(const Mcn16BwdArgs& a, hipStream_t st);

struct Mcn16DwArgs {
    McnLayout lay;
    int bf16;
    const int* count; int rows;
    const void* act_ws; const void* enc_ws; const void* dy_ws; const void* dsh_ws;
    size_t slot_bytes;
    float* grads;
    const unsigned* gmax_bits;
    float x_scale;              // scale of the X planes (0 / 1: none; MCNX3_SX for the hi planes of the split-f16 chains, dtype 3)
};
hipError_t mcn16_launch_dw(const Mcn16DwArgs& a, hipStream_t st);
hipError_t mcn16_launch_pack(const McnLayout& L, const float* params, void* packed_fwd, void* packed_bwd, int bf16, unsigned* range_flags, hipStream_t st);

#ifdef __HIPCC__
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

// The per-launch gradient scale of the reduced-precision backward / weight-gradient kernels: the power of two that puts
// max|d_out| (a device word written by composite_bwd) near 2^4.  The exponent is clamped so that a tiny but non-zero
// maximum (below ~2^-96) cannot overflow the scale to +inf (1 / SG would be 0 and every dY inf / NaN); the backward and
// the weight-gradient kernel MUST agree on SG, so both call this.
#define MCN16_SG_LOG2 4.f             // max|d_out| * SG lands in (2^(SG_LOG2 - 1), 2^SG_LOG2]
__device__ __forceinline__ float mcn16_grad_scale(float gmax) {
    return (gmax > 0.f && gmax < 3e38f) ? exp2f(fminf(MCN16_SG_LOG2 - ceilf(log2f(gmax)), 100.f)) : 1.f;
}

// channel of contraction position (k-step s, lane half h, element j)
__host__ __device__ __forceinline__ constexpr int mcn16_chan(int s, int h, int j) { return 16 * s + 8 * (j >> 2) + 4 * h + (j & 3); }

template <bool BF> struct Mcn16T;
template <> struct Mcn16T<false> {
    static __device__ __forceinline__ f32x16 mfma(const u32x4_t& a, const u32x4_t& b, const f32x16& c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    }
    // (vector conversion: one v_cvt_pk_f16_f32; two scalar casts become two v_cvt_f16_f32 + v_pack unless something packed follows)
    static __device__ __forceinline__ unsigned pack(float a, float b) {
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){a, b}, f16x2_t));
    }
    static __device__ __forceinline__ unsigned relu_pack(float a, float b) {
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        f16x2_t p = __builtin_convertvector((f32x2_t){a, b}, f16x2_t);
        p = __builtin_elementwise_max(p, (f16x2_t){(_Float16)0.f, (_Float16)0.f});
        return __builtin_bit_cast(unsigned, p);
    }
    static __device__ __forceinline__ float lo(unsigned w) { return (float)__builtin_bit_cast(f16x2_t, w)[0]; }
    static __device__ __forceinline__ float hi(unsigned w) { return (float)__builtin_bit_cast(f16x2_t, w)[1]; }
    static __device__ __forceinline__ unsigned short one(float x) { return __builtin_bit_cast(unsigned short, (_Float16)x); }
};
template <> struct Mcn16T<true> {
    static __device__ __forceinline__ f32x16 mfma(const u32x4_t& a, const u32x4_t& b, const f32x16& c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ unsigned pack(float a, float b) {
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){a, b}, bf16x2_t));
    }
    static __device__ __forceinline__ unsigned relu_pack(float a, float b) {
        // relu on the fp32 bit patterns (negative floats are negative integers): no canonicalising v_max, no packed bf16 max needed
        const int ia = __builtin_bit_cast(int, a), ib = __builtin_bit_cast(int, b);
        return pack(__builtin_bit_cast(float, ia > 0 ? ia : 0), __builtin_bit_cast(float, ib > 0 ? ib : 0));
    }
    static __device__ __forceinline__ float lo(unsigned w) { return __builtin_bit_cast(float, w << 16); }
    static __device__ __forceinline__ float hi(unsigned w) { return __builtin_bit_cast(float, w & 0xffff0000u); }
    static __device__ __forceinline__ unsigned short one(float x) { return __builtin_bit_cast(unsigned short, (__bf16)x); }
};

// per 16-bit half: 1 if the half is non-zero (inputs are >= +0 after the ReLU), else 0
__device__ __forceinline__ unsigned mcn16_nz(unsigned w) {
    unsigned r;
    asm("v_pk_min_u16 %0, %1, 1 op_sel_hi:[1,0]" : "=v"(r) : "v"(w));
    return r;
}
// per 16-bit half: a * b (low 16 bits); with b in {0,1} per half this applies a ReLU mask to a packed pair
__device__ __forceinline__ unsigned mcn16_pkmul(unsigned a, unsigned b) {
    unsigned r;
    asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// workspace store of one fragment / mask vector (non-temporal: written once, read by a later kernel; keeps the packed
// weights in L2).
template <class V>
__device__ __forceinline__ void mcn16_ws_store(const V& v, V* p) {
    __builtin_nontemporal_store(v, p);
}

// ---- the shared weight ring -------------------------------------------------------------------------------------
// Every wave issues the same two 1 KiB pieces per slab; all of a wave's LDS-DMA operations are these pieces, in program
// order, so "my pieces of slab G have landed" == at most 2 * (AHEAD - 1) younger pieces outstanding.  Other vector-memory
// operations of the wave (workspace stores) only make the counted wait more conservative, never less.
// A slab is synchronised EARLY, MCN16_PF fragments before the previous slab ends, so that the A-fragment prefetch
// (MCN16_PF fragments ahead of the MFMAs) runs straight across slab boundaries: with RING = 8 and AHEAD <= 6 the slot
// refilled at that point belongs to a slab every wave has already left.
#define MCN16_PF 2      // (measured 1 / 2 / 3 / 4 / 6 / 8: the backward, which is short of registers, is 2 % faster at 1-2 than at 4; the forwards do not care)
struct Mcn16Ring {
    const char* src;          // packed stream + this lane's byte offset inside a slab (piece PPW * wave, lane * 16)
    unsigned lds_piece;       // LDS byte offset of this wave's first piece inside a slab (wave-uniform)
    unsigned lds_base;        // LDS address of the ring
    int src_slab;             // stream slab that the next issue fetches (wraps at n_slabs)
    int n_slabs;
    unsigned issue_slot;      // ring slot (0..RING-1) of the next issue
    unsigned sync_slot;       // ring slot of the next slab to synchronise
    unsigned next_off;        // LDS byte offset of the slab synchronised last (the one after the slab being consumed)
    // (mcnerf_x3.h: the source address is split into a wave-uniform base -- scalar registers -- and the lane's 32-bit byte offset,
    //  the SADDR form of global_load_lds: no 64-bit per-lane pointer stays live across the layers)
    const char* ubase;        // packed stream + this wave's first piece inside a slab (wave-uniform)
    unsigned voff;            // lane * 16
    const char* piece_base;   // (spread refill) wave-uniform source address of piece 0 of the slab being refilled
};
typedef __attribute__((address_space(3))) void* mcn16_lds_ptr_t;
typedef const __attribute__((address_space(1))) void* mcn16_gbl_ptr_t;

// One 1 KiB LDS-DMA piece (64 lanes x 16 B; LDS destination = wave-uniform lds_dst + lane * 16) as inline asm: with
// the builtin form hipcc (ROCm 7.2) gives up counted lgkmcnt waits in the whole kernel (every LDS read is then waited
// for with lgkmcnt(0), i.e. the A-fragment prefetch stops overlapping) and may drain vmcnt(0) in front of LDS reads
// it cannot disambiguate.  The piece is invisible to the compiler's counters: completion is counted by hand (below).
__device__ __forceinline__ void mcn16_dma16(const char* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// the same with the non-temporal cache policy: for operands that are read exactly once (the weight-gradient streams)
__device__ __forceinline__ void mcn16_dma16_nt(const char* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// ... SADDR form: wave-uniform base (a scalar register pair) + the lane's 32-bit byte offset.  A stream whose pieces differ only in
// a uniform base keeps ONE vector register of addressing state instead of a 64-bit pointer per piece.
__device__ __forceinline__ void mcn16_dma16_nt_s(const char* ubase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(ubase), "s"(lds_dst) : "memory");
}
// 256-byte piece: one dword per lane (LDS destination = lds_dst + lane * 4)
__device__ __forceinline__ void mcn16_dma4(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void mcn16_ring_issue(Mcn16Ring& r, char* ring_lds) {
    const char* s = r.src + (size_t)r.src_slab * (MCN16_SLAB * 1024);
    const unsigned d = r.lds_base + r.issue_slot * (MCN16_SLAB * 1024) + r.lds_piece;
#pragma unroll
    for (int i = 0; i < MCN16_PPW; ++i) mcn16_dma16(s + i * 1024, d + i * 1024);
    r.src_slab = (r.src_slab + 1 == r.n_slabs) ? 0 : r.src_slab + 1;
    r.issue_slot = (r.issue_slot + 1) & (MCN16_RING - 1);
}
// Synchronise the next slab: wait for this wave's pieces of it, rendezvous (every wave's pieces have landed), refill
// the ring AHEAD slabs further on.  r.next_off = the slab's LDS byte offset.
__device__ __forceinline__ void mcn16_ring_sync(Mcn16Ring& r, char* ring_lds) {
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(MCN16_PPW * (MCN16_AHEAD - 1)) : "memory");
    r.next_off = r.sync_slot * (MCN16_SLAB * 1024);
    r.sync_slot = (r.sync_slot + 1) & (MCN16_RING - 1);
    mcn16_ring_issue(r, ring_lds);
}
__device__ __forceinline__ void mcn16_ring_start(Mcn16Ring& r, char* ring_lds, const void* packed, int n_slabs, int wave, int lane) {
    r.src = reinterpret_cast<const char*>(packed) + (MCN16_PPW * wave) * 1024 + lane * 16;
    r.lds_piece = (MCN16_PPW * wave) * 1024;
    r.lds_base = (unsigned)reinterpret_cast<size_t>((mcn16_lds_ptr_t)ring_lds);
    r.src_slab = 0; r.n_slabs = n_slabs; r.issue_slot = 0; r.sync_slot = 0; r.next_off = 0;
#pragma unroll
    for (int i = 0; i < MCN16_AHEAD; ++i) mcn16_ring_issue(r, ring_lds);
    mcn16_ring_sync(r, ring_lds);
}

// One output tile's MFMA chain position inside a layer of F fragments: fragment f is read from the slab being consumed
// (cur) or, once the next slab has been synchronised, from the next one.  The sync for slab q + 1 sits in front of the
// MFMA of local fragment min(16 q + 16 - PF, F - 1).
struct Mcn16Cursor { unsigned cur; };
template <int F>
__device__ __forceinline__ void mcn16_before_mfma(Mcn16Ring& r, char* ring_lds, Mcn16Cursor& c, int f) {
    // (f is a compile-time constant at every call site after unrolling)
    if ((f & (MCN16_SLAB - 1)) == 0) c.cur = r.next_off;
    const int q = f / MCN16_SLAB;
    const int sync_at = (MCN16_SLAB * q + MCN16_SLAB - MCN16_PF) < (F - 1) ? (MCN16_SLAB * q + MCN16_SLAB - MCN16_PF) : (F - 1);
    if (f == sync_at) mcn16_ring_sync(r, ring_lds);
}
// LDS byte offset of fragment f (which may lie in the slab after the one being consumed: allowed once f's slab is synced)
__device__ __forceinline__ unsigned mcn16_frag_off(const Mcn16Ring& r, const Mcn16Cursor& c, int f_now, int f) {
    const bool next = (f / MCN16_SLAB) != (f_now / MCN16_SLAB);
    return (next ? r.next_off : c.cur) + (f & (MCN16_SLAB - 1)) * 1024;
}
#endif
