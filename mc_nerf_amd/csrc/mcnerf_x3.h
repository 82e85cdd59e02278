// Split-f16 ("f16x3") mode of the NeRF MLP kernels on the register-chain / shared-ring / fragment-major architecture of
// mcnerf_16.h: the fp32-grade mode (every fp32 operand x = (hi + lo) / scale with hi = f16(x * scale), lo = f16(x * scale - hi):
// 22 significand bits; a product is hi*hi + hi*lo + lo*hi = three v_mfma_f32_32x32x16_f16 into one fp32 accumulator,
// relative product error ~2^-21 against fp32's 2^-24).  Everything mcnerf_16.h says about the orientation, the
// contraction-index permutation c(s,h,j) and the fragment-major workspaces holds here, with one difference per item:
//
//  * REGISTER-CHAINED LAYERS.  A wave still carries 32 samples through the network in registers, but a layer's input and
//    output are now TWO fragments per k-step (hi, lo): 128 + 128 registers at width 256.  The 256-wide net therefore
//    runs ONE wave per SIMD (4 waves = 128 samples per workgroup pass, 512 registers per lane); narrower nets keep 8 waves.
//  * SHARED WEIGHT STREAM.  The packed stream holds, per logical fragment, the hi piece (1 KiB) followed by the lo piece;
//    a ring slab is 8 logical fragments (16 pieces = 16 KiB, the same slab size and ring as the 16-bit mode), every wave
//    issues 16 / WAVES pieces per slab.  One slab = 24 MFMAs per wave.
//  * FRAGMENT-MAJOR WORKSPACES.  ws[slot][tile of 32 rows][part hi / lo][k-step][lane][8 x f16]: the two planes of a tile
//    are adjacent (2 * KS KiB per tile); the ReLU bit masks are those of the 16-bit mode (bit = hi half non-zero).
//
// Scales (powers of two, so they commute with every rounding): weights SW = 2^8 (lo stays a normal f16 down to
// |w| ~ 5e-4), activations SX = 2^3 (|x| <= 8188), gradients SG = the per-launch power of two of the 16-bit mode.
#pragma once
#include "mcnerf_16.h"

#define MCNX3_SW 256.0f
#define MCNX3_SX 8.0f
#define MCNX3_SLABF 8                  // logical fragments (hi + lo piece) per ring slab
#define MCNX3_PF 2                     // A fragments (pairs) read this many k-steps ahead of their MFMAs

// waves per workgroup by net width (the wave count decides rows per pass and ring pieces per wave): nets at least
// MCNX3_WIDE_MIN wide run one wave per SIMD with 512 registers per lane
#define MCNX3_WIDE_MIN 128
static inline constexpr int mcnx3_waves(int width) { return width >= MCNX3_WIDE_MIN ? 4 : 8; }

static inline int mcnx3_pad(int frags) { return (frags + MCNX3_SLABF - 1) / MCNX3_SLABF * MCNX3_SLABF; }
// The 16-bit streams with the segments padded to whole x3 slabs (first_frag / total_frags count LOGICAL fragments of 2 KiB).
static inline Mcn16Stream mcnx3_repad(Mcn16Stream st) {
    int total = 0;
    for (int s = 0; s < st.nseg; ++s) {
        st.seg[s].first_frag = total;
        total += mcnx3_pad(st.seg[s].tiles * (st.seg[s].a.ksteps + st.seg[s].b.ksteps));
    }
    st.total_frags = total;
    return st;
}
static inline Mcn16Stream mcnx3_fwd_stream(const McnLayout& L) { return mcnx3_repad(mcn16_fwd_stream(L)); }
static inline Mcn16Stream mcnx3_bwd_stream(const McnLayout& L) { return mcnx3_repad(mcn16_bwd_stream(L)); }

// Workspace geometry (bytes): twice the 16-bit planes; the sh.2 outputs are kept as the fp32 accumulator tile (4 KiB per tile).
static inline size_t mcnx3_slot_bytes(long long capacity, int width) { return 2 * mcn16_slot_bytes(capacity, width); }
static inline size_t mcnx3_enc_bytes(long long capacity) { return 2 * mcn16_enc_bytes(capacity); }
static inline size_t mcnx3_sh_bytes(long long capacity) { return (size_t)mcn16_cap_tiles(capacity) * 4096; }
static inline size_t mcnx3_dsh_bytes(long long capacity) { return 2 * mcn16_dsh_bytes(capacity); }

hipError_t mcnx3_launch_fwd(const Mcn16FwdArgs& a, hipStream_t st);
hipError_t mcnx3_launch_bwd(const Mcn16BwdArgs& a, hipStream_t st);
hipError_t mcnx3_launch_dw(const Mcn16DwArgs& a, hipStream_t st);
hipError_t mcnx3_launch_pack(const McnLayout& L, const float* params, void* packed_fwd, void* packed_bwd, unsigned* range_flags, hipStream_t st);

#ifdef __HIPCC__
typedef float f32x2_t __attribute__((ext_vector_type(2)));

// (a, b) * 1 -> packed hi halves and packed lo halves (a in the low 16 bits)
__device__ __forceinline__ void mcnx3_split2(float a, float b, unsigned& hi, unsigned& lo) {
    const f16x2_t h = __builtin_convertvector((f32x2_t){a, b}, f16x2_t);
    const f32x2_t back = __builtin_convertvector(h, f32x2_t);
    const f16x2_t l = __builtin_convertvector((f32x2_t){a - back[0], b - back[1]}, f16x2_t);
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
}
// value of element e (0 / 1) of a (hi, lo) word pair
__device__ __forceinline__ float mcnx3_val(unsigned hi, unsigned lo, int e) {
    const f16x2_t h = __builtin_bit_cast(f16x2_t, hi), l = __builtin_bit_cast(f16x2_t, lo);
    return (float)h[e] + (float)l[e];
}
// v - (float)(half HI ? 1 : 0 of the packed f16 word hw) in ONE instruction: v_fma_mix_f32 computes hw.half * (-1) + v with the
// f16 source converted on the fly; the product is exact, so the single rounding is the subtraction's (bit-identical to
// v_cvt_f32_f16 + v_sub_f32, one vector instruction less per element in the (hi, lo) split of every layer output).
template <int HI>
__device__ __forceinline__ float mcnx3_residual(float v, unsigned hw) {
    float r;
    if (HI) asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hw), "v"(v));
    else asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hw), "v"(v));
    return r;
}
__device__ __forceinline__ float mcnx3_relu(float x) {       // integer max: no canonicalising v_max in front
    const int i = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, i > 0 ? i : 0);
}
__device__ __forceinline__ f32x16 mcnx3_mfma(const u32x4_t& a, const u32x4_t& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}
// acc += (ah + al) (bh + bl) without the lo * lo term; smallest terms first
__device__ __forceinline__ void mcnx3_mfma3(f32x16& acc, const u32x4_t& ah, const u32x4_t& al, const u32x4_t& bh, const u32x4_t& bl) {
    acc = mcnx3_mfma(al, bh, acc);
    acc = mcnx3_mfma(ah, bl, acc);
    acc = mcnx3_mfma(ah, bh, acc);
}

// ---- the shared weight ring with PPW pieces per wave and slab (Mcn16Ring state, mcnerf_16.h)
// One 1 KiB piece, source = wave-uniform base (scalar register pair) + the lane's byte offset (SADDR form), LDS destination through M0.
__device__ __forceinline__ void mcnx3_dma16(const char* ubase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(ubase), "s"(lds_dst) : "memory");
}
template <int PPW>
__device__ __forceinline__ void mcnx3_ring_issue(Mcn16Ring& r) {
    const char* s = r.ubase + (size_t)r.src_slab * (MCN16_SLAB * 1024);
    const unsigned d = r.lds_base + r.issue_slot * (MCN16_SLAB * 1024) + r.lds_piece;
#pragma unroll
    for (int i = 0; i < PPW; ++i) mcnx3_dma16(s + i * 1024, r.voff, d + i * 1024);
    r.src_slab = (r.src_slab + 1 == r.n_slabs) ? 0 : r.src_slab + 1;
    r.issue_slot = (r.issue_slot + 1) & (MCN16_RING - 1);
}
template <int PPW>
__device__ __forceinline__ void mcnx3_ring_sync(Mcn16Ring& r) {
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(PPW * (MCN16_AHEAD - 1)) : "memory");
    r.next_off = r.sync_slot * (MCN16_SLAB * 1024);
    r.sync_slot = (r.sync_slot + 1) & (MCN16_RING - 1);
    mcnx3_ring_issue<PPW>(r);
}
template <int PPW>
__device__ __forceinline__ void mcnx3_ring_start(Mcn16Ring& r, char* ring_lds, const void* packed, int n_slabs, int wave, int lane) {
    r.ubase = reinterpret_cast<const char*>(packed) + (PPW * wave) * 1024;
    r.voff = lane * 16;
    r.src = nullptr;
    r.lds_piece = (PPW * wave) * 1024;
    r.lds_base = (unsigned)reinterpret_cast<size_t>((mcn16_lds_ptr_t)ring_lds);
    r.src_slab = 0; r.n_slabs = n_slabs; r.issue_slot = 0; r.sync_slot = 0; r.next_off = 0;
#pragma unroll
    for (int i = 0; i < MCN16_AHEAD; ++i) mcnx3_ring_issue<PPW>(r);
    mcnx3_ring_sync<PPW>(r);
}
// Fragment f of a layer of F logical fragments: the slab after the one being consumed is synchronised PF fragments before
// the current one ends, so the A prefetch runs across slab boundaries (as mcn16_before_mfma).
template <int F>
__host__ __device__ constexpr int mcnx3_sync_at(int q) {
    return (MCNX3_SLABF * q + MCNX3_SLABF - MCNX3_PF) < (F - 1) ? (MCNX3_SLABF * q + MCNX3_SLABF - MCNX3_PF) : (F - 1);
}
// (immediate form: the refill's pieces are issued back to back at the synchronisation point)
template <int F, int PPW>
__device__ __forceinline__ void mcnx3_before_mfma(Mcn16Ring& r, Mcn16Cursor& c, int f) {
    if ((f & (MCNX3_SLABF - 1)) == 0) c.cur = r.next_off;
    if (f == mcnx3_sync_at<F>(f / MCNX3_SLABF)) mcnx3_ring_sync<PPW>(r);
}
// SPREAD form, used by the layer bodies.  A vector-memory instruction costs the wave that issues it ~60 cycles of issue time
// (MI355X_MICROARCH.md, "LDS-DMA piece issue cost"), of which only the MFMA in flight (32 cycles) is covered when the wave is
// alone on its SIMD; four pieces back to back at every synchronisation point leave the matrix pipe idle for ~200 cycles per
// slab of 768.  So the synchronisation (counted wait + barrier) stays where it was and the refill's pieces follow ONE PER
// SECOND MFMA GAP: piece i of the slab synchronised in front of fragment s goes out in gap 3 s + 1 + 2 i of the layer (gap =
// 3 f + g in front of MFMA g of fragment f), the pieces a layer's last gaps cannot take at its end (mcnx3_layer_end).
#define MCNX3_DMA_STEP 2
// The refill's addresses are set up ONCE per slab at the synchronisation point -- the wave-uniform source address in a scalar
// register pair, the LDS destination in M0 (nothing else in these kernels uses M0; the other LDS-DMA helpers save and restore it) --
// and every piece is then a single instruction: the instruction's immediate offset advances both the global and the LDS address,
// and a slab's pieces are 1 KiB apart in both.  (A per-lane 64-bit source pointer kept across the layers was the one value the
// 512-register saving forward spilled: reloaded from scratch at every slab of the sigma head, each reload a full vmcnt(0) drain.)
template <int PPW>
__device__ __forceinline__ void mcnx3_ring_piece(Mcn16Ring& r, int i) {
    switch (i) {              // (immediate offsets)
        case 0: asm volatile("global_load_lds_dwordx4 %0, %1" ::"v"(r.voff), "s"(r.piece_base) : "memory"); break;
        case 1: asm volatile("global_load_lds_dwordx4 %0, %1 offset:1024" ::"v"(r.voff), "s"(r.piece_base) : "memory"); break;
        case 2: asm volatile("global_load_lds_dwordx4 %0, %1 offset:2048" ::"v"(r.voff), "s"(r.piece_base) : "memory"); break;
        default: asm volatile("global_load_lds_dwordx4 %0, %1 offset:3072" ::"v"(r.voff), "s"(r.piece_base) : "memory"); break;
    }
}
template <int F, int PPW>
__device__ __forceinline__ void mcnx3_before_mfma_spread(Mcn16Ring& r, Mcn16Cursor& c, int f) {
    static_assert(PPW <= 4, "immediate offsets of the refill pieces");
    if ((f & (MCNX3_SLABF - 1)) == 0) c.cur = r.next_off;
    if (f == mcnx3_sync_at<F>(f / MCNX3_SLABF)) {
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(PPW * (MCN16_AHEAD - 1)) : "memory");
        r.next_off = r.sync_slot * (MCN16_SLAB * 1024);
        r.sync_slot = (r.sync_slot + 1) & (MCN16_RING - 1);
        r.piece_base = r.ubase + (size_t)r.src_slab * (MCN16_SLAB * 1024);
        asm volatile("s_mov_b32 m0, %0" ::"s"(r.lds_base + r.issue_slot * (MCN16_SLAB * 1024) + r.lds_piece) : "memory");
        r.src_slab = (r.src_slab + 1 == r.n_slabs) ? 0 : r.src_slab + 1;
        r.issue_slot = (r.issue_slot + 1) & (MCN16_RING - 1);
    }
}
// called in every MFMA gap `gap` (= 3 f + g) of a layer: at most one piece
template <int F, int PPW>
__device__ __forceinline__ void mcnx3_gap_dma(Mcn16Ring& r, int gap) {
    constexpr int NQ = (F + MCNX3_SLABF - 1) / MCNX3_SLABF;
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int i = 0; i < PPW; ++i)
            if (gap == 3 * mcnx3_sync_at<F>(q) + 1 + MCNX3_DMA_STEP * i) mcnx3_ring_piece<PPW>(r, i);
}
// the pieces whose gap lies beyond the layer's last one
template <int F, int PPW>
__device__ __forceinline__ void mcnx3_layer_end(Mcn16Ring& r) {
    constexpr int NQ = (F + MCNX3_SLABF - 1) / MCNX3_SLABF;
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int i = 0; i < PPW; ++i)
            if (3 * mcnx3_sync_at<F>(q) + 1 + MCNX3_DMA_STEP * i >= 3 * F) mcnx3_ring_piece<PPW>(r, i);
}
// LDS byte offset of the hi piece of fragment f (lo piece: + 1024)
__device__ __forceinline__ unsigned mcnx3_frag_off(const Mcn16Ring& r, const Mcn16Cursor& c, int f_now, int f) {
    const bool next = (f / MCNX3_SLABF) != (f_now / MCNX3_SLABF);
    return (next ? r.next_off : c.cur) + (f & (MCNX3_SLABF - 1)) * 2048;
}

// The 63 (+1 pad) encoded channels of one sample, each sin / cos within an fp32 rounding of the true value: one fp64 sin / cos per
// axis, the octaves by the fp64 double-angle step (mcn_sincos_octaves: ~240 instructions per sample instead of 30 fp64 reductions +
// fp32 polynomials = ~1200; round 4 had halved those 30 by splitting them over a sample's two lanes, at the price of 9 shuffles and of
// registers the 256-wide chains did not have -- the recurrence is cheaper than that split on every width and needs neither), times
// the BARF weight of f (model/net_block.py:22-33 order: x, y, z, then per axis sin f = 0..9, cos f = 0..9).
__device__ __forceinline__ void mcnx3_encode_values(const float (&p)[3], const float (&bw)[MCN_NFREQ], float (&E)[64]) {
    E[0] = p[0]; E[1] = p[1]; E[2] = p[2]; E[63] = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float S[MCN_NFREQ], C[MCN_NFREQ];
        mcn_sincos_octaves<MCN_NFREQ>(p[a], S, C);
#pragma unroll
        for (int f = 0; f < MCN_NFREQ; ++f) {
            E[3 + 20 * a + f] = S[f] * bw[f];
            E[3 + 20 * a + 10 + f] = C[f] * bw[f];
        }
    }
}
#endif
