// C ABI of libmcnerf.so (see include/mcnerf.h).  Thin argument checking + launcher calls; every
// kernel is enqueued on the caller's stream, nothing here synchronises or allocates.
#include "../../include/mcnerf.h"
#include "mcnerf_kernels.h"
#include "mcnerf_16.h"
#include "mcnerf_x3.h"
#define MCN_ACT_STRIDE(capacity, width) ((size_t)(capacity) * (width))
#include <stdio.h>
#include <string.h>

static thread_local char g_err[512] = "";

static int fail(const char* where, const char* what) {
    snprintf(g_err, sizeof(g_err), "%s: %s", where, what);
    return -1;
}
static int check(const char* where, hipError_t e) {
    if (e == hipSuccess) return 0;
    snprintf(g_err, sizeof(g_err), "%s: HIP error %d (%s)", where, (int)e, hipGetErrorString(e));
    return -2;
}
static bool net_ok(int depth, int width, int skip) {
    if (!(depth >= 1 && depth <= MCN_MAXD && (width == 32 || width == 64 || width == 128 || width == 256))) return false;
    if (skip < -1) return false;
    if (skip >= MCN_SKIP_MASK && ((skip >> 8) >> depth) != 0) return false;          // -1, a layer index, or a mask of layers < depth
    return mcn_topo_deg(skip) <= MCN_MAXDEG && mcn_topo_nfreq(skip) <= MCN_NFREQ;
}
// the register-chain families (f16 / bf16 / f16x3) take at most one skip layer and SH degrees 0 .. 2 (their kernels have the geometry
// of degree 2 and 10 frequencies; a smaller net is scattered into it when its weights are packed: mcnerf_common.h)
static bool net16_ok(int depth, int width, int skip) {
    return net_ok(depth, width, skip) && mcn_single_skip(mcn_skip_mask(depth, skip)) != -2 && mcn_topo_deg(skip) <= 2;
}
#define REQ(cond, name) do { if (!(cond)) return fail(name, "invalid argument: " #cond); } while (0)

extern "C" {

int mcnerf_abi_version(void) { return MCNERF_ABI_VERSION; }
const char* mcnerf_last_error(void) { return g_err; }

long long mcnerf_param_count(int depth, int width, int skip) {
    if (!net_ok(depth, width, skip)) return -1;
    return mcn_make_layout(depth, width, skip).n_params;
}
long long mcnerf_packed_count(int depth, int width, int skip) {
    if (!net_ok(depth, width, skip)) return -1;
    return mcn_make_layout(depth, width, skip).n_packed;
}
int mcnerf_tile_rows(int width) { return mcn_mlp_tile_rows(width); }

int mcnerf_param_offsets(int depth, int width, int skip, long long* offsets) {
    REQ(net_ok(depth, width, skip) && offsets, "mcnerf_param_offsets");
    const McnLayout L = mcn_make_layout(depth, width, skip);
    int k = 0;
    for (int i = 0; i < depth; ++i) { offsets[k++] = L.pW[i]; offsets[k++] = L.pB[i]; }
    offsets[k++] = L.pWs1; offsets[k++] = L.pBs1; offsets[k++] = L.pWs2; offsets[k++] = L.pBs2;
    offsets[k++] = L.pWc1; offsets[k++] = L.pBc1; offsets[k++] = L.pWc2; offsets[k++] = L.pBc2;
    return 0;
}

int mcnerf_pack_weights(int depth, int width, int skip, const float* params, float* packed, void* stream) {
    REQ(net_ok(depth, width, skip) && params && packed, "mcnerf_pack_weights");
    return check("mcnerf_pack_weights", mcn_launch_pack(mcn_make_layout(depth, width, skip), params, packed, (hipStream_t)stream));
}


int mcnerf_raygen_fwd(const float* pose, const float* kinv, const int64_t* pix, int n, int W,
                      float* rays_d, float* rays_o, void* stream) {
    REQ(pose && kinv && pix && rays_d && rays_o && n >= 0 && W > 0, "mcnerf_raygen_fwd");
    McnRaygenArgs a = {pose, kinv, (const long long*)pix, n, W, rays_d, rays_o};
    return check("mcnerf_raygen_fwd", mcn_launch_raygen_fwd(a, (hipStream_t)stream));
}
int mcnerf_raygen_bwd(const float* pose, const float* kinv, const int64_t* pix, int n, int W,
                      const float* d_rays_d, const float* d_rays_o, float* d_pose, float* d_kinv, void* stream) {
    REQ(pose && kinv && pix && d_rays_d && d_rays_o && d_pose && d_kinv && n >= 0 && W > 0, "mcnerf_raygen_bwd");
    McnRaygenBwdArgs a = {pose, kinv, (const long long*)pix, n, W, d_rays_d, d_rays_o, d_pose, d_kinv};
    return check("mcnerf_raygen_bwd", mcn_launch_raygen_bwd(a, (hipStream_t)stream));
}

int mcnerf_mlp_fwd(int depth, int width, int skip, const float* params, const float* packed,
                   const float* rays_o, const float* rays_d, const float* zgrid, const float* jitter,
                   const float* barf_w, const int32_t* idx, const int32_t* count, int max_rows,
                   int n_rays, int S, float* out,
                   float* act_save, long long capacity, float* enc_save, float* sh_save, uint32_t* mask_save, void* stream) {
    REQ(net_ok(depth, width, skip), "mcnerf_mlp_fwd");
    REQ(params && packed && rays_o && rays_d && zgrid && barf_w && out && n_rays >= 0 && S > 0, "mcnerf_mlp_fwd");
    REQ((idx == nullptr) == (count == nullptr), "mcnerf_mlp_fwd");
    REQ(!idx || max_rows >= 0, "mcnerf_mlp_fwd");
    REQ((long long)n_rays * S < (1ll << 31), "mcnerf_mlp_fwd");
    if (act_save) {
        REQ(enc_save && sh_save && mask_save, "mcnerf_mlp_fwd");
        REQ(capacity >= (idx ? (long long)max_rows : (long long)n_rays * S), "mcnerf_mlp_fwd");
    }
    McnMlpFwdArgs a;
    a.lay = mcn_make_layout(depth, width, skip);
    a.params = params; a.packed = packed; a.rays_o = rays_o; a.rays_d = rays_d; a.zgrid = zgrid; a.jitter = jitter;
    a.barf_w = barf_w; a.idx = (const int2*)idx; a.count = count; a.max_rows = max_rows; a.n_rays = n_rays; a.S = S;
    a.out = out; a.act_save = act_save; a.act_stride = MCN_ACT_STRIDE(capacity, width); a.enc_save = enc_save; a.sh_save = sh_save; a.mask_save = mask_save;
    return check("mcnerf_mlp_fwd", mcn_launch_mlp_fwd(a, (hipStream_t)stream));
}

int mcnerf_sample_perm(int64_t* out, long long n, int batch, const uint32_t* seed, void* stream) {
    REQ(out && seed && batch >= 0 && n >= batch && n <= (1ll << 31), "mcnerf_sample_perm");
    return check("mcnerf_sample_perm", mcn_launch_sample_perm((long long*)out, n, batch, seed, (hipStream_t)stream));
}

int mcnerf_train_loss(const float* pd, const float* pt_gt, int np, int H, int W, int normalise,
                      const float* rgb_c, const float* rgb_f, const float* gt, int nrgb,
                      float* out, float* d_pd, float* d_c, float* d_f, void* stream) {
    REQ(np >= 0 && (np == 0 || (pd && pt_gt && d_pd)) && rgb_c && gt && nrgb > 0 && out && d_c && (!rgb_f || d_f) && H > 0 && W > 0, "mcnerf_train_loss");
    return check("mcnerf_train_loss", mcn_launch_train_loss(pd, pt_gt, np, H, W, normalise, rgb_c, rgb_f, gt, nrgb, out, d_pd, d_c, d_f, (hipStream_t)stream));
}
int mcnerf_scale3(float* a, int na, float* b, int nb, float* c, int nc, const float* g, void* stream) {
    REQ(g && na >= 0 && nb >= 0 && nc >= 0 && (na == 0 || a) && (nb == 0 || b), "mcnerf_scale3");
    return check("mcnerf_scale3", mcn_launch_scale3(a, na, b, nb, c, c ? nc : 0, g, (hipStream_t)stream));
}

int mcnerf_upload_f32(float* dst, const float* host_vals, int n, void* stream) {
    REQ(dst && host_vals && n >= 0 && n <= 16, "mcnerf_upload_f32");
    return check("mcnerf_upload_f32", mcn_launch_upload_f32(dst, host_vals, n, (hipStream_t)stream));
}

int mcnerf_encode(const float* x, const float* barf_w, int n, int n_freqs, float* out, void* stream) {
    REQ(x && barf_w && out && n >= 0 && n_freqs >= 0 && n_freqs <= 16, "mcnerf_encode");
    return check("mcnerf_encode", mcn_launch_encode(x, barf_w, n, n_freqs, out, (hipStream_t)stream));
}
int mcnerf_mlp_apply(int depth, int width, int skip, const float* params, const float* packed, const float* x_enc,
                     const float* dirs, int n, float* out, void* stream) {
    REQ(net_ok(depth, width, skip) && params && packed && x_enc && dirs && out && n >= 0, "mcnerf_mlp_apply");
    McnMlpFwdArgs a;
    a.lay = mcn_make_layout(depth, width, skip);
    a.params = params; a.packed = packed; a.rays_o = dirs; a.rays_d = dirs; a.zgrid = dirs; a.jitter = nullptr;      // positions are not used:
    a.barf_w = dirs; a.idx = nullptr; a.count = nullptr; a.max_rows = 0; a.n_rays = n; a.S = 1;                       // the encodings come from x_enc
    a.out = out; a.act_save = nullptr; a.act_stride = 0; a.enc_save = nullptr; a.sh_save = nullptr; a.mask_save = nullptr;
    a.enc_in = x_enc;
    return check("mcnerf_mlp_apply", mcn_launch_mlp_fwd(a, (hipStream_t)stream));
}

int mcnerf_encode_bwd(const float* x, const float* barf_w, int n, int n_freqs, const float* d_out, float* d_x, void* stream) {
    REQ(x && barf_w && d_out && d_x && n >= 0 && n_freqs >= 0 && n_freqs <= 16, "mcnerf_encode_bwd");
    return check("mcnerf_encode_bwd", mcn_launch_encode_bwd(x, barf_w, n, n_freqs, d_out, d_x, (hipStream_t)stream));
}
// CorseFine_NeRF.forward with the operands of its backward saved (exact-fp32 workspaces, as mcnerf_mlp_fwd's), and that backward
int mcnerf_mlp_apply_save(int depth, int width, int skip, const float* params, const float* packed, const float* x_enc,
                          const float* dirs, int n, float* out, float* act_save, long long capacity, float* enc_save,
                          float* sh_save, uint32_t* mask_save, void* stream) {
    REQ(net_ok(depth, width, skip) && params && packed && x_enc && dirs && out && n >= 0, "mcnerf_mlp_apply_save");
    REQ(act_save && enc_save && sh_save && mask_save && capacity >= n, "mcnerf_mlp_apply_save");
    McnMlpFwdArgs a;
    a.lay = mcn_make_layout(depth, width, skip);
    a.params = params; a.packed = packed; a.rays_o = dirs; a.rays_d = dirs; a.zgrid = dirs; a.jitter = nullptr;
    a.barf_w = dirs; a.idx = nullptr; a.count = nullptr; a.max_rows = 0; a.n_rays = n; a.S = 1;
    a.out = out; a.act_save = act_save; a.act_stride = MCN_ACT_STRIDE(capacity, width); a.enc_save = enc_save; a.sh_save = sh_save; a.mask_save = mask_save;
    a.enc_in = x_enc;
    return check("mcnerf_mlp_apply_save", mcn_launch_mlp_fwd(a, (hipStream_t)stream));
}
int mcnerf_mlp_apply_bwd(int depth, int width, int skip, const float* params, const float* packed, const float* dirs, const float* zero,
                         int n, const float* out, const float* d_out, const uint32_t* mask_save, long long capacity,
                         const float* enc_save, const float* sh_save, float* dy_save, float* dsh_save,
                         float* d_x_enc, float* d_dirs, void* stream) {
    REQ(net_ok(depth, width, skip) && params && packed && dirs && zero && out && d_out && n >= 0, "mcnerf_mlp_apply_bwd");
    REQ(mask_save && enc_save && sh_save && dy_save && dsh_save && d_x_enc && d_dirs && capacity >= n, "mcnerf_mlp_apply_bwd");
    McnMlpBwdArgs a;
    a.lay = mcn_make_layout(depth, width, skip);
    a.params = params; a.packed = packed; a.rays_o = dirs; a.rays_d = dirs; a.zgrid = zero; a.jitter = nullptr;      // one sample per "ray" at z = 0
    a.barf_w = dirs; a.idx = nullptr; a.count = nullptr; a.max_rows = 0; a.n_rays = n; a.S = 1;
    a.out = out; a.d_out = d_out; a.mask_save = mask_save; a.act_stride = MCN_ACT_STRIDE(capacity, width);
    a.enc_save = enc_save; a.sh_save = sh_save; a.dy_save = dy_save; a.dsh_save = dsh_save;
    a.d_rays_o = nullptr; a.d_rays_d = d_dirs; a.gmax_bits = nullptr; a.d_enc_out = d_x_enc;
    return check("mcnerf_mlp_apply_bwd", mcn_launch_mlp_bwd(a, (hipStream_t)stream));
}


int mcnerf_mlp_bwd(int depth, int width, int skip, const float* params, const float* packed,
                   const float* rays_o, const float* rays_d, const float* zgrid, const float* jitter,
                   const float* barf_w, const int32_t* idx, const int32_t* count, int max_rows,
                   int n_rays, int S, const float* out, const float* d_out,
                   const uint32_t* mask_save, long long capacity, const float* enc_save, const float* sh_save,
                   float* dy_save, float* dsh_save, float* d_rays_o, float* d_rays_d, void* stream) {
    REQ(net_ok(depth, width, skip), "mcnerf_mlp_bwd");
    REQ(params && packed && rays_o && rays_d && zgrid && barf_w && out && d_out && n_rays >= 0 && S > 0, "mcnerf_mlp_bwd");
    REQ(mask_save && enc_save && sh_save && dy_save && dsh_save, "mcnerf_mlp_bwd");
    REQ((idx == nullptr) == (count == nullptr), "mcnerf_mlp_bwd");
    REQ(capacity >= (idx ? (long long)max_rows : (long long)n_rays * S), "mcnerf_mlp_bwd");
    McnMlpBwdArgs a;
    a.lay = mcn_make_layout(depth, width, skip);
    a.params = params; a.packed = packed; a.rays_o = rays_o; a.rays_d = rays_d; a.zgrid = zgrid; a.jitter = jitter;
    a.barf_w = barf_w; a.idx = (const int2*)idx; a.count = count; a.max_rows = max_rows; a.n_rays = n_rays; a.S = S;
    a.out = out; a.d_out = d_out; a.mask_save = mask_save; a.act_stride = MCN_ACT_STRIDE(capacity, width);
    a.enc_save = enc_save; a.sh_save = sh_save; a.dy_save = dy_save; a.dsh_save = dsh_save;
    a.d_rays_o = d_rays_o; a.d_rays_d = d_rays_d; a.gmax_bits = nullptr;
    return check("mcnerf_mlp_bwd", mcn_launch_mlp_bwd(a, (hipStream_t)stream));
}


int mcnerf_mlp_dw(int depth, int width, int skip, const int32_t* count, int rows,
                  const float* act_save, const float* enc_save, const float* dy_save, const float* dsh_save,
                  long long capacity, float* grads, void* stream) {
    REQ(net_ok(depth, width, skip), "mcnerf_mlp_dw");
    REQ(act_save && enc_save && dy_save && dsh_save && grads && rows >= 0 && capacity >= rows, "mcnerf_mlp_dw");
    McnDwArgs a;
    a.lay = mcn_make_layout(depth, width, skip);
    a.count = count; a.rows = rows; a.act_save = act_save; a.enc_save = enc_save; a.dy_save = dy_save;
    a.dsh_save = dsh_save; a.act_stride = MCN_ACT_STRIDE(capacity, width); a.grads = grads; a.split16 = false; a.gmax_bits = nullptr;
    return check("mcnerf_mlp_dw", mcn_launch_dw(a, (hipStream_t)stream));
}


// ---- register-chain modes: single-pass 16-bit (mcnerf_16.h; dtype 0 = f16, 1 = bf16), split-f16 "f16x3" (mcnerf_x3.h; dtype 2), and
//      dtype 3 "f16x3h": the split-f16 forward / backward chains saving only the hi plane of every operand (16-bit workspace layout),
//      the weight gradient by the single-pass f16 kernel on those planes
static bool dtype_ok(int dtype) { return dtype >= 0 && dtype <= 3; }
static bool x3_chain(int dtype) { return dtype == 2 || dtype == 3; }
long long mcnerf_packed_bytes_16(int depth, int width, int skip, int dtype, int backward) {
    if (!net16_ok(depth, width, skip) || !dtype_ok(dtype)) return -1;
    const McnLayout L = mcn_make_layout(depth, width, skip);
    if (x3_chain(dtype)) return (long long)(backward ? mcnx3_bwd_stream(L) : mcnx3_fwd_stream(L)).total_frags * 2048;
    return (long long)(backward ? mcn16_bwd_stream(L) : mcn16_fwd_stream(L)).total_frags * 1024;
}
int mcnerf_pack_weights_16(int depth, int width, int skip, const float* params, void* packed_fwd, void* packed_bwd,
                           int dtype, uint32_t* range_flags, void* stream) {
    REQ(net16_ok(depth, width, skip) && params && packed_fwd && packed_bwd && dtype_ok(dtype), "mcnerf_pack_weights_16");
    if (x3_chain(dtype)) return check("mcnerf_pack_weights_16", mcnx3_launch_pack(mcn_make_layout(depth, width, skip), params, packed_fwd, packed_bwd, range_flags, (hipStream_t)stream));
    return check("mcnerf_pack_weights_16", mcn16_launch_pack(mcn_make_layout(depth, width, skip), params, packed_fwd, packed_bwd, dtype, range_flags, (hipStream_t)stream));
}
static size_t slot_bytes_of(int dtype, long long capacity, int width) { return dtype == 2 ? mcnx3_slot_bytes(capacity, width) : mcn16_slot_bytes(capacity, width); }
long long mcnerf_ws_bytes_16(int depth, int width, int dtype, long long capacity, int which) {
    if (!net_ok(depth, width, 0) || capacity < 0 || !dtype_ok(dtype)) return -1;
    const bool x3 = dtype == 2;
    switch (which) {
        case 0: return (long long)(depth + 2) * (long long)slot_bytes_of(dtype, capacity, width);
        case 1: return (long long)(x3 ? mcnx3_enc_bytes(capacity) : mcn16_enc_bytes(capacity));
        case 2: return (long long)(depth + 2) * (long long)mcn16_mask_slot_bytes(capacity, width);
        case 3: return (long long)(x3 ? mcnx3_dsh_bytes(capacity) : mcn16_dsh_bytes(capacity));
        case 4: return (long long)(x3_chain(dtype) ? mcnx3_sh_bytes(capacity) : mcn16_dsh_bytes(capacity));      // (the chains' fp32 sh.2 tile)
    }
    return -1;
}
int mcnerf_mlp_fwd_16(int depth, int width, int skip, int dtype, const float* params, const void* packed_fwd,
                      const float* rays_o, const float* rays_d, const float* zgrid, const float* jitter,
                      const float* barf_w, const int32_t* idx, const int32_t* count, int max_rows,
                      int n_rays, int S, float* out,
                      void* act_ws, long long capacity, void* enc_ws, uint32_t* mask_ws, void* sh_ws, void* stream) {
    REQ(net16_ok(depth, width, skip) && dtype_ok(dtype), "mcnerf_mlp_fwd_16");
    REQ(params && packed_fwd && rays_o && rays_d && zgrid && barf_w && out && n_rays >= 0 && S > 0, "mcnerf_mlp_fwd_16");
    REQ((idx == nullptr) == (count == nullptr), "mcnerf_mlp_fwd_16");
    REQ(!idx || max_rows >= 0, "mcnerf_mlp_fwd_16");
    REQ((long long)n_rays * S < (1ll << 31), "mcnerf_mlp_fwd_16");
    REQ((act_ws == nullptr) == (enc_ws == nullptr) && (act_ws == nullptr) == (mask_ws == nullptr) && (act_ws == nullptr) == (sh_ws == nullptr), "mcnerf_mlp_fwd_16");
    if (act_ws) REQ(capacity >= (idx ? (long long)max_rows : (long long)n_rays * S), "mcnerf_mlp_fwd_16");
    Mcn16FwdArgs a;
    a.lay = mcn_make_layout(depth, width, skip);
    a.params = params; a.packed = packed_fwd; a.bf16 = dtype;
    a.stream_slabs = x3_chain(dtype) ? mcnx3_fwd_stream(a.lay).total_frags / MCNX3_SLABF : mcn16_fwd_stream(a.lay).total_frags / MCN16_SLAB;
    a.rays_o = rays_o; a.rays_d = rays_d; a.zgrid = zgrid; a.jitter = jitter; a.barf_w = barf_w;
    a.idx = (const int2*)idx; a.count = count; a.max_rows = max_rows; a.n_rays = n_rays; a.S = S; a.out = out;
    a.act_ws = act_ws; a.slot_bytes = slot_bytes_of(dtype, capacity, width); a.enc_ws = enc_ws;
    a.mask_ws = mask_ws; a.mask_slot_words = mcn16_mask_slot_bytes(capacity, width) / 4; a.sh_ws = sh_ws;
    return check("mcnerf_mlp_fwd_16", x3_chain(dtype) ? mcnx3_launch_fwd(a, (hipStream_t)stream) : mcn16_launch_fwd(a, (hipStream_t)stream));
}
int mcnerf_mlp_bwd_16(int depth, int width, int skip, int dtype, const float* params, const void* packed_bwd,
                      const float* rays_o, const float* rays_d, const float* zgrid, const float* jitter,
                      const float* barf_w, const int32_t* idx, const int32_t* count, int max_rows,
                      int n_rays, int S, const float* out, const float* d_out,
                      const uint32_t* mask_ws, long long capacity, const void* enc_ws, const void* sh_ws,
                      void* dy_ws, void* dsh_ws, float* d_rays_o, float* d_rays_d, const uint32_t* gmax_bits, void* stream) {
    REQ(net16_ok(depth, width, skip) && dtype_ok(dtype), "mcnerf_mlp_bwd_16");
    REQ(params && packed_bwd && gmax_bits && rays_o && rays_d && zgrid && barf_w && out && d_out && n_rays >= 0 && S > 0, "mcnerf_mlp_bwd_16");
    REQ(mask_ws && enc_ws && sh_ws && dy_ws && dsh_ws, "mcnerf_mlp_bwd_16");
    REQ((idx == nullptr) == (count == nullptr), "mcnerf_mlp_bwd_16");
    REQ(capacity >= (idx ? (long long)max_rows : (long long)n_rays * S), "mcnerf_mlp_bwd_16");
    Mcn16BwdArgs a;
    a.lay = mcn_make_layout(depth, width, skip);
    a.params = params; a.packed = packed_bwd; a.bf16 = dtype;
    a.stream_slabs = x3_chain(dtype) ? mcnx3_bwd_stream(a.lay).total_frags / MCNX3_SLABF : mcn16_bwd_stream(a.lay).total_frags / MCN16_SLAB;
    a.rays_o = rays_o; a.rays_d = rays_d; a.zgrid = zgrid; a.jitter = jitter; a.barf_w = barf_w;
    a.idx = (const int2*)idx; a.count = count; a.max_rows = max_rows; a.n_rays = n_rays; a.S = S;
    a.out = out; a.d_out = d_out; a.mask_ws = mask_ws; a.mask_slot_words = mcn16_mask_slot_bytes(capacity, width) / 4;
    a.enc_ws = enc_ws; a.sh_ws = sh_ws; a.dy_ws = dy_ws; a.slot_bytes = slot_bytes_of(dtype, capacity, width); a.dsh_ws = dsh_ws;
    a.d_rays_o = d_rays_o; a.d_rays_d = d_rays_d; a.gmax_bits = gmax_bits;
    return check("mcnerf_mlp_bwd_16", x3_chain(dtype) ? mcnx3_launch_bwd(a, (hipStream_t)stream) : mcn16_launch_bwd(a, (hipStream_t)stream));
}
int mcnerf_mlp_dw_16(int depth, int width, int skip, int dtype, const int32_t* count, int rows,
                     const void* act_ws, const void* enc_ws, const void* dy_ws, const void* dsh_ws,
                     long long capacity, float* grads, const uint32_t* gmax_bits, void* stream) {
    REQ(net16_ok(depth, width, skip) && dtype_ok(dtype), "mcnerf_mlp_dw_16");
    REQ(act_ws && enc_ws && dy_ws && dsh_ws && grads && gmax_bits && rows >= 0 && capacity >= rows, "mcnerf_mlp_dw_16");
    Mcn16DwArgs a;
    a.lay = mcn_make_layout(depth, width, skip);
    a.bf16 = dtype == 3 ? 0 : dtype; a.count = count; a.rows = rows; a.act_ws = act_ws; a.enc_ws = enc_ws; a.dy_ws = dy_ws; a.dsh_ws = dsh_ws;
    a.slot_bytes = slot_bytes_of(dtype, capacity, width); a.grads = grads; a.gmax_bits = gmax_bits;
    a.x_scale = dtype == 3 ? MCNX3_SX : 1.f;           // (dtype 3: f16 hi planes of the split-f16 chains, activations x 2^3)
    return check("mcnerf_mlp_dw_16", dtype == 2 ? mcnx3_launch_dw(a, (hipStream_t)stream) : mcn16_launch_dw(a, (hipStream_t)stream));
}

int mcnerf_composite_fwd(const float* sig_rgb, const float* rays_d, const float* zgrid, const float* jitter,
                         const float* eps, const float* eps_sel, int N, int S, int white_back,
                         float* rgb, float* depth, float* opacity, float* w_sel, uint32_t* wmax_bits, void* stream) {
    REQ(sig_rgb && rays_d && zgrid && eps && rgb && N >= 0 && S > 0, "mcnerf_composite_fwd");
    REQ((depth == nullptr) == (opacity == nullptr), "mcnerf_composite_fwd");
    REQ(!eps_sel || w_sel, "mcnerf_composite_fwd");
    McnCompositeArgs a = {sig_rgb, rays_d, zgrid, jitter, eps, eps_sel, N, S, white_back, rgb, depth, opacity, w_sel, wmax_bits};
    return check("mcnerf_composite_fwd", mcn_launch_composite_fwd(a, (hipStream_t)stream));
}
int mcnerf_composite_bwd(const float* sig_rgb, const float* zgrid, const float* jitter, const float* eps,
                         const float* d_rgb, int N, int S, int white_back, float* d_sig_rgb, uint32_t* gmax_bits, void* stream) {
    REQ(sig_rgb && zgrid && eps && d_rgb && d_sig_rgb && N >= 0 && S > 0, "mcnerf_composite_bwd");
    REQ((size_t)4 * 5 * S * sizeof(float) <= 160 * 1024, "mcnerf_composite_bwd");
    McnCompositeBwdArgs a = {sig_rgb, zgrid, jitter, eps, d_rgb, N, S, white_back, d_sig_rgb, gmax_bits};
    return check("mcnerf_composite_bwd", mcn_launch_composite_bwd(a, (hipStream_t)stream));
}

int mcnerf_select_fine(const float* w_sel, const uint32_t* wmax_bits, float thresh, int N, int Sc, int scale,
                       float sigma_default, int32_t* ray_counts, int32_t* ray_offsets,
                       int32_t* idx, int32_t* count, float* out_f, void* stream) {
    REQ(w_sel && wmax_bits && ray_counts && ray_offsets && idx && count && N >= 0 && Sc > 0 && scale > 0, "mcnerf_select_fine");
    REQ((long long)N * Sc * scale < (1ll << 31), "mcnerf_select_fine");
    McnSelectArgs a = {w_sel, wmax_bits, thresh, N, Sc, scale, sigma_default, ray_counts, ray_offsets, (int2*)idx, count, out_f};
    return check("mcnerf_select_fine", mcn_launch_select(a, (hipStream_t)stream));
}
int mcnerf_cap_gather(const int32_t* idx_in, const int64_t* perm, int keep, int32_t* idx_out, int32_t* count, void* stream) {
    REQ(idx_in && perm && idx_out && count && keep >= 0, "mcnerf_cap_gather");
    return check("mcnerf_cap_gather", mcn_launch_cap_gather((const int2*)idx_in, (const long long*)perm, keep, (int2*)idx_out, count, (hipStream_t)stream));
}

long long mcnerf_cap_ws_words(void) { return MCN_CAP_WS; }
int mcnerf_cap_random(const int32_t* idx_in, const int32_t* count, int max_rows, int keep, const uint32_t* seed,
                      uint32_t* ws, int32_t* idx_out, int32_t* count_out, void* stream) {
    REQ(idx_in && count && seed && ws && idx_out && count_out && max_rows >= 0 && keep >= 0, "mcnerf_cap_random");
    return check("mcnerf_cap_random", mcn_launch_cap_random((const int2*)idx_in, count, max_rows, keep, seed, ws, (int2*)idx_out, count_out, (hipStream_t)stream));
}

int mcnerf_gather_gt(const uint8_t* image, int channels, const int64_t* pix, int n, float* out, void* stream) {
    REQ(image && pix && out && n >= 0 && (channels == 3 || channels == 4), "mcnerf_gather_gt");
    return check("mcnerf_gather_gt", mcn_launch_gather_gt(image, channels, (const long long*)pix, n, out, (hipStream_t)stream));
}

int mcnerf_camera_fwd(const float* wpose, const float* wpose_intr, const float* wfx, const float* wfy, const float* wux,
                      const float* wuy, int C, int H, int W, float* K, float* Kinv, float* pose, float* calib,
                      const float* wpts_intr, const float* wpts_extr, int P, float* pix_intr, float* pix_extr, void* stream) {
    REQ(wpose && wpose_intr && wfx && wfy && wux && wuy && K && Kinv && pose && calib && C >= 0 && H > 0 && W > 0, "mcnerf_camera_fwd");
    REQ((wpts_intr == nullptr) == (pix_intr == nullptr) && (wpts_extr == nullptr) == (pix_extr == nullptr) && P >= 0, "mcnerf_camera_fwd");
    McnCameraArgs a = {wpose, wpose_intr, wfx, wfy, wux, wuy, C, H, W, K, Kinv, pose, calib, wpts_intr, wpts_extr, P, pix_intr, pix_extr};
    return check("mcnerf_camera_fwd", mcn_launch_camera_fwd(a, (hipStream_t)stream));
}
int mcnerf_camera_bwd(const float* wpose, const float* wpose_intr, const float* wfx, const float* wfy, const float* wux,
                      const float* wuy, int C, int H, int W, const float* dK, const float* dKinv, const float* dpose,
                      const float* dcalib, const float* wpts_intr, const float* wpts_extr, int P, const float* dpix_intr,
                      const float* dpix_extr, float* d_wpose, float* d_wpose_intr, float* d_wfx, float* d_wfy, float* d_wux,
                      float* d_wuy, void* stream) {
    REQ(wpose && wpose_intr && wfx && wfy && wux && wuy && C >= 0 && H > 0 && W > 0 && P >= 0, "mcnerf_camera_bwd");
    REQ(d_wpose && d_wpose_intr && d_wfx && d_wfy && d_wux && d_wuy, "mcnerf_camera_bwd");
    REQ((!dpix_intr || wpts_intr) && (!dpix_extr || wpts_extr), "mcnerf_camera_bwd");
    McnCameraArgs a = {wpose, wpose_intr, wfx, wfy, wux, wuy, C, H, W, nullptr, nullptr, nullptr, nullptr, wpts_intr, wpts_extr, P, nullptr, nullptr};
    McnCameraGrads g = {dK, dKinv, dpose, dcalib, dpix_intr, dpix_extr, d_wpose, d_wpose_intr, d_wfx, d_wfy, d_wux, d_wuy};
    return check("mcnerf_camera_bwd", mcn_launch_camera_bwd(a, g, (hipStream_t)stream));
}
int mcnerf_reproj_loss_fwd(const float* pd, const float* gt, int n, int H, int W, float* loss, void* stream) {
    REQ(pd && gt && loss && n > 0 && H > 0 && W > 0, "mcnerf_reproj_loss_fwd");
    return check("mcnerf_reproj_loss_fwd", mcn_launch_reproj_loss_fwd(pd, gt, n, H, W, loss, (hipStream_t)stream));
}
int mcnerf_reproj_loss_bwd(const float* pd, const float* gt, int n, int H, int W, const float* dloss, float* d_pd, void* stream) {
    REQ(pd && gt && dloss && d_pd && n > 0 && H > 0 && W > 0, "mcnerf_reproj_loss_bwd");
    return check("mcnerf_reproj_loss_bwd", mcn_launch_reproj_loss_bwd(pd, gt, n, H, W, dloss, d_pd, (hipStream_t)stream));
}

int mcnerf_radam_step(int n_tensors, float* const* params, const float* const* grads, float* const* exp_avg,
                      float* const* exp_avg_sq, const long long* sizes, float lr, float beta1, float beta2, float eps,
                      float weight_decay, float step_size, int rectified, uint32_t* guard, int phase, void* stream) {
    REQ(n_tensors >= 0 && params && grads && exp_avg && exp_avg_sq && sizes && (phase & ~15) == 0, "mcnerf_radam_step");
    for (int i = 0; i < n_tensors; ++i)
        REQ(params[i] && grads[i] && exp_avg[i] && exp_avg_sq[i] && sizes[i] >= 0, "mcnerf_radam_step");
    auto table = [&](int t0, int& blocks) {                           // 64 tensors per launch
        McnRadamTable t;
        t.n_tensors = n_tensors - t0 < MCN_RADAM_MAXT ? n_tensors - t0 : MCN_RADAM_MAXT;
        t.rectified = rectified; t.lr = lr; t.beta1 = beta1; t.beta2 = beta2; t.eps = eps; t.wd = weight_decay;
        t.step_size = step_size;
        blocks = 0;
        for (int i = 0; i < t.n_tensors; ++i) {
            t.p[i] = params[t0 + i]; t.g[i] = grads[t0 + i]; t.m[i] = exp_avg[t0 + i]; t.v[i] = exp_avg_sq[t0 + i];
            t.n[i] = sizes[t0 + i];
            t.first_block[i] = blocks;
            blocks += (int)((sizes[t0 + i] + MCN_RADAM_CHUNK - 1) / MCN_RADAM_CHUNK);
        }
        return t;
    };
    McnRadamTable none;
    none.n_tensors = 0;
    if (guard && (phase & 1)) {
        const int rc = check("mcnerf_radam_step", mcn_launch_radam(none, 0, guard, 1, (hipStream_t)stream));
        if (rc) return rc;
    }
    // every check of the call precedes every update: a non-finite gradient in a later chunk of 64 tensors must not find the
    // earlier chunks already updated
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 0 ? !(guard && (phase & 2)) : !(phase & 4)) continue;
        for (int t0 = 0; t0 < n_tensors; t0 += MCN_RADAM_MAXT) {
            int blocks;
            const McnRadamTable t = table(t0, blocks);
            const bool last = t0 + MCN_RADAM_MAXT >= n_tensors;
            const int ph = pass == 0 ? 2 : (4 | (last ? (phase & 8) : 0));
            const int rc = check("mcnerf_radam_step", mcn_launch_radam(t, blocks, guard, ph, (hipStream_t)stream));
            if (rc) return rc;
        }
    }
    return 0;
}

int mcnerf_sync_finish(float* arena, long long n_grad, int n_flags, int world, const float* local_flags, int32_t* asym, void* stream) {
    REQ(arena && local_flags && asym && n_grad >= 0 && n_flags >= 0 && world >= 1, "mcnerf_sync_finish");
    return check("mcnerf_sync_finish", mcn_launch_sync_finish(arena, n_grad, n_flags, (float)world, local_flags, (int*)asym, (hipStream_t)stream));
}

}  // extern "C"
