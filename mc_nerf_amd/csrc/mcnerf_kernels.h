// Internal launcher interface between the C-ABI (api.hip) and the kernel translation units.
#pragma once
#include "mcnerf_common.h"

struct McnMlpFwdArgs {
    McnLayout lay;
    const float* params;      // flat parameter buffer (biases, sigma.2 / sh.2 biases are read from here)
    const float* packed;      // packed weights (mcn_launch_pack)
    const float* rays_o;      // [n_rays,3]
    const float* rays_d;      // [n_rays,3]
    const float* zgrid;       // [S]  linspace(near, far, S)
    const float* jitter;      // [n_rays] or null
    const float* barf_w;      // [10] per-frequency mask
    const int2* idx;          // [rows] (ray, sample) pairs, or null = dense ray-major
    const int* count;         // device row count (with idx), or null
    int max_rows;             // capacity of idx / upper bound on *count
    int n_rays, S;
    float* out;               // [n_rays,S,4] (sigma_raw, r, g, b), written at ray*S + sample
    float* act_save;          // [(depth+2)][act_stride] post-ReLU activations, or null (no-grad path)
    size_t act_stride;        // floats between layers = capacity * width
    float* enc_save;          // [capacity][64]
    float* sh_save;           // [capacity][32]
    unsigned int* mask_save;  // [(depth+2)][capacity][width/32] ReLU masks (bit c of word g = column 32g+c is > 0)
    const float* enc_in = nullptr;   // fp32 kernel only: caller-supplied encodings [rows][63] instead of the fused positional encoding
};
hipError_t mcn_launch_encode(const float* x, const float* barf_w, int n, int n_freqs, float* out, hipStream_t st);
hipError_t mcn_launch_encode_bwd(const float* x, const float* barf_w, int n, int n_freqs, const float* d_out, float* d_x, hipStream_t st);
hipError_t mcn_launch_sample_perm(long long* out, long long n, int batch, const unsigned* seed, hipStream_t st);
#define MCN_LOSS_BLOCKS 128          // (include/mcnerf.h: MCNERF_TRAIN_LOSS_OUT = 4 + MCN_LOSS_BLOCKS floats of `out`)
hipError_t mcn_launch_train_loss(const float* pd, const float* ptg, int np, int H, int W, int normalise, const float* rgb_c, const float* rgb_f,
                                 const float* gt, int nrgb, float* out, float* d_pd, float* d_c, float* d_f, hipStream_t st);
hipError_t mcn_launch_scale3(float* a, int na, float* b, int nb, float* c, int nc, const float* g, hipStream_t st);
hipError_t mcn_launch_upload_f32(float* dst, const float* host_vals, int n, hipStream_t st);
hipError_t mcn_launch_mlp_fwd(const McnMlpFwdArgs& a, hipStream_t st);
int mcn_mlp_tile_rows(int width);

struct McnMlpBwdArgs {
    McnLayout lay;
    const float* params;
    const float* packed;
    const float* rays_o;
    const float* rays_d;
    const float* zgrid;
    const float* jitter;
    const float* barf_w;
    const int2* idx;
    const int* count;
    int max_rows;
    int n_rays, S;
    const float* out;         // forward output [n_rays,S,4]
    const float* d_out;       // upstream gradient [n_rays,S,4]
    const unsigned int* mask_save;   // ReLU masks written by the forward
    size_t act_stride;        // capacity * width (stride between the layers of dy_save; masks use act_stride / 32)
    const float* enc_save;    // [capacity][64] encoded inputs (already BARF-weighted)
    const float* sh_save;
    float* dy_save;           // [(depth+2)][act_stride] pre-activation gradients (dW operands)
    float* dsh_save;          // [capacity][32] gradient of the sh.2 outputs (cols 0..26) and of sigma_raw (col 27)
    float* d_rays_o;          // [n_rays,3] accumulated with atomics (may be null)
    float* d_rays_d;          // [n_rays,3]
    const unsigned int* gmax_bits;   // split-f16 mode only: float bits of max|d_out| over the launch (device scalar)
    float* d_enc_out = nullptr;      // stand-alone CorseFine_NeRF backward: the encoded-input gradient [rows][63] is the result
                                     // (no positional-encoding backward; d_rays_d then receives the SH view-direction term alone)
};
hipError_t mcn_launch_mlp_bwd(const McnMlpBwdArgs& a, hipStream_t st);

struct McnDwArgs {
    McnLayout lay;
    const int* count;         // device row count or null
    int rows;                 // rows when count == null, capacity otherwise
    const float* act_save;
    const float* enc_save;
    const float* dy_save;
    const float* dsh_save;
    size_t act_stride;
    float* grads;             // flat gradient buffer (same layout as params), accumulated with atomics
    bool split16;             // (unused: the split-f16 mode lives in mlp_x3_dw.hip)
    const unsigned int* gmax_bits;   // split-f16 mode: float bits of max|d_out| (gradient scale), device scalar
};
hipError_t mcn_launch_dw(const McnDwArgs& a, hipStream_t st);

hipError_t mcn_launch_pack(const McnLayout& lay, const float* params, float* packed, hipStream_t st);
hipError_t mcn_launch_sync_finish(float* arena, long long n_grad, int n_flags, float world, const float* local, int* asym, hipStream_t st);

struct McnCompositeArgs {
    const float* sig_rgb;     // [N,S,4]
    const float* rays_d;      // [N,3]
    const float* zgrid;       // [S]
    const float* jitter;      // [N] or null
    const float* eps;         // [N,S] noise of the rgb composite
    const float* eps_sel;     // [N,S] noise of the selection weights, or null
    int N, S;
    int white_back;
    float* rgb;               // [N,3]
    float* depth;             // [N] or null
    float* opacity;           // [N] or null
    float* w_sel;             // [N,S] or null
    unsigned int* wmax_bits;  // running max of w_sel as float bits (weights are >= 0), or null
};
hipError_t mcn_launch_composite_fwd(const McnCompositeArgs& a, hipStream_t st);

struct McnCompositeBwdArgs {
    const float* sig_rgb;     // [N,S,4]
    const float* zgrid;
    const float* jitter;
    const float* eps;
    const float* d_rgb;       // [N,3]
    int N, S;
    int white_back;
    float* d_sig_rgb;         // [N,S,4]
    unsigned int* gmax_bits;  // running max of |d_sig_rgb| as float bits (for the split-f16 backward), or null
};
hipError_t mcn_launch_composite_bwd(const McnCompositeBwdArgs& a, hipStream_t st);

struct McnSelectArgs {
    const float* w_sel;       // [N,Sc]
    const unsigned int* wmax_bits;
    float thresh;
    int N, Sc, scale;
    float sigma_default;
    int* ray_counts;          // [N] workspace
    int* ray_offsets;         // [N] workspace
    int2* idx;                // [N*Sc*scale] out
    int* count;               // out (device)
    float* out_f;             // [N,Sc*scale,4] prefilled with (sigma_default,1,1,1), or null
};
hipError_t mcn_launch_select(const McnSelectArgs& a, hipStream_t st);
#define MCN_CAP_BLOCKS 2048     // workgroups (= list chunks) of the cap's passes
#define MCN_CAP_LT 131080       // per-chunk counts: keys below the threshold ...
#define MCN_CAP_EQ (MCN_CAP_LT + MCN_CAP_BLOCKS)      // ... and equal to it
#define MCN_CAP_WS (MCN_CAP_EQ + MCN_CAP_BLOCKS)      // uint32 words of mcn_launch_cap_random's workspace
hipError_t mcn_launch_cap_random(const int2* idx_in, const int* count, int max_rows, int keep, const unsigned* seed, unsigned* ws,
                                 int2* idx_out, int* count_out, hipStream_t st);
hipError_t mcn_launch_cap_gather(const int2* idx_in, const long long* perm, int keep, int2* idx_out, int* count, hipStream_t st);

struct McnRaygenArgs {
    const float* pose;        // [3,4] world->cam
    const float* kinv;        // [3,3]
    const long long* pix;     // [n] pixel ids (v*W+u)
    int n, W;
    float* rays_d;            // [n,3]
    float* rays_o;            // [n,3]
};
hipError_t mcn_launch_raygen_fwd(const McnRaygenArgs& a, hipStream_t st);
struct McnRaygenBwdArgs {
    const float* pose;
    const float* kinv;
    const long long* pix;
    int n, W;
    const float* d_rays_d;    // [n,3]
    const float* d_rays_o;    // [n,3]
    float* d_pose;            // [12] accumulated (atomics)
    float* d_kinv;            // [9]
};
hipError_t mcn_launch_raygen_bwd(const McnRaygenBwdArgs& a, hipStream_t st);

// ---- fused multi-tensor RAdam (optim.hip)
#define MCN_RADAM_MAXT 64
#define MCN_RADAM_CHUNK 4096
struct McnRadamTable {
    int n_tensors;
    int rectified;            // N_sma >= 5
    float lr, beta1, beta2, eps, wd, step_size;
    float* p[MCN_RADAM_MAXT];
    const float* g[MCN_RADAM_MAXT];
    float* m[MCN_RADAM_MAXT];
    float* v[MCN_RADAM_MAXT];
    long long n[MCN_RADAM_MAXT];
    int first_block[MCN_RADAM_MAXT];
};
hipError_t mcn_launch_radam(const McnRadamTable& t, int n_blocks, unsigned* guard, int phase, hipStream_t st);

// ---- fused camera parametrisation (camera.hip)
struct McnCameraArgs {
    const float* wpose;       // [C,6] se(3) of the world->cam poses
    const float* wpose_intr;  // [C,6] se(3) of the calibration poses
    const float* wfx; const float* wfy; const float* wux; const float* wuy;   // [C]
    int C, H, W;
    float* K;                 // [C,3,3]
    float* Kinv;              // [C,3,3]
    float* pose;              // [C,3,4]
    float* calib;             // [C,3,4]
    // calibration reprojection branch (model/mc_nerf.py:147-152, 236-267); any of these may be null
    const float* wpts_intr;   // [C,P,3] world points projected through (K, calib)
    const float* wpts_extr;   // [C,P,3] world points projected through (K, pose)
    int P;
    float* pix_intr;          // [C,P,2]
    float* pix_extr;          // [C,P,2]
};
struct McnCameraGrads {
    const float* dK; const float* dKinv; const float* dpose; const float* dcalib;     // upstream (any may be null)
    const float* dpix_intr; const float* dpix_extr;                                    // upstream of the reprojected pixels [C,P,2] (may be null)
    float* d_wpose; float* d_wpose_intr; float* d_wfx; float* d_wfy; float* d_wux; float* d_wuy;   // written
};
hipError_t mcn_launch_camera_fwd(const McnCameraArgs& a, hipStream_t st);
hipError_t mcn_launch_camera_bwd(const McnCameraArgs& a, const McnCameraGrads& g, hipStream_t st);

// fused reprojection loss (model/loss.py:45-58): mean((pd_x - gt_x)^2) / W^2 + mean((pd_y - gt_y)^2) / H^2 over n points
hipError_t mcn_launch_reproj_loss_fwd(const float* pd, const float* gt, int n, int H, int W, float* loss, hipStream_t st);
hipError_t mcn_launch_reproj_loss_bwd(const float* pd, const float* gt, int n, int H, int W, const float* dloss, float* d_pd, hipStream_t st);

hipError_t mcn_launch_gather_gt(const unsigned char* img, int channels, const long long* pix, int n, float* out, hipStream_t st);
