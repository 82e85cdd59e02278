/* mcnerf.h - C ABI of the MI355X-native MC-NeRF volumetric-rendering hot path (libmcnerf.so).
 *
 * The reference (SkylerGao/MC_NeRF) has NO plugin / FFI layer: its hot path is eager PyTorch inside
 * model/mc_nerf.py, model/net_block.py and model/net_utils.py.  The entry points below are therefore
 * this build's own boundary; each one names the reference code it replaces (file:line, relative to
 * the reference checkout).  The Python host classes in mc_nerf_amd/model mirror the reference's
 * MC_Model / NeRF_Model API and call these functions through ctypes.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch allocates inputs, outputs,
 *     gradients and workspaces); the library keeps no global state and frees nothing;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued asynchronously on it, there
 *     are no host synchronisations and no internal streams;
 *   - return value: 0 = OK, negative = error; mcnerf_last_error() returns a thread-local message;
 *   - all floating point is fp32; sample index pairs are int32 (ray, sample); pixel ids int64;
 *   - a net is described by (depth, width, skip): `depth` trunk layers of `width` units with the
 *     encoded input re-concatenated ([x_enc, h]) at layer index `skip` (model/net_block.py:51-59);
 *     widths 32, 64, 128, 256 and depth <= 8 are built.  `skip`: -1 = none, 0 < skip < depth = that
 *     layer, skip >= 256 = the mask form: bits 8 + l set for every such layer l (bit 8 itself, layer 0,
 *     is ignored and may serve as the marker of the form) -- the reference's `skips` is a list
 *     (model/net_block.py:45, 55-58) -- and, when bit 7 is set, the SH degree of the colour head in
 *     bits 4..6 (`MLP_deg` 0 .. 3, model/net_block.py:43, 75-76; 3 (deg + 1)^2 sh.2 outputs; default 2), and in
 *     bits 0..3 the number of encoding frequencies + 1 (`emb_freqs_xyz` 0 .. 10, model/net_block.py:11-18:
 *     3 + 6 F encoded channels; 0 = the default 10).  The exact-fp32 entry points take any mask, degree and
 *     frequency count, the `_16` entry points (f16 / bf16 / f16x3 register chains) at most one skip layer and
 *     degrees 0 .. 2 (any frequency count: their kernels have the geometry of degree 2 and 10 frequencies; a smaller
 *     net's tensors are scattered into it by mcnerf_pack_weights_16 -- zero weights where it has no channel / row --
 *     and its weight gradients gathered back by mcnerf_mlp_dw_16; `barf_w` is always 10 values there);
 *   - parameters of one net live in ONE flat fp32 buffer in the reference's state-dict order
 *     (xyz_encoding_{1..depth}.0.{weight,bias}, sigma.0.*, sigma.2.*, sh.0.*, sh.2.*), Linear weights
 *     [out][in] row-major; gradients use the same layout.
 */
#ifndef MCNERF_H
#define MCNERF_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define MCNERF_ABI_VERSION 6      /* 6: dtype 3 (f16x3h) of the register-chain entry points */

int mcnerf_abi_version(void);
const char* mcnerf_last_error(void);

/* Sizes (in floats) of the flat parameter buffer and of the packed-weight buffer of one net, and the
 * number of samples one workgroup tile covers (workspace capacities need no rounding; informative). */
long long mcnerf_param_count(int depth, int width, int skip);
long long mcnerf_packed_count(int depth, int width, int skip);
int mcnerf_tile_rows(int width);
/* Float offsets of the 2*depth+8 tensors (reference state-dict order) inside the flat parameter
 * buffer; each tensor starts on a 16-byte boundary, so the buffer has a few padding floats. */
int mcnerf_param_offsets(int depth, int width, int skip, long long* offsets);

/* Re-lays the Linear weights into MFMA operand-fragment order (forward + transposed copies).
 * Must be called after every parameter update, before mlp_fwd / mlp_bwd.  No reference
 * counterpart (torch's addmm reads nn.Linear.weight directly, model/net_block.py:69-74). */
int mcnerf_pack_weights(int depth, int width, int skip, const float* params, float* packed, void* stream);

/* Ray generation for the selected pixels of ONE camera.
 * Replaces MC_Model.get_rays + generate_rand_rays (model/mc_nerf.py:124-145, 327-345):
 * pose [3,4] = world->cam [R|t], kinv [3,3], pix [n] = v*W+u  ->  rays_d [n,3] (unit), rays_o [n,3]. */
int mcnerf_raygen_fwd(const float* pose, const float* kinv, const int64_t* pix, int n, int W,
                      float* rays_d, float* rays_o, void* stream);
/* Backward of the above; ACCUMULATES into d_pose [3,4] and d_kinv [3,3] (caller zeroes them). */
int mcnerf_raygen_bwd(const float* pose, const float* kinv, const int64_t* pix, int n, int W,
                      const float* d_rays_d, const float* d_rays_o, float* d_pose, float* d_kinv, void* stream);

/* Fused sample generation + positional encoding + MLP + SH colour for one net.
 * Replaces the gather/encode/MLP/scatter part of NeRF_Model.inference (model/mc_nerf.py:688-701),
 * SinCosEmbedding.forward (model/net_block.py:20-35), CorseFine_NeRF.forward (model/net_block.py:67-78)
 * and eval_sh (model/net_utils.py:103-191).
 *   rays_o, rays_d [n_rays,3]; zgrid [S] = linspace(near,far,S); jitter [n_rays] or NULL;
 *   barf_w [10] per-frequency mask (ones when BARF is off);
 *   idx  NULL  -> dense: every (ray, sample) of the [n_rays,S] grid is evaluated (coarse pass);
 *        !NULL -> [max_rows] int32 (ray, sample) pairs, *count of them valid (fine pass); workgroups
 *                 beyond *count exit, so no host sync is needed to size the launch;
 *   out  [n_rays,S,4] = (sigma_raw, r, g, b) written at (ray, sample); entries not listed in idx are
 *        left untouched (mcnerf_select_fine pre-fills the reference's defaults);
 *   act_save / enc_save / sh_save / mask_save: NULL for the no-grad path; otherwise workspaces of
 *        (depth+2)*capacity*width, capacity*64, capacity*32 floats and (depth+2)*capacity*width/32
 *        uint32 (1-bit ReLU masks) that receive what mcnerf_mlp_bwd / mcnerf_mlp_dw need
 *        (capacity >= number of evaluated samples). */
int mcnerf_mlp_fwd(int depth, int width, int skip, const float* params, const float* packed,
                   const float* rays_o, const float* rays_d, const float* zgrid, const float* jitter,
                   const float* barf_w, const int32_t* idx, const int32_t* count, int max_rows,
                   int n_rays, int S, float* out,
                   float* act_save, long long capacity, float* enc_save, float* sh_save, uint32_t* mask_save,
                   void* stream);

/* The two modules of model/net_block.py as stand-alone forward calls (exact fp32 MFMA; the render path never materialises
 * these tensors):
 *   mcnerf_encode   = SinCosEmbedding.forward (:20-35): x [n,3], barf_w [n_freqs] -> out [n, 3 + 6 n_freqs] (`emb_freqs_xyz`; 10 -> 63);
 *   mcnerf_mlp_apply = CorseFine_NeRF.forward (:67-78): x_enc [n,63], dirs [n,3] -> out [n,4] = (sigma_raw, r, g, b). */
int mcnerf_encode(const float* x, const float* barf_w, int n, int n_freqs, float* out, void* stream);

/* Stream-ordered upload of up to 16 host floats WITHOUT a host-device copy: the values travel as kernel arguments and a
 * one-wave kernel stores them to `dst` (device).  For the per-step host scalars of the train step (the ten BARF weights of
 * model/net_block.py:26-29, evaluated in fp32 on the host exactly as the reference does): a pageable hipMemcpy would block
 * the host until every kernel queued before it has finished, i.e. serialise host and device once per step. */
int mcnerf_upload_f32(float* dst, const float* host_vals, int n, void* stream);
int mcnerf_mlp_apply(int depth, int width, int skip, const float* params, const float* packed, const float* x_enc,
                     const float* dirs, int n, float* out, void* stream);

/* The same two modules DIFFERENTIATED (both are autograd-differentiable in the reference; the host classes wrap these in
 * torch.autograd.Functions, model/net_block.py):
 *   mcnerf_encode_bwd:     d_out [n, 3 + 6 n_freqs] -> d_x [n,3] (written);
 *   mcnerf_mlp_apply_save: mcnerf_mlp_apply that also fills the exact-fp32 workspaces of mcnerf_mlp_fwd (capacity >= n);
 *   mcnerf_mlp_apply_bwd:  d_out [n,4] -> d_x_enc [n,63] (written), d_dirs [n,3] (ACCUMULATED: zero it first; the SH view-direction
 *                          term) and dy_save / dsh_save, from which mcnerf_mlp_dw reduces the parameter gradients as for the
 *                          render path.  `zero` is a device float holding 0 (the sample depth of the degenerate one-sample ray). */
int mcnerf_encode_bwd(const float* x, const float* barf_w, int n, int n_freqs, const float* d_out, float* d_x, void* stream);
int mcnerf_mlp_apply_save(int depth, int width, int skip, const float* params, const float* packed, const float* x_enc,
                          const float* dirs, int n, float* out, float* act_save, long long capacity, float* enc_save,
                          float* sh_save, uint32_t* mask_save, void* stream);
int mcnerf_mlp_apply_bwd(int depth, int width, int skip, const float* params, const float* packed, const float* dirs, const float* zero,
                         int n, const float* out, const float* d_out, const uint32_t* mask_save, long long capacity,
                         const float* enc_save, const float* sh_save, float* dy_save, float* dsh_save,
                         float* d_x_enc, float* d_dirs, void* stream);

/* Backward of mcnerf_mlp_fwd wrt the activations (the dX chain): consumes d_out [n_rays,S,4],
 * writes the pre-activation gradients of every layer to dy_save ((depth+2)*capacity*width floats) and
 * dsh_save (capacity*32: d sh.2 outputs in columns 0..26, d sigma_raw in column 27) and ACCUMULATES
 * d_rays_o / d_rays_d [n_rays,3] (through the sample positions, the encoding and the SH view
 * direction; either may be NULL).  All weight gradients come from mcnerf_mlp_dw.
 * Replaces autograd through model/net_block.py:22-33, 67-78 and model/mc_nerf.py:602, 635, 690-691. */
int mcnerf_mlp_bwd(int depth, int width, int skip, const float* params, const float* packed,
                   const float* rays_o, const float* rays_d, const float* zgrid, const float* jitter,
                   const float* barf_w, const int32_t* idx, const int32_t* count, int max_rows,
                   int n_rays, int S, const float* out, const float* d_out,
                   const uint32_t* mask_save, long long capacity, const float* enc_save, const float* sh_save,
                   float* dy_save, float* dsh_save, float* d_rays_o, float* d_rays_d, void* stream);

/* Weight / bias gradients of one net: dW_l = dY_l^T X_l, db_l = sum_rows dY_l, ACCUMULATED into
 * `grads` (flat, same layout as the parameters; caller zeroes it once per step).
 * `count` NULL -> `rows` rows are valid; otherwise *count (<= rows). */
int mcnerf_mlp_dw(int depth, int width, int skip, const int32_t* count, int rows,
                  const float* act_save, const float* enc_save, const float* dy_save, const float* dsh_save,
                  long long capacity, float* grads, void* stream);

/* ---- Register-chain modes (csrc/mcnerf_16.h, csrc/mcnerf_x3.h), selected by `dtype`:
 *   0 = f16, 1 = bf16: ONE v_mfma_f32_32x32x16_{f16|bf16} per product, fp32 accumulate, biases / epilogues / outputs in
 *       fp32, 2-byte saved operands.  Results agree with the fp32 path to the operand rounding of the dtype (f16: ~1e-5
 *       in rgb on the reference's fixtures), NOT in general to the 1e-4 parity bar.
 *   2 = f16x3: the fp32-GRADE mode.  Every operand is hi + lo (two f16, 22 significand bits), a product is three f16
 *       MFMAs into one fp32 accumulator (~2^-21 relative product error), saved operands are a hi and a lo plane
 *       (4 bytes per value).  This is the mode that meets the 1e-4 parity bar at 16-bit MFMA rates ON EVERY OUTPUT: colours /
 *       depth / opacity (measured 3e-7), ray gradients and every parameter gradient (<= 1e-4 of a tensor's largest entry on
 *       the reference's goldens, measured <= 1.2e-5).
 *   3 = f16x3h: the forward and the backward (dX) chains of dtype 2, instruction for instruction -- colours, selection and ray
 *       gradients are bit-identical to dtype 2's -- but only the HI plane of every saved operand is written (in dtype 0's
 *       workspace layout: 2 bytes per value), and mcnerf_mlp_dw_16 is dtype 0's single-pass f16 kernel on those planes
 *       (activations carry the chains' 2^3 scale): the weight-gradient operands are rounded to 11 significand bits, everything
 *       that reaches a colour or a dX stays at 22.  Packed weights and the sh.2-output workspace (which 4) as dtype 2, the
 *       other workspaces as dtype 0.
 *       GATE of this mode (ONE statement; DESIGN.md 2, README.md, smoke() and the tests say the same): rendered colours /
 *       depth / opacity, the selection list and the ray (camera) gradients meet dtype 2's 1e-4 bar -- they are dtype 2's
 *       bits.  The PARAMETER gradients do NOT: (i) against dtype 2's they are gated at 1e-3 of a tensor's largest entry
 *       (tests/test_model_gpu.py::test_f16x3h_runs_the_f16x3_chains; measured <= 6e-4: unbiased rounding of the dW operands to
 *       11 bits); (ii) against the fp32 oracle, over ALL 40 tensors, at the multiples of the reference's own reorder noise
 *       that dtype 2 is gated at -- worst tensor <= 4 x 1.7e-3, median tensor <= 40 x 2.2e-5 of a tensor's largest entry
 *       (__graft_entry__.PARITY_GATES; measured at 256 rays: 2.5e-3 / 9.8e-5, dtype 2: 2.4e-3 / 3.0e-5, exact fp32: 1.4e-3 / 1.6e-5).
 *       bench.py reports this mode as its headline with dtype 2's line beside it (`f16x3_value`) and every mode's measured
 *       gradient errors (`parity.gradient_parity`).
 * Same reference code replaced as the fp32 entry points above (model/net_block.py:20-35, 67-78;
 * model/net_utils.py:103-191; model/mc_nerf.py:688-701).
 * Packed weights are two fragment STREAMS (forward order, backward = transposed order); workspaces are
 * fragment-major and sized by mcnerf_ws_bytes_16 (capacity is rounded up to whole 256-row passes inside). */
long long mcnerf_packed_bytes_16(int depth, int width, int skip, int dtype, int backward);
/* range_flags (may be NULL): depth + 3 words, one per packed weight tensor in state-dict order (trunk layers 0 .. depth-1,
 * sigma.0, sh.0, sh.2); word t is OR-ed with 1 when tensor t holds a weight outside the mode's operand range (f16: |w| > 65504,
 * f16x3: |w| > 65504 / 2^8 = 255.9, any mode: not finite).  Sticky: zero the words once, read them when a step was refused. */
int mcnerf_pack_weights_16(int depth, int width, int skip, const float* params, void* packed_fwd, void* packed_bwd,
                           int dtype, uint32_t* range_flags, void* stream);
/* which: 0 = activations or pre-activation gradients (all depth+2 slots), 1 = encodings, 2 = ReLU bit masks (all slots),
 *        3 = d(sh.2 outputs / sigma_raw), 4 = sh.2 outputs */
long long mcnerf_ws_bytes_16(int depth, int width, int dtype, long long capacity, int which);
/* mcnerf_mlp_fwd in these modes.  act_ws / enc_ws / mask_ws / sh_ws: NULL (all four) for the no-grad path. */
int mcnerf_mlp_fwd_16(int depth, int width, int skip, int dtype, const float* params, const void* packed_fwd,
                      const float* rays_o, const float* rays_d, const float* zgrid, const float* jitter,
                      const float* barf_w, const int32_t* idx, const int32_t* count, int max_rows,
                      int n_rays, int S, float* out,
                      void* act_ws, long long capacity, void* enc_ws, uint32_t* mask_ws, void* sh_ws, void* stream);
/* mcnerf_mlp_bwd in these modes: dy_ws / dsh_ws receive 16-bit (f16x3: hi + lo) gradients scaled by the power of two derived from
 * *gmax_bits (f16 range); d_rays_o / d_rays_d are accumulated in fp32. */
int mcnerf_mlp_bwd_16(int depth, int width, int skip, int dtype, const float* params, const void* packed_bwd,
                      const float* rays_o, const float* rays_d, const float* zgrid, const float* jitter,
                      const float* barf_w, const int32_t* idx, const int32_t* count, int max_rows,
                      int n_rays, int S, const float* out, const float* d_out,
                      const uint32_t* mask_ws, long long capacity, const void* enc_ws, const void* sh_ws,
                      void* dy_ws, void* dsh_ws, float* d_rays_o, float* d_rays_d, const uint32_t* gmax_bits, void* stream);
/* mcnerf_mlp_dw in these modes (fp32 accumulation and fp32 float-atomic output into `grads`). */
int mcnerf_mlp_dw_16(int depth, int width, int skip, int dtype, const int32_t* count, int rows,
                     const void* act_ws, const void* enc_ws, const void* dy_ws, const void* dsh_ws,
                     long long capacity, float* grads, const uint32_t* gmax_bits, void* stream);

/* Alpha compositing of [N,S] samples per ray.
 * Replaces NeRF_Model.inference's compositing (model/mc_nerf.py:705-727) and sigma2weights
 * (model/mc_nerf.py:729-736); the N(0,1) draws of sigma2weights are explicit inputs.
 *   eps [N,S]: draw of the rgb composite; eps_sel [N,S] or NULL: draw of the selection weights
 *   (model/mc_nerf.py:619 / 662), which are written to w_sel [N,S] and max-reduced into *wmax_bits
 *   (float bits; caller zeroes it);  depth/opacity [N] may be NULL (training). */
int mcnerf_composite_fwd(const float* sig_rgb, const float* rays_d, const float* zgrid, const float* jitter,
                         const float* eps, const float* eps_sel, int N, int S, int white_back,
                         float* rgb, float* depth, float* opacity, float* w_sel, uint32_t* wmax_bits, void* stream);
/* Backward of the rgb composite: d_rgb [N,3] -> d_sig_rgb [N,S,4].  gmax_bits (or NULL): max |d_sig_rgb| of the
 * launch as float bits, max-reduced into a caller-zeroed word (the gradient scale of mcnerf_mlp_bwd_16). */
int mcnerf_composite_bwd(const float* sig_rgb, const float* zgrid, const float* jitter, const float* eps,
                         const float* d_rgb, int N, int S, int white_back, float* d_sig_rgb, uint32_t* gmax_bits,
                         void* stream);

/* Weight-threshold fine-sample selection (model/mc_nerf.py:623-629): every coarse sample with
 * w_sel >= min(thresh, max w_sel) contributes `scale` fine samples (ray, j*scale + r), emitted in
 * torch.nonzero order into idx [N*Sc*scale] (int32 pairs) with the total in *count.  out_f
 * [N,Sc*scale,4] (or NULL) is pre-filled with (sigma_default, 1, 1, 1) (model/mc_nerf.py:692-694).
 * ray_counts / ray_offsets: int32 [N] workspaces. */
int mcnerf_select_fine(const float* w_sel, const uint32_t* wmax_bits, float thresh, int N, int Sc, int scale,
                       float sigma_default, int32_t* ray_counts, int32_t* ray_offsets,
                       int32_t* idx, int32_t* count, float* out_f, void* stream);
/* The random cap of model/mc_nerf.py:630-632: idx_out[i] = idx_in[perm[i]], i < keep; *count = keep. */
int mcnerf_cap_gather(const int32_t* idx_in, const int64_t* perm, int keep, int32_t* idx_out, int32_t* count, void* stream);

/* The same cap with the random subset drawn ON THE DEVICE (no host synchronisation): when *count > keep a uniformly
 * random subset of `keep` entries of idx_in[0 .. *count) is written to idx_out (order unspecified, as the order of the
 * list never reaches a result) and *count_out = keep; otherwise all *count entries are copied.  Entry i is ranked by a
 * 32-bit hash of (*seed, i); *seed is a device word (drawn from torch's device generator by the host classes).
 * ws: mcnerf_cap_ws_words() uint32 of scratch. */
long long mcnerf_cap_ws_words(void);
int mcnerf_cap_random(const int32_t* idx_in, const int32_t* count, int max_rows, int keep, const uint32_t* seed,
                      uint32_t* ws, int32_t* idx_out, int32_t* count_out, void* stream);

/* The pixel subset of a train step, randperm(H * W)[:batch] of model/mc_nerf.py:329, drawn on the device in one kernel:
 * out[i] = P(i), i < batch, with P a pseudo-random permutation of [0, n) keyed by the device word *seed (batch <= n <= 2^31):
 * distinct, uniformly distributed pixel ids in random order. */
int mcnerf_sample_perm(int64_t* out, long long n, int batch, const uint32_t* seed, void* stream);

/* Ground-truth colours of n pixels of ONE uint8 image resident in HBM:
 * image [H*W, channels] (channels 3 = RGB, 4 = RGBA composited on white as data/data_read.py:130-137),
 * pix [n] int64 -> out [n,3] fp32.  Replaces the per-step H2D image copy + gather (model/mc_nerf.py:379, 80). */
int mcnerf_gather_gt(const uint8_t* image, int channels, const int64_t* pix, int n, float* out, void* stream);

/* Fused camera parametrisation of all C cameras.
 * Replaces add_weights2intr / add_weights2pose / add_weights2calib_pose / se3_to_SE3 / taylor_A,B,C /
 * inverse_intrinsic (model/mc_nerf.py:171-210, 269-316) and the calibration reprojection branch get_reproject_pixels /
 * world2cam / cam2pix (model/mc_nerf.py:147-152, 236-267):
 *   K [C,3,3] = [[|W wfx|,0,|W/2 wux|],[0,|W wfy|,|H/2 wuy|],[0,0,1]], Kinv its analytic inverse,
 *   pose / calib [C,3,4] = se3_to_SE3(wpose / wpose_intr) with the reference's 11-term Taylor series,
 *   pix_intr [C,P,2] = pixels of wpts_intr [C,P,3] through (K, calib), pix_extr = wpts_extr through (K, pose)
 *   (each pair may be NULL: branch not evaluated). */
int mcnerf_camera_fwd(const float* wpose, const float* wpose_intr, const float* wfx, const float* wfy, const float* wux,
                      const float* wuy, int C, int H, int W, float* K, float* Kinv, float* pose, float* calib,
                      const float* wpts_intr, const float* wpts_extr, int P, float* pix_intr, float* pix_extr, void* stream);
/* Backward of the above: upstream dK / dKinv [C,3,3], dpose / dcalib [C,3,4], dpix_intr / dpix_extr [C,P,2] (any may be
 * NULL) -> gradients of the six parameter tensors (WRITTEN, not accumulated). */
int mcnerf_camera_bwd(const float* wpose, const float* wpose_intr, const float* wfx, const float* wfy, const float* wux,
                      const float* wuy, int C, int H, int W, const float* dK, const float* dKinv, const float* dpose,
                      const float* dcalib, const float* wpts_intr, const float* wpts_extr, int P, const float* dpix_intr,
                      const float* dpix_extr, float* d_wpose, float* d_wpose_intr, float* d_wfx, float* d_wfy, float* d_wux,
                      float* d_wuy, void* stream);
/* Reprojection loss of n = B*C*P points (model/loss.py:45-58): *loss = mean((pd_x-gt_x)^2)/W^2 + mean((pd_y-gt_y)^2)/H^2;
 * backward: d_pd [n,2] from the device scalar *dloss. */
int mcnerf_reproj_loss_fwd(const float* pd, const float* gt, int n, int H, int W, float* loss, void* stream);
int mcnerf_reproj_loss_bwd(const float* pd, const float* gt, int n, int H, int W, const float* dloss, float* d_pd, void* stream);

/* The whole loss of a NeRF-stage train step, MC_NeRF_Loss.forward with the keys {"intr", "rgb"} (model/loss.py:13-31), value and
 * gradients in one launch:  total = L_intr / (L_intr + 1e-8) [normalise != 0: the GLOBAL_OPTIM / FINE_TUNE rescaling of :20-23, its
 * denominator a detached constant; normalise == 0: L_intr itself] + mean((rgb_c - gt)^2) + mean((rgb_f - gt)^2) [rgb_f may be null:
 * coarse only, :37-41].  pd, pt_gt [np,2] pixels (np may be 0), rgb_*, gt [nrgb] floats (nrgb = 3 N).
 * out[MCNERF_TRAIN_LOSS_OUT]: out[0..2] = {total, L_intr, rgb term}; out[3] is an arrival counter that must be ZERO on entry (it
 * is zero again on exit, so one zero-initialised buffer serves every call) and out[4..] per-block partial sums -- the value is
 * added in block order, i.e. deterministic;  d_pd [np,2], d_c, d_f [nrgb] = d total / d input.
 * mcnerf_scale3: a, b, c (c may be null) *= *g in place -- the saved gradients times the upstream gradient of the total. */
#define MCNERF_TRAIN_LOSS_OUT 132
int mcnerf_train_loss(const float* pd, const float* pt_gt, int np, int H, int W, int normalise,
                      const float* rgb_c, const float* rgb_f, const float* gt, int nrgb,
                      float* out, float* d_pd, float* d_c, float* d_f, void* stream);
int mcnerf_scale3(float* a, int na, float* b, int nb, float* c, int nc, const float* g, void* stream);

/* Fused multi-tensor Rectified-Adam step (one launch for all tensors of a param group).
 * Replaces the per-tensor loop of RAdam.step (model/net_utils.py:38-99).  The arrays of n_tensors device
 * pointers / sizes live on the HOST; step_size and `rectified` (N_sma >= 5) are the host-side scalars of the
 * reference's step-size cache (model/net_utils.py:67-86) for the tensors' common step count.
 * guard (device uint32[2], or NULL): overflow guard of the reduced-precision modes -- guard[0] is raised when a gradient is
 * inf / NaN, an update then leaves parameters and moments untouched, and guard[1] counts refused steps (the caller
 * zero-initialises guard[1] once and reads it whenever it likes; no synchronisation here).
 * phase (bit set) lets ONE optimiser step that spans several calls (param groups, step counts) be refused as a whole:
 *   1 = clear guard[0] first, 2 = check this call's gradients, 4 = update this call's tensors, 8 = this call is the step's
 *   last update: count a refused step in guard[1].  A self-contained step is phase 15; a multi-call step runs (1|2), 2, ...
 *   over all its calls and then 4, ..., (4|8).  Within a call every check precedes every update. */
int mcnerf_radam_step(int n_tensors, float* const* params, const float* const* grads, float* const* exp_avg,
                      float* const* exp_avg_sq, const long long* sizes, float lr, float beta1, float beta2, float eps,
                      float weight_decay, float step_size, int rectified, uint32_t* guard, int phase, void* stream);

/* Finish of the data-parallel step's ONE gradient all-reduce (mc_nerf_amd/distributed.py; replaces the bucketed reducer of
 * DistributedDataParallel, /root/reference main.py:60-62): `arena` = [n_grad summed gradient floats | n_flags summed flags]
 * after the SUM all-reduce.  One launch divides the gradients by `world` (DDP's average) and adds 1 to *asym when some rank's
 * "produced a gradient" flags differ from this rank's `local_flags` (sum != world * local). */
int mcnerf_sync_finish(float* arena, long long n_grad, int n_flags, int world, const float* local_flags, int32_t* asym, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MCNERF_H */
