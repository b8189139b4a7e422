/*
 * nerf_amd.h -- C ABI of libnerf_amd.so: the MI355X (gfx950) NeRF volume-rendering hot path.
 *
 * This is the drop-in boundary.  The reference (1501454486/NeRF) is pure Python over
 * PyTorch, so there is no reference FFI to mirror; each entry point below replaces the
 * reference function it cites (paths relative to the reference root) and is what a
 * ctypes / cffi binding on the reference side binds.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous fp32 data unless the name ends in
 *     `_host`;  sizes are element counts;  `stream` is a hipStream_t (NULL = default stream);
 *   - entry points enqueue work on `stream` and return; scratch is passed in by the caller
 *     (nrf_*_workspace_bytes), and no entry point allocates, frees, synchronises or keeps state
 *     between calls -- with exactly THREE documented exceptions:
 *       1. nrf_kilo_train_forward_backward: on its first call per device it allocates 64 bytes of pinned
 *          host memory + one event (kept until the process ends), and every call WAITS on the host
 *          for the batch's sample count (a spin on that pinned word) before it enqueues the rest;
 *          calls on one device share the word, so calls that overlap on one device race;
 *       2. nrf_packed_fp16x2_status: copies one status word to the host and waits for `stream`;
 *       3. nrf_render_static: its workspace is a device array inside the library (per-device
 *          address resolved on first use), shared by all calls on that device;
 *   - return value: 0 on success, a negative nrf_status otherwise (nrf_strerror gives text);
 *     shapes are validated on the host before any launch;
 *   - thread-safe for distinct streams / buffers, apart from the three exceptions above.
 */
#ifndef NERF_AMD_H
#define NERF_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* nrf_stream_t; /* hipStream_t */

typedef enum nrf_status {
    NRF_OK = 0,
    NRF_ERR_NULL = -1,       /* a required pointer is NULL */
    NRF_ERR_SHAPE = -2,      /* a size is outside what the kernels support */
    NRF_ERR_LAUNCH = -3,     /* hipGetLastError() reported a launch failure */
    NRF_ERR_WORKSPACE = -4,  /* workspace too small */
    NRF_ERR_ALIGN = -5,      /* a pointer is not 16-byte aligned */
    NRF_ERR_STATIC_LIMIT = -6, /* nrf_render_static: more rays / samples than the static scratch holds */
    NRF_ERR_RUNTIME = -7,    /* a HIP runtime query failed (hipGetDevice, hipGetSymbolAddress) */
    NRF_ERR_RANGE = -8       /* nrf_packed_fp16x2_status: a weight lies outside fp16's range */
} nrf_status;

/* Layer order used by every `w[12]` / `b[12]` argument: the reference's state-dict layout
 * (src/models/nerf/network.py:22-47): 0..7 pts_linears.{0..7}, 8 feature_linear,
 * 9 alpha_linear, 10 views_linears.0, 11 rgb_linear; weights are nn.Linear (out, in)
 * row-major fp32, biases (out,). */
enum { NRF_NUM_LAYERS = 12 };

int nrf_version(void);
const char* nrf_strerror(int status);

/* ---- weights ------------------------------------------------------------------------ */

/* Number of floats of one model's weights in the MFMA streaming layout (incl. padding). */
int64_t nrf_packed_floats(void);

/* Permute one 8x256 NeRF MLP (network.py:9-47: 63->256, 4x 256->256, 319->256, 2x 256->256,
 * feature 256->256, alpha 256->1, views 283->128, rgb 128->3) from the state-dict layout
 * into the layout nrf_mlp_forward_* stream: per layer, per 32-row output block, per 32-wide
 * K block, 1 KiB pieces ordered exactly as one wavefront's v_mfma_f32_32x32x2_f32 A-operands
 * are consumed.  `w_host` / `b_host` are HOST arrays of 12 DEVICE pointers. */
int nrf_pack_weights(const float* const* w_host, const float* const* b_host, float* packed,
                     nrf_stream_t stream);

/* ---- ray generation: src/datasets/nerf/blender.py:76-101 ---------------------------- */

/* Rays [first_ray, first_ray + n_rays) of the row-major H x W pixel grid (or, when
 * `pixel_index` is non-NULL, of the listed int64 flat pixel indices = the random-pixel draw
 * of blender.py:82-85).  c2w_host: 12 doubles, rows of the 3x4 camera-to-world matrix.
 * Math in fp64, results cast to fp32 like blender.py:99-100.  xyz, viewdirs: (n_rays, 3). */
int nrf_raygen(const double* c2w_host, int H, int W, double focal, int64_t first_ray, int64_t n_rays,
               const int64_t* pixel_index, float* xyz, float* viewdirs, nrf_stream_t stream);

/* ---- stratified sampling: volume_renderer.py:209-220, 336-349 ----------------------- */

/* add_perturbation: z_out[r,i] = lower + (upper - lower) * u[r,i] with mid-point bins of
 * z_in.  z_in has `z_stride` floats between rays (0 = one shared table).  n <= 1024. */
int nrf_perturb_samples(const float* z_in, int64_t z_stride, const float* u, int64_t n_rays, int n,
                        float* z_out, nrf_stream_t stream);

/* ---- the 8x256 MLP with fused point construction + positional encoding -------------- */

/* volume_renderer.py:227-231 + _run_network :393-423 + Network.forward network.py:171-195 +
 * freq.py:31-32 + NeRF.forward network.py:49-74, in one kernel.  For ray r, sample s:
 * pts = rays_o[r] + z[r*z_stride + s] * (rays_d[r] / |rays_d[r]|); raw[r,s,:] = [r,g,b,sigma]
 * (no activation).  n_samples must be a multiple of 32.  raw: (n_rays, n_samples, 4). */
int nrf_mlp_forward_rays(const float* packed, const float* rays_o, const float* rays_d, const float* z,
                         int64_t z_stride, int64_t n_rays, int n_samples, float* raw, nrf_stream_t stream);

/* Network.forward on a flat batch (network.py:184-186): pts (n,3), dirs (n,3) used as given
 * (not normalised).  raw: (n, 4). */
int nrf_mlp_forward_points(const float* packed, const float* pts, const float* dirs, int64_t n_points,
                           float* raw, nrf_stream_t stream);

/* ---- bf16 MLP variant (BASELINE configs[4]) ------------------------------------------------ */

/* Same network and reference path as nrf_pack_weights / nrf_mlp_forward_rays (network.py:49-74, 171-195,
 * volume_renderer.py:227-231, 393-423), evaluated on v_mfma_f32_32x32x16_bf16: matrix weights and every layer
 * input (encodings, relu'd activations, feature vector) rounded to bf16 (nearest even), fp32 accumulation, fp32
 * biases, fp32 alpha / rgb heads.  n_samples must be a multiple of 64.  Stated tolerance against the fp32
 * reference: see tests/test_gpu_bf16.py. */
int64_t nrf_packed_bf16_bytes(void);
int nrf_pack_weights_bf16(const float* const* w, const float* const* b, void* packed_bf16, nrf_stream_t stream);
int nrf_mlp_forward_rays_bf16(const void* packed_bf16, const float* rays_o, const float* rays_d, const float* z,
                              int64_t z_stride, int64_t n_rays, int n_samples, float* raw, nrf_stream_t stream);
/* Network.forward on a flat batch (network.py:184-186) with the bf16 MLP: pts (n, 3), dirs (n, 3) used as given
 * (not normalised), any n.  raw: (n, 4). */
int nrf_mlp_forward_points_bf16(const void* packed_bf16, const float* pts, const float* dirs, int64_t n_points,
                                float* raw, nrf_stream_t stream);

/* ---- fp32-class MLP on the bf16 matrix cores ("bf16x3") ------------------------------------------------------------ */

/* Same network and reference path again (network.py:49-74, 171-195, volume_renderer.py:227-231, 393-423).  Every fp32
 * weight and every layer input is split into three bf16 terms (a = a0 + a1 + a2, 24 significand bits) and each product
 * is the six bf16 MFMAs with i + j <= 2 (v_mfma_f32_16x16x32_bf16 in the forward kernels), fp32 accumulation;
 * the dropped terms are 2^-24 of the product (fp32's own rounding unit).  fp32 biases and fp32 alpha / rgb heads.  Results
 * are fp32-class (as close to a float64 evaluation as the fp32 reference is: tests/test_gpu_bf16x3.py), at 6 / 16 of the
 * fp32 MFMA time on paper.  n_samples must be a multiple of 32.  The packed stream is private to the library (its layout
 * follows the kernels' MFMA shape): size it with nrf_packed_bf16x3_bytes(), fill it with nrf_pack_weights_bf16x3. */
int64_t nrf_packed_bf16x3_bytes(void);
int nrf_pack_weights_bf16x3(const float* const* w, const float* const* b, void* packed_x3, nrf_stream_t stream);
int nrf_mlp_forward_rays_bf16x3(const void* packed_x3, const float* rays_o, const float* rays_d, const float* z,
                                int64_t z_stride, int64_t n_rays, int n_samples, float* raw, nrf_stream_t stream);
int nrf_mlp_forward_points_bf16x3(const void* packed_x3, const float* pts, const float* dirs, int64_t n_points,
                                  float* raw, nrf_stream_t stream);
/* nrf_mlp_forward_rays_train on the three-term bf16 kernel: same acts / masks layout and sizes, for the fp32
 * backward entry points (nrf_mlp_backward). */
int nrf_mlp_forward_rays_train_bf16x3(const void* packed, const float* rays_o, const float* rays_d, const float* z,
                                      int64_t z_stride, int64_t n_rays, int n_samples, float* raw, float* acts,
                                      uint32_t* masks, nrf_stream_t stream);

/* ---- fp32-class MLP on the fp16 matrix cores ("fp16x2") ------------------------------------------------------------- */

/* Same network and reference path (network.py:49-74, 171-195, volume_renderer.py:227-231, 393-423).  Every fp32 weight
 * and every layer input is split into two fp16 terms (a ~ a_hi + a_lo / 2048, both rounded to nearest even: 22
 * significand bits) and each product is three v_mfma_f32_32x32x16_f16 (hi.hi, and hi.lo + lo.hi through a second
 * accumulator at scale 2^11), fp32 accumulation; fp32 biases and fp32 alpha / rgb heads.  Operand error 2^-22 per
 * product, below what an fp32 accumulation of a 256-term sum loses: results stay as close to a float64 evaluation as
 * the fp32 reference is (tests/test_gpu_fp16x2.py), at 3 / 16 of the fp32 MFMA time on paper.  Operands must lie in
 * fp16's range (|a| < 65520, weights and activations alike): beyond it the output is non-finite.
 * n_samples must be a multiple of 32. */
int64_t nrf_packed_fp16x2_bytes(void);
int nrf_pack_weights_fp16x2(const float* const* w, const float* const* b, void* packed_h2, nrf_stream_t stream);
/* Did every weight fit?  NRF_OK, or NRF_ERR_RANGE if the packer met |w| >= 65520 or a non-finite weight (the forward
 * entry points then return NaN for every point; so does a point whose activations leave the range).  Waits for the
 * stream: the packer itself stays asynchronous. */
int nrf_packed_fp16x2_status(const void* packed_h2, nrf_stream_t stream);
int nrf_mlp_forward_rays_fp16x2(const void* packed_h2, const float* rays_o, const float* rays_d, const float* z,
                                int64_t z_stride, int64_t n_rays, int n_samples, float* raw, nrf_stream_t stream);
int nrf_mlp_forward_points_fp16x2(const void* packed_h2, const float* pts, const float* dirs, int64_t n_points,
                                  float* raw, nrf_stream_t stream);
/* nrf_mlp_forward_rays_train on the two-term fp16 kernel: same acts / masks layout and sizes (the fp32 value each operand
 * split starts from is what is saved), for nrf_mlp_backward / nrf_mlp_backward_bf16x3. */
int nrf_mlp_forward_rays_train_fp16x2(const void* packed_h2, const float* rays_o, const float* rays_d, const float* z,
                                      int64_t z_stride, int64_t n_rays, int n_samples, float* raw, float* acts,
                                      uint32_t* masks, nrf_stream_t stream);

/* nrf_mlp_backward with the data-gradient chain (d h_{l-1} = W_l^T dY_l) on the three-term bf16 kernel; the weight-gradient
 * GEMMs are the same fp32 kernels.  packed: nrf_pack_weights (head rows); packed_t_x3: nrf_pack_weights_transposed_bf16x3
 * (nrf_packed_bwd_bf16x3_bytes() bytes).  Same workspace, acts / masks and gradient layout as nrf_mlp_backward. */
int64_t nrf_packed_bwd_bf16x3_bytes(void);
int nrf_pack_weights_transposed_bf16x3(const float* const* weights_host, void* packed_t_x3, nrf_stream_t stream);
int nrf_mlp_backward_bf16x3(const float* packed, const void* packed_t_x3, const float* grad_raw, const float* acts,
                            const uint32_t* masks, int64_t n_rays, int n_samples, void* workspace, int64_t workspace_bytes,
                            float* grad_flat, nrf_stream_t stream);

/* nrf_mlp_backward with the data-gradient chain and the wide weight-gradient GEMMs on the two-term fp16 kernels (the
 * GEMMs scale their dY operand per row set from maxima the chain leaves in the workspace).  Gradients are far below fp16's normal range, so the chain runs at a power-of-two loss scale chosen on the
 * device from max |grad_raw| of the call and every dY row is unscaled (exactly) before it is stored: callers see
 * ordinary fp32 gradients.  packed: nrf_pack_weights (head rows); packed_t_h2: nrf_pack_weights_transposed_fp16x2
 * (nrf_packed_bwd_fp16x2_bytes() bytes).  Same workspace, acts / masks and gradient layout as nrf_mlp_backward. */
int64_t nrf_packed_bwd_fp16x2_bytes(void);
int nrf_pack_weights_transposed_fp16x2(const float* const* weights_host, void* packed_t_h2, nrf_stream_t stream);
int nrf_mlp_backward_fp16x2(const float* packed, const void* packed_t_h2, const float* grad_raw, const float* acts,
                            const uint32_t* masks, int64_t n_rays, int n_samples, void* workspace, int64_t workspace_bytes,
                            float* grad_flat, nrf_stream_t stream);

/* ---- alpha compositing: volume_renderer.py:222-224, 236-265 / 291-334 --------------- */

/* sigma = relu(raw[...,3]); rgb = sigmoid(raw[...,:3]); delta = diff(z), last = 1e10;
 * alpha = 1 - exp(-sigma*delta); T = exclusive cumprod(1 - alpha); w = T*alpha;
 * rgb_map = sum w*rgb (+ 1 - acc if white_bkgd); depth = sum w*z; acc = sum w.
 * One wavefront per ray, 64 samples per step, wave-wide product scan.
 * n_samples multiple of 64, <= 1024.  weights (n_rays, n_samples) may be NULL. */
int nrf_composite(const float* raw, const float* z, int64_t z_stride, int64_t n_rays, int n_samples,
                  int white_bkgd, float* rgb_map, float* depth_map, float* acc_map, float* weights,
                  nrf_stream_t stream);

/* ---- importance resampling + merge: volume_renderer.py:363-391, 283-289 ------------- */

/* pdf = w/(sum(w)+1e-10); cdf = [0, cumsum(pdf)]; idx = clamp(searchsorted(cdf, u, right) - 1,
 * 0, 63); fine = bins[idx] + (bins[min(idx+1,63)] - bins[idx]) * u2; optional
 * add_perturbation with u_fine; z_out = sort(cat(z_coarse, fine)).  Bit-exact with the
 * reference on torch-CPU (8x4-lane fp32 sum, double running cumsum, no FMA contraction).
 * n_coarse must be 64; n_importance a multiple of 64 with n_coarse + n_importance <= 256.
 * bins / z_coarse: stride 0 = one shared 64-entry table.  fine_out (n_rays, n_importance)
 * and idx_out (int32) may be NULL. */
int nrf_sample_fine(const float* weights, const float* bins, int64_t bins_stride, const float* z_coarse,
                    int64_t zc_stride, const float* u, const float* u2, const float* u_fine, int64_t n_rays,
                    int n_coarse, int n_importance, float* z_out, float* fine_out, int32_t* idx_out,
                    nrf_stream_t stream);

/* ---- training step: backward kernels (what autograd does for loss.backward(), -------- */
/* ---- src/train/trainers/trainer.py:73, through the reference's torch ops) ----------- */

/* Number of floats of the transposed weight stream the backward chain reads. */
int64_t nrf_packed_bwd_floats(void);
/* W^T of views(:, :256), feature, pts7..pts1 in MFMA streaming order (see nrf_pack_weights). */
int nrf_pack_weights_transposed(const float* const* w_host, float* packed_t, nrf_stream_t stream);

/* Saved tensors of the training forward, sized for n_rays * n_samples points (whole 4-tile groups). */
int64_t nrf_mlp_train_tiles(int64_t n_rays, int n_samples);
int64_t nrf_mlp_acts_bytes(int64_t n_rays, int n_samples);
int64_t nrf_mlp_masks_bytes(int64_t n_rays, int n_samples);
/* nrf_mlp_forward_rays that also saves every layer's activated input (acts) and relu sign bits
 * (masks) for nrf_mlp_backward. */
int nrf_mlp_forward_rays_train(const float* packed, const float* rays_o, const float* rays_d, const float* z,
                               int64_t z_stride, int64_t n_rays, int n_samples, float* raw, float* acts,
                               uint32_t* masks, nrf_stream_t stream);

/* d(raw) from d(rgb_map) (n_rays,3), d(depth_map) (n_rays,), d(acc_map) (n_rays,) -- any may be NULL:
 * the backward of nrf_composite (volume_renderer.py:244-263 under autograd).  n_samples <= 512. */
int nrf_composite_backward(const float* raw, const float* z, int64_t z_stride, int64_t n_rays, int n_samples,
                           int white_bkgd, const float* grad_rgb, const float* grad_depth, const float* grad_acc,
                           float* grad_raw, nrf_stream_t stream);

/* Parameters of one model: 595,844.  grad_flat of nrf_mlp_backward has this many floats, laid out as the
 * reference state dict: per layer (include order above) weight (out, in) row-major, then bias. */
int64_t nrf_model_param_floats(void);
int64_t nrf_mlp_backward_workspace_bytes(int64_t n_rays, int n_samples);
/* Parameter gradients of one 8x256 MLP (network.py:49-74) from grad_raw (n_rays, n_samples, 4):
 * dgrad chain on the matrix cores + per-layer wgrad GEMMs + bias sums; overwrites grad_flat. */
int nrf_mlp_backward(const float* packed, const float* packed_t, const float* grad_raw, const float* acts,
                     const uint32_t* masks, int64_t n_rays, int n_samples, void* workspace, int64_t workspace_bytes,
                     float* grad_flat, nrf_stream_t stream);

/* grad[i] = 2 (x[i] - target[i]) / n: gradient of nn.MSELoss(mean) (src/train/trainers/nerf.py:25-27);
 * if loss_accum is non-NULL, adds mean((x - target)^2) to *loss_accum (zero it first).  Up to 65536 elements (training
 * batches) the sum is formed in a fixed order: the same bits on every run; beyond, one float atomic per wavefront. */
int nrf_mse_grad(const float* x, const float* target, int64_t n, float* grad, float* loss_accum, nrf_stream_t stream);

/* clip_grad_value_(clip_value) then one torch.optim.Adam step (trainer.py:74-75, optimizer.py:9-29) over a
 * flat buffer of n parameters; grads are first scaled by grad_scale (1/world after a sum all-reduce).
 * `step` counts from 1. */
int nrf_clip_adam(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float grad_scale,
                  float clip_value, float lr, float beta1, float beta2, float eps, int step, nrf_stream_t stream);

/* The same with an L2 regulariser on the tail [reg_begin, n) folded into the pass, for the KiloNeRF trainer whose
 * colour-regularised tensors (kilonerf/network.py:357-384, trainers/kilonerf.py:52-55) are the tail of its flat buffer:
 * before scale / clip / Adam, grads[i] += 2 reg_grad_coeff params[i] (written back, as nrf_l2_reg leaves it) and, if
 * loss_accum is non-NULL, *loss_accum += reg_loss_coeff * sum(params[i]^2) over the tail (the parameters before the
 * update).  With several ranks reg_grad_coeff = l2_reg * world (grad_scale divides it again), reg_loss_coeff = l2_reg.
 * All four buffers 16-byte aligned. */
int nrf_clip_adam_l2(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float grad_scale,
                     float clip_value, float lr, float beta1, float beta2, float eps, int step, int64_t reg_begin,
                     float reg_grad_coeff, float reg_loss_coeff, float* loss_accum, nrf_stream_t stream);

/* ---- KiloNeRF: src/models/kilonerf/network.py, src/models/kilonerf/renderer/volumn_renderer.py ---- */

/* Per-cell MFMA streaming layout of the 12 batched tensors of BatchedKiloNeRF (network.py:87-113),
 * `tensors_host`: HOST array of 12 DEVICE pointers in state-dict order: pts_linears_weights.0,
 * pts_linears_biases.0, pts_linears_weights.1, pts_linears_biases.1, feature_linear_{weights,biases},
 * sigma_linear_{weights,biases}, view_linears_{weights,biases}, rgb_linear_{weights,biases}. */
int64_t nrf_kilo_packed_floats(int num_cells);
int nrf_kilo_pack_weights(const float* const* tensors_host, int num_cells, float* packed, nrf_stream_t stream);

/* Network._get_index + flat index (network.py:314-334, 246-253), bit-exact: cell[i] in [0, rx*ry*rz).
 * aabb_host: {min xyz, max xyz}; res_host: {rx, ry, rz}. */
int nrf_kilo_route(const float* pts, int64_t n, const float* aabb_host, const int* res_host, int32_t* cell,
                   nrf_stream_t stream);

/* Network.forward (network.py:223-311): route, counting-sort by cell, evaluate each point with its cell's
 * tiny MLP; raw (n, 4) = [r, g, b, sigma] in input order.  dirs (n, 3) or, with dir_index (n,), (m, 3). */
int64_t nrf_kilo_forward_workspace_bytes(int64_t n_points, int num_cells);
int nrf_kilo_forward(const float* packed, int num_cells, const float* aabb_host, const int* res_host, const float* pts,
                     const float* dirs, const int32_t* dir_index, int64_t n_points, float* raw, void* workspace,
                     int64_t workspace_bytes, nrf_stream_t stream);
/* The same two calls with the cells' matrices on the fp16 matrix cores (network.py:123-167 again): every weight and
 * layer input as two fp16 terms, three v_mfma_f32_32x32x16_f16 per product, fp32 accumulate, fp32 biases and heads
 * (the "fp16x2" rule of nrf_mlp_forward_rays_fp16x2; fp32-class: tests/test_gpu_kilo.py holds the fp32 kernel's bars).
 * packed_h2 has nrf_kilo_packed_floats(num_cells) floats like the fp32 records.  A point whose operands leave fp16's
 * range (|a| >= 65520), and every point of a cell with such a weight, comes out as NaN. */
int nrf_kilo_pack_weights_fp16x2(const float* const* tensors_host, int num_cells, float* packed_h2, nrf_stream_t stream);
int nrf_kilo_forward_fp16x2(const float* packed_h2, int num_cells, const float* aabb_host, const int* res_host,
                            const float* pts, const float* dirs, const int32_t* dir_index, int64_t n_points, float* raw,
                            void* workspace, int64_t workspace_bytes, nrf_stream_t stream);

/* The same two calls at fp32 accuracy on the bf16 matrix cores (network.py:123-167 again): every weight and layer input
 * as THREE bf16 terms (a fp32 value is exactly their sum), the six products w_i x_j with i + j <= 2 as six
 * v_mfma_f32_32x32x16_bf16 per fp32 product, added small terms first into one fp32 accumulator; fp32 biases and heads
 * (the "bf16x3" rule of nrf_mlp_forward_rays_bf16x3: fp32's 24 significand bits and fp32's range, no guard).
 * packed_x3 has nrf_kilo_packed_bf16x3_floats(num_cells) floats (36 KiB of fragments + the small block per cell). */
int64_t nrf_kilo_packed_bf16x3_floats(int num_cells);
int nrf_kilo_pack_weights_bf16x3(const float* const* tensors_host, int num_cells, float* packed_x3, nrf_stream_t stream);
int nrf_kilo_forward_bf16x3(const float* packed_x3, int num_cells, const float* aabb_host, const int* res_host,
                            const float* pts, const float* dirs, const int32_t* dir_index, int64_t n_points, float* raw,
                            void* workspace, int64_t workspace_bytes, nrf_stream_t stream);

/* Network.forward with the direction encodings evaluated once per RAY instead of once per point.  The reference expands
 * viewdirs per point and embeds every copy (network.py:243, 276-283); the 27 direction features depend on the ray only.
 * `kind` names the records / kernel: 0 = nrf_kilo_forward (fp32), 1 = nrf_kilo_forward_fp16x2, 2 = nrf_kilo_forward_bf16x3.
 * nrf_kilo_encode_dirs writes, per row of dirs (n_dirs, 3), the view layer's direction operands exactly as that kernel's
 * lanes would compute them (kind 1: the two-term fp16 fragments plus the range guard's maximum per row; kind 2: the
 * three-term bf16 fragments) into `table` (nrf_kilo_dir_table_bytes(n_dirs) bytes, 16-byte aligned; one size for every
 * kind: 196 B per row); nrf_kilo_forward_encoded is that kind's forward (`packed` accordingly) reading the table through
 * dir_index (required) -- bit-identical raw, ~100 fewer VALU instructions per 32-point tile.  n_dirs <= 2^24. */
int64_t nrf_kilo_dir_table_bytes(int64_t n_dirs);
int nrf_kilo_encode_dirs(const float* dirs, int64_t n_dirs, int kind, void* table, nrf_stream_t stream);
int nrf_kilo_forward_encoded(const float* packed, int kind, int num_cells, const float* aabb_host, const int* res_host,
                             const float* pts, const void* dir_table, int64_t n_dirs, const int32_t* dir_index,
                             int64_t n_points, float* raw, void* workspace, int64_t workspace_bytes, nrf_stream_t stream);

/* nrf_kilo_pack_weights (and, with packed_t non-NULL, nrf_kilo_pack_weights_transposed) for the cells with
 * cell_mask[cell] != 0 only (int32 per cell, device); the other cells' records are left as they are. */
int nrf_kilo_pack_weights_cells(const float* const* tensors_host, int num_cells, const int32_t* cell_mask, float* packed,
                                float* packed_t, nrf_stream_t stream);

/* nrf_kilo_forward for a training step, whose parameters change every step (BatchedKiloNeRF.forward reads the live
 * nn.Parameters, kilonerf/network.py:123-167): between routing and the MLPs, the records of the cells this batch
 * routes to are packed from the 12 tensors (`tensors_host` as in nrf_kilo_pack_weights) into `packed` and, if non-NULL,
 * `packed_t` (nrf_kilo_pack_weights_transposed's layout, for nrf_kilo_backward) -- except cells that `packed_cells`
 * (int32 per cell, device, or NULL) marks as packed from these parameters already (nrf_kilo_pack_weights_cells with
 * the cells the occupancy grid can reach, launched while the host waits for the sample count).  Records of other
 * cells are left as they were: both buffers are only good for this batch's forward and backward. */
int nrf_kilo_forward_packing(const float* const* tensors_host, const int32_t* packed_cells, float* packed, float* packed_t,
                             int num_cells,
                             const float* aabb_host, const int* res_host, const float* pts, const float* dirs,
                             const int32_t* dir_index, int64_t n_points, float* raw, void* workspace,
                             int64_t workspace_bytes, nrf_stream_t stream);

/* Occupancy-grid marching + early-ray-termination compositing: what volumn_renderer.py:78-86, 107-113 delegate
 * to nerfacc (absent, unpinned).  This library's own rule: intervals [t0 + k*step, min(t0 + (k+1)*step, t1)) over
 * the ray/box overlap clipped to [near, far], kept when the occupancy voxel (uint8 grid, (rx,ry,rz) C order) at the
 * interval mid-point is non-zero; w_i = T_i * alpha_i while T_i >= ert_eps; depth = sum w t_mid / max(acc, eps). */
/* Two passes with one host read of `total` between them.  The workspace holds the scan's block sums and, from
 * pass 1, one 64-bit keep mask per ray and 64-step chunk; pass 2 given `counts` and that workspace writes the kept
 * samples from the masks (no second walk of the grid); with workspace == NULL it walks the grid again. */
int64_t nrf_kilo_march_workspace_bytes(int64_t n_rays, float t_near, float t_far, float step);
int nrf_kilo_march_count(const float* rays_o, const float* rays_d, int64_t n_rays, const uint8_t* occ,
                         const int* occ_res_host, const float* aabb_host, float t_near, float t_far, float step,
                         int32_t* counts, int32_t* offsets, int64_t* total, void* workspace, int64_t workspace_bytes,
                         nrf_stream_t stream);
/* The same pass 1 over a BIT-BRICK copy of the grid (no counterpart in the reference: the grid lives inside nerfacc there,
 * volumn_renderer.py:47-54): nrf_occ_pack_bricks turns the uint8 grid into one 64-bit word per 4 x 4 x 4 voxels (bit
 * (x & 3) * 16 + (y & 3) * 4 + (z & 3); nrf_occ_bricks_bytes(res) bytes, 8-byte aligned; 2 MB for 256^3), built once per
 * grid; a ray's 64-step chunk then touches about ten cache lines instead of forty.  Same counts, offsets, total and masks
 * as nrf_kilo_march_count on the uint8 grid, bit for bit; pass 2 (nrf_kilo_march_write with that workspace) is unchanged. */
int64_t nrf_occ_bricks_bytes(const int* occ_res_host);
int nrf_occ_pack_bricks(const uint8_t* occ, const int* occ_res_host, uint64_t* bricks, nrf_stream_t stream);
int nrf_kilo_march_count_bricks(const float* rays_o, const float* rays_d, int64_t n_rays, const uint64_t* bricks,
                                const int* occ_res_host, const float* aabb_host, float t_near, float t_far, float step,
                                int32_t* counts, int32_t* offsets, int64_t* total, void* workspace,
                                int64_t workspace_bytes, nrf_stream_t stream);
int nrf_kilo_march_write(const float* rays_o, const float* rays_d, int64_t n_rays, const uint8_t* occ,
                         const int* occ_res_host, const float* aabb_host, float t_near, float t_far, float step,
                         const int32_t* offsets, float* pts, int32_t* ray_idx, float* t_start, float* t_end,
                         const int32_t* counts, const void* workspace, int64_t workspace_bytes, nrf_stream_t stream);
int nrf_kilo_composite(const float* raw, const float* t_start, const float* t_end, const int32_t* offsets,
                       const int32_t* counts, int64_t n_rays, float ert_eps, int white_bkgd, float* rgb_map,
                       float* depth_map, float* acc_map, nrf_stream_t stream);

/* Occupancy-grid bake from the teacher NeRF: Renderer._evaluate_grid (kilonerf/renderer/volumn_renderer.py:142-232).
 * For cells [first_cell, first_cell + n_cells) of the (rx,ry,rz) grid (C order): 27 sub-points per cell (:169-176,
 * :194) through the teacher's packed fine MLP with zero view directions (:204-207); occ[cell] = any raw sigma >
 * threshold (:212-220).  `occ` points at the whole grid (uint8, rx*ry*rz). */
int64_t nrf_occ_bake_workspace_bytes(int64_t n_cells);
int nrf_occ_bake(const float* packed_teacher, const float* aabb_host, const int* res_host, float threshold,
                 int64_t first_cell, int64_t n_cells, uint8_t* occ, void* workspace, int64_t workspace_bytes,
                 nrf_stream_t stream);
/* nrf_occ_bake with the teacher on the fp16x2 kernel (packed_teacher_h2 from nrf_pack_weights_fp16x2): 3.7x the bake
 * rate; densities are fp32-class, so only cells whose maximum density sits within rounding of the threshold can differ. */
int nrf_occ_bake_fp16x2(const void* packed_teacher_h2, const float* aabb_host, const int* res_host, float threshold,
                        int64_t first_cell, int64_t n_cells, uint8_t* occ, void* workspace, int64_t workspace_bytes,
                        nrf_stream_t stream);
/* nrf_occ_bake with the teacher on the bf16x3 kernel (packed_teacher_x3 from nrf_pack_weights_bf16x3): the same
 * `_evaluate_grid` (volumn_renderer.py:142-232) with every operand carried exactly (24 bits, fp32's range), about twice the
 * fp32 MFMA kernel's bake rate. */
int nrf_occ_bake_bf16x3(const void* packed_teacher_x3, const float* aabb_host, const int* res_host, float threshold,
                        int64_t first_cell, int64_t n_cells, uint8_t* occ, void* workspace, int64_t workspace_bytes,
                        nrf_stream_t stream);

/* The NeRF renderer's own occupancy grid: Renderer._build_occ_grid (src/models/nerf/renderer/volume_renderer.py:156-197).
 * ONE point per voxel at torch.linspace(min, max, res)[i] per axis (:170-178), through `packed` (the network's
 * default = coarse model, network.py:171-176) with zero view directions (:183-185); occ = raw sigma > threshold (:195). */
int64_t nrf_occ_bake_vertices_workspace_bytes(int64_t n_cells);
int nrf_occ_bake_vertices(const float* packed, const float* aabb_host, const int* res_host, float threshold,
                          int64_t first_cell, int64_t n_cells, uint8_t* occ, void* workspace, int64_t workspace_bytes,
                          nrf_stream_t stream);

/* ---- KiloNeRF training: what loss.backward() does in src/train/trainers/trainer.py:68-75 for the KiloNeRF
 * wrapper (src/train/trainers/kilonerf.py:34-63) ---- */

/* A operands of the transposed chain (pts_linears 1, feature_linear, view_linears feature columns). */
int64_t nrf_kilo_packed_bwd_floats(int num_cells);
int nrf_kilo_pack_weights_transposed(const float* const* tensors_host, int num_cells, float* packed_t,
                                     nrf_stream_t stream);

/* Backward of nrf_kilo_composite (this library's own rule; the termination mask is a constant):
 * grad_rgb (n_rays, 3), grad_depth (n_rays,) or NULL, grad_acc (n_rays,) or NULL -> grad_raw (n, 4), every
 * sample written.  Autograd through volumn_renderer.py:89-116 under the rule above. */
int nrf_kilo_composite_backward(const float* raw, const float* t_start, const float* t_end, const int32_t* offsets,
                                const int32_t* counts, int64_t n_rays, float ert_eps, int white_bkgd,
                                const float* grad_rgb, const float* grad_depth, const float* grad_acc, float* grad_raw,
                                nrf_stream_t stream);

/* Backward of nrf_kilo_forward w.r.t. the 12 parameter tensors (autograd through BatchedKiloNeRF.forward,
 * kilonerf/network.py:123-167): `forward_workspace` is the workspace nrf_kilo_forward filled for these same
 * points (it holds their order by cell) and must be untouched since.  grads_host: HOST array of 12 DEVICE
 * pointers shaped like the parameter tensors (nrf_kilo_pack_weights order); every element is overwritten
 * (zeros for cells without points); partial sums per job of a few tiles, added in job order: no atomics,
 * reproducible for a given forward workspace. */
int64_t nrf_kilo_backward_workspace_bytes(int64_t n_points, int num_cells);
int nrf_kilo_backward(const float* packed, const float* packed_t, int num_cells, const float* pts, const float* dirs,
                      const int32_t* dir_index, int64_t n_points, const float* grad_raw, const void* forward_workspace,
                      int64_t forward_workspace_bytes, float* const* grads_host, void* workspace,
                      int64_t workspace_bytes, nrf_stream_t stream);

/* Compact form of a set of cells' slices of the 12 batched (cells, ...) tensors: record u = the 12 per-cell slices
 * of cell cells[u] back to back (nrf_kilo_record_floats() = 6212 floats), `cells` a DEVICE list of cell indices.
 * The indices MUST be distinct (scatter writes each listed cell once; duplicates race and are not detected); an index
 * outside [0, num_cells) is skipped on the device (gather: its record is zeros; scatter: nothing is written).  For the data-parallel gradient exchange: only cells some
 * rank has samples in carry a gradient, so the ranks all-reduce n_cells records instead of all 25.4 M values.
 * The reference has no counterpart (DistributedDataParallel all-reduces every parameter, trainer.py:14-21). */
int64_t nrf_kilo_record_floats(void);
int nrf_kilo_cells_gather(const float* const* tensors_host, const int32_t* cells, int n_cells, int num_cells,
                          float* records, nrf_stream_t stream);
int nrf_kilo_cells_scatter(const float* records, const int32_t* cells, int n_cells, int num_cells,
                           float* const* tensors_host, nrf_stream_t stream);

/* *loss_accum += coeff * sum(p^2), grad_accum[i] += 2 coeff p[i] (either may be NULL): the colour regulariser
 * l2_reg * Network.get_color_reg_loss() (kilonerf/network.py:357-384, trainers/kilonerf.py:52-55), one call per
 * tensor (view_linears_{weights,biases}, rgb_linear_{weights,biases}). */
int nrf_l2_reg(const float* params, int64_t n, float coeff, float* grad_accum, float* loss_accum, nrf_stream_t stream);

/* One KiloNeRF training forward + backward in one host call: the body of NetworkWrapper.forward
 * (src/train/trainers/kilonerf.py:34-63: render, mse(rgb) [+ mse(alpha) when target_alpha is non-NULL, the distillation
 * stage]) and loss.backward() (trainer.py:68-72) for one batch of rays -- nrf_kilo_march_count, the host read of the
 * sample count, nrf_kilo_march_write, nrf_kilo_forward_packing, nrf_kilo_composite, nrf_mse_grad,
 * nrf_kilo_composite_backward, nrf_kilo_backward, issued back to back from C; while the count travels to the host
 * the records of the cells marked in `reach` are repacked from the parameters (nrf_kilo_pack_weights_cells).  The
 * colour regulariser and the optimiser are nrf_clip_adam_l2, after the caller's gradient all-reduce.
 * Returns NRF_ERR_WORKSPACE with *n_samples set when the batch has more samples than max_samples: size the workspace
 * for that (nrf_kilo_train_workspace_bytes) and call again.  The call blocks the host once (the sample count); it
 * keeps a pinned word and an event per device for that, created on the first call on the device.  Calls from
 * several host threads on one device must not overlap. */
typedef struct nrf_kilo_train_args {
    const float* rays_o;         /* (n_rays, 3) */
    const float* rays_d;         /* (n_rays, 3) */
    int64_t n_rays;
    const uint8_t* occ;          /* occupancy grid, (occ_res) C order */
    const int* occ_res;          /* HOST {rx, ry, rz} of the occupancy grid */
    const float* aabb;           /* HOST {min xyz, max xyz} */
    const int* res;              /* HOST {rx, ry, rz} of the cell grid */
    int num_cells;
    float t_near, t_far, step, ert_eps;
    int white_bkgd;
    const float* target_rgb;     /* (n_rays, 3) */
    const float* target_alpha;   /* (n_rays,) or NULL (fine-tuning stage: rgb loss only) */
    const float* const* tensors; /* HOST array of the 12 parameter tensors (nrf_kilo_pack_weights order) */
    float* const* grads;         /* HOST array of 12 gradient tensors, every element overwritten */
    const int32_t* reach;        /* (num_cells,) int32 or NULL: cells to repack while the sample count travels */
    float* packed;               /* nrf_kilo_packed_floats(num_cells) floats; good for this batch only */
    float* packed_t;             /* nrf_kilo_packed_bwd_floats(num_cells) floats */
    float* rgb_map;              /* out (n_rays, 3) */
    float* depth_map;            /* out (n_rays,) */
    float* acc_map;              /* out (n_rays,) */
    float* loss_accum;           /* out: one float, zeroed here, then mse terms added */
    int32_t* touched;            /* out (num_cells,) int32 or NULL: 1 where the batch has samples */
    int32_t* live;               /* in/out (num_cells,) int32 or NULL: live[c] |= 3 where the batch has samples in c (nrf_kilo_clip_adam's mask: both bits) */
    int64_t* n_samples;          /* HOST out: samples of this batch */
    int64_t max_samples;
    void* workspace;
    int64_t workspace_bytes;
} nrf_kilo_train_args;
int64_t nrf_kilo_train_workspace_bytes(int64_t n_rays, int64_t max_samples, int num_cells, float t_near, float t_far,
                                       float step);
int nrf_kilo_train_forward_backward(const nrf_kilo_train_args* args, nrf_stream_t stream);

/* nrf_clip_adam_l2 for the 12 batched KiloNeRF tensors (trainer.py:74-75 + optimizer.py:9-29 over
 * BatchedKiloNeRF's parameters), one block per cell: params / grads / exp_avg / exp_avg_sq are HOST arrays of 12 DEVICE
 * tensors (nrf_kilo_pack_weights order, each 16-byte aligned).  `live` (int32 per cell, or NULL = every tensor of
 * every cell) is a two-bit mask: bit 0 clear -> the cell's slices of the eight unregularised tensors are skipped,
 * bit 1 clear -> its slices of the four colour-regularised tensors (the last four) are skipped, unless this call
 * applies the regulariser (then they are updated in every cell).  Skipping is exact, not an approximation, PROVIDED
 * the skipped slices' gradient and BOTH Adam moments are zero (Adam then computes p - lr * (0 / eps) = p).  That is
 * the caller's contract for all twelve tensors: bit 0 | bit 1 (= 3) for every cell a batch has sampled since the
 * moments were zeroed (nrf_kilo_train_args.live ORs 3 in), bit 1 for EVERY cell from the first call that applied the
 * regulariser (its gradient 2 l2 p is non-zero everywhere, so every cell has moments in the last four tensors from
 * then on), and all cells = 3 whenever moments are loaded from elsewhere (a checkpoint's exp_avg / exp_avg_sq).
 * With the regulariser on, the last four tensors get grads += 2 reg_grad_coeff p (written back) and
 * *loss_accum += reg_loss_coeff sum p^2 in every cell. */
int nrf_kilo_clip_adam(float* const* params, float* const* grads, float* const* exp_avg, float* const* exp_avg_sq, int num_cells,
                       const int32_t* live, float grad_scale, float clip_value, float lr, float beta1, float beta2, float eps,
                       int step, float reg_grad_coeff, float reg_loss_coeff, float* loss_accum, nrf_stream_t stream);

/* ---- whole forward path: volume_renderer.py:143-153 (render_coarse + render_fine) --- */

typedef struct nrf_render_args {
    const float* packed_coarse;  /* nrf_pack_weights of Network.model      */
    const float* packed_fine;    /* nrf_pack_weights of Network.model_fine */
    const float* rays_o;         /* (n_rays, 3) batch['xyz']                */
    const float* rays_d;         /* (n_rays, 3) batch['viewdirs'], un-normalised */
    const float* bins;           /* (n_coarse,) near/far table (self.t_bins) */
    const float* u;              /* (n_rays, n_importance) first torch.rand of importance_sampling */
    const float* u2;             /* (n_rays, n_importance) second one */
    const float* u_coarse;       /* (n_rays, n_coarse) or NULL: perturb>0 and is_training */
    const float* u_fine;         /* (n_rays, n_importance) or NULL */
    int64_t n_rays;
    int n_coarse, n_importance, white_bkgd;
    float* coarse_rgb;           /* (n_rays, 3) */
    float* coarse_depth;         /* (n_rays,)   */
    float* fine_rgb;             /* (n_rays, 3) */
    float* fine_depth;           /* (n_rays,)   */
    float* fine_acc;             /* (n_rays, 1) 'alpha_map' */
    void* workspace;             /* >= nrf_render_workspace_bytes(...) */
    int64_t workspace_bytes;
} nrf_render_args;

int64_t nrf_render_workspace_bytes(int64_t n_rays, int n_coarse, int n_importance);
int nrf_render(const nrf_render_args* args, nrf_stream_t stream);
/* The same path with bf16 MLPs: packed_coarse / packed_fine are nrf_pack_weights_bf16 streams (cast to const float*),
 * everything else as above (compositing, resampling and all outputs stay fp32). */
int nrf_render_bf16(const nrf_render_args* args, nrf_stream_t stream);
/* ... and with the bf16x3 MLPs: packed_coarse / packed_fine are nrf_pack_weights_bf16x3 streams. */
int nrf_render_bf16x3(const nrf_render_args* args, nrf_stream_t stream);
/* ... and with the fp16x2 MLPs: packed_coarse / packed_fine are nrf_pack_weights_fp16x2 streams. */
int nrf_render_fp16x2(const nrf_render_args* args, nrf_stream_t stream);


/* ---- the whole forward path without any allocation ------------------------------------------------------------------
 * No counterpart in the reference (its renderer allocates through torch at every op).  For callers that cannot
 * allocate or pass scratch: code under HIP graph capture, serving loops with a frozen allocator, harnesses that forbid
 * hipMalloc.  Same enqueue sequence and results as nrf_render (fp32 MLPs), but args->workspace / workspace_bytes are
 * IGNORED: the workspace is a device array inside the library, sized for NRF_RENDER_STATIC_MAX_RAYS rays (the
 * reference's chunk_size) at up to 256 samples per ray = 23.1 MB per GPU the library runs on, held until the process
 * ends.  More rays or samples -> NRF_ERR_STATIC_LIMIT (split the batch, or use nrf_render).  Calls no hipMalloc /
 * hipFree / allocating routine.  All calls on one GPU share that GPU's array: calls that overlap on one GPU (two
 * streams, two threads) race; serialise them.  Inputs and outputs are the caller's, as in nrf_render. */
#define NRF_RENDER_STATIC_MAX_RAYS 4096
int nrf_render_static(const nrf_render_args* args, nrf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NERF_AMD_H */
