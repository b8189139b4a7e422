// Loss gradient and the fused clip + Adam parameter update of the training step.
//
// Replaces src/train/trainers/nerf.py:25-27 (MSELoss, mean over R*3) with its gradient, and
// src/train/trainers/trainer.py:74-75 (`clip_grad_value_(params, 40)`; `optimizer.step()`) with
// torch.optim.Adam's defaults from src/train/optimizer.py:9-29 (lr 5e-4, betas 0.9/0.999, eps 1e-8,
// weight_decay 0) -- ~48 tensors x several foreach kernels in the reference -- by ONE pass over a
// flat parameter buffer: 16 B read + 12 B written per parameter, HBM-bound.
#include "nrf_common.h"

namespace {

__global__ __launch_bounds__(256) void mse_grad_kernel(const float* __restrict__ x, const float* __restrict__ gt, int64_t n,
                                                       float* __restrict__ grad, float* __restrict__ loss) {
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    float sq = 0.f;
    if (i < n) {
        const float d = x[i] - gt[i];
        grad[i] = 2.f * d / float(n);
        sq = d * d;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sq += __shfl_xor(sq, off);
    if ((threadIdx.x & 63) == 0 && loss) atomicAdd(loss, sq / float(n));
}

// torch.optim.Adam (single-tensor path): exp_avg.lerp_(g, 1-b1); exp_avg_sq = b2*v + (1-b2) g*g;
// denom = sqrt(v)/sqrt(1-b2^t) + eps; p -= (lr/(1-b1^t)) * m / denom.
__global__ __launch_bounds__(256) void clip_adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                        float grad_scale, float clip, float step_size, float beta1,
                                                        float beta2, float inv_bc2_sqrt, float eps) {
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float gi = g[i] * grad_scale;                       // mean over ranks (DDP semantics)
    gi = fminf(fmaxf(gi, -clip), clip);                 // clip_grad_value_
    const float mi = m[i] + (gi - m[i]) * (1.f - beta1);  // lerp_
    const float vi = v[i] * beta2 + (1.f - beta2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) * inv_bc2_sqrt + eps;
    p[i] = p[i] - step_size * (mi / denom);
}

}  // namespace

NRF_EXPORT int nrf_mse_grad(const float* x, const float* target, int64_t n, float* grad, float* loss_accum,
                            nrf_stream_t stream) {
    if (n < 0) return NRF_ERR_SHAPE;
    if (n == 0) return NRF_OK;
    if (!x || !target || !grad) return NRF_ERR_NULL;
    const int64_t blocks = (n + 255) / 256;
    if (blocks > 0x7fffffffLL) return NRF_ERR_SHAPE;
    nrf_launch_begin();
    mse_grad_kernel<<<dim3((unsigned)blocks), 256, 0, static_cast<hipStream_t>(stream)>>>(x, target, n, grad, loss_accum);
    return nrf_launch_status();
}

NRF_EXPORT int nrf_clip_adam(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                             float grad_scale, float clip_value, float lr, float beta1, float beta2, float eps, int step,
                             nrf_stream_t stream) {
    if (n < 0 || step < 1 || !(clip_value > 0.f)) return NRF_ERR_SHAPE;
    if (n == 0) return NRF_OK;
    if (!params || !grads || !exp_avg || !exp_avg_sq) return NRF_ERR_NULL;
    const int64_t blocks = (n + 255) / 256;
    if (blocks > 0x7fffffffLL) return NRF_ERR_SHAPE;
    const double bc1 = 1.0 - pow(double(beta1), step), bc2 = 1.0 - pow(double(beta2), step);
    nrf_launch_begin();
    clip_adam_kernel<<<dim3((unsigned)blocks), 256, 0, static_cast<hipStream_t>(stream)>>>(
        params, grads, exp_avg, exp_avg_sq, n, grad_scale, clip_value, float(double(lr) / bc1), beta1, beta2,
        float(1.0 / sqrt(bc2)), eps);
    return nrf_launch_status();
}
