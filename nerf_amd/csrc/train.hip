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

// The same for batches of training size (n <= kMseOneBlock elements) in ONE workgroup: every thread owns a fixed set of
// elements, the partial sums meet in a fixed tree and one thread adds the total to the loss word -- no float atomics, so the
// loss of a batch has the same bits on every run and every stream (with one atomic per wavefront their order, and the last
// bit of the sum, depended on which wavefront came first).
constexpr int64_t kMseOneBlock = 1 << 16;
__global__ __launch_bounds__(1024) void mse_grad_one_block_kernel(const float* __restrict__ x, const float* __restrict__ gt,
                                                                  int64_t n, float* __restrict__ grad, float* __restrict__ loss) {
    __shared__ float s_part[16];
    float sq = 0.f;
    for (int64_t i0 = threadIdx.x; i0 < n; i0 += 4 * 1024) {   // four elements per thread and round, their loads in flight together
        float d[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t i = i0 + u * 1024;
            d[u] = i < n ? x[i] - gt[i] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t i = i0 + u * 1024;
            if (i < n) grad[i] = 2.f * d[u] / float(n);
            sq = fmaf(d[u], d[u], sq);
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sq += __shfl_xor(sq, off);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = sq;
    __syncthreads();
    if (threadIdx.x == 0 && loss) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += s_part[w];
        *loss += t / float(n);
    }
}

// torch.optim.Adam (single-tensor path): exp_avg.lerp_(g, 1-b1); exp_avg_sq = b2*v + (1-b2) g*g;
// denom = sqrt(v)/sqrt(1-b2^t) + eps; p -= (lr/(1-b1^t)) * m / denom.
struct AdamArgs {
    float grad_scale, clip, step_size, beta1, beta2, inv_bc2_sqrt, eps;
    // L2 regulariser of the tail [reg_begin, n) folded into the same pass (KiloNeRF's colour regulariser: the grad word
    // becomes g + reg_c2 * p, as nrf_l2_reg would have left it, before scale / clip / Adam; loss += reg_loss * sum p^2)
    int64_t reg_begin;
    float reg_c2, reg_loss;
};

__device__ __forceinline__ void adam_one(float& p, float& g, float& m, float& v, bool reg, const AdamArgs& a, float& sq) {
    if (reg) {
        sq = fmaf(p, p, sq);
        g = fmaf(a.reg_c2, p, g);
    }
    float gi = g * a.grad_scale;                        // mean over ranks (DDP semantics)
    gi = fminf(fmaxf(gi, -a.clip), a.clip);             // clip_grad_value_
    const float mi = m + (gi - m) * (1.f - a.beta1);    // lerp_
    const float vi = v * a.beta2 + (1.f - a.beta2) * gi * gi;
    m = mi;
    v = vi;
    const float denom = sqrtf(vi) * a.inv_bc2_sqrt + a.eps;
    p = p - a.step_size * (mi / denom);
}

// One block = 4096 consecutive parameters: four float4 of each of the four arrays per thread, all sixteen loads issued
// before the first use (16 B read + 12 B written per parameter; the regularised tail also writes its grad word back).
// No loop: a grid-stride form with 2048 blocks ran at 5.4 TB/s where this shape gives the memory system every load
// at once.  REG = the launch has a regularised tail; only its blocks reduce p^2 and touch the loss word (one
// hardware fp32 atomic per block, fire and forget).
constexpr int kAdamUnroll = 4;
template <bool REG>
__global__ __launch_bounds__(256) void clip_adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, int64_t n, AdamArgs a, float* __restrict__ loss) {
    const int64_t n4 = n >> 2;
    const int64_t base = int64_t(blockIdx.x) * (256 * kAdamUnroll) + threadIdx.x;
    float4 *p4 = reinterpret_cast<float4*>(p), *g4 = reinterpret_cast<float4*>(g), *m4 = reinterpret_cast<float4*>(m),
           *v4 = reinterpret_cast<float4*>(v);
    float4 pi[kAdamUnroll], gi[kAdamUnroll], mi[kAdamUnroll], vi[kAdamUnroll];
#pragma unroll
    for (int u = 0; u < kAdamUnroll; ++u) {
        const int64_t i = base + u * 256;
        if (i < n4) pi[u] = p4[i], gi[u] = g4[i], mi[u] = m4[i], vi[u] = v4[i];
    }
    float sq = 0.f;
    const bool block_reg = REG && (int64_t(blockIdx.x) + 1) * (1024 * kAdamUnroll) > a.reg_begin;   // block-uniform
#pragma unroll
    for (int u = 0; u < kAdamUnroll; ++u) {
        const int64_t i = base + u * 256;
        if (i >= n4) continue;
        const int64_t e = i << 2;
        adam_one(pi[u].x, gi[u].x, mi[u].x, vi[u].x, block_reg && e >= a.reg_begin, a, sq);
        adam_one(pi[u].y, gi[u].y, mi[u].y, vi[u].y, block_reg && e + 1 >= a.reg_begin, a, sq);
        adam_one(pi[u].z, gi[u].z, mi[u].z, vi[u].z, block_reg && e + 2 >= a.reg_begin, a, sq);
        adam_one(pi[u].w, gi[u].w, mi[u].w, vi[u].w, block_reg && e + 3 >= a.reg_begin, a, sq);
        p4[i] = pi[u], m4[i] = mi[u], v4[i] = vi[u];
        if (block_reg && e + 3 >= a.reg_begin) g4[i] = gi[u];
    }
    if (blockIdx.x == gridDim.x - 1) {   // the n % 4 tail
        const int64_t i = (n4 << 2) + threadIdx.x;
        if (i < n) {
            float ps = p[i], gs = g[i], ms = m[i], vs = v[i];
            const bool reg = REG && i >= a.reg_begin;
            adam_one(ps, gs, ms, vs, reg, a, sq);
            p[i] = ps, m[i] = ms, v[i] = vs;
            if (reg) g[i] = gs;
        }
    }
    if (block_reg && loss) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) sq += __shfl_xor(sq, off);
        __shared__ float s_part[4];
        if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = sq;
        __syncthreads();
        if (threadIdx.x == 0) unsafeAtomicAdd(loss, a.reg_loss * (s_part[0] + s_part[1] + s_part[2] + s_part[3]));
    }
}

// unaligned buffers (views at odd offsets): one element per thread
__global__ __launch_bounds__(256) void clip_adam_scalar_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                               float* __restrict__ v, int64_t n, AdamArgs a) {
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float pi = p[i], gi = g[i], mi = m[i], vi = v[i], sq = 0.f;
    adam_one(pi, gi, mi, vi, false, a, sq);
    p[i] = pi, m[i] = mi, v[i] = vi;
}

int launch_adam(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float grad_scale, float clip_value,
                float lr, float beta1, float beta2, float eps, int step, int64_t reg_begin, float reg_grad_coeff,
                float reg_loss_coeff, float* loss_accum, hipStream_t stream) {
    if (n < 0 || step < 1 || !(clip_value > 0.f) || reg_begin < 0 || reg_begin > n) return NRF_ERR_SHAPE;
    if (n == 0) return NRF_OK;
    if (!params || !grads || !exp_avg || !exp_avg_sq) return NRF_ERR_NULL;
    const double bc1 = 1.0 - pow(double(beta1), step), bc2 = 1.0 - pow(double(beta2), step);
    const AdamArgs a{grad_scale, clip_value, float(double(lr) / bc1), beta1, beta2, float(1.0 / sqrt(bc2)), eps,
                     reg_begin, 2.f * reg_grad_coeff, reg_loss_coeff};
    const bool reg = reg_begin < n && (reg_grad_coeff != 0.f || (reg_loss_coeff != 0.f && loss_accum));
    const bool vec = nrf_aligned16(params) && nrf_aligned16(grads) && nrf_aligned16(exp_avg) && nrf_aligned16(exp_avg_sq);
    if (reg && !vec) return NRF_ERR_ALIGN;
    nrf_launch_begin();
    if (!vec) {
        const int64_t blocks = (n + 255) / 256;
        if (blocks > 0x7fffffffLL) return NRF_ERR_SHAPE;
        clip_adam_scalar_kernel<<<dim3((unsigned)blocks), 256, 0, stream>>>(params, grads, exp_avg, exp_avg_sq, n, a);
    } else {
        const int64_t blocks = n / 4 / (256 * kAdamUnroll) + 1;   // the last block also takes the n % 4 tail
        if (blocks > 0x7fffffffLL) return NRF_ERR_SHAPE;
        const unsigned grid = unsigned(blocks);
        if (reg)
            clip_adam_kernel<true><<<dim3(grid), 256, 0, stream>>>(params, grads, exp_avg, exp_avg_sq, n, a, loss_accum);
        else
            clip_adam_kernel<false><<<dim3(grid), 256, 0, stream>>>(params, grads, exp_avg, exp_avg_sq, n, a, nullptr);
    }
    return nrf_launch_status();
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
    if (n <= kMseOneBlock)
        mse_grad_one_block_kernel<<<1, 1024, 0, static_cast<hipStream_t>(stream)>>>(x, target, n, grad, loss_accum);
    else
        mse_grad_kernel<<<dim3((unsigned)blocks), 256, 0, static_cast<hipStream_t>(stream)>>>(x, target, n, grad, loss_accum);
    return nrf_launch_status();
}

NRF_EXPORT int nrf_clip_adam(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                             float grad_scale, float clip_value, float lr, float beta1, float beta2, float eps, int step,
                             nrf_stream_t stream) {
    return launch_adam(params, const_cast<float*>(grads), exp_avg, exp_avg_sq, n, grad_scale, clip_value, lr, beta1, beta2, eps,
                       step, n, 0.f, 0.f, nullptr, static_cast<hipStream_t>(stream));   // no regularised tail: grads is only read
}

NRF_EXPORT int nrf_clip_adam_l2(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float grad_scale,
                                float clip_value, float lr, float beta1, float beta2, float eps, int step, int64_t reg_begin,
                                float reg_grad_coeff, float reg_loss_coeff, float* loss_accum, nrf_stream_t stream) {
    return launch_adam(params, grads, exp_avg, exp_avg_sq, n, grad_scale, clip_value, lr, beta1, beta2, eps, step, reg_begin,
                       reg_grad_coeff, reg_loss_coeff, loss_accum, static_cast<hipStream_t>(stream));
}
