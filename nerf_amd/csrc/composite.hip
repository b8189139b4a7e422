// Activation + alpha compositing as a wavefront scan (HBM-bound).
//
// Replaces volume_renderer.py:222-224 (delta), :244-265 (coarse) and :305-334 (fine):
// a chain of ~15 elementwise / cumprod / reduction PyTorch ops, each a full HBM pass, becomes
// one pass: 20 B read per sample (raw float4 + z), 20 B written per ray (+4 B/sample when the
// coarse weights are kept for importance resampling).
//
// One wavefront per ray.  Samples are taken 64 at a time (lane = sample, so the float4 load
// is one contiguous 1 KiB per wave-instruction and the z load 256 B); the exclusive
// transmittance product is a 6-step wave scan carried across the chunks of a ray.
#include "nrf_common.h"

namespace {

// The kernel is one HBM pass, but with libm's expf / IEEE division and ds_bpermute scans it was bound by its ~640
// instructions per ray, not by memory.  exp and the sigmoid's reciprocal are the hardware v_exp_f32 / v_rcp_f32
// (1 ulp each; the argument scaling adds |x| * 6e-8 relative to exp(x), which alpha = 1 - exp(-x) and the sigmoid
// turn into < 3e-8 absolute), the scans run on DPP (nrf_common.h).
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + fast_exp(-x)); }

template <int NCHUNK>
__global__ __launch_bounds__(256) void composite_kernel(const float4* __restrict__ raw, const float* __restrict__ z,
                                                        int64_t z_stride, int64_t n_rays, int white_bkgd,
                                                        float* __restrict__ rgb_map, float* __restrict__ depth_map,
                                                        float* __restrict__ acc_map, float* __restrict__ weights) {
    constexpr int S = NCHUNK * 64;
    const int lane = threadIdx.x & 63;
    const int64_t ray = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (ray >= n_rays) return;  // wave-uniform
    const float4* rr = raw + ray * S;
    const float* zr = z + ray * z_stride;

    float4 rv[NCHUNK];
    float zv[NCHUNK], zn[NCHUNK];
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c) {  // all loads of the ray in flight before any use
        const int s = c * 64 + lane;
        rv[c] = rr[s];
        zv[c] = zr[s];
        zn[c] = zr[s + 1 < S ? s + 1 : s];
    }

    float carry = 1.f;  // product of (1 - alpha) over all earlier chunks
    float sr = 0.f, sg = 0.f, sb = 0.f, sa = 0.f, sd = 0.f;
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c) {
        const int s = c * 64 + lane;
        const float delta = (s + 1 < S) ? zn[c] - zv[c] : 1e10f;          // :222-224
        const float sigma = fmaxf(rv[c].w, 0.f);                          // :244
        const float alpha = 1.f - fast_exp(-sigma * delta);               // :248
        const float om = 1.f - alpha;
        const float p = nrf_wave_scan_mul(om);  // inclusive product scan over the 64 lanes
        const float T = carry * nrf_lane_shr1(1.f, p);                    // :249-250 exclusive cumprod
        carry *= nrf_read_lane63(p);
        const float w = T * alpha;                                        // :252
        if (weights) weights[ray * S + s] = w;
        sr += w * fast_sigmoid(rv[c].x);                                  // :246 sigmoid, :254
        sg += w * fast_sigmoid(rv[c].y);
        sb += w * fast_sigmoid(rv[c].z);
        sa += w;                                                          // :256
        sd += w * zv[c];                                                  // :263
    }
    sr = nrf_wave_scan_add(sr), sg = nrf_wave_scan_add(sg), sb = nrf_wave_scan_add(sb), sa = nrf_wave_scan_add(sa),
    sd = nrf_wave_scan_add(sd);
    if (lane == 63) {  // the inclusive sums end in lane 63
        const float bg = white_bkgd ? 1.f - sa : 0.f;                     // :259-261
        rgb_map[ray * 3 + 0] = sr + bg;
        rgb_map[ray * 3 + 1] = sg + bg;
        rgb_map[ray * 3 + 2] = sb + bg;
        depth_map[ray] = sd;
        acc_map[ray] = sa;
    }
}

// Backward of the compositing: d(raw) from d(rgb_map), d(depth_map), d(acc_map) (any may be
// NULL).  Same one-wave-per-ray layout; recomputes alpha / T / w from raw and z (nothing saved),
// then dL/d alpha_i = T_i * (G_i - S_i) with G_i = dL/dw_i and the division-free suffix recurrence
// S_i = alpha_{i+1} G_{i+1} + (1 - alpha_{i+1}) S_{i+1} (alpha = 1 exactly at the 1e10 last
// interval, so the textbook T/(1 - alpha) form would produce inf/NaN), evaluated as a wave-wide
// scan of affine maps carried across the chunks of a ray in reverse order.
template <int NCHUNK>
__global__ __launch_bounds__(256) void composite_bwd_kernel(const float4* __restrict__ raw, const float* __restrict__ z,
                                                            int64_t z_stride, int64_t n_rays, int white_bkgd,
                                                            const float* __restrict__ g_rgb,
                                                            const float* __restrict__ g_depth,
                                                            const float* __restrict__ g_acc, float4* __restrict__ g_raw) {
    constexpr int S = NCHUNK * 64;
    const int lane = threadIdx.x & 63;
    const int64_t ray = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (ray >= n_rays) return;
    const float4* rr = raw + ray * S;
    const float* zr = z + ray * z_stride;
    const float gr = g_rgb ? g_rgb[ray * 3 + 0] : 0.f, gg = g_rgb ? g_rgb[ray * 3 + 1] : 0.f,
                gb = g_rgb ? g_rgb[ray * 3 + 2] : 0.f;
    const float gd = g_depth ? g_depth[ray] : 0.f, ga = g_acc ? g_acc[ray] : 0.f;
    const float bgc = white_bkgd ? 1.f : 0.f;  // rgb_map = sum w c + (1 - sum w) * white

    float4 rv[NCHUNK];
    float alpha[NCHUNK], om[NCHUNK], T[NCHUNK], G[NCHUNK], delta[NCHUNK], cr[NCHUNK], cg[NCHUNK], cb[NCHUNK];
    float carry = 1.f;
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c) {
        const int s = c * 64 + lane;
        rv[c] = rr[s];
        const float zv = zr[s];
        const float zn = zr[s + 1 < S ? s + 1 : s];
        delta[c] = (s + 1 < S) ? zn - zv : 1e10f;
        const float sigma = fmaxf(rv[c].w, 0.f);
        alpha[c] = 1.f - expf(-sigma * delta[c]);
        om[c] = 1.f - alpha[c];
        float p = om[c];
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const float t = __shfl_up(p, off);
            if (lane >= off) p *= t;
        }
        float ex = __shfl_up(p, 1);
        if (lane == 0) ex = 1.f;
        T[c] = carry * ex;
        carry *= __shfl(p, 63);
        cr[c] = 1.f / (1.f + expf(-rv[c].x));
        cg[c] = 1.f / (1.f + expf(-rv[c].y));
        cb[c] = 1.f / (1.f + expf(-rv[c].z));
        G[c] = gr * (cr[c] - bgc) + gg * (cg[c] - bgc) + gb * (cb[c] - bgc) + gd * zv + ga;  // dL/dw_i
    }
    float scarry = 0.f;  // S entering from the samples behind this chunk
#pragma unroll
    for (int c = NCHUNK - 1; c >= 0; --c) {
        // R_i = f_i(f_{i+1}(... f_63(scarry))) with f_i(x) = om_i x + alpha_i G_i: suffix scan of (A, B)
        float A = om[c], B = alpha[c] * G[c];
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const float A2 = __shfl_down(A, off), B2 = __shfl_down(B, off);
            if (lane + off < 64) {
                B = fmaf(A, B2, B);
                A *= A2;
            }
        }
        const float R = fmaf(A, scarry, B);
        float Si = __shfl_down(R, 1);
        if (lane == 63) Si = scarry;
        scarry = __shfl(R, 0);
        const float w = T[c] * alpha[c];
        const float dalpha = T[c] * (G[c] - Si);
        // alpha = 1 - exp(-relu(raw_sigma) delta): d alpha / d raw_sigma = delta (1 - alpha) for raw_sigma > 0
        const float dsig = rv[c].w > 0.f ? dalpha * delta[c] * om[c] : 0.f;
        g_raw[ray * S + c * 64 + lane] = make_float4(gr * w * cr[c] * (1.f - cr[c]), gg * w * cg[c] * (1.f - cg[c]),
                                                     gb * w * cb[c] * (1.f - cb[c]), dsig);
    }
}

}  // namespace

NRF_EXPORT int nrf_composite(const float* raw, const float* z, int64_t z_stride, int64_t n_rays, int n_samples,
                             int white_bkgd, float* rgb_map, float* depth_map, float* acc_map, float* weights,
                             nrf_stream_t stream_) {
    if (!raw || !z || !rgb_map || !depth_map || !acc_map) return NRF_ERR_NULL;
    if (!nrf_aligned16(raw)) return NRF_ERR_ALIGN;
    if (n_rays < 0 || n_samples <= 0 || n_samples % 64 != 0 || n_samples > 1024 || z_stride < 0) return NRF_ERR_SHAPE;
    if (z_stride != 0 && z_stride < n_samples) return NRF_ERR_SHAPE;
    if (n_rays == 0) return NRF_OK;
    const int64_t blocks = (n_rays + 3) / 4;
    if (blocks > 0x7fffffffLL) return NRF_ERR_SHAPE;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const float4* r4 = reinterpret_cast<const float4*>(raw);
#define NRF_COMPOSITE_CASE(N)                                                                              \
    case N:                                                                                                \
        composite_kernel<N><<<dim3((unsigned)blocks), 256, 0, stream>>>(r4, z, z_stride, n_rays, white_bkgd, rgb_map, \
                                                                        depth_map, acc_map, weights);      \
        break;
    nrf_launch_begin();
    switch (n_samples / 64) {
        NRF_COMPOSITE_CASE(1)
        NRF_COMPOSITE_CASE(2)
        NRF_COMPOSITE_CASE(3)
        NRF_COMPOSITE_CASE(4)
        NRF_COMPOSITE_CASE(5)
        NRF_COMPOSITE_CASE(6)
        NRF_COMPOSITE_CASE(7)
        NRF_COMPOSITE_CASE(8)
        NRF_COMPOSITE_CASE(9)
        NRF_COMPOSITE_CASE(10)
        NRF_COMPOSITE_CASE(11)
        NRF_COMPOSITE_CASE(12)
        NRF_COMPOSITE_CASE(13)
        NRF_COMPOSITE_CASE(14)
        NRF_COMPOSITE_CASE(15)
        NRF_COMPOSITE_CASE(16)
        default:
            return NRF_ERR_SHAPE;
    }
#undef NRF_COMPOSITE_CASE
    return nrf_launch_status();
}

NRF_EXPORT int nrf_composite_backward(const float* raw, const float* z, int64_t z_stride, int64_t n_rays, int n_samples,
                                      int white_bkgd, const float* grad_rgb, const float* grad_depth,
                                      const float* grad_acc, float* grad_raw, nrf_stream_t stream_) {
    if (n_rays < 0 || n_samples <= 0 || n_samples % 64 != 0 || n_samples > 512 || z_stride < 0) return NRF_ERR_SHAPE;
    if (z_stride != 0 && z_stride < n_samples) return NRF_ERR_SHAPE;
    if (n_rays == 0) return NRF_OK;
    if (!raw || !z || !grad_raw) return NRF_ERR_NULL;
    if (!nrf_aligned16(raw) || !nrf_aligned16(grad_raw)) return NRF_ERR_ALIGN;
    const int64_t blocks = (n_rays + 3) / 4;
    if (blocks > 0x7fffffffLL) return NRF_ERR_SHAPE;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const float4* r4 = reinterpret_cast<const float4*>(raw);
    float4* g4 = reinterpret_cast<float4*>(grad_raw);
#define NRF_CBWD_CASE(N)                                                                                       \
    case N:                                                                                                    \
        composite_bwd_kernel<N><<<dim3((unsigned)blocks), 256, 0, stream>>>(r4, z, z_stride, n_rays, white_bkgd, grad_rgb, \
                                                                            grad_depth, grad_acc, g4);         \
        break;
    nrf_launch_begin();
    switch (n_samples / 64) {
        NRF_CBWD_CASE(1)
        NRF_CBWD_CASE(2)
        NRF_CBWD_CASE(3)
        NRF_CBWD_CASE(4)
        NRF_CBWD_CASE(5)
        NRF_CBWD_CASE(6)
        NRF_CBWD_CASE(7)
        NRF_CBWD_CASE(8)
        default:
            return NRF_ERR_SHAPE;
    }
#undef NRF_CBWD_CASE
    return nrf_launch_status();
}
