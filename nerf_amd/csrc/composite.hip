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

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

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
        const float alpha = 1.f - expf(-sigma * delta);                   // :248
        const float om = 1.f - alpha;
        float p = om;  // inclusive product scan over the 64 lanes
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const float t = __shfl_up(p, off);
            if (lane >= off) p *= t;
        }
        float ex = __shfl_up(p, 1);
        if (lane == 0) ex = 1.f;
        const float T = carry * ex;                                       // :249-250 exclusive cumprod
        carry *= __shfl(p, 63);
        const float w = T * alpha;                                        // :252
        if (weights) weights[ray * S + s] = w;
        sr += w / (1.f + expf(-rv[c].x));                                 // :246 sigmoid, :254
        sg += w / (1.f + expf(-rv[c].y));
        sb += w / (1.f + expf(-rv[c].z));
        sa += w;                                                          // :256
        sd += w * zv[c];                                                  // :263
    }
    sr = wave_sum(sr), sg = wave_sum(sg), sb = wave_sum(sb), sa = wave_sum(sa), sd = wave_sum(sd);
    if (lane == 0) {
        const float bg = white_bkgd ? 1.f - sa : 0.f;                     // :259-261
        rgb_map[ray * 3 + 0] = sr + bg;
        rgb_map[ray * 3 + 1] = sg + bg;
        rgb_map[ray * 3 + 2] = sb + bg;
        depth_map[ray] = sd;
        acc_map[ray] = sa;
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
