// Stratified-sample perturbation, importance resampling (CDF + searchsorted in LDS) and the
// coarse/fine merge sort -- one wavefront per ray, bit-exact with the reference on torch-CPU.
//
// Replaces volume_renderer.py:336-349 (add_perturbation), :363-391 (importance_sampling) and
// :283-289 (cat + sort).  HBM traffic per ray: 64 weights + 2x128 uniforms in, 192 z out.
//
// Bit-exactness contract (oracle/nerf_oracle.py models):
//   * sum(w) over 64 bins is torch-CPU's 8-lane x 4-accumulator fp32 cascade;
//   * cumsum(pdf) is a sequential fp64 running sum, each prefix rounded to fp32;
//   * every fp32 op is a separately rounded IEEE op (explicit __f*_rn: no FMA contraction),
//     the division is correctly rounded.
#include "nrf_common.h"

namespace {

__device__ __forceinline__ void wave_sync() {
    // LDS accesses of one wave execute in order; this only stops compiler reordering.
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// lower + (upper - lower) * u with mid-point bins (volume_renderer.py:345-348)
__device__ __forceinline__ float perturb_one(float prev, float cur, float next, bool first, bool last, float u) {
    const float upper = last ? cur : __fmul_rn(0.5f, __fadd_rn(next, cur));
    const float lower = first ? cur : __fmul_rn(0.5f, __fadd_rn(cur, prev));
    return __fadd_rn(lower, __fmul_rn(__fsub_rn(upper, lower), u));
}

__global__ __launch_bounds__(256) void perturb_kernel(const float* __restrict__ z_in, int64_t z_stride,
                                                      const float* __restrict__ u, int64_t n_rays, int n,
                                                      float* __restrict__ z_out) {
    const int64_t idx = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (idx >= n_rays * n) return;
    const int64_t ray = idx / n;
    const int i = int(idx - ray * n);
    const float* zr = z_in + ray * z_stride;
    const float cur = zr[i];
    const float prev = i > 0 ? zr[i - 1] : cur;
    const float next = i + 1 < n ? zr[i + 1] : cur;
    z_out[idx] = perturb_one(prev, cur, next, i == 0, i + 1 == n, u[idx]);
}

template <int NI>  // n_importance = 64 * NI
__global__ __launch_bounds__(256) void sample_fine_kernel(const float* __restrict__ weights, const float* __restrict__ bins,
                                                          int64_t bins_stride, const float* __restrict__ z_coarse,
                                                          int64_t zc_stride, const float* __restrict__ u,
                                                          const float* __restrict__ u2, const float* __restrict__ u_fine,
                                                          int64_t n_rays, float* __restrict__ z_out,
                                                          float* __restrict__ fine_out, int32_t* __restrict__ idx_out) {
    constexpr int NC = 64, NIMP = 64 * NI, NOUT = NC + NIMP;
    __shared__ float s_cdf[4][80];
    __shared__ float s_bins[4][64];
    __shared__ float s_sort[4][256];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t ray = int64_t(blockIdx.x) * 4 + wv;
    if (ray >= n_rays) return;  // wave-uniform; LDS regions are wave-private, no block barrier used
    float* cdf = s_cdf[wv];
    float* sbin = s_bins[wv];
    float* srt = s_sort[wv];

    const float w = weights[ray * NC + lane];
    sbin[lane] = bins[ray * bins_stride + lane];
    srt[lane] = z_coarse[ray * zc_stride + lane];

    // ---- sum(w): torch-CPU cascade.  v[i] = x[i] + x[32+i]; lane partial l = ((v[l]+v[8+l])+v[16+l])+v[24+l];
    //      total = (((0 + L0) + L1) + ... + L7)                                   (volume_renderer.py:374)
    const float v = __fadd_rn(w, __shfl_down(w, 32));
    float part = __fadd_rn(v, __shfl_down(v, 8));
    part = __fadd_rn(part, __shfl_down(v, 16));
    part = __fadd_rn(part, __shfl_down(v, 24));
    float tot = 0.f;
#pragma unroll
    for (int l = 0; l < 8; ++l) tot = __fadd_rn(tot, __shfl(part, l));
    const float pdf = __fdiv_rn(w, __fadd_rn(tot, 1e-10f));

    // ---- cdf = [0, cumsum(pdf)]: sequential fp64 running sum, prefix k rounded to fp32 (:375-376)
    double acc = 0.0;
#pragma unroll
    for (int jj = 0; jj < 64; ++jj) {
        const float pj = __shfl(pdf, jj);
        acc = __dadd_rn(acc, jj <= lane ? double(pj) : 0.0);
    }
    cdf[lane + 1] = float(acc);
    if (lane == 0) cdf[0] = 0.f;
    wave_sync();

    // ---- searchsorted(cdf, u, right=True) - 1, clamp, gather bins, lerp (:379-390)
    float fine[NI];
#pragma unroll
    for (int c = 0; c < NI; ++c) {
        const int i = c * 64 + lane;
        const float uu = u[ray * NIMP + i];
        int lo = 0, hi = NC + 1;  // upper bound over the 65 cdf entries
#pragma unroll
        for (int it = 0; it < 7; ++it) {
            if (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (cdf[mid] <= uu) lo = mid + 1; else hi = mid;
            }
        }
        int idx = lo - 1;
        idx = idx < 0 ? 0 : (idx > NC - 1 ? NC - 1 : idx);
        const float b_lo = sbin[idx];
        const float b_hi = sbin[idx + 1 > NC - 1 ? NC - 1 : idx + 1];
        fine[c] = __fadd_rn(b_lo, __fmul_rn(__fsub_rn(b_hi, b_lo), u2[ray * NIMP + i]));
        if (idx_out) idx_out[ray * NIMP + i] = idx;
        srt[NC + i] = fine[c];
    }
    if (u_fine) {  // add_perturbation on the (unsorted) fine samples, render_fine :283-284
        wave_sync();
        float pf[NI];
#pragma unroll
        for (int c = 0; c < NI; ++c) {
            const int i = c * 64 + lane;
            const float prev = i > 0 ? srt[NC + i - 1] : fine[c];
            const float next = i + 1 < NIMP ? srt[NC + i + 1] : fine[c];
            pf[c] = perturb_one(prev, fine[c], next, i == 0, i + 1 == NIMP, u_fine[ray * NIMP + i]);
        }
        wave_sync();
#pragma unroll
        for (int c = 0; c < NI; ++c) {
            fine[c] = pf[c];
            srt[NC + c * 64 + lane] = pf[c];
        }
    }
    if (fine_out) {
#pragma unroll
        for (int c = 0; c < NI; ++c) fine_out[ray * NIMP + c * 64 + lane] = fine[c];
    }
#pragma unroll
    for (int i = NOUT + lane; i < 256; i += 64) srt[i] = __builtin_inff();
    wave_sync();

    // ---- sort(cat(coarse, fine)) (:288-289): bitonic network over 256 LDS slots, 2 pairs per lane per stage
    for (int k = 2; k <= 256; k <<= 1) {
        for (int jd = k >> 1; jd > 0; jd >>= 1) {
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
                const int p = lane + 64 * pp;
                const int i = ((p & ~(jd - 1)) << 1) | (p & (jd - 1));
                const int q = i | jd;
                const float a = srt[i], b = srt[q];
                const bool up = (i & k) == 0;
                if ((a > b) == up) {
                    srt[i] = b;
                    srt[q] = a;
                }
            }
            wave_sync();
        }
    }
#pragma unroll
    for (int i = lane; i < NOUT; i += 64) z_out[ray * NOUT + i] = srt[i];
}

}  // namespace

NRF_EXPORT int nrf_perturb_samples(const float* z_in, int64_t z_stride, const float* u, int64_t n_rays, int n,
                                   float* z_out, nrf_stream_t stream) {
    if (!z_in || !u || !z_out) return NRF_ERR_NULL;
    if (n_rays < 0 || n <= 0 || n > 1024 || z_stride < 0 || (z_stride != 0 && z_stride < n)) return NRF_ERR_SHAPE;
    if (n_rays == 0) return NRF_OK;
    const int64_t blocks = (n_rays * n + 255) / 256;
    if (blocks > 0x7fffffffLL) return NRF_ERR_SHAPE;
    nrf_launch_begin();
    perturb_kernel<<<dim3((unsigned)blocks), 256, 0, static_cast<hipStream_t>(stream)>>>(z_in, z_stride, u, n_rays, n, z_out);
    return nrf_launch_status();
}

NRF_EXPORT int nrf_sample_fine(const float* weights, const float* bins, int64_t bins_stride, const float* z_coarse,
                               int64_t zc_stride, const float* u, const float* u2, const float* u_fine, int64_t n_rays,
                               int n_coarse, int n_importance, float* z_out, float* fine_out, int32_t* idx_out,
                               nrf_stream_t stream_) {
    if (!weights || !bins || !z_coarse || !u || !u2 || !z_out) return NRF_ERR_NULL;
    if (n_coarse != 64 || n_importance <= 0 || n_importance % 64 != 0 || n_coarse + n_importance > 256)
        return NRF_ERR_SHAPE;
    if (n_rays < 0 || bins_stride < 0 || zc_stride < 0) return NRF_ERR_SHAPE;
    if ((bins_stride != 0 && bins_stride < 64) || (zc_stride != 0 && zc_stride < 64)) return NRF_ERR_SHAPE;
    if (n_rays == 0) return NRF_OK;
    const int64_t blocks = (n_rays + 3) / 4;
    if (blocks > 0x7fffffffLL) return NRF_ERR_SHAPE;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const dim3 grid((unsigned)blocks);
    nrf_launch_begin();
    switch (n_importance / 64) {
        case 1:
            sample_fine_kernel<1><<<grid, 256, 0, stream>>>(weights, bins, bins_stride, z_coarse, zc_stride, u, u2, u_fine,
                                                            n_rays, z_out, fine_out, idx_out);
            break;
        case 2:
            sample_fine_kernel<2><<<grid, 256, 0, stream>>>(weights, bins, bins_stride, z_coarse, zc_stride, u, u2, u_fine,
                                                            n_rays, z_out, fine_out, idx_out);
            break;
        case 3:
            sample_fine_kernel<3><<<grid, 256, 0, stream>>>(weights, bins, bins_stride, z_coarse, zc_stride, u, u2, u_fine,
                                                            n_rays, z_out, fine_out, idx_out);
            break;
        default:
            return NRF_ERR_SHAPE;
    }
    return nrf_launch_status();
}
