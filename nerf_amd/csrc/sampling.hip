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

// ---- cross-lane exchange for the register sort: value of lane (lane ^ MASK) ------------------
template <int MASK>
__device__ __forceinline__ float lane_xor(float v) {
    const int x = __float_as_int(v);
    int r;
    if constexpr (MASK == 1) r = __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xf, 0xf, true);         // quad_perm [1,0,3,2]
    else if constexpr (MASK == 2) r = __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xf, 0xf, true);    // quad_perm [2,3,0,1]
    else if constexpr (MASK == 3) r = __builtin_amdgcn_update_dpp(x, x, 0x1B, 0xf, 0xf, true);    // quad_perm [3,2,1,0]
    else if constexpr (MASK == 7) r = __builtin_amdgcn_update_dpp(x, x, 0x141, 0xf, 0xf, true);   // row_half_mirror
    else if constexpr (MASK == 15) r = __builtin_amdgcn_update_dpp(x, x, 0x140, 0xf, 0xf, true);  // row_mirror
    else if constexpr (MASK == 8) r = __builtin_amdgcn_update_dpp(x, x, 0x128, 0xf, 0xf, true);   // row_ror:8
    else if constexpr (MASK == 4 || MASK == 16 || MASK == 31)
        r = __builtin_amdgcn_ds_swizzle(x, 0x1f | (MASK << 10));                                  // bit-mask mode: lane ^ MASK
    else r = __shfl_xor(x, MASK);
    return __int_as_float(r);
}

// One compare-exchange stage of the ascending bitonic network over 256 keys, 4 per lane (key index = 4 * lane + r).
// MASK = lane distance (xor), REV: the partner holds the key in register 3 - r (mirror step).  The lane with the
// smaller index keeps the minimum: med3(a, b, -inf) = min, med3(a, b, +inf) = max, `sel` = -+inf by lane bit.
template <int MASK, bool REV>
__device__ __forceinline__ void sort_stage_lanes(float (&e)[4], float sel) {
    float p[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) p[r] = lane_xor<MASK>(e[REV ? 3 - r : r]);
#pragma unroll
    for (int r = 0; r < 4; ++r) e[r] = __builtin_amdgcn_fmed3f(e[r], p[r], sel);
}
__device__ __forceinline__ void cx(float& a, float& b) {
    const float lo = fminf(a, b), hi = fmaxf(a, b);
    a = lo, b = hi;
}
__device__ __forceinline__ void sort_tail(float (&e)[4]) {  // key distances 2 and 1: inside the lane
    cx(e[0], e[2]), cx(e[1], e[3]);
    cx(e[0], e[1]), cx(e[2], e[3]);
}

// sort(cat(coarse, fine)) (:288-289), values only, so any assignment of the keys to (lane, register) slots will do:
// no staging.  36 compare-exchange stages, 21 of them across lanes (DPP / swizzle), ~230 VALU instructions.
__device__ __forceinline__ void sort256(float (&e)[4], int lane) {
    const float ninf = -__builtin_inff(), pinf = __builtin_inff();
    const float s1 = lane & 1 ? pinf : ninf, s2 = lane & 2 ? pinf : ninf, s4 = lane & 4 ? pinf : ninf;
    const float s8 = lane & 8 ? pinf : ninf, s16 = lane & 16 ? pinf : ninf, s32 = lane & 32 ? pinf : ninf;
    cx(e[0], e[1]), cx(e[2], e[3]);                                   // k = 2
    cx(e[0], e[3]), cx(e[1], e[2]), cx(e[0], e[1]), cx(e[2], e[3]);   // k = 4: mirror, then distance 1
    sort_stage_lanes<1, true>(e, s1), sort_tail(e);                                                  // k = 8
    sort_stage_lanes<3, true>(e, s2), sort_stage_lanes<1, false>(e, s1), sort_tail(e);               // k = 16
    sort_stage_lanes<7, true>(e, s4), sort_stage_lanes<2, false>(e, s2), sort_stage_lanes<1, false>(e, s1), sort_tail(e);
    sort_stage_lanes<15, true>(e, s8), sort_stage_lanes<4, false>(e, s4), sort_stage_lanes<2, false>(e, s2);
    sort_stage_lanes<1, false>(e, s1), sort_tail(e);                                                 // k = 64
    sort_stage_lanes<31, true>(e, s16), sort_stage_lanes<8, false>(e, s8), sort_stage_lanes<4, false>(e, s4);
    sort_stage_lanes<2, false>(e, s2), sort_stage_lanes<1, false>(e, s1), sort_tail(e);              // k = 128
    sort_stage_lanes<63, true>(e, s32), sort_stage_lanes<16, false>(e, s16), sort_stage_lanes<8, false>(e, s8);
    sort_stage_lanes<4, false>(e, s4), sort_stage_lanes<2, false>(e, s2), sort_stage_lanes<1, false>(e, s1), sort_tail(e);
}

__device__ __forceinline__ double shfl_up_f64(double v, int off) {
    const long long b = __double_as_longlong(v);
    const int lo = __shfl_up(int(b), off), hi = __shfl_up(int(b >> 32), off);
    return __longlong_as_double((long long)(unsigned(lo)) | ((long long)hi << 32));
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
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t ray = int64_t(blockIdx.x) * 4 + wv;
    if (ray >= n_rays) return;  // wave-uniform; LDS regions are wave-private, no block barrier used
    float* cdf = s_cdf[wv];
    float* sbin = s_bins[wv];

    const float w = weights[ray * NC + lane];
    sbin[lane] = bins[ray * bins_stride + lane];
    const float zc = z_coarse[ray * zc_stride + lane];
    float uu[NI], uv[NI];
#pragma unroll
    for (int c = 0; c < NI; ++c) uu[c] = u[ray * NIMP + c * 64 + lane], uv[c] = u2[ray * NIMP + c * 64 + lane];

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

    // ---- cdf = [0, cumsum(pdf)] (:375-376).  The reference's prefix k is a SEQUENTIAL fp64 running sum rounded to
    //      fp32.  A 6-step fp64 wave scan adds the same non-negative terms in another order: the two sums differ by
    //      at most ~64 ulp64 (1.4e-14 relative), so they round to the same fp32 unless the scan result lies within
    //      that distance of a rounding boundary (the mid-point to a neighbouring float).  Such rays (and rays with a
    //      negative or NaN term, where the bound does not hold) take the sequential path; probability ~2e-4 per ray.
    double run = double(pdf);
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const double t = shfl_up_f64(run, off);
        if (lane >= off) run = __dadd_rn(run, t);
    }
    float cdf_k = float(run);
    bool risky = !(pdf >= 0.f);
    if (run != 0.0) {
        const unsigned b = __float_as_uint(cdf_k);
        const double fu = double(__uint_as_float(b + 1u)), fd = double(__uint_as_float(b ? b - 1u : b)), f = double(cdf_k);
        const double margin = 1e-13 * run;
        risky = risky || fabs(run - 0.5 * (f + fu)) <= margin || fabs(run - 0.5 * (f + fd)) <= margin;
    }
    if (__any(risky)) {
        double acc = 0.0;
        for (int jj = 0; jj < 64; ++jj) {
            const float pj = __shfl(pdf, jj);
            acc = __dadd_rn(acc, jj <= lane ? double(pj) : 0.0);
        }
        cdf_k = float(acc);
    }
    cdf[lane + 1] = cdf_k;
    if (lane == 0) cdf[0] = 0.f;
    wave_sync();

    // ---- searchsorted(cdf, u, right=True) - 1, clamp, gather bins, lerp (:379-390)
    float fine[NI];
#pragma unroll
    for (int c = 0; c < NI; ++c) {
        const int i = c * 64 + lane;
        int lo = 0, hi = NC + 1;  // upper bound over the 65 cdf entries
#pragma unroll
        for (int it = 0; it < 7; ++it) {
            if (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (cdf[mid] <= uu[c]) lo = mid + 1; else hi = mid;
            }
        }
        int idx = lo - 1;
        idx = idx < 0 ? 0 : (idx > NC - 1 ? NC - 1 : idx);
        const float b_lo = sbin[idx];
        const float b_hi = sbin[idx + 1 > NC - 1 ? NC - 1 : idx + 1];
        fine[c] = __fadd_rn(b_lo, __fmul_rn(__fsub_rn(b_hi, b_lo), uv[c]));
        if (idx_out) idx_out[ray * NIMP + i] = idx;
    }
    if (u_fine) {  // add_perturbation on the (unsorted) fine samples, render_fine :283-284; sample i = c * 64 + lane
        float pf[NI];
#pragma unroll
        for (int c = 0; c < NI; ++c) {
            // neighbours in sample order; shuffles stay outside divergent code
            const float last_of_prev = c > 0 ? __shfl(fine[c > 0 ? c - 1 : 0], 63) : fine[c];
            const float first_of_next = c + 1 < NI ? __shfl(fine[c + 1 < NI ? c + 1 : c], 0) : fine[c];
            float prev = __shfl_up(fine[c], 1), next = __shfl_down(fine[c], 1);
            if (lane == 0) prev = last_of_prev;
            if (lane == 63) next = first_of_next;
            pf[c] = perturb_one(prev, fine[c], next, c == 0 && lane == 0, c == NI - 1 && lane == 63,
                                u_fine[ray * NIMP + c * 64 + lane]);
        }
#pragma unroll
        for (int c = 0; c < NI; ++c) fine[c] = pf[c];
    }
    if (fine_out) {
#pragma unroll
        for (int c = 0; c < NI; ++c) fine_out[ray * NIMP + c * 64 + lane] = fine[c];
    }

    // ---- sort(cat(coarse, fine)) (:288-289) in registers; key 4 * lane + r comes out in ascending order
    float e[4];
    e[0] = zc;
#pragma unroll
    for (int c = 0; c < 3; ++c) e[1 + c] = c < NI ? fine[c < NI ? c : 0] : __builtin_inff();
    sort256(e, lane);
    if (4 * lane < NOUT) {
        float* dst = z_out + ray * NOUT + 4 * lane;
        if ((reinterpret_cast<uintptr_t>(z_out) & 15u) == 0) *reinterpret_cast<float4*>(dst) = make_float4(e[0], e[1], e[2], e[3]);
        else dst[0] = e[0], dst[1] = e[1], dst[2] = e[2], dst[3] = e[3];
    }
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
