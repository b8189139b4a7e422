// Shared pieces of the KiloNeRF kernels (kilo.hip: forward / marching / compositing, kilo_bwd.hip: backward).
#pragma once
#include <cmath>

#include "mlp_chain.h"

namespace {

constexpr int kKiloSteps = 24;                 // float4-per-lane pieces of one cell's weights
constexpr int kKiloSmall = 288;                // biases + heads, padded
constexpr int kKiloCellFloats = kKiloSteps * 256 + kKiloSmall;  // 6432
constexpr int kKOffBias = 0;                   // [4 layers: pts0, pts1, feature, view][h][16]
constexpr int kKOffSigW = 128;                 // [h][16]
constexpr int kKOffRgbW = 160;                 // [c][h][16]
constexpr int kKOffSigB = 256, kKOffRgbB = 257;
constexpr int kJobTiles = 16;                  // tiles of one cell handled per weight load

struct KiloSrc {
    const float* t[12];  // KILO tensor order: w0 b0 w1 b1 wf bf ws bs wv bv wr br, each (cells, ...)
};

// A cell's parameters as one record: the 12 tensors' per-cell slices back to back (6212 floats).
constexpr int kRecOff[13] = {0, 2016, 2048, 3072, 3104, 4128, 4160, 4192, 4193, 6081, 6113, 6209, 6212};
#define REC_OFF(t) (kRecOff[t])   // only ever indexed with literals: folds at compile time, no device-side array
constexpr int kRecFloats = kRecOff[12];
static_assert(kRecOff[1] == 32 * 63 && kRecOff[3] - kRecOff[2] == 32 * 32 && kRecOff[9] - kRecOff[8] == 32 * 59 &&
                  kRecOff[11] - kRecOff[10] == 96,
              "record layout = the 12 tensors' per-cell slices");

// Coalesced copy of one cell's slices of the 12 (cells, ...) tensors into LDS (block of 256 threads); the packers
// then permute out of LDS instead of gathering 4-byte elements from twelve places in HBM.
template <bool ALL = true>   // ALL = false: only the three matrices the transposed packer reads
__device__ __forceinline__ void stage_cell_record(const KiloSrc& s, int cell, float* rec) {
#define STAGE(K)                                                                                        \
    for (int i = threadIdx.x; i < REC_OFF(K + 1) - REC_OFF(K); i += 256)                                \
        rec[REC_OFF(K) + i] = s.t[K][int64_t(cell) * (REC_OFF(K + 1) - REC_OFF(K)) + i];
    STAGE(2) STAGE(4) STAGE(8)
    if (ALL) {
        STAGE(0) STAGE(1) STAGE(3) STAGE(5) STAGE(6) STAGE(7) STAGE(9) STAGE(10) STAGE(11)
    }
#undef STAGE
}

struct Grid {
    float mn[3], mx[3];
    int res[3];
    float inv[3];  // 1 / (max - min) where that edge is a power of two (x / e == x * (1 / e) bit for bit), else 0
    int pinv;      // all three edges are: the voxel / cell formulas multiply instead of dividing
};

// One 32x32 output block of a tiny layer: Y = init + sum_k W . x over NKB k-blocks of 32.
template <int NKB>
__device__ __forceinline__ f32x16 tiny_layer(const float4* w, f32x16 init, const f32x16* x) {
    f32x16 y = init;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 a = w[kb * 4 + q];
            y = NRF_MFMA(a.x, x[kb][4 * q + 0], y);
            y = NRF_MFMA(a.y, x[kb][4 * q + 1], y);
            y = NRF_MFMA(a.z, x[kb][4 * q + 2], y);
            y = NRF_MFMA(a.w, x[kb][4 * q + 3], y);
        }
    return y;
}

__attribute__((unused)) Grid make_grid(const float* aabb, const int* res) {
    Grid g;
    g.pinv = 1;
    for (int a = 0; a < 3; ++a) {
        g.mn[a] = aabb[a], g.mx[a] = aabb[3 + a], g.res[a] = res[a];
        const float e = g.mx[a] - g.mn[a];
        int ex = 0;
        const bool pow2 = e > 0.f && std::frexp(e, &ex) == 0.5f && ex > -100 && ex < 100;
        g.inv[a] = pow2 ? 1.0f / e : 0.f;
        g.pinv = g.pinv && pow2;
    }
    return g;
}

__attribute__((unused)) inline int64_t a256(int64_t n) { return (n + 255) & ~int64_t(255); }


// workspace of nrf_kilo_forward: cell (n) | perm (n) | hist, seg_start, cursor, job_prefix (ncell + 1 each)
struct KiloSortWs {
    int *cell, *perm, *hist, *seg_start, *cursor, *job_prefix;
};
__attribute__((unused)) inline KiloSortWs kilo_sort_ws(void* workspace, int64_t n_points, int num_cells) {
    char* wsb = static_cast<char*>(workspace);
    const int64_t tab = a256((int64_t(num_cells) + 1) * 4);
    KiloSortWs w;
    w.cell = reinterpret_cast<int*>(wsb);
    w.perm = reinterpret_cast<int*>(wsb + a256(n_points * 4));
    char* t = wsb + 2 * a256(n_points * 4);
    w.hist = reinterpret_cast<int*>(t);
    w.seg_start = reinterpret_cast<int*>(t + tab);
    w.cursor = reinterpret_cast<int*>(t + 2 * tab);
    w.job_prefix = reinterpret_cast<int*>(t + 3 * tab);
    return w;
}

// One 64-sample chunk of a packed ray: alpha, exclusive transmittance T (carried across chunks in `carry`) and
// the sigmoid colours.  Shared by the forward and both passes of the backward compositing so that the three
// agree bit for bit on which samples early termination keeps (T >= ert_eps).  Hardware exp / rcp and DPP scans
// (composite.hip has the error budget); call with all 64 lanes active, `in` = lane has a sample.
struct PackedChunk {
    float alpha, T, cr, cg, cb;
};
__device__ __forceinline__ float packed_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
__device__ __forceinline__ PackedChunk packed_chunk(float4 rv, float delta, bool in, float& carry) {
    PackedChunk c;
    c.alpha = in ? 1.f - packed_exp(-fmaxf(rv.w, 0.f) * delta) : 0.f;  // volumn_renderer.py:99-100
    const float p = nrf_wave_scan_mul(1.f - c.alpha);
    c.T = carry * nrf_lane_shr1(1.f, p);
    carry *= nrf_read_lane63(p);
    c.cr = __builtin_amdgcn_rcpf(1.f + packed_exp(-rv.x));
    c.cg = __builtin_amdgcn_rcpf(1.f + packed_exp(-rv.y));
    c.cb = __builtin_amdgcn_rcpf(1.f + packed_exp(-rv.z));
    return c;
}
__device__ __forceinline__ float wave_total(float v) { return nrf_read_lane63(nrf_wave_scan_add(v)); }

}  // namespace

// kilo_bwd.hip, for nrf_kilo_forward_packing / nrf_kilo_pack_weights_cells (kilo.hip): the transposed records of the
// cells with active[cell] != 0 and done[cell] == 0 (either may be NULL) from the 12 parameter tensors (host array of
// device pointers)
void kilo_pack_t_launch(const float* const* tensors, int num_cells, const int* active, const int* done, float* packed_t,
                        hipStream_t stream);
