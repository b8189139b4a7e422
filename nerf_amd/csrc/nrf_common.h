// Shared definitions of the HIP sources behind include/nerf_amd.h (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/nerf_amd.h"

#define NRF_EXPORT extern "C" __attribute__((visibility("default")))

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// hipGetLastError() is sticky per host thread and also reports errors left behind by the caller's
// own HIP calls (torch leaves benign ones, e.g. hipErrorNotReady from event queries), so every
// entry point clears it before its first launch and reads it after its last.
static inline void nrf_launch_begin() { (void)hipGetLastError(); }
static inline int nrf_launch_status() {
    return hipGetLastError() == hipSuccess ? NRF_OK : NRF_ERR_LAUNCH;
}

static inline bool nrf_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ------------------------------------------------------------------------------------
// Wave64 scans on DPP (data-parallel primitives: the cross-lane operand is part of the VALU instruction, so a
// scan step is one v_mul / v_add instead of ds_bpermute + s_waitcnt + select + op).  Lanes whose source lies
// outside the row / bank keep `identity`.  Steps: Kogge-Stone inside each row of 16 (row_shr 1, 2, 4, 8), then
// lane 15 of rows 0 and 2 into rows 1 and 3 (row_bcast:15), then lane 31 into rows 2 and 3 (row_bcast:31).
// All 64 lanes must be active.
// ------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float nrf_dpp(float identity, float v) {
    return __int_as_float(
        __builtin_amdgcn_update_dpp(__float_as_int(identity), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float nrf_wave_scan_mul(float p) {  // inclusive product over lanes 0..lane
    p *= nrf_dpp<0x111>(1.f, p);
    p *= nrf_dpp<0x112>(1.f, p);
    p *= nrf_dpp<0x114>(1.f, p);
    p *= nrf_dpp<0x118>(1.f, p);
    p *= nrf_dpp<0x142, 0xa>(1.f, p);
    p *= nrf_dpp<0x143, 0xc>(1.f, p);
    return p;
}
__device__ __forceinline__ float nrf_wave_scan_add(float p) {  // inclusive sum; lane 63 holds the wave total
    p += nrf_dpp<0x111>(0.f, p);
    p += nrf_dpp<0x112>(0.f, p);
    p += nrf_dpp<0x114>(0.f, p);
    p += nrf_dpp<0x118>(0.f, p);
    p += nrf_dpp<0x142, 0xa>(0.f, p);
    p += nrf_dpp<0x143, 0xc>(0.f, p);
    return p;
}
__device__ __forceinline__ float nrf_lane_shr1(float identity, float v) { return nrf_dpp<0x138>(identity, v); }  // wave_shr:1
// v + (v of the lane 32 away): the two lane halves of a wavefront added, in every lane, by ONE v_permlane32_swap and an add
// (__shfl_xor(v, 32) is a ds_bpermute: a trip through the LDS crossbar and a wait).  {lo, hi} + {hi, lo}: the same sum in both halves.
__device__ __forceinline__ float nrf_add_halves(float v) {
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
}
__device__ __forceinline__ float nrf_read_lane63(float v) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// ------------------------------------------------------------------------------------
// Packed ("streaming") layout of one 8x256 MLP, in floats.  See mlp_fp32.hip.
//
// The A-operand stream: one *step* = 256 floats = 1 KiB = one float4 per lane = the weight
// operands of 4 consecutive v_mfma_f32_32x32x2_f32.  Layers are contiguous so a wavefront
// reads the whole network as one linear stream.
// ------------------------------------------------------------------------------------
namespace nrf {

constexpr int kStepFloats = 256;
constexpr int kStepsL0 = 8 * 2 * 4;      // 8 out blocks x 2 PE k-blocks x 4
constexpr int kStepsHidden = 8 * 8 * 4;  // 256 -> 256
constexpr int kStepsSkip = 8 * 10 * 4;   // (64 PE + 256) -> 256
constexpr int kStepsViews = 4 * 9 * 4;   // (256 + 32 dir PE) -> 128
constexpr int kStreamSteps = kStepsL0 + 7 * kStepsHidden + kStepsSkip + kStepsViews;  // 2320
constexpr int kPrefetch = 8;             // steps of zero padding after each packed stream (the v1 register ring's depth;
                                         // today only a pad the LDS-DMA prefetch may run into past the last layer)

// stream offsets (in steps) of each MFMA layer: pts0..7, feature, views
constexpr int kLayerStep[10] = {
    0,
    kStepsL0,
    kStepsL0 + 1 * kStepsHidden,
    kStepsL0 + 2 * kStepsHidden,
    kStepsL0 + 3 * kStepsHidden,
    kStepsL0 + 4 * kStepsHidden,                 // pts5 (skip)
    kStepsL0 + 4 * kStepsHidden + kStepsSkip,    // pts6
    kStepsL0 + 5 * kStepsHidden + kStepsSkip,    // pts7
    kStepsL0 + 6 * kStepsHidden + kStepsSkip,    // feature
    kStepsL0 + 7 * kStepsHidden + kStepsSkip,    // views
};

// after the stream (+ kPrefetch steps of zero padding so the ring may run past the end):
constexpr int64_t kOffBias = int64_t(kStreamSteps + kPrefetch) * kStepFloats;  // 9 x 256 + 128, order [ob][h][r]
constexpr int64_t kOffAlphaW = kOffBias + 9 * 256 + 128;                       // 256, order [kb][h][t]
constexpr int64_t kOffAlphaB = kOffAlphaW + 256;                               // 16 (1 used)
constexpr int64_t kOffRgbW = kOffAlphaB + 16;                                  // 3 x 128, order [c][kb][h][t]
constexpr int64_t kOffRgbB = kOffRgbW + 3 * 128;                               // 16 (3 used)
constexpr int64_t kPackedFloats = kOffRgbB + 16;
constexpr int kSmallFloats = int(kPackedFloats - kOffBias);

// backward (dgrad) stream: views^T 8x4x4 steps, then feature^T and pts7^T..pts1^T at 8x8x4 each
constexpr int kBwdSteps = 8 * 4 * 4 + 8 * kStepsHidden;  // 2176
constexpr int64_t kBwdPackedFloats = int64_t(kBwdSteps + kPrefetch) * kStepFloats;

// row of a 32x32 MFMA accumulator held in register t of lane-half h
__host__ __device__ constexpr int acc_row(int t, int h) { return (t & 3) + 8 * (t >> 2) + 4 * h; }

// 16x16x32 form (m16.h): which of a 32-feature block's slots sits at element jj (0..7) of lane group g (0..3) of a B / A
// fragment.  A wave holds a block of a layer's output as four 16 x 16 accumulators (two 16-row halves mh x two 16-point
// halves nb); register 8 nb + 4 mh + r of lane (c, g) is feature 16 mh + 4 g + r of point 16 nb + c, and the eight
// registers of one nb are, in order, the lane's eight k-values of the next layer's B fragment.
__host__ __device__ constexpr int m16_slot(int g, int jj) { return 16 * (jj >> 2) + 4 * g + (jj & 3); }

// Kernel PE slot n of lane-half h -> column of the reference encoding
// cat([x, sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(2^(L-1) x)]) (3 wide each), or -1.
// Half 0 holds the sin terms, half 1 the cos terms; slots 3L, 3L+1 hold the raw input.
__host__ __device__ inline int pe_col(int n, int h, int L) {
    if (n < 3 * L) return 3 + 6 * (n / 3) + 3 * h + (n % 3);
    if (n == 3 * L) return h;               // x | y
    if (n == 3 * L + 1) return h ? -1 : 2;  // z | pad
    return -1;
}


}  // namespace nrf

// mlp_bf16x3.hip, for nrf_mlp_backward_bf16x3 (mlp_bwd.hip): the data-gradient chain on the three-term bf16 kernel
int launch_dgrad_x3(const float* packed, const void* packed_t, const float* grad_raw, const unsigned* masks, int64_t n_points,
                    float* dys, hipStream_t stream);
// mlp_fp16x2.hip, for nrf_mlp_backward_fp16x2: the same chain on the two-term fp16 kernel at a device-chosen loss scale
int launch_dgrad_h2(const float* packed, const void* packed_t, const float* grad_raw, const unsigned* masks, int64_t n_points,
                    float* dys, unsigned* gmax, hipStream_t stream);
