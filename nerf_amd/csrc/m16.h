// The v_mfma_f32_16x16x32 form of the register-chained MLP kernels (mlp_bf16x3.hip: mlp_bf16x3_m16_kernel; mlp_bf16.hip).
// Same chain, same stream positions, same VALU work as the 32x32x16 form; the matrix shape is the change: under these
// kernels' load the chip holds its clock by the energy a FLOP costs, and this shape costs less at the same cycles per
// FLOP (MI355X_MICROARCH.md, DVFS give-back item 7; tools/ubench/mfma_shape_clock.hip).  A step at stream position
// (ob, kb, s2) is rows 32 ob + 16 s2 .. + 15 against ALL 32 features of k-block kb (32x32 form: 32 rows against the 16
// features of half s2), multiplied into two 16-point accumulators.  Layout: nrf_common.h, m16_slot.
#pragma once
#include "mlp_chain.h"

namespace {

struct Blk16 {   // kept as ONE 16-register value (four separate quarters: the allocator parked finished quarters in AGPRs / scratch)
    f32x16 v;
    __device__ __forceinline__ float operator[](int t) const { return v[t]; }
};
// the two 16 x 16 accumulators (point halves nb = 0, 1) of row half MH of a block
template <int MH>
__device__ __forceinline__ void blk16_get(const f32x16& y, f32x4& a, f32x4& b) {
    a = __builtin_shufflevector(y, y, 4 * MH, 4 * MH + 1, 4 * MH + 2, 4 * MH + 3);
    b = __builtin_shufflevector(y, y, 8 + 4 * MH, 8 + 4 * MH + 1, 8 + 4 * MH + 2, 8 + 4 * MH + 3);
}
template <int MH>
__device__ __forceinline__ void blk16_put(f32x16& y, const f32x4& a, const f32x4& b) {
#pragma unroll
    for (int t = 0; t < 4; ++t) y[4 * MH + t] = a[t], y[8 + 4 * MH + t] = b[t];
}
template <int MH>
__device__ __forceinline__ void blk16_get(const Blk16& y, f32x4& a, f32x4& b) { blk16_get<MH>(y.v, a, b); }
template <int MH>
__device__ __forceinline__ void blk16_put(Blk16& y, const f32x4& a, const f32x4& b) { blk16_put<MH>(y.v, a, b); }

template <class T> struct is_blk16 { static constexpr bool value = false; };
template <> struct is_blk16<Blk16> { static constexpr bool value = true; };
// where register t of a block's lane sits inside the block's saved rows ([row][32 points] floats), relative to the lane's
// base -- 32x32 form: lane (j, h), base 4 h * 32 + j, row acc_row(t, 0); 16x16x32 form: lane (c, g), base 4 g * 32 + c,
// row 16 mh + r and point half nb of t = 8 nb + 4 mh + r
template <bool M16>
__host__ __device__ constexpr int saved_row_off(int t) {
    return M16 ? (16 * ((t >> 2) & 1) + (t & 3)) * 32 + 16 * (t >> 3) : ((t & 3) + 8 * (t >> 2)) * 32;
}

// the small fp32 block of the 16x16x32 stream, in floats from its start
constexpr int kS16AlphaW = 9 * 512 + 256;         // biases [layer 0..9][block][g][16] before it (views_linears: 4 blocks)
constexpr int kS16AlphaB = kS16AlphaW + 256;      // alpha weights [block][g][8]
constexpr int kS16RgbW = kS16AlphaB + 16;         // rgb weights [c][block 0..3][g][8]
constexpr int kS16RgbB = kS16RgbW + 3 * 128;
constexpr int kS16Floats = kS16RgbB + 16;         // 5536
struct PackSmall16 {
    const float* bias[10];  // pts0..7, feature, views
    const float* alpha_w;
    const float* alpha_b;
    const float* rgb_w;
    const float* rgb_b;
    float* dst;
};
__attribute__((unused)) __global__ void pack_small16_kernel(PackSmall16 p) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= kS16Floats) return;
    float v = 0.f;
    if (idx < kS16AlphaW) {
        const int layer = idx >> 9, t = idx & 15, g = (idx >> 4) & 3, sb = (idx >> 6) & 7;
        v = p.bias[layer][32 * sb + m16_slot(g, t & 7)];
    } else if (idx < kS16AlphaB) {
        const int i = idx - kS16AlphaW;
        v = p.alpha_w[32 * (i >> 5) + m16_slot((i >> 3) & 3, i & 7)];
    } else if (idx < kS16RgbW) {
        v = idx == kS16AlphaB ? p.alpha_b[0] : 0.f;
    } else if (idx < kS16RgbB) {
        const int i = idx - kS16RgbW;
        v = p.rgb_w[(i >> 7) * 128 + 32 * ((i >> 5) & 3) + m16_slot((i >> 3) & 3, i & 7)];
    } else {
        v = idx - kS16RgbB < 3 ? p.rgb_b[idx - kS16RgbB] : 0.f;
    }
    p.dst[idx] = v;
}

// An encoding block made in the 32x32 form (register t of lane (j, h): slot acc_row(t, h) of point j) handed over to the
// 16x16x32 form: register 8 nb + 4 mh + r of lane (c, g) is register 4 (2 mh + (g >> 1)) + r of lane 32 (g & 1) + 16 nb + c.
// Per register pair (A = 4 (2 mh) + r, B = A + 4): rows of 16 lanes [A0 A1 A2 A3], [B0 B1 B2 B3] -> v_permlane32_swap ->
// [A0 A1 B0 B1], [A2 A3 B2 B3] -> v_permlane16_swap -> [A0 A2 B0 B2] (nb = 0), [A1 A3 B1 B3] (nb = 1).  Two VALU
// instructions per pair, no LDS.
__attribute__((unused)) __device__ __forceinline__ f32x16 to_m16(const f32x16& e) {
    f32x16 r;
#pragma unroll
    for (int mh = 0; mh < 2; ++mh)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const auto s32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(e[8 * mh + rr]), __float_as_uint(e[8 * mh + 4 + rr]), false, false);
            const auto s16 = __builtin_amdgcn_permlane16_swap(s32[0], s32[1], false, false);
            r[4 * mh + rr] = __uint_as_float(s16[0]);
            r[8 + 4 * mh + rr] = __uint_as_float(s16[1]);
        }
    return r;
}
// A block's 16 relu-sign bits per lane travel in memory in the 32x32 form's order (bit t of lane (j, h): slot acc_row(t, h)
// of point j), whichever form wrote or reads them: the bit-level inverse of to_m16, nibble by nibble (bits 4 mh .. + 3
// and 8 + 4 mh .. + 3 of the 16x16x32 word are one register quartet of each point half).
__attribute__((unused)) __device__ __forceinline__ unsigned m16_mask_to_old(unsigned n) {
    unsigned o = 0;
#pragma unroll
    for (int mh = 0; mh < 2; ++mh) {
        const auto s16 = __builtin_amdgcn_permlane16_swap((n >> (4 * mh)) & 0xfu, (n >> (8 + 4 * mh)) & 0xfu, false, false);
        const auto s32 = __builtin_amdgcn_permlane32_swap(s16[0], s16[1], false, false);
        o |= (s32[0] << (8 * mh)) | (s32[1] << (8 * mh + 4));
    }
    return o;
}
__attribute__((unused)) __device__ __forceinline__ unsigned m16_mask_from_old(unsigned o) {
    unsigned n = 0;
#pragma unroll
    for (int mh = 0; mh < 2; ++mh) {
        const auto s32 = __builtin_amdgcn_permlane32_swap((o >> (8 * mh)) & 0xfu, (o >> (8 * mh + 4)) & 0xfu, false, false);
        const auto s16 = __builtin_amdgcn_permlane16_swap(s32[0], s32[1], false, false);
        n |= (s16[0] << (4 * mh)) | (s16[1] << (8 + 4 * mh));
    }
    return n;
}
// Head sums of the 16x16x32 form: a lane holds partial sums of TWO points (v0: point c, v1: point 16 + c) over its group's
// features.  Returns, in even groups, the sum of v0 over the four groups and in odd groups that of v1 (lane l < 32 then
// holds point l's): the first exchange hands the neighbour group the half it keeps.
__device__ __forceinline__ float add_groups(float v0, float v1, bool odd) {
    const float mine = (odd ? v1 : v0) + __shfl_xor(odd ? v0 : v1, 16);
    return mine + __shfl_xor(mine, 32);
}

}  // namespace
