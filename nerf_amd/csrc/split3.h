// Three-term bf16 operand split shared by the bf16x3 kernels (mlp_bf16x3.hip: forward / data-gradient chains,
// mlp_bwd.hip: weight-gradient GEMMs).  fp32 a = a0 + a1 + a2 exactly (a0 = bf16(a), a1 = bf16(a - a0),
// a2 = bf16(a - a0 - a1): 3 x 8 significand bits, both subtractions exact in fp32); a product keeps the six terms
// a_i b_j with i + j <= 2, added small terms first into one fp32 accumulator block.
#pragma once
#include "nrf_common.h"

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

namespace {

__device__ __forceinline__ unsigned pk_bf16(float lo, float hi) {
    const bf16x2 v = {(__bf16)lo, (__bf16)hi};  // v_cvt_pk_bf16_f32, round to nearest even
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float bf_lo(unsigned r) { return __uint_as_float(r << 16); }
__device__ __forceinline__ float bf_hi(unsigned r) { return __uint_as_float(r & 0xffff0000u); }

// a pair of fp32 values -> the packed pair of each of its three bf16 terms (residuals are exact in fp32)
__device__ __forceinline__ void split_pair(float a, float b, unsigned& p0, unsigned& p1, unsigned& p2) {
    p0 = pk_bf16(a, b);
    const float ra = __fsub_rn(a, bf_lo(p0)), rb = __fsub_rn(b, bf_hi(p0));
    p1 = pk_bf16(ra, rb);
    p2 = pk_bf16(__fsub_rn(ra, bf_lo(p1)), __fsub_rn(rb, bf_hi(p1)));
}
// the same split with the two residual subtractions as one v_pk_add_f32 each (9 instead of 11 instructions per pair; same
// bits: the packed op is two IEEE fp32 subtractions) -- for the kernels whose splits are not hand-sliced into MFMA gaps
typedef float f32x2_s3 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_pair_pk(float a, float b, unsigned& p0, unsigned& p1, unsigned& p2) {
    p0 = pk_bf16(a, b);
    const f32x2_s3 r1 = (f32x2_s3){a, b} - (f32x2_s3){bf_lo(p0), bf_hi(p0)};
    p1 = pk_bf16(r1[0], r1[1]);
    const f32x2_s3 r2 = r1 - (f32x2_s3){bf_lo(p1), bf_hi(p1)};
    p2 = pk_bf16(r2[0], r2[1]);
}
// The residuals by v_dot2c_f32_bf16: acc += lo(p) * lo(sel) + hi(p) * hi(sel) with sel = {-1, 0} or {0, -1} subtracts one
// half of the packed word from its fp32 source IN ONE instruction (shift / mask + subtract are two; per pair 7 instead of
// 11).  Bit-identical to split_pair on every finite fp32 value, denormals included (tools/ubench/bf16_split_dot2.hip: a
// stride-7 sweep of all 2^32 patterns).  The selectors must be REGISTERS the compiler cannot see through (SplitSel::make,
// once per kernel): folded into the inline constant `-1.0` the instruction reads fp32 -1.0 = {0, -1} for both.
struct SplitSel {
    unsigned lo, hi;   // {-1, 0}, {0, -1} as bf16 pairs
    static __device__ __forceinline__ SplitSel make() {
        SplitSel s{0x0000BF80u, 0xBF800000u};
        asm volatile("" : "+s"(s.lo), "+s"(s.hi));   // scalar registers: the instruction takes one as src0
        return s;
    }
};
__device__ __forceinline__ float minus_bf(float a, unsigned p, unsigned sel) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, p), __builtin_bit_cast(bf16x2, sel), a, false);
}
__device__ __forceinline__ void split_pair_dot2(const SplitSel& k, float a, float b, unsigned& p0, unsigned& p1, unsigned& p2) {
    p0 = pk_bf16(a, b);
    const float ra = minus_bf(a, p0, k.lo), rb = minus_bf(b, p0, k.hi);
    p1 = pk_bf16(ra, rb);
    p2 = pk_bf16(minus_bf(ra, p1, k.lo), minus_bf(rb, p1, k.hi));
}
__device__ __forceinline__ void split_block_x3(const SplitSel& k, const f32x16& v, u32x4 (&out)[2][3]) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned p0, p1, p2;
            split_pair_dot2(k, v[8 * s + 2 * i], v[8 * s + 2 * i + 1], p0, p1, p2);
            out[s][0][i] = p0, out[s][1][i] = p1, out[s][2][i] = p2;
        }
}
// a 32-row block held as 16 accumulator-order registers -> the B fragments [k16-step][term] of its three terms
__device__ __forceinline__ void split_block_x3(const f32x16& v, u32x4 (&out)[2][3]) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned p0, p1, p2;
            split_pair_pk(v[8 * s + 2 * i], v[8 * s + 2 * i + 1], p0, p1, p2);
            out[s][0][i] = p0, out[s][1][i] = p1, out[s][2][i] = p2;
        }
}
}  // namespace

#define NRF_MFMA_X3(a, b, c) \
    __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, (a)), __builtin_bit_cast(bf16x8, (b)), (c), 0, 0, 0)
