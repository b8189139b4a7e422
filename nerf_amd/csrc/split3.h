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
}  // namespace

#define NRF_MFMA_X3(a, b, c) \
    __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, (a)), __builtin_bit_cast(bf16x8, (b)), (c), 0, 0, 0)
