// Two-term fp16 operand split of the fp16x2 kernels (mlp_fp16x2.hip).  fp32 a ~ a_hi + a_lo / 2048 with
// a_hi = fp16(a) and a_lo = fp16((a - a_hi) * 2048), both round-to-nearest-even: 2 x 11 significand bits, a relative
// representation error <= 2^-22 (1.5e-7 measured, tools/ubench/f16_probe.hip; fp32's own is 2^-24).  The residual
// a - a_hi is exact in fp32; scaling it by 2^11 keeps a_lo a NORMAL fp16 number over the whole normal range of a_hi
// (unscaled it would go subnormal below |a| = 2^-3 and lose bits), so the scheme holds its precision for
// 6.1e-5 <= |a| < 65520 and degrades to an absolute error <= 2^-36 below (the matrix cores honour fp16 subnormals:
// same probe).  |a| >= 65520 is outside the format: a_hi = inf and the output is non-finite.  A product keeps
// hi.hi (main accumulator) and hi.lo + lo.hi (a second accumulator at scale 2^11, folded in as acc += 2^-11 * acc_lo);
// the dropped lo.lo term is 2^-22 of the product.
#pragma once
#include "nrf_common.h"

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr float kLoScale = 2048.f, kLoUnscale = 1.f / 2048.f;

__device__ __forceinline__ unsigned pk_f16(float lo, float hi) {
    const f16x2 v = {(_Float16)lo, (_Float16)hi};  // v_cvt_pk_f16_f32, round to nearest even
    return __builtin_bit_cast(unsigned, v);
}
// a - (the fp16 in the low / high half of `hi`), exact: v_fma_mix_f32 reads the fp16 source in place (no unpack).
// The asm operands are VALU results (the activation, a v_cvt_pk), never MFMA results: the compiler's MFMA -> VALU
// hazard pass does not look inside inline asm (mlp_chain.h: relu16).
__device__ __forceinline__ float f16_residual_lo(float a, unsigned hi) {
    float r;
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(a), "v"(hi));
    return r;
}
__device__ __forceinline__ float f16_residual_hi(float a, unsigned hi) {
    float r;
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(a), "v"(hi));
    return r;
}
// {fp16(ra * 2048), fp16(rb * 2048)}: scale and conversion in one instruction per half (v_fma_mixlo/hi_f16)
__device__ __forceinline__ unsigned pk_f16_scaled(float ra, float rb) {
    unsigned lo;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(lo) : "v"(ra), "v"(kLoScale));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(lo) : "v"(rb), "v"(kLoScale));
    return lo;
}
// a pair of fp32 values -> the packed pairs of its two fp16 terms: 5 VALU instructions
__device__ __forceinline__ void split_pair_f16(float a, float b, unsigned& hi, unsigned& lo) {
    hi = pk_f16(a, b);
    lo = pk_f16_scaled(f16_residual_lo(a, hi), f16_residual_hi(b, hi));
}
// ovf = max(ovf, |a|, |b|) in one instruction: the range guard's running maximum of what was fed to a split
__device__ __forceinline__ void track_max(float& ovf, float a, float b) {
    asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(ovf) : "v"(a), "v"(b));
}
constexpr float kF16Limit = 65520.f;   // |a| >= this rounds to fp16 infinity

// a 32-row block held as 16 accumulator-order registers -> the B fragments [k16-step][term] of its two terms
__device__ __forceinline__ void split_block_f16(const f32x16& v, u32x4 (&out)[2][2]) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned hi, lo;
            split_pair_f16(v[8 * s + 2 * i], v[8 * s + 2 * i + 1], hi, lo);
            out[s][0][i] = hi, out[s][1][i] = lo;
        }
}
__device__ __forceinline__ void split_block_f16(const f32x16& v, u32x4 (&out)[2][2], float& ovf) {
#pragma unroll
    for (int t = 0; t < 16; t += 2) track_max(ovf, v[t], v[t + 1]);
    split_block_f16(v, out);
}
// The backward's power-of-two loss scale (mlp_fp16x2.hip: mlp_dgrad_h2_kernel; mlp_bwd.hip: wgrad_h2p_kernel): S with
// 2^S * max in [2^target, 2^(target+1)) from the bits of a non-negative float maximum; 0 for max = 0, subnormal or not
// finite.  Both kernels derive the chain's scale from the same word with the same formula.
__device__ __forceinline__ int scale_exponent(unsigned max_bits, int target) {
    const int e = int((max_bits >> 23) & 0xffu);
    if (e == 0 || e == 255) return 0;
    const int S = target - (e - 127);
    return S > 100 ? 100 : S < -100 ? -100 : S;
}
__device__ __forceinline__ float exp2i(int S) { return __uint_as_float(unsigned(127 + S) << 23); }   // |S| <= 126
// words of the chain's scale block (256 B of workspace, zeroed per call): max |d raw|, then max |scaled dY| per row set
constexpr int kGmaxLayer0 = 1, kGmaxFeat = 9, kGmaxViews = 10;   // dY_l at 1 + l (l = 0 .. 7)
constexpr int kChainScaleTarget = 4;    // the data-gradient chain: max |d raw| -> [16, 32)
constexpr int kWgradScaleTarget = 11;   // a weight-gradient GEMM's dY operand: its row set's max -> [2048, 4096)
}  // namespace

#define NRF_MFMA_F16(a, b, c) \
    __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, (a)), __builtin_bit_cast(f16x8, (b)), (c), 0, 0, 0)
