// Is the three-term bf16 split with its residuals taken by v_dot2c_f32_bf16 (acc += lo(p) * -1 + hi(p) * 0: one instruction
// per element and level instead of shift / mask + subtract) bit-identical to the shift / mask / subtract form (split3.h)?
// Every finite fp32 bit pattern of a stride-7 sweep (613 M values) + the patterns around each exponent boundary.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench/bf16_split_dot2.hip -o tools/ubench/bf16_split_dot2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pk_bf16(float lo, float hi) {
    const bf16x2 v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float bf_lo(unsigned r) { return __uint_as_float(r << 16); }
__device__ __forceinline__ float bf_hi(unsigned r) { return __uint_as_float(r & 0xffff0000u); }
__device__ __forceinline__ void split_ref(float a, float b, unsigned& p0, unsigned& p1, unsigned& p2) {
    p0 = pk_bf16(a, b);
    const float ra = __fsub_rn(a, bf_lo(p0)), rb = __fsub_rn(b, bf_hi(p0));
    p1 = pk_bf16(ra, rb);
    p2 = pk_bf16(__fsub_rn(ra, bf_lo(p1)), __fsub_rn(rb, bf_hi(p1)));
}
// the selectors {-1, 0} and {0, -1} as registers the compiler cannot fold: folded into an inline constant (`-1.0`) the
// instruction takes the fp32 pattern 0xBF800000, i.e. {0, -1}, for BOTH (seen in the first version of this probe)
__device__ __forceinline__ unsigned opaque(unsigned v) {
    asm volatile("" : "+v"(v));
    return v;
}
__device__ __forceinline__ float minus_lo(float a, unsigned p) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, p), __builtin_bit_cast(bf16x2, opaque(0x0000BF80u)), a, false);
}
__device__ __forceinline__ float minus_hi(float a, unsigned p) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, p), __builtin_bit_cast(bf16x2, opaque(0xBF800000u)), a, false);
}
__device__ __forceinline__ void split_dot2(float a, float b, unsigned& p0, unsigned& p1, unsigned& p2) {
    p0 = pk_bf16(a, b);
    const float ra = minus_lo(a, p0), rb = minus_hi(b, p0);
    p1 = pk_bf16(ra, rb);
    p2 = pk_bf16(minus_lo(ra, p1), minus_hi(rb, p1));
}

__global__ void k(unsigned long long* out, unsigned stride) {
    // out[0]: mismatching pairs, out[1]: of them with both |a|, |b| >= 2^-100, out[2]: first mismatching a's bits
    unsigned long long bad = 0, bad_normal = 0;
    const uint64_t n = (uint64_t(1) << 32) / stride;
    for (uint64_t i = blockIdx.x * uint64_t(blockDim.x) + threadIdx.x; i < n; i += uint64_t(gridDim.x) * blockDim.x) {
        const unsigned ba = unsigned(i * stride), bb = ba * 2654435761u + 12345u;
        const float a = __uint_as_float(ba), b = __uint_as_float(bb);
        if (!(fabsf(a) < 3.0e38f) || !(fabsf(b) < 3.0e38f)) continue;
        unsigned r0, r1, r2, d0, d1, d2;
        split_ref(a, b, r0, r1, r2);
        split_dot2(a, b, d0, d1, d2);
        if (r0 != d0 || r1 != d1 || r2 != d2) {
            ++bad;
            if (fabsf(a) >= 7.9e-31f && fabsf(b) >= 7.9e-31f) {
                ++bad_normal;
                atomicMin(out + 2, (unsigned long long)ba);
            }
        }
    }
    atomicAdd(out, bad);
    atomicAdd(out + 1, bad_normal);
}

int main() {
    unsigned long long* out;
    hipMalloc(&out, 24);
    unsigned long long h[3] = {0, 0, ~0ull};
    hipMemcpy(out, h, 24, hipMemcpyHostToDevice);
    k<<<2048, 256>>>(out, 7);
    k<<<2048, 256>>>(out, 65521);   // a second, coarser sweep with other partners
    hipMemcpy(h, out, 24, hipMemcpyDeviceToHost);
    printf("pairs whose split differs: %llu; of them with |a|, |b| >= 2^-100: %llu (first a: 0x%08llx)\n", h[0], h[1], h[2]);
    return 0;
}
