// Probe for the fp16 two-term split (mlp_fp16x2.hip): does v_mfma_f32_32x32x16_f16 honour fp16 subnormal operands, and
// do v_cvt_pk_f16_f32 / v_fma_mix* round to nearest even?  Prints what the hardware does; no pass/fail.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void mfma_k(float av, float bv, float* out) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) a[i] = (_Float16)av, b[i] = (_Float16)bv;
    f32x16 z = {0};
    f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, z, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = acc[0];
}
__global__ void split_k(const float* in, unsigned* out, int n) {
    int i = threadIdx.x + blockIdx.x * blockDim.x;
    if (2 * i + 1 >= n) return;
    float a = in[2 * i], b = in[2 * i + 1];
    h2 v = {(_Float16)a, (_Float16)b};
    unsigned hi = __builtin_bit_cast(unsigned, v);
    float ra, rb;
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(ra) : "v"(a), "v"(hi));
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(rb) : "v"(b), "v"(hi));
    unsigned lo;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(lo) : "v"(ra), "v"(2048.f));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(lo) : "v"(rb), "v"(2048.f));
    out[2 * i] = hi, out[2 * i + 1] = lo;
}
static float h2f(unsigned short h) { _Float16 x; __builtin_memcpy(&x, &h, 2); return (float)x; }
int main() {
    float* d; hipMalloc(&d, 64);
    const float tests[][2] = {{1.0f, 1.0f}, {9.5367431640625e-07f /* 2^-20 */, 1.0f}, {3.0517578125e-05f /* 2^-15 */, 1.0f},
                              {5.9604644775390625e-08f /* 2^-24 */, 1.0f}, {9.5367431640625e-07f, 9.5367431640625e-07f}, {6.103515625e-05f, 1.0f}};
    for (auto& t : tests) {
        mfma_k<<<1, 64>>>(t[0], t[1], d);
        float r; hipMemcpy(&r, d, 4, hipMemcpyDeviceToHost);
        printf("mfma a=%g b=%g: acc=%g expected(16 terms)=%g\n", t[0], t[1], r, 16.0 * double(t[0]) * double(t[1]));
    }
    const int n = 1 << 16;
    float* hin = new float[n]; unsigned* hout = new unsigned[n];
    unsigned s = 12345u;
    for (int i = 0; i < n; ++i) {
        s = s * 1664525u + 1013904223u;
        const float u = (s >> 8) * (1.0f / 16777216.f) * 2.f - 1.f;
        s = s * 1664525u + 1013904223u;
        const int e = int(s >> 27) - 24;   // 2^-24 .. 2^7
        hin[i] = ldexpf(u, e);
    }
    float* din; unsigned* dout; hipMalloc(&din, n * 4); hipMalloc(&dout, n * 4);
    hipMemcpy(din, hin, n * 4, hipMemcpyHostToDevice);
    split_k<<<n / 2 / 256, 256>>>(din, dout, n);
    hipMemcpy(hout, dout, n * 4, hipMemcpyDeviceToHost);
    int bad = 0; double worst = 0;
    for (int p = 0; p < n / 2; ++p)
        for (int k = 0; k < 2; ++k) {
            const float a = hin[2 * p + k];
            const unsigned short hb = (hout[2 * p] >> (16 * k)) & 0xffff, lb = (hout[2 * p + 1] >> (16 * k)) & 0xffff;
            const _Float16 he = (_Float16)a;   // host: round to nearest even
            const _Float16 le = (_Float16)((a - (float)he) * 2048.f);
            unsigned short heb, leb; __builtin_memcpy(&heb, &he, 2); __builtin_memcpy(&leb, &le, 2);
            if (heb != hb || leb != lb) { if (bad < 5) printf("mismatch a=%g: hi %04x vs %04x lo %04x vs %04x\n", a, hb, heb, lb, leb); ++bad; }
            const double rec = double(h2f(hb)) + double(h2f(lb)) / 2048.0;
            const double rel = fabs(rec - double(a)) / fabs(double(a));
            if (fabs(a) > 1e-4 && rel > worst) worst = rel;
        }
    printf("split: %d mismatches vs host RNE of %d; worst relative reconstruction error for |a| > 1e-4: %.3g (2^-22 = %.3g)\n", bad, n, worst, ldexp(1.0, -22));
    return 0;
}
