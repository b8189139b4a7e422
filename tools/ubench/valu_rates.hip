// Issue rate of a few VALU instructions the operand splits use, relative to v_fma_f32: 8 independent chains per wave, one
// wave per SIMD (256 workgroups of 256), 64 instructions per loop iteration.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rates.hip -o tools/ubench/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
template <int WHICH>
__global__ __launch_bounds__(256, 1) void k(float* sink, int iters) {
    float x[8];
    unsigned u[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x + i, u[i] = threadIdx.x * 977u + i;
    const float m = 1.0001f;
    unsigned sel = 0x0000BF80u;
    asm volatile("" : "+s"(sel));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (WHICH == 0) {
#define S(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(m));
                REP8(S)
#undef S
            } else if (WHICH == 1) {
#define S(i) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(x[i]) : "s"(sel), "v"(u[i]));
                REP8(S)
#undef S
            } else if (WHICH == 2) {
#define S(i) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(x[i]), "v"(x[(i + 1) & 7]));
                REP8(S)
#undef S
            } else if (WHICH == 3) {
#define S(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*reinterpret_cast<double*>(&x[i & 6])) : "v"(*reinterpret_cast<double*>(&x[(i + 2) & 6])));
                REP8(S)
#undef S
            } else if (WHICH == 4) {
#define S(i) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(u[i]) : "v"(u[(i + 1) & 7]));
                REP8(S)
#undef S
            } else {
#define S(i) asm volatile("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(x[i]) : "v"(x[(i + 1) & 7]), "v"(u[i]));
                REP8(S)
#undef S
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += x[i] + u[i];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int WHICH>
float run(float* sink, int iters) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    k<WHICH><<<256, 256>>>(sink, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<WHICH><<<256, 256>>>(sink, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float* sink;
    if (hipMalloc(&sink, 256 * 256 * 4) != hipSuccess) return 1;
    const int iters = 20000;   // 1.28 M instructions per wave
    const float f = run<0>(sink, iters);
    printf("v_fma_f32          %.3f ms  1.00\n", f);
    const float d = run<1>(sink, iters);
    printf("v_dot2c_f32_bf16   %.3f ms  %.2f\n", d, d / f);
    const float c = run<2>(sink, iters);
    printf("v_cvt_pk_bf16_f32  %.3f ms  %.2f\n", c, c / f);
    const float p = run<3>(sink, iters);
    printf("v_pk_add_f32       %.3f ms  %.2f\n", p, p / f);
    const float l = run<4>(sink, iters);
    printf("v_lshlrev_b32      %.3f ms  %.2f\n", l, l / f);
    const float x = run<5>(sink, iters);
    printf("v_fma_mix_f32      %.3f ms  %.2f\n", x, x / f);
    return 0;
}
