// Two waves on one SIMD, one issuing a DEPENDENT chain of v_mfma_f32_32x32x16_bf16 (one accumulator), the other a chain of
// plain VALU instructions: does the SIMD overlap them (time = max) or add them up (time = sum)?  512 threads per block,
// one block per CU: waves 0..3 are the MFMA waves, 4..7 the VALU waves (waves go to SIMDs round-robin); modes: both,
// MFMA only, VALU only, and the VALU work as independent instructions (4 chains) instead of one chain.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/two_wave_overlap.hip -o tools/ubench/two_wave_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int CHAINS>   // MODE bit 0: MFMA waves work, bit 1: VALU waves work
__global__ __launch_bounds__(512, 1) void k(float* sink, int iters) {
    const int wave = threadIdx.x >> 6;
    float out = 0.f;
    if (wave < 4) {
        if (MODE & 1) {
            f32x16 c = {};
            u32x4 a = {0x3c003c00u + threadIdx.x, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}, b = a;
            for (int i = 0; i < iters; ++i)
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\t"
                             "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\t"
                             "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\t"
                             "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0"
                             : "+v"(c) : "v"(a), "v"(b));
            out = c[0];
        }
    } else if (MODE & 2) {
        float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
        const float m = 1.0001f, ad = 0.5f;
        for (int i = 0; i < iters; ++i) {   // 56 VALU per 8 MFMAs = 7 per MFMA slot (28 of an MFMA's 32 cycles)
#pragma unroll
            for (int j = 0; j < 14; ++j) {
                if (CHAINS == 1) {
                    asm volatile("v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2"
                                 : "+v"(x0) : "v"(m), "v"(ad));
                } else {
                    asm volatile("v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5"
                                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(m), "v"(ad));
                }
            }
        }
        out = x0 + x1 + x2 + x3;
    }
    sink[blockIdx.x * 512 + threadIdx.x] = out;
}

template <int MODE, int CHAINS>
float run(float* sink, int iters) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    k<MODE, CHAINS><<<256, 512>>>(sink, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<MODE, CHAINS><<<256, 512>>>(sink, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float* sink;
    if (hipMalloc(&sink, 256 * 512 * 4) != hipSuccess) return 1;
    const int iters = 20000;   // 160 k MFMAs = 5.1 M matrix-pipe cycles per MFMA wave; 1.12 M VALU = 4.5 M cycles per VALU wave
    printf("MFMA chain only        %.3f ms\n", run<1, 1>(sink, iters));
    printf("VALU chain only        %.3f ms\n", run<2, 1>(sink, iters));
    printf("VALU 4 chains only     %.3f ms\n", run<2, 4>(sink, iters));
    printf("both, VALU one chain   %.3f ms\n", run<3, 1>(sink, iters));
    printf("both, VALU 4 chains    %.3f ms\n", run<3, 4>(sink, iters));
    return 0;
}
