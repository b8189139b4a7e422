// Microbenchmark: v_mfma_f32_32x32x2_f32 throughput with NACC independent accumulator chains,
// one wave per SIMD (256 threads per block, 512-register budget), every CU busy.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256, 1) void chain(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int j = 0; j < 16; ++j) acc[i][j] = threadIdx.x * 0.001f + i;
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i)
        for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
void run(float* d, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    chain<NACC><<<256, 256>>>(d, 10, 1e-3f, 1e-3f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    chain<NACC><<<256, 256>>>(d, iters, 1e-3f, 1e-3f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double n = double(iters) * 16 * NACC;           // MFMAs per wave
    double flops = n * 4096.0 * 1024;                // 1024 waves
    printf("NACC=%d  %.3f ms  %.1f TFLOP/s  cycles/MFMA @2.4GHz = %.2f\n", NACC, ms, flops / ms / 1e9, ms * 1e-3 * 2.4e9 / n);
}

int main() {
    float* d; hipMalloc(&d, 256 * 256 * 4);
    for (int rep = 0; rep < 2; ++rep) { run<1>(d, 4000); run<2>(d, 2000); run<4>(d, 1000); }
    return 0;
}
