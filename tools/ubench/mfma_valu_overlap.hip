// How many independent VALU instructions hide in the gap of a v_mfma_f32_32x32x16_bf16 stream (one wave per SIMD, two
// accumulator chains), and does it matter whether the accumulators are VGPRs or AGPRs?  NF fillers per MFMA.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_valu_overlap.hip -o tools/ubench/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define FILL1 "v_max_i32 %5, %7, %5\n\t"
#define FILL2 FILL1 "v_max_i32 %6, %7, %6\n\t"

template <int ACC_AGPR, int NF>
__global__ __launch_bounds__(256, 1) void k(unsigned long long* out, float* sink) {
    f32x16 c0 = {}, c1 = {};
    u32x4 a = {0x3c003c00u + threadIdx.x, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
    u32x4 b0 = {0x3c003c00u, 0x3c003c00u + threadIdx.x, 0x3c003c00u, 0x3c003c00u}, b1 = b0;
    int f0 = threadIdx.x, f1 = threadIdx.x * 3, lim = 5;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 512; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#define BODY(CC)                                                                                                         \
    if (NF == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %2, %4, %1"        \
                              : CC(c0), CC(c1) : "v"(a), "v"(b0), "v"(b1), "v"(f0), "v"(f1), "v"(lim));                    \
    if (NF == 2) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\t" FILL2 "v_mfma_f32_32x32x16_bf16 %1, %2, %4, %1\n\t" FILL2 \
                              : CC(c0), CC(c1) : "v"(a), "v"(b0), "v"(b1), "v"(f0), "v"(f1), "v"(lim));                    \
    if (NF == 4) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\t" FILL2 FILL2 "v_mfma_f32_32x32x16_bf16 %1, %2, %4, %1\n\t" FILL2 FILL2 \
                              : CC(c0), CC(c1) : "v"(a), "v"(b0), "v"(b1), "v"(f0), "v"(f1), "v"(lim));                    \
    if (NF == 6) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\t" FILL2 FILL2 FILL2 "v_mfma_f32_32x32x16_bf16 %1, %2, %4, %1\n\t" FILL2 FILL2 FILL2 \
                              : CC(c0), CC(c1) : "v"(a), "v"(b0), "v"(b1), "v"(f0), "v"(f1), "v"(lim));                    \
    if (NF == 8) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\t" FILL2 FILL2 FILL2 FILL2 "v_mfma_f32_32x32x16_bf16 %1, %2, %4, %1\n\t" FILL2 FILL2 FILL2 FILL2 \
                              : CC(c0), CC(c1) : "v"(a), "v"(b0), "v"(b1), "v"(f0), "v"(f1), "v"(lim));
#define CV(x) "+v"(x)
#define CA(x) "+a"(x)
            if (ACC_AGPR) { BODY(CA) } else { BODY(CV) }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;
    sink[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[3] + f0 + f1;
}

template <int ACC_AGPR, int NF> void run() {
    unsigned long long* out; float* sink;
    (void)hipMalloc(&out, 256 * 4 * 8); (void)hipMalloc(&sink, 256 * 256 * 4);
    for (int r = 0; r < 2; ++r) k<ACC_AGPR, NF><<<256, 256>>>(out, sink);
    (void)hipDeviceSynchronize();
    unsigned long long h[1024]; (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < 1024; ++i) s += double(h[i]);
    printf("acc in %s, %d VALU fillers per MFMA: %.2f cycles per MFMA\n", ACC_AGPR ? "AGPR" : "VGPR", NF, s / 1024 / 4096);
}
int main() {
    run<0, 0>(); run<0, 2>(); run<0, 4>(); run<0, 6>(); run<0, 8>();
    run<1, 0>(); run<1, 2>(); run<1, 4>(); run<1, 6>(); run<1, 8>();
    return 0;
}
