// Does v_mfma_f32_32x32x16_bf16 issue slower when its B (or A) operand comes from an AGPR while the accumulator is a
// VGPR?  One wave per SIMD, two independent accumulator chains (the shape of mlp_bf16_kernel's inner loop), 4096 MFMAs.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_agpr_src.hip -o tools/ubench/mfma_agpr_src
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(unsigned long long* out, float* sink) {
    f32x16 c0 = {}, c1 = {};
    u32x4 a = {0x3c003c00u + threadIdx.x, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
    u32x4 b0 = {0x3c003c00u, 0x3c003c00u + threadIdx.x, 0x3c003c00u, 0x3c003c00u}, b1 = b0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 256; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (MODE == 0) {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %2, %4, %1"
                             : "+v"(c0), "+v"(c1) : "v"(a), "v"(b0), "v"(b1));
            } else if (MODE == 1) {   // B from AGPR
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %2, %4, %1"
                             : "+v"(c0), "+v"(c1) : "v"(a), "a"(b0), "a"(b1));
            } else if (MODE == 2) {   // A from AGPR
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %2, %4, %1"
                             : "+v"(c0), "+v"(c1) : "a"(a), "v"(b0), "v"(b1));
            } else {                  // everything AGPR (acc too)
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %2, %4, %1"
                             : "+a"(c0), "+a"(c1) : "v"(a), "v"(b0), "v"(b1));
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;
    sink[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[3];
}

template <int MODE> void run(const char* name) {
    unsigned long long* out; float* sink;
    (void)hipMalloc(&out, 256 * 4 * 8); (void)hipMalloc(&sink, 256 * 256 * 4);
    for (int r = 0; r < 2; ++r) k<MODE><<<256, 256>>>(out, sink);
    (void)hipDeviceSynchronize();
    unsigned long long h[1024]; (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < 1024; ++i) s += double(h[i]);
    printf("%-28s %.2f cycles per MFMA\n", name, s / 1024 / 4096);
}
int main() {
    run<0>("A vgpr, B vgpr, C vgpr");
    run<1>("A vgpr, B agpr, C vgpr");
    run<2>("A agpr, B vgpr, C vgpr");
    run<3>("A vgpr, B vgpr, C agpr");
    return 0;
}
