// Does a chain of v_mfma_f32_32x32x16_bf16 that all accumulate into the SAME block issue at the rate of two alternating
// chains (one wave per SIMD)?  And what does one ds_read_b128 per gap cost either of them?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_chain.hip -o tools/ubench/mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int CHAINS, int LDS>
__global__ __launch_bounds__(256, 1) void k(unsigned long long* out, float* sink) {
    __shared__ u32x4 sm[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) sm[i] = u32x4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
    __syncthreads();
    f32x16 c0 = {}, c1 = {};
    u32x4 a = {0x3c003c00u + threadIdx.x, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
    u32x4 b0 = {0x3c003c00u, 0x3c003c00u + threadIdx.x, 0x3c003c00u, 0x3c003c00u}, b1 = b0;
    u32x4 w = a;
    const unsigned p = unsigned(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) void*)sm)) + (threadIdx.x & 63) * 16;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 512; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (CHAINS == 1)
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\tv_mfma_f32_32x32x16_bf16 %0, %2, %4, %0"
                             : "+v"(c0), "+v"(c1) : "v"(a), "v"(b0), "v"(b1));
            else
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %2, %4, %1"
                             : "+v"(c0), "+v"(c1) : "v"(a), "v"(b0), "v"(b1));
            if (LDS) {
                // the wait covers reads issued at least two MFMAs earlier
                if (j == 3) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w) :: "memory"); b1 = w; }
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(w) : "v"(p), "n"(0));
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;
    sink[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[3] + w[0];
}

template <int CHAINS, int LDS> void run() {
    unsigned long long* out; float* sink;
    (void)hipMalloc(&out, 256 * 4 * 8); (void)hipMalloc(&sink, 256 * 256 * 4);
    for (int r = 0; r < 2; ++r) k<CHAINS, LDS><<<256, 256>>>(out, sink);
    (void)hipDeviceSynchronize();
    unsigned long long h[1024]; (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < 1024; ++i) s += double(h[i]);
    printf("%d chain(s), %s: %.2f ticks per MFMA\n", CHAINS, LDS ? "one ds_read_b128 per 2 MFMAs" : "no LDS", s / 1024 / 4096);
}
int main() {
    run<2, 0>(); run<1, 0>(); run<2, 1>(); run<1, 1>();
    return 0;
}
