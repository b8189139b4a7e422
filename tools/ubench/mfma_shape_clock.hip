// Does the bf16 MFMA SHAPE change what the chip delivers under the load of mlp_bf16x3_kernel's step?
// (MI355X_MICROARCH.md, DVFS give-back item 7: the 16x16x32 loop held a higher clock than the 32x32x16 loop at equal
// cycles per FLOP.)  A model of the kernel's step on random data: three ds_read_b128 of weight fragments per step out of
// an LDS ring, the six products of the three-term split, B fragments and accumulators in registers (192 + 128), NF VALU
// fillers per 32 matrix cycles, one 4-wave workgroup per CU, one wave per SIMD.
//   SHAPE 0: 6 x v_mfma_f32_32x32x16_bf16 per step into one 32 x 32 block
//   SHAPE 1: 12 x v_mfma_f32_16x16x32_bf16 per step (the same 16 rows x 32 k of weights against two 16-point blocks)
// Both do 6 x 32768 FLOP per step.  Wall time by HIP events after two seconds of warm-up launches; in-kernel clock from
// s_memtime / s_memrealtime.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_shape_clock.hip -o tools/ubench/mfma_shape_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kLdsBytes = 120 * 1024;
constexpr int kSteps = 1160;   // steps per "tile", as the real stream

#define M32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, (a)), __builtin_bit_cast(bf16x8, (b)), (c), 0, 0, 0)
#define M16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, (a)), __builtin_bit_cast(bf16x8, (b)), (c), 0, 0, 0)

template <int SHAPE, int NF>
__global__ __launch_bounds__(256, 1) void k(const u32x4* __restrict__ rnd, int tiles, unsigned long long* clk, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < kLdsBytes / 16; i += 256) reinterpret_cast<u32x4*>(smem)[i] = rnd[(blockIdx.x * 131 + i) & 0xffff];
    __syncthreads();
    u32x4 X[8][2][3];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int c = 0; c < 3; ++c) X[a][b][c] = rnd[(threadIdx.x * 48 + a * 6 + b * 3 + c + blockIdx.x * 7) & 0xffff];
    f32x16 Y[8];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int t = 0; t < 16; ++t) Y[a][t] = 0.f;
    int f0 = threadIdx.x, f1 = threadIdx.x * 3, f2 = 7, f3 = 11, lim = 5;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    const u32x4* base = reinterpret_cast<const u32x4*>(smem) + lane;
    u32x4 wq[4][3];   // the fragments of steps s .. s + 1, read two steps ahead of their MFMAs (as the kernel does)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < 3; ++t) wq[i][t] = base[(i * 3 + t) * 64];
    for (int tile = 0; tile < tiles; ++tile) {
        for (int so = 0; so < kSteps / 40 * 40; so += 40) {   // 40 steps = one pass over the 120-KiB ring
#pragma unroll
            for (int si = 0; si < 40; ++si) {
                const int ob = (si >> 2) & 7, kb = si & 7, s2 = (si >> 3) & 1;
                const u32x4 w0 = wq[si & 3][0], w1 = wq[si & 3][1], w2 = wq[si & 3][2];
                const int nx = (si + 2) % 40;
                __builtin_amdgcn_sched_barrier(0);
#define RD(t) wq[nx & 3][t] = base[(nx * 3 + t) * 64];
#define FILL                                                                           \
    if (NF >= 1) asm volatile("v_max_i32 %0, %1, %0" : "+v"(f0) : "v"(lim));           \
    if (NF >= 2) asm volatile("v_max_i32 %0, %1, %0" : "+v"(f1) : "v"(lim));           \
    if (NF >= 3) asm volatile("v_add_u32 %0, %1, %0" : "+v"(f2) : "v"(f0));            \
    if (NF >= 4) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(f3) : "v"(f1));            \
    if (NF >= 5) asm volatile("v_sub_u32 %0, %1, %0" : "+v"(f0) : "v"(f3));            \
    if (NF >= 6) asm volatile("v_add_u32 %0, %1, %0" : "+v"(f1) : "v"(f2));
                if (SHAPE == 0) {
#define GAP(wa, xb, R) Y[ob] = M32(wa, xb, Y[ob]); R FILL __builtin_amdgcn_sched_barrier(0);
                    GAP(w2, X[kb][s2][0], RD(0)) GAP(w0, X[kb][s2][2], RD(1)) GAP(w1, X[kb][s2][1], RD(2))
                    GAP(w1, X[kb][s2][0], ) GAP(w0, X[kb][s2][1], ) GAP(w0, X[kb][s2][0], )
#undef GAP
                } else {
                    // 16 rows (block 2 ob + s2) x 32 k (step kb) against point blocks nb = 0, 1: X[kb][nb][term]
                    f32x4 y0 = __builtin_shufflevector(Y[ob], Y[ob], 0 + 4 * 0, 1, 2, 3);
                    f32x4 y1 = __builtin_shufflevector(Y[ob], Y[ob], 8, 9, 10, 11);
                    if (s2) {
                        y0 = __builtin_shufflevector(Y[ob], Y[ob], 4, 5, 6, 7);
                        y1 = __builtin_shufflevector(Y[ob], Y[ob], 12, 13, 14, 15);
                    }
#define GAP(wa, t, R) y0 = M16(wa, X[kb][0][t], y0); y1 = M16(wa, X[kb][1][t], y1); R FILL __builtin_amdgcn_sched_barrier(0);
                    GAP(w2, 0, RD(0)) GAP(w0, 2, RD(1)) GAP(w1, 1, RD(2)) GAP(w1, 0, ) GAP(w0, 1, ) GAP(w0, 0, )
#undef GAP
                    if (s2) {
#pragma unroll
                        for (int t = 0; t < 4; ++t) Y[ob][4 + t] = y0[t], Y[ob][12 + t] = y1[t];
                    } else {
#pragma unroll
                        for (int t = 0; t < 4; ++t) Y[ob][t] = y0[t], Y[ob][8 + t] = y1[t];
                    }
                }
            }
        }
        // keep the accumulators bounded (a tile of the real kernel starts from biases)
#pragma unroll
        for (int a = 0; a < 8; ++a)
#pragma unroll
            for (int t = 0; t < 16; ++t) Y[a][t] = Y[a][t] * 1e-3f;
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) {
        clk[(blockIdx.x * 4 + threadIdx.x / 64) * 2] = c1 - c0;
        clk[(blockIdx.x * 4 + threadIdx.x / 64) * 2 + 1] = r1 - r0;
    }
    float s = float(f0 + f1 + f2 + f3);
#pragma unroll
    for (int a = 0; a < 8; ++a) s += Y[a][0] + Y[a][5] + Y[a][10] + Y[a][15];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
}

static u32x4* g_rnd;
static unsigned long long* g_clk;
static float* g_sink;

template <int SHAPE, int NF> void run(int tiles) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<SHAPE, NF>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    // warm-up: about two seconds of back-to-back launches
    float ms = 0.f;
    (void)hipEventRecord(e0);
    k<SHAPE, NF><<<256, 256, kLdsBytes>>>(g_rnd, tiles, g_clk, g_sink);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
    const int warm = int(2000.f / ms) + 1;
    for (int i = 0; i < warm; ++i) k<SHAPE, NF><<<256, 256, kLdsBytes>>>(g_rnd, tiles, g_clk, g_sink);
    const int reps = 10;
    (void)hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) k<SHAPE, NF><<<256, 256, kLdsBytes>>>(g_rnd, tiles, g_clk, g_sink);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    std::vector<unsigned long long> h(2048);
    (void)hipMemcpy(h.data(), g_clk, 2048 * 8, hipMemcpyDeviceToHost);
    double cyc = 0, rt = 0;
    for (int i = 0; i < 1024; ++i) cyc += double(h[2 * i]), rt += double(h[2 * i + 1]);
    const double steps = double(tiles) * (kSteps / 40 * 40);
    const double flop = steps * 6 * 32768.0 * 1024;
    printf("shape %s  fillers/gap %d : %.3f ms  %.1f TFLOP/s bf16  cycles/step %.1f  in-kernel clock %.3f GHz\n",
           SHAPE ? "16x16x32" : "32x32x16", NF, ms, flop / (ms * 1e-3) / 1e12, cyc / 1024 / steps, cyc / rt * 0.1);
    fflush(stdout);
}

int main() {
    std::vector<unsigned> h(65536 * 4);
    srand(1);
    for (auto& v : h) {   // two bf16 per word, magnitudes in [2^-3, 2), random signs and mantissas
        auto one = [] { return unsigned(((rand() & 1) << 15) | ((124 + (rand() & 3)) << 7) | (rand() & 127)); };
        v = one() | (one() << 16);
    }
    (void)hipMalloc(&g_rnd, h.size() * 4);
    (void)hipMemcpy(g_rnd, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    (void)hipMalloc(&g_clk, 2048 * 8);
    (void)hipMalloc(&g_sink, 256 * 256 * 4);
    const int tiles = 40;
    for (int pass = 0; pass < 2; ++pass) {
        run<0, 0>(tiles); run<1, 0>(tiles);
        run<0, 3>(tiles); run<1, 3>(tiles);
        run<0, 5>(tiles); run<1, 5>(tiles);
    }
    return 0;
}
