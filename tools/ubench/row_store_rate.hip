// What does HBM absorb of the training forward's store pattern when nothing else runs?  1024 waves (256 workgroups of four,
// one per CU), each writing its own tile's 2528 rows of 128 B (two rows, 512 B apart, per store instruction, rows in the
// MFMA block order of mlp_chain.h: store_block) tile after tile, as the SAVE kernels do -- and the same bytes as plain
// consecutive 256-byte pieces, each with and without the non-temporal hint.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/row_store_rate.hip -o tools/ubench/row_store_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int kRows = 2528;

template <int PATTERN, bool NT>
__global__ __launch_bounds__(256, 1) void k(float* buf, int64_t n_tiles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    for (int64_t tg = blockIdx.x; tg * 4 + wave < n_tiles; tg += gridDim.x) {
        const int64_t tile = tg * 4 + wave;
        float* base = buf + tile * (int64_t(kRows) * 32);
        const float v = float(tile);
        if (PATTERN == 0) {   // block order: rows 32 b + (t & 3) + 8 (t >> 2) + 4 h
            for (int b = 0; b < kRows / 32; ++b)
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    float* p = base + (32 * b + (t & 3) + 8 * (t >> 2) + 4 * h) * 32 + j;
                    if (NT) __builtin_nontemporal_store(v, p);
                    else *p = v;
                }
        } else {              // consecutive 256-byte pieces
            for (int r = 0; r < kRows / 2; ++r) {
                float* p = base + r * 64 + lane;
                if (NT) __builtin_nontemporal_store(v, p);
                else *p = v;
            }
        }
    }
}

template <int PATTERN, bool NT>
void run(const char* name, float* buf, int64_t n_tiles) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    k<PATTERN, NT><<<256, 256>>>(buf, n_tiles);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 3; ++i) k<PATTERN, NT><<<256, 256>>>(buf, n_tiles);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = 3.0 * double(n_tiles) * kRows * 128;
    printf("%-28s %8.3f ms per pass  %6.2f TB/s\n", name, ms / 3, bytes / (ms * 1e-3) / 1e12);
}

int main() {
    const int64_t n_tiles = 24576;   // the fine launch of a 4096-ray step: 786,432 points
    float* buf = nullptr;
    if (hipMalloc(&buf, size_t(n_tiles) * kRows * 128) != hipSuccess) return 1;
    run<0, false>("block order", buf, n_tiles);
    run<0, true>("block order, nt", buf, n_tiles);
    run<1, false>("consecutive", buf, n_tiles);
    run<1, true>("consecutive, nt", buf, n_tiles);
    hipFree(buf);
    return 0;
}
