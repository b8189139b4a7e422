// Phase breakdown of mlp_bf16_kernel: a -DBF_STAMP build of the real kernel source stamps s_memtime at the phase
// boundaries of one tile per wave (the 5th); this harness runs it on junk data and prints, per phase, the median
// cycles over all waves beside the cycles its MFMAs alone would take (32 per v_mfma_f32_32x32x16_bf16).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 -DBF_STAMP \
//         tools/ubench/bf16_phase.hip -o tools/ubench/bf16_phase
#include "../../nerf_amd/csrc/mlp_bf16.hip"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
int main() {
    const int64_t R = 65536; const int S = 192;
    const size_t bytes = size_t(nrf_packed_bf16_bytes());
    std::vector<unsigned short> h(bytes / 2);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned short)(0x3c00 + ((i * 2654435761u) >> 20) % 0x300) ^ ((i & 1) << 15);  // +-0.01..0.04 bf16
    char* packed; float *o, *d, *z, *raw;
    (void)hipMalloc(&packed, bytes); (void)hipMemcpy(packed, h.data(), bytes, hipMemcpyHostToDevice);
    std::vector<float> small(nrf::kSmallFloats, 0.01f);
    (void)hipMemcpy(packed + kBSmallFloatOff * 4, small.data(), small.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> ho(R * 3), hd(R * 3), hz(R * S);
    for (int64_t i = 0; i < R * 3; ++i) { ho[i] = 0.1f * ((i * 37) % 13) - 0.5f; hd[i] = 0.3f + 0.05f * ((i * 11) % 7); }
    for (int64_t i = 0; i < R * S; ++i) hz[i] = 2.f + 4.f * float(i % S) / (S - 1);
    (void)hipMalloc(&o, R * 12); (void)hipMalloc(&d, R * 12); (void)hipMalloc(&z, R * S * 4); (void)hipMalloc(&raw, R * S * 16);
    (void)hipMemcpy(o, ho.data(), R * 12, hipMemcpyHostToDevice); (void)hipMemcpy(d, hd.data(), R * 12, hipMemcpyHostToDevice);
    (void)hipMemcpy(z, hz.data(), R * S * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e0);
        int st = nrf_mlp_forward_rays_bf16(packed, o, d, z, S, R, S, raw, nullptr);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (st) { printf("status %d\n", st); return 1; }
        if (rep && ms < best) best = ms;
    }
    printf("kernel %.3f ms  %.1f TFLOP/s\n", best, double(R) * S * 1186816.0 / best / 1e9);
#ifdef BF_STAMP
    const char* env = getenv("NRF_BF16_SETS");
    const int NW = 1024; (void)env;                          // waves on the chip (NSETS = 2)
    const int mf = NW == 1024 ? 2 : 1;                       // MFMAs per step per wave
    std::vector<unsigned long long> st(256 * 8 * 128);
    (void)hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_bf_stamps), st.size() * 8);
    const char* names[13] = {"loads+encode", "L0", "pts1", "pts2", "pts3", "pts4", "pts5(skip)", "pts6", "pts7", "feature", "alpha head", "views+rgb", "store"};
    const int mfma[13] = {0, 64, 256, 256, 256, 256, 320, 256, 256, 256, 0, 144, 0};
    double tot = 0, totm = 0;
    for (int p = 0; p < 13; ++p) {
        std::vector<long long> v;
        for (int w = 0; w < NW; ++w) v.push_back((long long)(st[w * 128 + p + 1] - st[w * 128 + p]));
        std::sort(v.begin(), v.end());
        printf("%-14s median %7lld  p10 %7lld  p90 %7lld   mfma-only %6d  overhead %6lld\n", names[p], v[NW / 2], v[NW / 10], v[NW * 9 / 10],
               mfma[p] * 16 * mf, v[NW / 2] - mfma[p] * 16 * mf);
        tot += v[NW / 2]; totm += mfma[p] * 16 * mf;
    }
    printf("tile total %.0f cycles, own MFMAs %.0f (%.1f %%); at two waves per SIMD the partner's MFMAs fill the rest\n", tot, totm, 100 * totm / tot);
    // per output block of the 8 layers of the pair loop: stamps at 16 + 10 L + {0..7 block starts, 8 before the last
    // block's epilogue, 9 after it}; then the phase stamp 3 + L
    for (int L = 0; L < 8; ++L) {
        printf("pts%d:", L + 1);
        for (int k = 0; k < 10; ++k) {
            std::vector<long long> v;
            const int i0 = 16 + 10 * L + k;
            for (int w = 0; w < NW; ++w) {
                const unsigned long long end = k < 9 ? st[w * 128 + i0 + 1] : st[w * 128 + 3 + L];
                v.push_back((long long)(end - st[w * 128 + i0]));
            }
            std::sort(v.begin(), v.end());
            printf(" %5lld", v[NW / 2]);
        }
        printf("   (8 blocks, last epilogue, return; MFMA-only %d per block)\n", L == 4 ? 1280 : 1024);
    }
#endif
    return 0;
}
