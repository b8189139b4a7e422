// Ablation harness for mlp_fwd_kernel: compiled once per -DNRF_ABLATE_* flag, times the kernel on
// 32768 rays x 64 samples of junk data (outputs are not checked -- ablated builds are wrong by design).
#include "../../nerf_amd/csrc/mlp_fp32.hip"
#include <cstdio>
#include <vector>
int main() {
    const int64_t R = 32768; const int S = 64;
    std::vector<float> h(nrf::kPackedFloats);
    for (size_t i = 0; i < h.size(); ++i) h[i] = ((i * 2654435761u) % 2001) / 20000.f - 0.05f;
    float *packed, *o, *d, *z, *raw;
    (void)hipMalloc(&packed, h.size() * 4); (void)hipMemcpy(packed, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> ho(R * 3, 0.1f), hd(R * 3, 0.5f), hz(S);
    for (int i = 0; i < S; ++i) hz[i] = 2.f + 4.f * i / (S - 1);
    (void)hipMalloc(&o, R * 12); (void)hipMalloc(&d, R * 12); (void)hipMalloc(&z, S * 4); (void)hipMalloc(&raw, R * S * 16);
    (void)hipMemcpy(o, ho.data(), R * 12, hipMemcpyHostToDevice); (void)hipMemcpy(d, hd.data(), R * 12, hipMemcpyHostToDevice);
    (void)hipMemcpy(z, hz.data(), S * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e0);
        int st = nrf_mlp_forward_rays(packed, o, d, z, 0, R, S, raw, nullptr);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (st) { printf("status %d\n", st); return 1; }
        if (rep && ms < best) best = ms;
    }
    const double flop = double(R) * S * 1186816.0;
    printf("%-14s %.3f ms  %.1f TFLOP/s  %.1f%% of 157.3\n", NRF_ABLATE_NAME, best, flop / best / 1e9, flop / best / 1e9 / 1.573);
    return 0;
}
