// Accuracy of the hardware v_sin_f32 / v_cos_f32 (input in revolutions) against double precision, for the
// positional encoding's arguments: x in [-6, 6], octaves 2^0 .. 2^9.  The revolutions are formed without the
// large-argument loss: rev(x) = x / 2pi in hi + lo parts once, then rev(2^k x) = fract(2^k hi) + 2^k lo.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o hw_sincos_error hw_sincos_error.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

__global__ void k(const float* x, int n, int oct, float* s, float* c) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float C_hi = 0.15915494309189535f;                  // fl(1 / 2pi)
    const float C_lo = (float)(0.15915494309189533576888 - (double)C_hi);
    const float r_hi = x[i] * C_hi;
    const float r_lo = fmaf(x[i], C_hi, -r_hi) + x[i] * C_lo;
    const float sc = (float)(1 << oct);
    const float t = r_hi * sc;                                 // exact
    const float f = t - floorf(t);                             // exact: fract
    const float r = fmaf(r_lo, sc, f);
    s[i] = __builtin_amdgcn_sinf(r);
    c[i] = __builtin_amdgcn_cosf(r);
}

int main() {
    const int n = 1 << 22;
    std::vector<float> x(n), s(n), c(n);
    unsigned long long st = 88172645463325252ull;
    for (int i = 0; i < n; ++i) {
        st ^= st << 13, st ^= st >> 7, st ^= st << 17;
        x[i] = (float)((double)(st >> 11) / 9007199254740992.0 * 12.0 - 6.0);
    }
    float *dx, *ds, *dc;
    hipMalloc(&dx, n * 4), hipMalloc(&ds, n * 4), hipMalloc(&dc, n * 4);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    for (int oct = 0; oct < 10; ++oct) {
        k<<<n / 256, 256>>>(dx, n, oct, ds, dc);
        hipMemcpy(s.data(), ds, n * 4, hipMemcpyDeviceToHost);
        hipMemcpy(c.data(), dc, n * 4, hipMemcpyDeviceToHost);
        double es = 0, ec = 0, rs = 0, rc = 0;
        for (int i = 0; i < n; ++i) {
            const double a = (double)x[i] * (double)(1 << oct);
            const double e1 = fabs((double)s[i] - sin(a)), e2 = fabs((double)c[i] - cos(a));
            es = fmax(es, e1), ec = fmax(ec, e2), rs += e1 * e1, rc += e2 * e2;
        }
        printf("octave %d: max abs err sin %.3e cos %.3e   rms sin %.3e cos %.3e\n", oct, es, ec, sqrt(rs / n), sqrt(rc / n));
    }
    return 0;
}
