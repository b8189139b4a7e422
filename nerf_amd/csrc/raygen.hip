// Ray generation: pixel grid -> (origin, un-normalised world direction).
//
// Replaces src/datasets/nerf/blender.py:76-101 (numpy float64 on a DataLoader worker + a
// 15 MB host-to-device copy per 800x800 frame) with one kernel that writes the rays where
// they are consumed: 24 B per ray, HBM-bound.  Math in fp64 with separately rounded ops
// (like numpy), results cast to fp32 (blender.py:99-100).
#include "nrf_common.h"

namespace {

struct Cam {
    double r[9];  // rotation rows
    double t[3];  // origin
    double focal, half_w, half_h;
};

__global__ __launch_bounds__(256) void raygen_kernel(Cam cam, int W, int64_t first_ray, int64_t n_rays,
                                                     const int64_t* __restrict__ pixel_index,
                                                     float* __restrict__ xyz, float* __restrict__ dirs) {
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n_rays) return;
    const int64_t pix = pixel_index ? pixel_index[i] : first_ray + i;
    const int64_t v = pix / W, u = pix - v * W;                       // row-major: v slow, u fast (:88-90)
    const double cx = __ddiv_rn(__dsub_rn(double(u), cam.half_w), cam.focal);      // (u - W/2) / focal
    const double cy = __ddiv_rn(-__dsub_rn(double(v), cam.half_h), cam.focal);     // -(v - H/2) / focal
    const double cz = -1.0;                                                        // (:94)
#pragma unroll
    for (int k = 0; k < 3; ++k) {                                                  // R @ dir (:95)
        const double d = __dadd_rn(__dadd_rn(__dmul_rn(cam.r[3 * k], cx), __dmul_rn(cam.r[3 * k + 1], cy)),
                                   __dmul_rn(cam.r[3 * k + 2], cz));
        dirs[i * 3 + k] = float(d);
        xyz[i * 3 + k] = float(cam.t[k]);                                          // (:96)
    }
}

}  // namespace

NRF_EXPORT int nrf_raygen(const double* c2w, int H, int W, double focal, int64_t first_ray, int64_t n_rays,
                          const int64_t* pixel_index, float* xyz, float* viewdirs, nrf_stream_t stream) {
    if (!c2w || !xyz || !viewdirs) return NRF_ERR_NULL;
    if (H <= 0 || W <= 0 || !(focal > 0.0) || n_rays < 0 || first_ray < 0) return NRF_ERR_SHAPE;
    if (!pixel_index && first_ray + n_rays > int64_t(H) * W) return NRF_ERR_SHAPE;
    if (n_rays == 0) return NRF_OK;
    Cam cam;
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) cam.r[3 * r + c] = c2w[4 * r + c];
        cam.t[r] = c2w[4 * r + 3];
    }
    cam.focal = focal;
    cam.half_w = W / 2.0;
    cam.half_h = H / 2.0;
    const int64_t blocks = (n_rays + 255) / 256;
    if (blocks > 0x7fffffffLL) return NRF_ERR_SHAPE;
    nrf_launch_begin();
    raygen_kernel<<<dim3((unsigned)blocks), 256, 0, static_cast<hipStream_t>(stream)>>>(cam, W, first_ray, n_rays,
                                                                                      pixel_index, xyz, viewdirs);
    return nrf_launch_status();
}
