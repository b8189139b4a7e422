// Library-level entry points: version / error text and the whole coarse+fine forward path
// (the intended body of Renderer.forward, volume_renderer.py:143-153) as one enqueue sequence.
#include "nrf_common.h"

#include <atomic>

NRF_EXPORT int nrf_version(void) { return 100; }

NRF_EXPORT const char* nrf_strerror(int status) {
    switch (status) {
        case NRF_OK: return "ok";
        case NRF_ERR_NULL: return "a required pointer is NULL";
        case NRF_ERR_SHAPE: return "size outside the supported range (see include/nerf_amd.h)";
        case NRF_ERR_LAUNCH: return "HIP kernel launch failed";
        case NRF_ERR_WORKSPACE: return "workspace too small (see nrf_render_workspace_bytes)";
        case NRF_ERR_ALIGN: return "pointer not 16-byte aligned";
        case NRF_ERR_STATIC_LIMIT:
            return "nrf_render_static: n_rays or sample count beyond the library's static scratch "
                   "(NRF_RENDER_STATIC_MAX_RAYS rays, n_coarse + n_importance <= 256); use nrf_render with a workspace";
        case NRF_ERR_RUNTIME: return "a HIP runtime call failed (hipGetDevice / hipGetSymbolAddress)";
        case NRF_ERR_RANGE: return "NRF_ERR_RANGE: fp16x2: a weight lies outside fp16's range (|w| >= 65520 or not finite); use bf16x3 or fp32";
        default: return "unknown nrf_status";
    }
}

namespace {

constexpr int64_t align256(int64_t n) { return (n + 255) & ~int64_t(255); }

struct Workspace {
    int64_t raw = 0, z_coarse = 0, w_coarse = 0, acc_coarse = 0, z_fine = 0, total = 0;
    constexpr Workspace(int64_t R, int nc, int ni) {
        int64_t off = 0;
        raw = off; off += align256(R * (nc + ni) * 4 * 4);   // raw (R, S, 4), sized for the fine pass
        z_coarse = off; off += align256(R * nc * 4);         // only used when perturbed
        w_coarse = off; off += align256(R * nc * 4);
        acc_coarse = off; off += align256(R * 4);
        z_fine = off; off += align256(R * (nc + ni) * 4);
        total = off;
    }
};

}  // namespace

NRF_EXPORT int64_t nrf_render_workspace_bytes(int64_t n_rays, int n_coarse, int n_importance) {
    if (n_rays < 0 || n_coarse <= 0 || n_importance <= 0) return NRF_ERR_SHAPE;
    return Workspace(n_rays, n_coarse, n_importance).total;
}

enum MlpKind { MLP_FP32 = 0, MLP_BF16 = 1, MLP_BF16X3 = 2, MLP_FP16X2 = 3 };

static int render_path(const nrf_render_args* a, MlpKind kind, nrf_stream_t stream) {
    if (!a) return NRF_ERR_NULL;
    if (!a->packed_coarse || !a->packed_fine || !a->rays_o || !a->rays_d || !a->bins || !a->u || !a->u2 ||
        !a->coarse_rgb || !a->coarse_depth || !a->fine_rgb || !a->fine_depth || !a->fine_acc || !a->workspace)
        return NRF_ERR_NULL;
    if (a->n_rays < 0 || a->n_coarse != 64 || a->n_importance <= 0 || a->n_importance % 64 != 0 ||
        a->n_coarse + a->n_importance > 256)
        return NRF_ERR_SHAPE;
    const Workspace ws(a->n_rays, a->n_coarse, a->n_importance);
    if (a->workspace_bytes < ws.total) return NRF_ERR_WORKSPACE;
    if (!nrf_aligned16(a->workspace)) return NRF_ERR_ALIGN;
    if (a->n_rays == 0) return NRF_OK;
    char* base = static_cast<char*>(a->workspace);
    float* raw = reinterpret_cast<float*>(base + ws.raw);
    float* zc_buf = reinterpret_cast<float*>(base + ws.z_coarse);
    float* w_c = reinterpret_cast<float*>(base + ws.w_coarse);
    float* acc_c = reinterpret_cast<float*>(base + ws.acc_coarse);
    float* z_f = reinterpret_cast<float*>(base + ws.z_fine);
    const int nc = a->n_coarse, ni = a->n_importance;
    int st;
    // render_coarse (volume_renderer.py:201-265)
    const float* zc = a->bins;
    int64_t zc_stride = 0;
    if (a->u_coarse) {
        if ((st = nrf_perturb_samples(a->bins, 0, a->u_coarse, a->n_rays, nc, zc_buf, stream))) return st;
        zc = zc_buf;
        zc_stride = nc;
    }
    auto mlp = [&](const float* packed, const float* z, int64_t z_stride, int n_samples) {
        if (kind == MLP_BF16)
            return nrf_mlp_forward_rays_bf16(packed, a->rays_o, a->rays_d, z, z_stride, a->n_rays, n_samples, raw, stream);
        if (kind == MLP_BF16X3)
            return nrf_mlp_forward_rays_bf16x3(packed, a->rays_o, a->rays_d, z, z_stride, a->n_rays, n_samples, raw, stream);
        if (kind == MLP_FP16X2)
            return nrf_mlp_forward_rays_fp16x2(packed, a->rays_o, a->rays_d, z, z_stride, a->n_rays, n_samples, raw, stream);
        return nrf_mlp_forward_rays(packed, a->rays_o, a->rays_d, z, z_stride, a->n_rays, n_samples, raw, stream);
    };
    if ((st = mlp(a->packed_coarse, zc, zc_stride, nc))) return st;
    if ((st = nrf_composite(raw, zc, zc_stride, a->n_rays, nc, a->white_bkgd, a->coarse_rgb, a->coarse_depth, acc_c, w_c,
                            stream)))
        return st;
    // render_fine (:267-334)
    if ((st = nrf_sample_fine(w_c, a->bins, 0, zc, zc_stride, a->u, a->u2, a->u_fine, a->n_rays, nc, ni, z_f, nullptr,
                              nullptr, stream)))
        return st;
    if ((st = mlp(a->packed_fine, z_f, nc + ni, nc + ni))) return st;
    return nrf_composite(raw, z_f, nc + ni, a->n_rays, nc + ni, a->white_bkgd, a->fine_rgb, a->fine_depth, a->fine_acc,
                         nullptr, stream);
}

NRF_EXPORT int nrf_render(const nrf_render_args* a, nrf_stream_t stream) { return render_path(a, MLP_FP32, stream); }

// Same path with both MLPs evaluated by nrf_mlp_forward_rays_bf16: packed_coarse / packed_fine point at
// nrf_pack_weights_bf16 streams (BASELINE configs[4]).
NRF_EXPORT int nrf_render_bf16(const nrf_render_args* a, nrf_stream_t stream) { return render_path(a, MLP_BF16, stream); }
NRF_EXPORT int nrf_render_bf16x3(const nrf_render_args* a, nrf_stream_t stream) { return render_path(a, MLP_BF16X3, stream); }
NRF_EXPORT int nrf_render_fp16x2(const nrf_render_args* a, nrf_stream_t stream) { return render_path(a, MLP_FP16X2, stream); }

// ---- nrf_render without an allocation anywhere (no counterpart in the reference) ---------------------------------------
// nrf_render's kernels need a workspace (raw, coarse weights, depths: ~4.4 KB per ray) that the caller normally brings.
// Callers that cannot allocate (HIP graph capture, serving loops with a frozen allocator, harnesses that forbid
// hipMalloc) get the same enqueue sequence over a file-scope device array instead: kStaticBytes per GPU the library
// runs on, held until the process ends.  `scratch` below is only the host-side stand-in of that array; its device
// address is asked for once per device (hipGetSymbolAddress with that device current: every GPU has its own copy at
// its own address) and only that pointer is ever handed to kernels.
namespace {
constexpr int kStaticMaxRays = NRF_RENDER_STATIC_MAX_RAYS;
constexpr int64_t kStaticBytes = Workspace(kStaticMaxRays, 64, 192).total;
constexpr int kMaxDevices = 64;
}  // namespace

static __device__ __align__(256) unsigned char scratch[kStaticBytes];

static std::atomic<void*> g_scratch_ptr[kMaxDevices];

NRF_EXPORT int nrf_render_static(const nrf_render_args* a, nrf_stream_t stream) {
    if (!a) return NRF_ERR_NULL;
    if (a->n_rays > kStaticMaxRays || a->n_coarse + (int64_t)a->n_importance > 256) return NRF_ERR_STATIC_LIMIT;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return NRF_ERR_RUNTIME;
    void* ptr = g_scratch_ptr[dev].load(std::memory_order_acquire);
    if (!ptr) {   // first call on this device (two threads racing here both get the same address)
        if (hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)) != hipSuccess || !ptr) return NRF_ERR_RUNTIME;
        g_scratch_ptr[dev].store(ptr, std::memory_order_release);
    }
    nrf_render_args b = *a;
    b.workspace = ptr;
    b.workspace_bytes = kStaticBytes;
    // Every workspace region is fully written by the stage that produces it before the stage that reads it
    // (raw by the MLP, weights / acc by compositing, depths by the resampler), for exactly n_rays rows: nothing has
    // to start at zero, so what the previous call left behind is never read.
    return render_path(&b, MLP_FP32, stream);
}
