// One KiloNeRF training forward + backward as ONE host call.
//
// The reference's step (src/train/trainers/trainer.py:68-75 around src/train/trainers/kilonerf.py:34-63) is a chain
// of about thirty small launches per 4096-ray batch here (march, route / sort, tiny MLPs, compositing, losses, their
// backward), most of them 5-15 us on the GPU.  Issued one by one from Python the host needs 10-20 us per launch, the
// GPU runs dry after the step's one host read (the sample count) and again in the middle of the chain, and the step
// takes 0.66 ms for 0.50 ms of GPU work.  This entry point issues the whole chain from C (2-3 us per launch), and
// puts the work that does not depend on the sample count -- repacking the cells the occupancy grid can reach from
// the updated parameters -- between the count's read-back and the wait for it.
#include <atomic>

#include "kilo_common.h"

namespace {

struct Carve {
    char* p;
    template <class T>
    T* take(int64_t n) {
        T* r = reinterpret_cast<T*>(p);
        p += a256(n * int64_t(sizeof(T)));
        return r;
    }
};

struct TrainWs {
    int32_t *counts, *offsets, *ray_idx;
    int64_t* total;
    float *pts, *t_start, *t_end, *raw, *g_raw, *g_rgb, *g_acc;
    void *march, *fwd, *bwd;
    int64_t march_bytes, fwd_bytes, bwd_bytes, bytes;
};

// max_samples sizes every per-sample buffer; the forward / backward workspaces are sized for it too (their layout is
// computed from the actual count, which is never larger)
int carve_train_ws(void* base, int64_t n_rays, int64_t max_samples, int num_cells, float t_near, float t_far, float step,
                   TrainWs& w) {
    w.march_bytes = nrf_kilo_march_workspace_bytes(n_rays, t_near, t_far, step);
    w.fwd_bytes = nrf_kilo_forward_workspace_bytes(max_samples, num_cells);
    w.bwd_bytes = nrf_kilo_backward_workspace_bytes(max_samples, num_cells);
    if (w.march_bytes < 0 || w.fwd_bytes < 0 || w.bwd_bytes < 0) return NRF_ERR_SHAPE;
    Carve c{static_cast<char*>(base)};
    w.total = c.take<int64_t>(1);
    w.counts = c.take<int32_t>(n_rays);
    w.offsets = c.take<int32_t>(n_rays);
    w.g_rgb = c.take<float>(n_rays * 3);
    w.g_acc = c.take<float>(n_rays);
    w.march = c.take<char>(w.march_bytes);
    w.pts = c.take<float>(max_samples * 3);
    w.ray_idx = c.take<int32_t>(max_samples);
    w.t_start = c.take<float>(max_samples);
    w.t_end = c.take<float>(max_samples);
    w.raw = c.take<float>(max_samples * 4);
    w.g_raw = c.take<float>(max_samples * 4);
    w.fwd = c.take<char>(w.fwd_bytes);
    w.bwd = c.take<char>(w.bwd_bytes);
    w.bytes = c.p - static_cast<char*>(base);
    return NRF_OK;
}

// per device: a pinned word for the sample count and the event that says it has arrived
struct CountSlot {
    std::atomic<int64_t*> host{nullptr};
    std::atomic<hipEvent_t> event{nullptr};
};
CountSlot g_slot[64];

int count_slot(int dev, int64_t*& host, hipEvent_t& ev) {
    if (dev < 0 || dev >= 64) return NRF_ERR_RUNTIME;
    host = g_slot[dev].host.load(std::memory_order_acquire);
    ev = g_slot[dev].event.load(std::memory_order_acquire);
    if (host && ev) return NRF_OK;
    int64_t* h = nullptr;
    hipEvent_t e = nullptr;
    if (hipHostMalloc(reinterpret_cast<void**>(&h), 64, hipHostMallocDefault) != hipSuccess) return NRF_ERR_RUNTIME;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
        (void)hipHostFree(h);
        return NRF_ERR_RUNTIME;
    }
    int64_t* expect_h = nullptr;
    if (!g_slot[dev].host.compare_exchange_strong(expect_h, h, std::memory_order_acq_rel)) {
        (void)hipHostFree(h);            // another thread got there first
        (void)hipEventDestroy(e);
        while (!(e = g_slot[dev].event.load(std::memory_order_acquire))) {}
        h = expect_h;
    } else {
        g_slot[dev].event.store(e, std::memory_order_release);
    }
    host = h, ev = e;
    return NRF_OK;
}

__global__ __launch_bounds__(256) void touched_kernel(const int* __restrict__ hist, int num_cells, int32_t* __restrict__ touched,
                                                      int32_t* __restrict__ live) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= num_cells) return;
    const int t = hist[i] != 0;
    if (touched) touched[i] = t;
    if (live && t) live[i] |= 3;   // both of nrf_kilo_clip_adam's bits: all twelve tensors of the cell have a gradient
}

// ---- optimiser pass over the cell grid ----------------------------------------------------------------------------
// torch.optim.Adam over all 25.4 M parameters moves 28 B per parameter, and for most of the grid it moves them to
// compute nothing: a cell that no batch has sampled since the optimiser was created has g = 0, exp_avg = 0 and
// exp_avg_sq = 0 in its eight unregularised tensors, and Adam's update of such an element is p - lr * (0 / eps) = p,
// bit for bit.  One block per cell: the unregularised slices only where bit 0 of `live` says the cell has ever had a
// gradient; the four colour-regularised slices in every cell when this call applies the regulariser (g = 2 l2 p
// everywhere), otherwise where bit 1 says they have a gradient or moments (every cell, once a regularised step ran).  Same arithmetic per element as clip_adam_kernel
// (train.hip); the regulariser folded in as in nrf_clip_adam_l2.
struct CellAdam {
    float* p[12];
    float* g[12];
    float* m[12];
    float* v[12];
    float grad_scale, clip, step_size, beta1, beta2, inv_bc2_sqrt, eps, reg_c2, reg_loss;
};

__device__ __forceinline__ void cell_adam_one(float& p, float& g, float& m, float& v, bool reg, const CellAdam& a, float& sq) {
    if (reg) {
        sq = fmaf(p, p, sq);
        g = fmaf(a.reg_c2, p, g);
    }
    float gi = g * a.grad_scale;
    gi = fminf(fmaxf(gi, -a.clip), a.clip);
    const float mi = m + (gi - m) * (1.f - a.beta1);
    const float vi = v * a.beta2 + (1.f - a.beta2) * gi * gi;
    m = mi;
    v = vi;
    const float denom = sqrtf(vi) * a.inv_bc2_sqrt + a.eps;
    p = p - a.step_size * (mi / denom);
}

template <int K, bool REG>
__device__ __forceinline__ void cell_adam_slice(const CellAdam& a, int cell, float& sq) {
    constexpr int n = REC_OFF(K + 1) - REC_OFF(K);
    const int64_t base = int64_t(cell) * n;
    float *p = a.p[K] + base, *g = a.g[K] + base, *m = a.m[K] + base, *v = a.v[K] + base;
    if constexpr (n % 4 == 0) {   // the slice starts on a 16-byte boundary when the tensor does (checked by the caller)
        float4 *p4 = reinterpret_cast<float4*>(p), *g4 = reinterpret_cast<float4*>(g), *m4 = reinterpret_cast<float4*>(m),
               *v4 = reinterpret_cast<float4*>(v);
        for (int i = threadIdx.x; i < n / 4; i += 256) {
            float4 pi = p4[i], gi = g4[i], mi = m4[i], vi = v4[i];
            cell_adam_one(pi.x, gi.x, mi.x, vi.x, REG, a, sq);
            cell_adam_one(pi.y, gi.y, mi.y, vi.y, REG, a, sq);
            cell_adam_one(pi.z, gi.z, mi.z, vi.z, REG, a, sq);
            cell_adam_one(pi.w, gi.w, mi.w, vi.w, REG, a, sq);
            p4[i] = pi, m4[i] = mi, v4[i] = vi;
            if (REG) g4[i] = gi;
        }
    } else {
        for (int i = threadIdx.x; i < n; i += 256) {
            float pi = p[i], gi = g[i], mi = m[i], vi = v[i];
            cell_adam_one(pi, gi, mi, vi, REG, a, sq);
            p[i] = pi, m[i] = mi, v[i] = vi;
            if (REG) g[i] = gi;
        }
    }
}

template <bool REG>
__global__ __launch_bounds__(256) void kilo_adam_kernel(CellAdam a, const int32_t* __restrict__ live, float* __restrict__ loss) {
    const int cell = blockIdx.x;
    float sq = 0.f;
    const int mask = live ? live[cell] : 3;   // bit 0: the eight unregularised tensors, bit 1: the four regularised ones
    if (mask & 1) {
        cell_adam_slice<0, false>(a, cell, sq);
        cell_adam_slice<1, false>(a, cell, sq);
        cell_adam_slice<2, false>(a, cell, sq);
        cell_adam_slice<3, false>(a, cell, sq);
        cell_adam_slice<4, false>(a, cell, sq);
        cell_adam_slice<5, false>(a, cell, sq);
        cell_adam_slice<6, false>(a, cell, sq);
        cell_adam_slice<7, false>(a, cell, sq);
    }
    if (REG || (mask & 2)) {   // without a regulariser in this call: only where they have a gradient or moments (bit 1)
        cell_adam_slice<8, REG>(a, cell, sq);
        cell_adam_slice<9, REG>(a, cell, sq);
        cell_adam_slice<10, REG>(a, cell, sq);
        cell_adam_slice<11, REG>(a, cell, sq);
    }
    if (REG && loss) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) sq += __shfl_xor(sq, off);
        __shared__ float s_part[4];
        if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = sq;
        __syncthreads();
        if (threadIdx.x == 0) unsafeAtomicAdd(loss, a.reg_loss * (s_part[0] + s_part[1] + s_part[2] + s_part[3]));
    }
}

}  // namespace

NRF_EXPORT int64_t nrf_kilo_train_workspace_bytes(int64_t n_rays, int64_t max_samples, int num_cells, float t_near, float t_far,
                                                  float step) {
    if (n_rays <= 0 || max_samples < 0 || num_cells <= 0) return NRF_ERR_SHAPE;
    TrainWs w;
    const int rc = carve_train_ws(nullptr, n_rays, max_samples, num_cells, t_near, t_far, step, w);
    return rc != NRF_OK ? rc : w.bytes;
}

NRF_EXPORT int nrf_kilo_train_forward_backward(const nrf_kilo_train_args* a, nrf_stream_t stream_) {
    if (!a) return NRF_ERR_NULL;
    if (a->n_rays <= 0 || a->max_samples < 0 || a->num_cells <= 0) return NRF_ERR_SHAPE;
    if (!a->rays_o || !a->rays_d || !a->occ || !a->occ_res || !a->aabb || !a->res || !a->target_rgb || !a->tensors ||
        !a->grads || !a->packed || !a->packed_t || !a->rgb_map || !a->depth_map || !a->acc_map || !a->loss_accum ||
        !a->n_samples || !a->workspace)
        return NRF_ERR_NULL;
    if (!nrf_aligned16(a->workspace)) return NRF_ERR_ALIGN;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    TrainWs w;
    int rc = carve_train_ws(a->workspace, a->n_rays, a->max_samples, a->num_cells, a->t_near, a->t_far, a->step, w);
    if (rc != NRF_OK) return rc;
    if (a->workspace_bytes < w.bytes) return NRF_ERR_WORKSPACE;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return NRF_ERR_RUNTIME;
    int64_t* count_host = nullptr;
    hipEvent_t arrived = nullptr;
    if ((rc = count_slot(dev, count_host, arrived)) != NRF_OK) return rc;

    // pass 1 of the march, the read-back of the sample count, and -- while that travels -- the records of the cells
    // the occupancy grid can reach, from the parameters as the optimiser left them
    rc = nrf_kilo_march_count(a->rays_o, a->rays_d, a->n_rays, a->occ, a->occ_res, a->aabb, a->t_near, a->t_far, a->step,
                              w.counts, w.offsets, w.total, w.march, w.march_bytes, stream);
    if (rc != NRF_OK) return rc;
    // The host spins on the pinned word the copy writes (a blocking hipEventSynchronize sleeps, and its wake-up costs
    // the GPU tens of idle microseconds per step).  Nothing guarantees that an 8-byte device-to-host copy lands as ONE
    // write, so the word is seeded with all ones and read as two 32-bit halves: it counts as arrived only when NEITHER
    // half still holds the seed (high half first would otherwise read as 0x00000000FFFFFFFF = 4.29e9 samples).  A
    // count whose low half really is 0xFFFFFFFF is picked up by the event query below instead.
    volatile uint32_t* half = reinterpret_cast<volatile uint32_t*>(count_host);
    half[0] = 0xFFFFFFFFu;
    half[1] = 0xFFFFFFFFu;
    if (hipMemcpyAsync(count_host, w.total, sizeof(int64_t), hipMemcpyDeviceToHost, stream) != hipSuccess ||
        hipEventRecord(arrived, stream) != hipSuccess)
        return NRF_ERR_RUNTIME;
    if (a->reach) {
        rc = nrf_kilo_pack_weights_cells(a->tensors, a->num_cells, a->reach, a->packed, a->packed_t, stream);
        if (rc != NRF_OK) return rc;
    }
    (void)hipMemsetAsync(a->loss_accum, 0, sizeof(float), stream);
    for (unsigned spins = 1; half[0] == 0xFFFFFFFFu || half[1] == 0xFFFFFFFFu; ++spins) {
        __builtin_ia32_pause();
        if ((spins & 0xfff) == 0) {   // every few microseconds: has the stream failed, or finished without the word showing up?
            const hipError_t q = hipEventQuery(arrived);
            if (q == hipSuccess) {
                if (hipEventSynchronize(arrived) != hipSuccess) return NRF_ERR_RUNTIME;
                break;
            }
            if (q != hipErrorNotReady) return NRF_ERR_RUNTIME;
        }
    }
    (void)hipGetLastError();   // hipErrorNotReady from the queries is not an error
    const int64_t n = int64_t((uint64_t(half[1]) << 32) | uint64_t(half[0]));
    if (n < 0) return NRF_ERR_RUNTIME;
    *a->n_samples = n;
    if (n > a->max_samples) return NRF_ERR_WORKSPACE;   // *n_samples says how many the batch has: size for that, call again

    if (n > 0) {
        rc = nrf_kilo_march_write(a->rays_o, a->rays_d, a->n_rays, a->occ, a->occ_res, a->aabb, a->t_near, a->t_far, a->step,
                                  w.offsets, w.pts, w.ray_idx, w.t_start, w.t_end, w.counts, w.march, w.march_bytes, stream);
        if (rc != NRF_OK) return rc;
        rc = nrf_kilo_forward_packing(a->tensors, a->reach, a->packed, a->packed_t, a->num_cells, a->aabb, a->res, w.pts,
                                      a->rays_d, w.ray_idx, n, w.raw, w.fwd, w.fwd_bytes, stream);
        if (rc != NRF_OK) return rc;
    }
    rc = nrf_kilo_composite(w.raw, w.t_start, w.t_end, w.offsets, w.counts, a->n_rays, a->ert_eps, a->white_bkgd, a->rgb_map,
                            a->depth_map, a->acc_map, stream);
    if (rc != NRF_OK) return rc;
    // trainers/kilonerf.py:43-61: mse(rgb) [+ mse(alpha) in the distillation stage]; the regulariser rides in the optimiser
    rc = nrf_mse_grad(a->rgb_map, a->target_rgb, a->n_rays * 3, w.g_rgb, a->loss_accum, stream);
    if (rc != NRF_OK) return rc;
    if (a->target_alpha) {
        rc = nrf_mse_grad(a->acc_map, a->target_alpha, a->n_rays, w.g_acc, a->loss_accum, stream);
        if (rc != NRF_OK) return rc;
    }
    if (n > 0) {
        rc = nrf_kilo_composite_backward(w.raw, w.t_start, w.t_end, w.offsets, w.counts, a->n_rays, a->ert_eps, a->white_bkgd,
                                         w.g_rgb, nullptr, a->target_alpha ? w.g_acc : nullptr, w.g_raw, stream);
        if (rc != NRF_OK) return rc;
    }
    rc = nrf_kilo_backward(a->packed, a->packed_t, a->num_cells, w.pts, a->rays_d, w.ray_idx, n, w.g_raw, w.fwd, w.fwd_bytes,
                           a->grads, w.bwd, w.bwd_bytes, stream);
    if (rc != NRF_OK) return rc;
    if (a->touched || a->live) {
        if (n > 0) {
            nrf_launch_begin();
            touched_kernel<<<dim3(unsigned((a->num_cells + 255) / 256)), 256, 0, stream>>>(
                kilo_sort_ws(w.fwd, n, a->num_cells).hist, a->num_cells, a->touched, a->live);
            if ((rc = nrf_launch_status()) != NRF_OK) return rc;
        } else if (a->touched) {
            (void)hipMemsetAsync(a->touched, 0, size_t(a->num_cells) * 4, stream);
        }
    }
    return NRF_OK;
}

NRF_EXPORT int nrf_kilo_clip_adam(float* const* params, float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                                  int num_cells, const int32_t* live, float grad_scale, float clip_value, float lr, float beta1,
                                  float beta2, float eps, int step, float reg_grad_coeff, float reg_loss_coeff,
                                  float* loss_accum, nrf_stream_t stream) {
    if (!params || !grads || !exp_avg || !exp_avg_sq) return NRF_ERR_NULL;
    if (num_cells <= 0 || step < 1 || !(clip_value > 0.f)) return NRF_ERR_SHAPE;
    CellAdam a;
    for (int i = 0; i < 12; ++i) {
        if (!params[i] || !grads[i] || !exp_avg[i] || !exp_avg_sq[i]) return NRF_ERR_NULL;
        if (!nrf_aligned16(params[i]) || !nrf_aligned16(grads[i]) || !nrf_aligned16(exp_avg[i]) || !nrf_aligned16(exp_avg_sq[i]))
            return NRF_ERR_ALIGN;
        a.p[i] = params[i], a.g[i] = grads[i], a.m[i] = exp_avg[i], a.v[i] = exp_avg_sq[i];
    }
    const double bc1 = 1.0 - pow(double(beta1), step), bc2 = 1.0 - pow(double(beta2), step);
    a.grad_scale = grad_scale, a.clip = clip_value, a.step_size = float(double(lr) / bc1), a.beta1 = beta1, a.beta2 = beta2;
    a.inv_bc2_sqrt = float(1.0 / sqrt(bc2)), a.eps = eps, a.reg_c2 = 2.f * reg_grad_coeff, a.reg_loss = reg_loss_coeff;
    const bool reg = reg_grad_coeff != 0.f || (reg_loss_coeff != 0.f && loss_accum);
    nrf_launch_begin();
    if (reg)
        kilo_adam_kernel<true><<<dim3(unsigned(num_cells)), 256, 0, static_cast<hipStream_t>(stream)>>>(a, live, loss_accum);
    else
        kilo_adam_kernel<false><<<dim3(unsigned(num_cells)), 256, 0, static_cast<hipStream_t>(stream)>>>(a, live, nullptr);
    return nrf_launch_status();
}
