// KiloNeRF backward: packed compositing, the per-cell tiny MLPs, and the colour regulariser.
//
// What autograd does for the reference's KiloNeRF training step (src/train/trainers/kilonerf.py:34-63 ->
// loss.backward() in src/train/trainers/trainer.py:68-75) through Renderer.forward
// (kilonerf/renderer/volumn_renderer.py:57-123) and BatchedKiloNeRF.forward (kilonerf/network.py:123-167):
// the reference materialises `W[model_indices]` per point and lets bmm's backward scatter-add 24.8 KB of
// gradient per POINT into the (4096, out, in) parameter tensors.  Here the points are already counting-sorted
// by cell (nrf_kilo_forward's workspace), so one wavefront owns one cell: it recomputes the forward chain of a
// 32-point tile in registers (cheaper than saving activations), runs the transposed chain for the data
// gradients (dX^T = W^T . dY^T, the same register-chained MFMA form as the forward), and accumulates the
// cell's weight gradients dW = dY^T . X in MFMA accumulators across all tiles of the cell.  The contraction of
// dW runs over POINTS, which sit on the lane axis of both operands, so each operand takes one trip through a
// wave-private LDS tile ([feature row][32 points], 36-float rows: conflict-free b128 reads) to turn it.
// Bias gradients of the two layers whose input has a spare encoding slot come for free: that slot is set to
// 1.0 in the turned copy.  A cell's segment is cut into jobs of a few tiles (one wavefront each, so one crowded
// cell does not serialise the launch); every job writes its partial gradient record and a second kernel adds a
// cell's records in job order into the 12 reference-layout tensors, writing every element exactly once (zeros
// for cells without samples): no atomics, no memset, and for a given point order (the forward's workspace)
// bit-reproducible.
//
// Packed compositing backward follows this repository's own forward rule (kilo.hip; the
// reference's is inside nerfacc: parity unpinned): the termination mask is a constant, w_i = T_i alpha_i.
#include "kilo_common.h"

namespace {

constexpr int kKiloTSteps = 12;                       // W1^T, Wf^T, Wv[:, :32]^T: 4 float4-per-lane pieces each
constexpr int kKiloCellTFloats = kKiloTSteps * 256;   // 3072
constexpr int kTileStride = 36;                       // floats per LDS row: 32 points + 4 pad
constexpr int kTileRows = 96;                         // 64 input rows + 32 gradient rows
constexpr int kRowDy = 64;

struct KiloGrad {
    float* t[12];  // KILO tensor order, reference layouts
};

// One job's partial gradients: the 12 tensors' per-cell slices back to back (kRecOff, kilo_common.h).
constexpr int kRecStride = 6224;                          // 64-byte multiple
static_assert(kRecFloats == 6212, "per-cell parameter count");

// single block: exclusive prefix of ceil(count / (32 * tiles_per_job)) over the cells
__global__ void kilo_job_scan_kernel(const int* __restrict__ seg_start, int ncell, int tiles_per_job,
                                     int* __restrict__ job_prefix, int* __restrict__ job_cell) {
    __shared__ int s_a[1024];
    const int tid = threadIdx.x;
    const int per = (ncell + 1023) / 1024, span = 32 * tiles_per_job;
    int jobs = 0;
    for (int k = 0; k < per; ++k) {
        const int c = tid * per + k;
        if (c < ncell) jobs += (seg_start[c + 1] - seg_start[c] + span - 1) / span;
    }
    s_a[tid] = jobs;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int a = tid >= off ? s_a[tid - off] : 0;
        __syncthreads();
        s_a[tid] += a;
        __syncthreads();
    }
    int b = s_a[tid] - jobs;
    for (int k = 0; k < per; ++k) {
        const int c = tid * per + k;
        if (c < ncell) {
            job_prefix[c] = b;
            const int nj = (seg_start[c + 1] - seg_start[c] + span - 1) / span;
            for (int q = 0; q < nj; ++q) job_cell[b + q] = c;   // the backward kernel's job -> cell table
            b += nj;
        }
    }
    if (tid == 1023) {
        job_prefix[ncell] = s_a[1023];
        job_prefix[ncell + 1] = 0;   // kilo_bwd_kernel's job counter
    }
}

// grads[t][cell] = sum of the cell's job records, in job order
__global__ __launch_bounds__(256) void kilo_grad_reduce_kernel(const float* __restrict__ partial,
                                                               const int* __restrict__ job_prefix, KiloGrad g) {
    const int cell = blockIdx.x;
    const int j0 = job_prefix[cell], j1 = job_prefix[cell + 1];
    // jobs outside, elements inside: the 25 loads a thread makes per job are independent and in flight together
    // (elements outside made every thread walk its jobs as 25 separate dependent-latency chains)
    constexpr int kPer = (kRecFloats + 255) / 256;
    float sum[kPer];
#pragma unroll
    for (int k = 0; k < kPer; ++k) sum[k] = 0.f;
    for (int j = j0; j < j1; ++j) {
        const float* rec = partial + int64_t(j) * kRecStride;
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            const int e = threadIdx.x + 256 * k;
            sum[k] += e < kRecFloats ? rec[e] : 0.f;
        }
    }
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
        const int e = threadIdx.x + 256 * k;
        if (e >= kRecFloats) break;
        const float acc = sum[k];
        int t = 0, off = 0, size = REC_OFF(1);
#define REC_CASE(k) if (e >= REC_OFF(k)) t = k, off = REC_OFF(k), size = REC_OFF(k + 1) - REC_OFF(k);
        REC_CASE(1) REC_CASE(2) REC_CASE(3) REC_CASE(4) REC_CASE(5) REC_CASE(6) REC_CASE(7) REC_CASE(8) REC_CASE(9)
        REC_CASE(10) REC_CASE(11)
#undef REC_CASE
        g.t[t][int64_t(cell) * size + (e - off)] = acc;
    }
}

// A operands of the transposed chain: lane (row = input feature, half h), MFMA t contracts over output feature
// acc_row(t, h) -- the order in which the accumulator layout presents dY^T as the B operand.
__global__ __launch_bounds__(256) void kilo_pack_t_kernel(KiloSrc s, int num_cells, const int* __restrict__ active,
                                                          const int* __restrict__ done, float* __restrict__ packed_t) {
    __shared__ float rec[kRecFloats];
    const int cell = blockIdx.x;
    if ((active && active[cell] == 0) || (done && done[cell] != 0)) return;   // see kilo_pack_kernel
    stage_cell_record<false>(s, cell, rec);
    __syncthreads();
    for (int idx = threadIdx.x; idx < kKiloCellTFloats; idx += 256) {
        const int e = idx & 3, lane = (idx >> 2) & 63, step = idx >> 8;
        const int h = lane >> 5, row = lane & 31, t = 4 * (step & 3) + e;
        const int k = acc_row(t, h);
        float v;
        if (step < 4) v = rec[REC_OFF(2) + k * 32 + row];        // pts_linears 1
        else if (step < 8) v = rec[REC_OFF(4) + k * 32 + row];   // feature_linear
        else v = rec[REC_OFF(8) + k * 59 + row];                 // view_linears, feature columns
        packed_t[int64_t(cell) * kKiloCellTFloats + idx] = v;
    }
}

__device__ __forceinline__ void wave_sync() {
    // LDS accesses of one wave execute in order; this only stops compiler reordering.
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// accumulator-layout block (register t of half h = feature acc_row(t, h), lane = point j) -> rows row0 .. row0+31
__device__ __forceinline__ void tile_store(float* T, int row0, const f32x16& v, int j, int h) {
#pragma unroll
    for (int t = 0; t < 16; ++t) T[(row0 + acc_row(t, h)) * kTileStride + j] = v[t];
}

// D[out][in] += sum over the 32 points of dY[out][pt] * X[in][pt]: MFMA m of the block contracts points
// (h * 16 + m), so a lane reads its row's 16 points of its half as 4 x b128.
__device__ __forceinline__ f32x16 wgrad_block(const float* T, int a_row0, int b_row0, f32x16 acc, int j, int h) {
    const float4* ap = reinterpret_cast<const float4*>(T + (a_row0 + j) * kTileStride + h * 16);
    const float4* bp = reinterpret_cast<const float4*>(T + (b_row0 + j) * kTileStride + h * 16);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 a = ap[q], b = bp[q];
        acc = NRF_MFMA(a.x, b.x, acc);
        acc = NRF_MFMA(a.y, b.y, acc);
        acc = NRF_MFMA(a.z, b.z, acc);
        acc = NRF_MFMA(a.w, b.w, acc);
    }
    return acc;
}

// tiny_layer with the A operands fetched from the cell's (L2-resident) record as they are needed: the backward
// kernel has no registers to keep them for the whole job.  `wp` already points at this lane's float4.
template <int NKB>
__device__ __forceinline__ f32x16 tiny_layer_g(const float4* wp, f32x16 init, const f32x16* x) {
#ifndef KB_NO_SCHED_BARRIER
    __builtin_amdgcn_sched_barrier(0);  // keep this layer's operand loads from being hoisted above the previous layer
#endif
    f32x16 y = init;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 a = wp[(kb * 4 + q) * 64];
            y = NRF_MFMA(a.x, x[kb][4 * q + 0], y);
            y = NRF_MFMA(a.y, x[kb][4 * q + 1], y);
            y = NRF_MFMA(a.z, x[kb][4 * q + 2], y);
            y = NRF_MFMA(a.w, x[kb][4 * q + 3], y);
        }
    return y;
}

// Sum over the 32 lanes of a lane half, on DPP (no LDS round trips: the job record reduces 100 values this way):
// Kogge-Stone inside each row of 16, then lane 15 of rows 0 / 2 into rows 1 / 3.  The total of half h is in lane
// 32 h + 31 ONLY; callers read it there.
__device__ __forceinline__ float half_sum_at31(float v) {
    v += nrf_dpp<0x111>(0.f, v);
    v += nrf_dpp<0x112>(0.f, v);
    v += nrf_dpp<0x114>(0.f, v);
    v += nrf_dpp<0x118>(0.f, v);
    v += nrf_dpp<0x142, 0xa>(0.f, v);
    return v;
}

#ifdef KB_STAMP   // diagnostic builds (tools/debug/kilo_bwd_phases.py): s_memtime ticks per phase, summed over all waves
__device__ unsigned long long g_kb_stamps[8];
#define KB_T(i)                                                                       \
    do {                                                                              \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();                 \
        if (lane == 0) atomicAdd(&g_kb_stamps[i], now_ - kb_last);                    \
        kb_last = now_;                                                               \
    } while (0)
#else
#define KB_T(i) do {} while (0)
#endif

template <bool INDEXED>
__global__ __launch_bounds__(256, 1) void kilo_bwd_kernel(const float* __restrict__ packed,
                                                          const float* __restrict__ packed_t, int ncell,
                                                          const int* __restrict__ seg_start,
                                                          const int* __restrict__ job_prefix, int tiles_per_job,
                                                          const int* __restrict__ perm, const float* __restrict__ pts,
                                                          const float* __restrict__ dirs, const int* __restrict__ dir_index,
                                                          const float4* __restrict__ grad_raw, float* __restrict__ partial,
                                                          int* __restrict__ job_counter, const int* __restrict__ job_cell) {
    // dynamic LDS: [4 waves][tile] then [4 waves][the job's cell record: forward operands, biases, head rows]
    extern __shared__ __attribute__((aligned(16))) float s_dyn[];
    const int lane = threadIdx.x & 63;
    const int j = lane & 31, h = lane >> 5;
    float* T = s_dyn + (threadIdx.x >> 6) * (kTileRows * kTileStride);
    float* Wl = s_dyn + 4 * (kTileRows * kTileStride) + (threadIdx.x >> 6) * kKiloCellFloats;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // encoding slot behind LDS row (32 * blk + j): the (t, h) with acc_row(t, h) == j
    const int slot_h = (j >> 2) & 1, slot_t = (j & 3) + 4 * (j >> 3);

    // Jobs are 1..tiles_per_job tiles long and a wave sees only three or four of them: dealt round-robin, the waves
    // with the long ones finish well after the rest.  Each wave takes the next job off a counter instead (the records
    // are indexed by job, so the result does not depend on who ran what).
    const int total_jobs = job_prefix[ncell];
#ifdef KB_STAMP
    unsigned long long kb_last = __builtin_amdgcn_s_memtime();
#endif
    // (no branch around the atomic: every lane adds, lane 0 adds 1 and the others 0 -- one atomic per wave after the
    // compiler's wave reduction, and the control flow stays uniform.)  The next job's number is asked for at the start
    // of the current job, so its round trip to L2 is over long before it is needed.
    int next_job = atomicAdd(job_counter, lane == 0 ? 1 : 0);
    for (;;) {
        // (taken in cell order: handing out the longest jobs first was measured and is slower, 0.198 vs 0.187 ms --
        // neighbouring jobs share a cell's records in L2, and the heavy cells then all run at once)
        const int job = __builtin_amdgcn_readfirstlane(next_job);
        if (job >= total_jobs) break;
        next_job = atomicAdd(job_counter, lane == 0 ? 1 : 0);
        KB_T(0);   // job fetch
        const int cell = job_cell[job];              // (one load; the search over job_prefix was twelve dependent ones)
        const int seg_e = seg_start[cell + 1];
        const int first = seg_start[cell] + (job - job_prefix[cell]) * 32 * tiles_per_job;
        const int job_e = first + 32 * tiles_per_job < seg_e ? first + 32 * tiles_per_job : seg_e;
        KB_T(1);   // cell search
        f32x16 dW0[2] = {zero16, zero16}, dW1 = zero16, dWf = zero16, dWv[2] = {zero16, zero16};
        f32x16 db1 = zero16, dbf = zero16, dsw = zero16, drw[3] = {zero16, zero16, zero16};
        float dsb = 0.f, drb[3] = {0.f, 0.f, 0.f};
        {
            const float* cw = packed + int64_t(cell) * kKiloCellFloats;
            const int last = job_e - 1;
            // Gather pipeline: perm -> (point, dir index, d raw) -> direction is three dependent loads; the next tile's
            // chain is issued piecewise between the phases of this tile (reads past the job's last tile are clamped to
            // its last element: valid memory, results unused).  The job's first chain starts here, ahead of the record
            // staging, so its last round trip (the direction) runs under the staging's.
            auto tile_id = [&](int t) { const int k = t + j; return unsigned(perm[k < last ? k : last]); };
            unsigned id_n = tile_id(first);
            unsigned di_n = INDEXED ? unsigned(dir_index[id_n]) : id_n;
            float pn[3] = {pts[id_n * 3u], pts[id_n * 3u + 1], pts[id_n * 3u + 2]};
            float4 gr_n = grad_raw[id_n];
            float dn[3];
            KB_T(3);   // first gather chain (index, point, d raw)
            // Stage the cell's forward record in this wave's LDS slice once per job, by LDS-DMA: 25 pieces of 1 KiB that
            // need no registers and are all in flight together (through registers it was three dependent rounds of
            // eight float4 -- the register file is full of accumulators -- and 18 % of the wave's time); the tile loop
            // then reads operands at LDS latency instead of L2 latency per layer.
            {
                wave_sync();   // the previous job's reads of this slice are done (same wave, in order)
                const char* src = reinterpret_cast<const char*>(cw) + lane * 16;
                const unsigned dst = __builtin_amdgcn_readfirstlane(unsigned(reinterpret_cast<uintptr_t>((lds_vptr)Wl)));
#pragma unroll
                for (int s = 0; s < kKiloSteps + 1; ++s) glds16(src + s * 1024, dst + s * 1024);
                const float4 tail1 = reinterpret_cast<const float4*>(cw)[kKiloSteps * 64 + 64 + (lane & 7)];   // last 128 B
                dn[0] = dirs[di_n * 3u], dn[1] = dirs[di_n * 3u + 1], dn[2] = dirs[di_n * 3u + 2];
                // the DMA pieces are invisible to the compiler's wait counting: this covers them (and the loads above)
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(dn[0]), "+v"(dn[1]), "+v"(dn[2]) : : "memory");
                if (lane < 8) reinterpret_cast<float4*>(Wl)[kKiloSteps * 64 + 64 + lane] = tail1;
                wave_sync();
            }
            KB_T(2);   // record staging
            const float4* w4_cell = reinterpret_cast<const float4*>(Wl) + lane;
            const float4* wt4_cell = reinterpret_cast<const float4*>(packed_t + int64_t(cell) * kKiloCellTFloats) + lane;
            const float* sm_cell = Wl + kKiloSteps * 256;
            for (int t0 = first; t0 < job_e; t0 += 32) {
                const bool valid = t0 + j < job_e;
                const float px = pn[0], py = pn[1], pz = pn[2], dx = dn[0], dy = dn[1], dz = dn[2];
                float4 gr = gr_n;
                if (!valid) gr = make_float4(0.f, 0.f, 0.f, 0.f);   // tail lanes repeat the last point: no gradient
                id_n = tile_id(t0 + 32);                            // next tile, stage 1
                // The cell's record is loop-invariant, and hoisting its 60 float4 out of the tile loop is exactly what the
                // register file cannot take: launder the offset so every tile re-reads it (LDS; L2 for the transposed part).
                int again = 0;
                asm volatile("" : "+v"(again));
                const float4* w4 = w4_cell + again;
                const float4* wt4 = wt4_cell + again;
                const float* sm = sm_cell + again;
                auto bias = [&](int l) { return *reinterpret_cast<const f32x16*>(sm + kKOffBias + l * 32 + h * 16); };

                // ---- forward chain again (kilo_mlp_kernel), keeping every layer's input ----
                f32x16 x[2];
                encode_pe_doubling<10>(px, py, pz, h, x);
                tile_store(T, 0, x[0], j, h);
                {
                    f32x16 x1 = x[1];
                    if (h == 1) x1[15] = 1.f;           // spare slot (n = 31, half 1): the ones column -> d(bias 0)
                    tile_store(T, 32, x1, j, h);
                }
                const f32x16 h0 = relu16(tiny_layer_g<2>(w4, bias(0), x));
                const f32x16 h1 = relu16(tiny_layer_g<1>(w4 + 8 * 64, bias(1), &h0));
                x[0] = tiny_layer_g<1>(w4 + 12 * 64, bias(2), &h1);                    // feature, no activation
                encode_pe_doubling<4>(dx, dy, dz, h, x + 1);
                const f32x16 hv = relu16(tiny_layer_g<2>(w4 + 16 * 64, bias(3), x));

                // next tile, stage 2 (its index arrived during the forward chain)
                di_n = INDEXED ? unsigned(dir_index[id_n]) : id_n;
                pn[0] = pts[id_n * 3u], pn[1] = pts[id_n * 3u + 1], pn[2] = pts[id_n * 3u + 2];
                gr_n = grad_raw[id_n];

                // ---- data gradients ----
                const f32x16 sw = *reinterpret_cast<const f32x16*>(sm + kKOffSigW + h * 16);
                f32x16 rw[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) rw[c] = *reinterpret_cast<const f32x16*>(sm + kKOffRgbW + c * 32 + h * 16);
                f32x16 d_hv;
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const float v = fmaf(rw[2][t], gr.z, fmaf(rw[1][t], gr.y, rw[0][t] * gr.x));
                    d_hv[t] = hv[t] > 0.f ? v : 0.f;
                    drw[0][t] = fmaf(gr.x, hv[t], drw[0][t]);
                    drw[1][t] = fmaf(gr.y, hv[t], drw[1][t]);
                    drw[2][t] = fmaf(gr.z, hv[t], drw[2][t]);
                    dsw[t] = fmaf(gr.w, h1[t], dsw[t]);
                }
                drb[0] += gr.x, drb[1] += gr.y, drb[2] += gr.z, dsb += gr.w;
                const f32x16 d_f = tiny_layer_g<1>(wt4 + 8 * 64, zero16, &d_hv);
                f32x16 d_h1 = tiny_layer_g<1>(wt4 + 4 * 64, zero16, &d_f);
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    d_h1[t] = h1[t] > 0.f ? fmaf(sw[t], gr.w, d_h1[t]) : 0.f;
                    db1[t] += d_h1[t];
                    dbf[t] += d_f[t];
                }
                f32x16 d_h0 = tiny_layer_g<1>(wt4, zero16, &d_h1);
#pragma unroll
                for (int t = 0; t < 16; ++t) d_h0[t] = h0[t] > 0.f ? d_h0[t] : 0.f;

                // next tile, stage 3
                dn[0] = dirs[di_n * 3u], dn[1] = dirs[di_n * 3u + 1], dn[2] = dirs[di_n * 3u + 2];

                KB_T(4);   // forward chain + data gradients
#ifndef KB_ABL_NOWGRAD  // timing-only ablation builds (tools/debug/kilo_bwd_time.py): results are wrong
                // ---- weight gradients: turn both operands through the LDS tile, one layer at a time ----
                tile_store(T, kRowDy, d_h0, j, h);                 // layer 0: X = encoding (+ ones), rows 0..63
                wave_sync();
                dW0[0] = wgrad_block(T, kRowDy, 0, dW0[0], j, h);
                dW0[1] = wgrad_block(T, kRowDy, 32, dW0[1], j, h);
                wave_sync();
                tile_store(T, 0, h0, j, h);                        // layer 1
                tile_store(T, kRowDy, d_h1, j, h);
                wave_sync();
                dW1 = wgrad_block(T, kRowDy, 0, dW1, j, h);
                wave_sync();
                tile_store(T, 0, h1, j, h);                        // feature layer
                tile_store(T, kRowDy, d_f, j, h);
                wave_sync();
                dWf = wgrad_block(T, kRowDy, 0, dWf, j, h);
                wave_sync();
                tile_store(T, 0, x[0], j, h);                      // view layer: X = [feature; direction encoding (+ ones)]
                {
                    f32x16 x1 = x[1];
                    if (h == 1) x1[13] = 1.f;           // spare slot (n = 13, half 1) -> d(view bias)
                    tile_store(T, 32, x1, j, h);
                }
                tile_store(T, kRowDy, d_hv, j, h);
                wave_sync();
                dWv[0] = wgrad_block(T, kRowDy, 0, dWv[0], j, h);
                dWv[1] = wgrad_block(T, kRowDy, 32, dWv[1], j, h);
                wave_sync();
#endif
                KB_T(5);   // weight gradients
            }
        }

        // ---- this job's record; accumulator register t of half h = output acc_row(t, h), lane j = input row
        //      (32 * blk + j) of the turned tile ----
        float* rec = partial + int64_t(job) * kRecStride;
        const int col0[2] = {pe_col(slot_t, slot_h, 10), pe_col(16 + slot_t, slot_h, 10)};
        const int colv = pe_col(slot_t, slot_h, 4);
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int o = acc_row(t, h);
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                if (col0[b] >= 0) rec[REC_OFF(0) + o * 63 + col0[b]] = dW0[b][t];
            }
            if (j == 31) rec[REC_OFF(1) + o] = dW0[1][t];                    // ones column of layer 0
            rec[REC_OFF(2) + o * 32 + j] = dW1[t];
            rec[REC_OFF(4) + o * 32 + j] = dWf[t];
            rec[REC_OFF(8) + o * 59 + j] = dWv[0][t];
            if (colv >= 0) rec[REC_OFF(8) + o * 59 + 32 + colv] = dWv[1][t];
            if (j == 29) rec[REC_OFF(9) + o] = dWv[1][t];                    // ones column of the view layer
            const float b1 = half_sum_at31(db1[t]), bf = half_sum_at31(dbf[t]), s_w = half_sum_at31(dsw[t]);
            const float r0 = half_sum_at31(drw[0][t]), r1 = half_sum_at31(drw[1][t]), r2 = half_sum_at31(drw[2][t]);
            if (j == 31) {
                rec[REC_OFF(3) + o] = b1;
                rec[REC_OFF(5) + o] = bf;
                rec[REC_OFF(6) + o] = s_w;
                rec[REC_OFF(10) + o] = r0;
                rec[REC_OFF(10) + 32 + o] = r1;
                rec[REC_OFF(10) + 64 + o] = r2;
            }
        }
        // (both halves hold every point once: half 0's total is the sum over the tile's points)
        const float sbs = half_sum_at31(dsb), rb0 = half_sum_at31(drb[0]), rb1 = half_sum_at31(drb[1]), rb2 = half_sum_at31(drb[2]);
        if (lane == 31) {
            rec[REC_OFF(7)] = sbs;
            rec[REC_OFF(11)] = rb0, rec[REC_OFF(11) + 1] = rb1, rec[REC_OFF(11) + 2] = rb2;
        }
        KB_T(6);   // the job's record
    }
}

// ---- packed compositing backward (forward: kilo_composite_kernel) ---------------------------
// Per ray: pass 1 repeats the forward sums; pass 2 walks the samples again with
//   dL/dw_i = G_i = g_rgb . (c_i - white) + g_acc + g_depth * (t_mid_i / den - [acc > eps] depth / den),
//   dL/dsigma_i = delta_i * ((1 - alpha_i) T_i G_i - sum_{k > i} G_k w_k)   for sigma_i > 0 and T_i >= ert_eps,
//   dL/dc_i = g_rgb * w_i * c_i (1 - c_i)                                    (c = sigmoid of the raw colour),
// the suffix sum taken as (total - inclusive prefix).  Terminated samples get zero.
__global__ __launch_bounds__(256) void kilo_composite_bwd_kernel(const float4* __restrict__ raw, const float* __restrict__ t_start,
                                                                 const float* __restrict__ t_end,
                                                                 const int* __restrict__ offsets, const int* __restrict__ counts,
                                                                 int64_t n_rays, float ert_eps, int white_bkgd,
                                                                 const float* __restrict__ g_rgb, const float* __restrict__ g_depth,
                                                                 const float* __restrict__ g_acc, float4* __restrict__ grad_raw) {
    const int lane = threadIdx.x & 63;
    const int64_t ray = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (ray >= n_rays) return;
    const int base = offsets[ray], n = counts[ray];
    if (n == 0) return;
    float carry = 1.f, sr = 0.f, sg = 0.f, sb = 0.f, sa = 0.f, sd = 0.f;
    for (int c0 = 0; c0 < n && carry >= ert_eps; c0 += 64) {
        const int s = c0 + lane;
        const bool in = s < n;
        const float4 rv = in ? raw[base + s] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float ts = in ? t_start[base + s] : 0.f, te = in ? t_end[base + s] : 0.f;
        const PackedChunk k = packed_chunk(rv, te - ts, in, carry);
        const float w = k.T >= ert_eps ? k.T * k.alpha : 0.f;
        sr += w * k.cr, sg += w * k.cg, sb += w * k.cb;
        sa += w;
        sd += w * 0.5f * (ts + te);
    }
    sr = wave_total(sr), sg = wave_total(sg), sb = wave_total(sb), sa = wave_total(sa), sd = wave_total(sd);
    const float gr = g_rgb[ray * 3], gg = g_rgb[ray * 3 + 1], gb = g_rgb[ray * 3 + 2];
    const float ga = g_acc ? g_acc[ray] : 0.f, gd = g_depth ? g_depth[ray] : 0.f;
    const float eps32 = 1.1920929e-07f;
    const float den = fmaxf(sa, eps32);
    const float dcoef = gd / den;
    const float dsub = sa > eps32 ? gd * sd / (den * den) : 0.f;
    const float cst = ga - dsub - (white_bkgd ? gr + gg + gb : 0.f);
    const float total = gr * sr + gg * sg + gb * sb + cst * sa + dcoef * sd;

    carry = 1.f;
    float prefix = 0.f;
    int c0 = 0;
    for (; c0 < n && carry >= ert_eps; c0 += 64) {
        const int s = c0 + lane;
        const bool in = s < n;
        const float4 rv = in ? raw[base + s] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float ts = in ? t_start[base + s] : 0.f, te = in ? t_end[base + s] : 0.f;
        const float delta = te - ts;
        const PackedChunk k = packed_chunk(rv, delta, in, carry);
        const float alpha = k.alpha, T = k.T, cr = k.cr, cg = k.cg, cb = k.cb;
        const bool live = in && T >= ert_eps;
        const float w = live ? T * alpha : 0.f;
        const float G = gr * cr + gg * cg + gb * cb + cst + dcoef * 0.5f * (ts + te);
        const float q = nrf_wave_scan_add(G * w);  // inclusive prefix of G w over the ray
        const float incl = prefix + q;
        prefix += nrf_read_lane63(q);
        float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live) {
            out.x = gr * w * cr * (1.f - cr);
            out.y = gg * w * cg * (1.f - cg);
            out.z = gb * w * cb * (1.f - cb);
            if (rv.w > 0.f) out.w = delta * ((1.f - alpha) * T * G - (total - incl));
        }
        if (in) grad_raw[base + s] = out;
    }
    for (int s = c0 + lane; s < n; s += 64) grad_raw[base + s] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// loss += coeff * sum(p^2); grad += 2 coeff p   (kilonerf/network.py:378-382 under l2_reg, trainers/kilonerf.py:52-55)
template <bool VEC4>  // VEC4: both pointers 16-byte aligned; the n % 4 tail is handled by the last scalar lanes
__global__ __launch_bounds__(256) void l2_reg_kernel(const float* __restrict__ p, int64_t n, float coeff,
                                                     float* __restrict__ grad, float* __restrict__ loss) {
    float acc = 0.f;
    const int64_t tid = int64_t(blockIdx.x) * blockDim.x + threadIdx.x, stride = int64_t(gridDim.x) * blockDim.x;
    const float c2 = 2.f * coeff;
    if (VEC4) {
        const int64_t n4 = n >> 2;
        const float4* p4 = reinterpret_cast<const float4*>(p);
        float4* g4 = reinterpret_cast<float4*>(grad);
        // four float4 per thread and iteration in flight; few blocks, because every block ends in one atomic on
        // the loss word and those serialise (1024 blocks: 40 us for this kernel, whatever its loop did)
        int64_t i = tid;
        for (; i + 3 * stride < n4; i += 4 * stride) {
            float4 v[4], g[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = p4[i + u * stride];
            if (grad) {
#pragma unroll
                for (int u = 0; u < 4; ++u) g[u] = g4[i + u * stride];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc = fmaf(v[u].x, v[u].x, acc), acc = fmaf(v[u].y, v[u].y, acc);
                acc = fmaf(v[u].z, v[u].z, acc), acc = fmaf(v[u].w, v[u].w, acc);
                if (grad) {
                    g[u].x = fmaf(c2, v[u].x, g[u].x), g[u].y = fmaf(c2, v[u].y, g[u].y);
                    g[u].z = fmaf(c2, v[u].z, g[u].z), g[u].w = fmaf(c2, v[u].w, g[u].w);
                    g4[i + u * stride] = g[u];
                }
            }
        }
        for (; i < n4; i += stride) {
            const float4 v = p4[i];
            acc = fmaf(v.x, v.x, acc), acc = fmaf(v.y, v.y, acc), acc = fmaf(v.z, v.z, acc), acc = fmaf(v.w, v.w, acc);
            if (grad) {
                float4 g = g4[i];
                g.x = fmaf(c2, v.x, g.x), g.y = fmaf(c2, v.y, g.y), g.z = fmaf(c2, v.z, g.z), g.w = fmaf(c2, v.w, g.w);
                g4[i] = g;
            }
        }
        for (int64_t i = (n4 << 2) + tid; i < n; i += stride) {
            const float v = p[i];
            acc = fmaf(v, v, acc);
            if (grad) grad[i] = fmaf(c2, v, grad[i]);
        }
    } else {
        for (int64_t i = tid; i < n; i += stride) {
            const float v = p[i];
            acc = fmaf(v, v, acc);
            if (grad) grad[i] = fmaf(c2, v, grad[i]);
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
    __shared__ float s_part[4];
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0 && loss) atomicAdd(loss, coeff * (s_part[0] + s_part[1] + s_part[2] + s_part[3]));
}

// ---- compact gradient exchange: the per-cell slices of the 12 tensors of a list of cells as records ----------
// One workgroup per listed cell; every slice is a contiguous run on both sides (record = the 12 slices back to
// back, kRecOff), so both directions are plain coalesced copies.
template <bool GATHER>
__global__ __launch_bounds__(256) void kilo_cells_copy_kernel(KiloSrc s, const int* __restrict__ cells, int num_cells,
                                                              float* __restrict__ records) {
    const int cell = cells[blockIdx.x];
    float* rec = records + int64_t(blockIdx.x) * kRecFloats;
    if (cell < 0 || cell >= num_cells) {   // a stale / foreign index: touch no tensor; a gathered record reads as zeros
        if (GATHER)
            for (int i = threadIdx.x; i < kRecFloats; i += 256) rec[i] = 0.f;
        return;
    }
#define COPY(K)                                                                                              \
    for (int i = threadIdx.x; i < REC_OFF(K + 1) - REC_OFF(K); i += 256) {                                   \
        float* g = const_cast<float*>(s.t[K]) + int64_t(cell) * (REC_OFF(K + 1) - REC_OFF(K)) + i;           \
        if (GATHER) rec[REC_OFF(K) + i] = *g; else *g = rec[REC_OFF(K) + i];                                 \
    }
    COPY(0) COPY(1) COPY(2) COPY(3) COPY(4) COPY(5) COPY(6) COPY(7) COPY(8) COPY(9) COPY(10) COPY(11)
#undef COPY
}

}  // namespace

// for nrf_kilo_forward_packing (kilo.hip): the transposed records of the cells with active[cell] != 0
void kilo_pack_t_launch(const float* const* tensors, int num_cells, const int* active, const int* done, float* packed_t,
                        hipStream_t stream) {
    KiloSrc s;
    for (int i = 0; i < 12; ++i) s.t[i] = tensors[i];
    kilo_pack_t_kernel<<<dim3(unsigned(num_cells)), 256, 0, stream>>>(s, num_cells, active, done, packed_t);
}


NRF_EXPORT int64_t nrf_kilo_record_floats(void) { return kRecFloats; }

static int cells_copy(bool gather, const float* const* tensors, const int32_t* cells, int n_cells, int num_cells,
                      float* records, nrf_stream_t stream) {
    if (n_cells < 0 || num_cells <= 0 || n_cells > num_cells) return NRF_ERR_SHAPE;
    if (n_cells == 0) return NRF_OK;
    if (!tensors || !cells || !records) return NRF_ERR_NULL;
    KiloSrc s;
    for (int i = 0; i < 12; ++i) {
        if (!tensors[i]) return NRF_ERR_NULL;
        s.t[i] = tensors[i];
    }
    nrf_launch_begin();
    if (gather)
        kilo_cells_copy_kernel<true><<<dim3(unsigned(n_cells)), 256, 0, static_cast<hipStream_t>(stream)>>>(s, cells, num_cells, records);
    else
        kilo_cells_copy_kernel<false><<<dim3(unsigned(n_cells)), 256, 0, static_cast<hipStream_t>(stream)>>>(s, cells, num_cells, records);
    return nrf_launch_status();
}

NRF_EXPORT int nrf_kilo_cells_gather(const float* const* tensors_host, const int32_t* cells, int n_cells, int num_cells,
                                     float* records, nrf_stream_t stream) {
    return cells_copy(true, tensors_host, cells, n_cells, num_cells, records, stream);
}

NRF_EXPORT int nrf_kilo_cells_scatter(const float* records, const int32_t* cells, int n_cells, int num_cells,
                                      float* const* tensors_host, nrf_stream_t stream) {
    return cells_copy(false, tensors_host, cells, n_cells, num_cells, const_cast<float*>(records), stream);
}

NRF_EXPORT int64_t nrf_kilo_packed_bwd_floats(int num_cells) {
    return num_cells <= 0 ? NRF_ERR_SHAPE : int64_t(num_cells) * kKiloCellTFloats;
}

NRF_EXPORT int nrf_kilo_pack_weights_transposed(const float* const* tensors, int num_cells, float* packed_t,
                                                nrf_stream_t stream) {
    if (!tensors || !packed_t) return NRF_ERR_NULL;
    if (num_cells <= 0) return NRF_ERR_SHAPE;
    KiloSrc s;
    for (int i = 0; i < 12; ++i) {
        if (!tensors[i]) return NRF_ERR_NULL;
        s.t[i] = tensors[i];
    }
    if (!nrf_aligned16(packed_t)) return NRF_ERR_ALIGN;
    nrf_launch_begin();
    kilo_pack_t_kernel<<<dim3(unsigned(num_cells)), 256, 0, static_cast<hipStream_t>(stream)>>>(s, num_cells, nullptr, nullptr, packed_t);
    return nrf_launch_status();
}

// tiles per job: 4 (128 points) unless that would make more than ~8192 + num_cells jobs
static int bwd_tiles_per_job(int64_t n_points) {
    const int64_t t = (n_points + 32 * 8192 - 1) / (32 * 8192);
    return int(t < 4 ? 4 : t);
}

// workspace of nrf_kilo_backward: job_prefix (ncell + 1), the job counter | job records | job -> cell table.
// The job bound assumes the shortest jobs (4 tiles), so it is monotone in n_points: a workspace sized for a sample
// capacity (nrf_kilo_train_workspace_bytes) is large enough for every smaller batch.
static int64_t bwd_prefix_bytes(int num_cells) { return a256((int64_t(num_cells) + 2) * 4); }
static int64_t bwd_max_jobs(int64_t n_points, int num_cells) {
    return n_points / (32 * 4) + (n_points < num_cells ? n_points : num_cells);
}
NRF_EXPORT int64_t nrf_kilo_backward_workspace_bytes(int64_t n_points, int num_cells) {
    if (n_points < 0 || num_cells <= 0) return NRF_ERR_SHAPE;
    const int64_t max_jobs = bwd_max_jobs(n_points, num_cells);
    return bwd_prefix_bytes(num_cells) + a256(max_jobs * kRecStride * 4) + a256(max_jobs * 4);
}

// Gradients of all 12 parameter tensors for the batch nrf_kilo_forward just evaluated: `forward_workspace` is that
// call's workspace, untouched since (it holds the points' cell order).  grads[i]: device tensor shaped like
// tensors[i]; every element is overwritten.
NRF_EXPORT int nrf_kilo_backward(const float* packed, const float* packed_t, int num_cells, const float* pts,
                                 const float* dirs, const int32_t* dir_index, int64_t n_points, const float* grad_raw,
                                 const void* forward_workspace, int64_t forward_workspace_bytes, float* const* grads,
                                 void* workspace, int64_t workspace_bytes, nrf_stream_t stream_) {
    if (n_points < 0 || num_cells <= 0 || num_cells > 8192 || n_points > (1LL << 28)) return NRF_ERR_SHAPE;
    if (!packed || !packed_t || !grads || !forward_workspace || !workspace) return NRF_ERR_NULL;
    if (n_points > 0 && (!pts || !dirs || !grad_raw)) return NRF_ERR_NULL;
    if (forward_workspace_bytes < nrf_kilo_forward_workspace_bytes(n_points, num_cells)) return NRF_ERR_WORKSPACE;
    if (workspace_bytes < nrf_kilo_backward_workspace_bytes(n_points, num_cells)) return NRF_ERR_WORKSPACE;
    if (!nrf_aligned16(packed) || !nrf_aligned16(packed_t) || !nrf_aligned16(grad_raw) || !nrf_aligned16(forward_workspace) ||
        !nrf_aligned16(workspace))
        return NRF_ERR_ALIGN;
    KiloGrad g;
    for (int i = 0; i < 12; ++i) {
        if (!grads[i]) return NRF_ERR_NULL;
        g.t[i] = grads[i];
    }
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const KiloSortWs ws = kilo_sort_ws(const_cast<void*>(forward_workspace), n_points, num_cells);
    int* job_prefix = static_cast<int*>(workspace);
    float* partial = reinterpret_cast<float*>(static_cast<char*>(workspace) + bwd_prefix_bytes(num_cells));
    int* job_cell = reinterpret_cast<int*>(static_cast<char*>(workspace) + bwd_prefix_bytes(num_cells) +
                                           a256(bwd_max_jobs(n_points, num_cells) * kRecStride * 4));
    const int tpj = bwd_tiles_per_job(n_points);
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (cus <= 0) cus = 256;
    constexpr int kBwdLds = 4 * (kTileRows * kTileStride + kKiloCellFloats) * 4;   // 158,208 B of the CU's 160 KiB
    static_assert(kKiloSmall == 288 && kBwdLds <= 160 * 1024, "LDS budget / staging of the 72 float4 behind the operands");
    nrf_launch_begin();
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&kilo_bwd_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            kBwdLds) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&kilo_bwd_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            kBwdLds) != hipSuccess)
        return NRF_ERR_LAUNCH;
    if (n_points == 0) {  // no forward ran: every segment is empty
        (void)hipMemsetAsync(job_prefix, 0, size_t(num_cells + 2) * 4, stream);
    } else {
        kilo_job_scan_kernel<<<1, 1024, 0, stream>>>(ws.seg_start, num_cells, tpj, job_prefix, job_cell);
        if (dir_index)
            kilo_bwd_kernel<true><<<dim3(unsigned(cus)), 256, kBwdLds, stream>>>(packed, packed_t, num_cells, ws.seg_start, job_prefix,
                                                                           tpj, ws.perm, pts, dirs, dir_index,
                                                                           reinterpret_cast<const float4*>(grad_raw), partial,
                                                                           job_prefix + num_cells + 1, job_cell);
        else
            kilo_bwd_kernel<false><<<dim3(unsigned(cus)), 256, kBwdLds, stream>>>(packed, packed_t, num_cells, ws.seg_start, job_prefix,
                                                                            tpj, ws.perm, pts, dirs, dir_index,
                                                                            reinterpret_cast<const float4*>(grad_raw), partial,
                                                                           job_prefix + num_cells + 1, job_cell);
    }
    kilo_grad_reduce_kernel<<<dim3(unsigned(num_cells)), 256, 0, stream>>>(partial, job_prefix, g);
    return nrf_launch_status();
}

NRF_EXPORT int nrf_kilo_composite_backward(const float* raw, const float* t_start, const float* t_end,
                                           const int32_t* offsets, const int32_t* counts, int64_t n_rays, float ert_eps,
                                           int white_bkgd, const float* grad_rgb, const float* grad_depth,
                                           const float* grad_acc, float* grad_raw, nrf_stream_t stream) {
    if (n_rays < 0) return NRF_ERR_SHAPE;
    if (n_rays == 0) return NRF_OK;
    if (!offsets || !counts || !grad_rgb) return NRF_ERR_NULL;
    if (!nrf_aligned16(raw) || !nrf_aligned16(grad_raw)) return NRF_ERR_ALIGN;
    const int64_t blocks = (n_rays + 3) / 4;
    if (blocks > 0x7fffffffLL) return NRF_ERR_SHAPE;
    nrf_launch_begin();
    kilo_composite_bwd_kernel<<<dim3((unsigned)blocks), 256, 0, static_cast<hipStream_t>(stream)>>>(
        reinterpret_cast<const float4*>(raw), t_start, t_end, offsets, counts, n_rays, ert_eps, white_bkgd, grad_rgb,
        grad_depth, grad_acc, reinterpret_cast<float4*>(grad_raw));
    return nrf_launch_status();
}

NRF_EXPORT int nrf_l2_reg(const float* params, int64_t n, float coeff, float* grad_accum, float* loss_accum,
                          nrf_stream_t stream) {
    if (n < 0) return NRF_ERR_SHAPE;
    if (n == 0) return NRF_OK;
    if (!params) return NRF_ERR_NULL;
    const int64_t blocks = (n + 255) / 256;
    nrf_launch_begin();
    const unsigned grid = unsigned(blocks < 512 ? blocks : 512);
    if (nrf_aligned16(params) && (!grad_accum || nrf_aligned16(grad_accum)))
        l2_reg_kernel<true><<<dim3(grid), 256, 0, static_cast<hipStream_t>(stream)>>>(params, n, coeff, grad_accum, loss_accum);
    else
        l2_reg_kernel<false><<<dim3(grid), 256, 0, static_cast<hipStream_t>(stream)>>>(params, n, coeff, grad_accum, loss_accum);
    return nrf_launch_status();
}

#ifdef KB_STAMP
NRF_EXPORT int nrf_debug_kb_stamps(unsigned long long* out8, int reset) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_kb_stamps), sizeof(g_kb_stamps)) != hipSuccess) return NRF_ERR_RUNTIME;
    if (reset) {
        unsigned long long z[8] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_kb_stamps), z, sizeof(z)) != hipSuccess) return NRF_ERR_RUNTIME;
    }
    return NRF_OK;
}
#endif
