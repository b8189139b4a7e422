// KiloNeRF: 16^3 tiny MLPs routed by cell, occupancy-grid ray marching, packed compositing.
//
// Replaces (reference paths): kilonerf/network.py:314-334 + :246-253 (_get_index, flat index),
// :123-167 (BatchedKiloNeRF.forward: per point it gathers that point's full weight set,
// 24.8 KB of HBM traffic per POINT, and does 1-row bmm's), :223-311 (Network.forward), and the
// nerfacc calls of kilonerf/renderer/volumn_renderer.py:78-86, 107-113 (occupancy-grid sampling,
// early ray termination, packed rendering).
//
// MI355X shape:
//   * points are counting-sorted by cell (LDS histogram -> 4096-entry scan -> scatter), so a
//     wavefront works on 32 points of ONE cell: the tiny MLP becomes 96 chained
//     v_mfma_f32_32x32x2_f32 (same transposed register chain as the big MLP, one 32-row output
//     block per layer) whose A operands -- the cell's whole weight set, 24 float4 per lane --
//     are loaded ONCE per job and stay in registers for up to kJobTiles tiles;
//   * ray marching is one wavefront per ray, 64 steps per iteration, ballot + popcount
//     compaction, so the packed sample list is written in order without atomics;
//   * packed compositing is the same wave scan as composite.hip over a ragged segment.
// nerfacc is absent and unpinned: the sampling / termination / packed-rendering rules below
// are this repo's own specification, checked against oracle/kilo_oracle.py.
#include <cmath>

#include "kilo_common.h"
#include "split2h.h"
#include "split3.h"

namespace {

// `active` / `done` (optional, one int per cell): only cells with active != 0 and done == 0 are packed, the others'
// records are left as they are (the training step packs the cells its batch can reach, nrf_kilo_pack_weights_cells,
// nrf_kilo_forward_packing)
__global__ __launch_bounds__(256) void kilo_pack_kernel(KiloSrc s, int num_cells, const int* __restrict__ active,
                                                        const int* __restrict__ done, float* __restrict__ packed) {
    __shared__ float rec[kRecFloats];
    const int cell = blockIdx.x;
    if ((active && active[cell] == 0) || (done && done[cell] != 0)) return;
    stage_cell_record(s, cell, rec);
    __syncthreads();
    for (int idx = threadIdx.x; idx < kKiloCellFloats; idx += 256) {
        float v = 0.f;
        if (idx < kKiloSteps * 256) {
            const int e = idx & 3, lane = (idx >> 2) & 63, step = idx >> 8;
            const int h = lane >> 5, row = lane & 31, q = step & 3, t = 4 * q + e;
            if (step < 8) {  // pts0: 63 encoding columns, kernel slot order
                const int c = pe_col((step >> 2) * 16 + t, h, 10);
                if (c >= 0) v = rec[REC_OFF(0) + row * 63 + c];
            } else if (step < 12) {
                v = rec[REC_OFF(2) + row * 32 + acc_row(t, h)];
            } else if (step < 16) {
                v = rec[REC_OFF(4) + row * 32 + acc_row(t, h)];
            } else if (step < 20) {  // view layer, feature columns 0..31
                v = rec[REC_OFF(8) + row * 59 + acc_row(t, h)];
            } else {  // view layer, direction encoding columns 32..58
                const int c = pe_col(t, h, 4);
                if (c >= 0) v = rec[REC_OFF(8) + row * 59 + 32 + c];
            }
        } else {
            const int i = idx - kKiloSteps * 256;
            if (i < 128) {
                const int layer = i >> 5, h = (i >> 4) & 1, r = i & 15;
                const int off = layer == 0 ? REC_OFF(1) : layer == 1 ? REC_OFF(3) : layer == 2 ? REC_OFF(5) : REC_OFF(9);
                v = rec[off + acc_row(r, h)];
            } else if (i < 160) {
                const int h = ((i - 128) >> 4) & 1, r = i & 15;
                v = rec[REC_OFF(6) + acc_row(r, h)];
            } else if (i < 256) {
                const int j = i - 160, c = j >> 5, h = (j >> 4) & 1, r = j & 15;
                v = rec[REC_OFF(10) + c * 32 + acc_row(r, h)];
            } else if (i == kKOffSigB) {
                v = rec[REC_OFF(7)];
            } else if (i < kKOffRgbB + 3) {
                v = rec[REC_OFF(11) + (i - kKOffRgbB)];
            }
        }
        packed[int64_t(cell) * kKiloCellFloats + idx] = v;
    }
}

// kilonerf/network.py:322-334, op for op in fp32: (p - min) / (max - min) * res, clamp(min=0),
// min(res - 1), truncate.  Bit-exact (IEEE subtract / divide / multiply, no contraction).
__device__ __forceinline__ int route_cell(const Grid& g, float x, float y, float z) {
    const float p[3] = {x, y, z};
    int idx[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float n = __fdiv_rn(__fsub_rn(p[a], g.mn[a]), __fsub_rn(g.mx[a], g.mn[a]));  // (the kernel is bound by
        // its loads: the exact-reciprocal form the march uses for power-of-two boxes measured no faster here)
        float s = __fmul_rn(n, float(g.res[a]));
        s = s < 0.f ? 0.f : s;                  // torch.clamp(min=0): NaN stays NaN -> (int) 0 below
        const float hi = float(g.res[a]) - 1.0f;
        s = s > hi ? hi : s;
        idx[a] = int(s);
        if (!(s == s)) idx[a] = 0;
    }
    return (idx[0] * g.res[1] + idx[1]) * g.res[2] + idx[2];  // :251-253
}

// Consecutive points are consecutive samples of neighbouring rays, so a CONTIGUOUS chunk of kSortChunk points
// touches only a dozen cells: a block owns one chunk, histograms it in LDS and sends one global atomic per touched
// cell (a grid-stride block would see every occupied cell and send ~900).
constexpr int kSortChunk = 4096;
// the routing kernel keeps nothing per point in registers, so its chunk is a run-time multiple of 256: large for a
// frame's 27 M points (zeroing and flushing the 4096-entry LDS histogram costs a block as much as routing 4096
// points), small for a training batch (176 k points in 16384-point chunks are 11 workgroups on 256 CUs: 14 -> 49 us)
static int route_chunk(int64_t n) { return n >= (int64_t(1) << 22) ? 16384 : n >= (int64_t(1) << 19) ? 4096 : 1024; }
__global__ __launch_bounds__(256) void kilo_route_kernel(Grid g, const float* __restrict__ pts, int64_t n, int chunk,
                                                         int* __restrict__ cell, int* __restrict__ hist) {
    extern __shared__ int s_hist[];
    const int ncell = g.res[0] * g.res[1] * g.res[2];
    if (hist)
        for (int i = threadIdx.x; i < ncell; i += blockDim.x) s_hist[i] = 0;
    __syncthreads();
    const int64_t begin = int64_t(blockIdx.x) * chunk;
    const int lane = threadIdx.x & 63;
    for (int k = 0; k < chunk / 256; ++k) {  // every lane runs every round: the run detection needs full waves
        const int64_t i = begin + k * 256 + threadIdx.x;
        const bool in = i < n;
        int c = -1;
        if (in) {
            struct P3 { float x, y, z; };   // one 12-byte load per lane instead of three strided 4-byte ones
            const P3 p = reinterpret_cast<const P3*>(pts)[i];
            c = route_cell(g, p.x, p.y, p.z);
            cell[i] = c;
        }
        if (hist) {  // one LDS atomic per run of equal cells (per point they collide 64-wide: a ray stays in a cell)
            const int prev = __shfl_up(c, 1);
            const bool head = lane == 0 || c != prev;
            const unsigned long long heads = __ballot(head);
            const unsigned long long after = lane == 63 ? 0ull : (heads >> (lane + 1)) << (lane + 1);
            const int next_head = after ? __ffsll((long long)after) - 1 : 64;
            if (head && c >= 0) atomicAdd(&s_hist[c], next_head - lane);
        }
    }
    __syncthreads();
    if (hist)
        for (int i = threadIdx.x; i < ncell; i += blockDim.x)
            if (s_hist[i]) atomicAdd(&hist[i], s_hist[i]);
}

// single block: exclusive scan of the per-cell counts -> segment starts, cursors and the job prefix
__global__ void kilo_cell_scan_kernel(const int* __restrict__ hist, int ncell, int job_tiles, int* __restrict__ seg_start,
                                      int* __restrict__ cursor, int* __restrict__ job_prefix) {
    __shared__ int s_a[1024], s_b[1024];
    const int tid = threadIdx.x;
    const int per = (ncell + 1023) / 1024;
    int cnt = 0, jobs = 0;
    for (int k = 0; k < per; ++k) {
        const int c = tid * per + k;
        if (c < ncell) {
            cnt += hist[c];
            jobs += (hist[c] + 32 * job_tiles - 1) / (32 * job_tiles);
        }
    }
    s_a[tid] = cnt, s_b[tid] = jobs;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int a = tid >= off ? s_a[tid - off] : 0, b = tid >= off ? s_b[tid - off] : 0;
        __syncthreads();
        s_a[tid] += a, s_b[tid] += b;
        __syncthreads();
    }
    int a = s_a[tid] - cnt, b = s_b[tid] - jobs;  // exclusive
    for (int k = 0; k < per; ++k) {
        const int c = tid * per + k;
        if (c < ncell) {
            seg_start[c] = a, cursor[c] = a, job_prefix[c] = b;
            a += hist[c];
            b += (hist[c] + 32 * job_tiles - 1) / (32 * job_tiles);
        }
    }
    if (tid == 1023) seg_start[ncell] = s_a[1023], job_prefix[ncell] = s_b[1023];
}

// Counting-sort scatter, same chunking.  Per-point atomics on cursor[cell] would collide 64-wide and a return-value
// atomic on a device-scope counter is the expensive operation here (1.1 M of them at one per run of equal cells per
// wavefront: 0.42 ms per frame).  A block (1) counts its chunk per cell in LDS -- one LDS atomic per run: consecutive
// points stay in a cell for ~25 steps, the head lane of a run adds the run's length --, (2) reserves the chunk's
// range of every touched cell with ONE global atomic each (~12 per 4096 points) and (3) hands out positions from
// the LDS cursors, again one LDS atomic per run.  The order inside a cell is whatever order the atomics land in,
// as before; the cells are held in registers between (1) and (3).
__global__ __launch_bounds__(256) void kilo_scatter_kernel(const int* __restrict__ cell, int64_t n, int ncell,
                                                           int* __restrict__ cursor, int* __restrict__ perm) {
    extern __shared__ int s_cur[];
    constexpr int kRounds = kSortChunk / 256;
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < ncell; i += 256) s_cur[i] = 0;
    __syncthreads();
    const int64_t begin = int64_t(blockIdx.x) * kSortChunk;
    int c[kRounds], head_lane[kRounds], run_len[kRounds];
#pragma unroll
    for (int k = 0; k < kRounds; ++k) {  // every lane runs every round: shuffles and ballots need full waves
        const int64_t i = begin + k * 256 + threadIdx.x;
        c[k] = i < n ? cell[i] : -1;
    }
#pragma unroll
    for (int k = 0; k < kRounds; ++k) {
        const int prev = __shfl_up(c[k], 1);
        const bool head = lane == 0 || c[k] != prev;
        const unsigned long long heads = __ballot(head);
        const unsigned long long upto = heads & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));
        head_lane[k] = 63 - __clzll(upto);                                   // last head at or before this lane
        const unsigned long long after = lane == 63 ? 0ull : (heads >> (lane + 1)) << (lane + 1);
        const int next_head = after ? __ffsll((long long)after) - 1 : 64;    // first head after this lane
        run_len[k] = head && c[k] >= 0 ? next_head - lane : 0;               // > 0 on the head lane of a run
        if (run_len[k]) atomicAdd(&s_cur[c[k]], run_len[k]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ncell; i += 256)
        if (s_cur[i]) s_cur[i] = atomicAdd(&cursor[i], s_cur[i]);            // count -> first position of the chunk's range
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kRounds; ++k) {
        int base = 0;
        if (run_len[k]) base = atomicAdd(&s_cur[c[k]], run_len[k]);
        base = __shfl(base, head_lane[k]);
        if (c[k] >= 0) perm[base + (lane - head_lane[k])] = int(begin + k * 256 + threadIdx.x);
    }
}

// ---- direction encodings once per RAY (nrf_kilo_encode_dirs) --------------------------------------------------------
// The 27 direction features (network.py:243: viewdirs expanded per point, then embedded per point) depend on the ray
// only, and a ray puts ~25 consecutive samples into a cell: evaluated per point they were 3 of the 11 trig iterations
// of a tile (~100 of its ~540 VALU instructions, plus an operand split in the fp16 form).  The table holds, per
// direction row, exactly what the kernels' lanes would have computed: 2 lane halves x 16 dwords = 128 B --
//   fp32 form:   the 16 slots of encode_pe_doubling<4> (the B operands of the view layer's last k-block);
//   fp16x2 form: split_block_f16 of those slots, [k16-step][term][4 dwords], and behind the table one float per row,
//                the range guard's maximum of the row's operands;
// produced by the same device functions, so a table-fed tile is bit-identical to a self-encoding one.  A tile loads its
// 64 B per lane at its top (four global_load_dwordx4; neighbouring points share rays, so the rows come from L1 / L2) and
// uses them three layers later.
constexpr int kDirRowDwordsMax = 48;   // the bf16x3 form's rows: [k16-step][term] x 4 dwords per lane half (below)
// one size for every form: rows of the widest form, then the fp16 form's guard column
__attribute__((unused)) inline int64_t dir_guard_offset(int64_t n_dirs) { return a256(n_dirs * kDirRowDwordsMax * 4); }
__attribute__((unused)) inline int64_t dir_table_bytes(int64_t n_dirs) { return dir_guard_offset(n_dirs) + a256(n_dirs * 4); }

template <int KIND>   // 0: fp32 slots, 1: fp16x2 fragments + guard, 2: bf16x3 fragments (48 dwords per row)
__global__ __launch_bounds__(256) void kilo_dir_table_kernel(const float* __restrict__ dirs, int64_t n_dirs,
                                                             unsigned* __restrict__ table, float* __restrict__ guard) {
    const int lane = threadIdx.x & 63;
    const int j = lane & 31, h = lane >> 5;
    const int64_t row = (int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6)) * 32 + j;
    const int64_t src = row < n_dirs ? row : n_dirs - 1;   // every lane runs (the encoding swaps between lane halves)
    f32x16 e[1];
    encode_pe_doubling<4>(dirs[src * 3], dirs[src * 3 + 1], dirs[src * 3 + 2], h, e);
    constexpr int NQ = KIND == 2 ? 6 : 4;
    u32x4 out[NQ];
    if constexpr (KIND == 1) {
        float ovf = 0.f;
        u32x4 f[2][2];
        split_block_f16(e[0], f, ovf);
        out[0] = f[0][0], out[1] = f[0][1], out[2] = f[1][0], out[3] = f[1][1];
        ovf = fmaxf(ovf, __shfl_xor(ovf, 32));
        if (row < n_dirs && h == 0) guard[row] = ovf;
    } else if constexpr (KIND == 2) {
        u32x4 f[2][3];
        split_block_x3(SplitSel::make(), e[0], f);
#pragma unroll
        for (int q = 0; q < 6; ++q) out[q] = f[q / 3][q % 3];
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int c = 0; c < 4; ++c) out[q][c] = __float_as_uint(e[0][4 * q + c]);
    }
    if (row < n_dirs) {
        u32x4* dst = reinterpret_cast<u32x4*>(table + (row * 2 + h) * (NQ * 4));
#pragma unroll
        for (int q = 0; q < NQ; ++q) dst[q] = out[q];
    }
}

// heads: dot products over the 16 registers of a block, two terms per v_pk_fma_f32
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float dot16(const f32x16& w, const f32x16& x) {
    f32x2 acc = {0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 16; t += 2) acc = __builtin_elementwise_fma((f32x2){w[t], w[t + 1]}, (f32x2){x[t], x[t + 1]}, acc);
    return acc[0] + acc[1];
}

// Jobs come off a counter (the workspace's hist[ncell], zero since the memset that precedes the routing), not round-robin:
// jobs are 1 .. job_tiles tiles long and a static deal left the waves resident for 84 % of the launch.  A wave holds the
// number of its next job already (the ticket was drawn a job earlier) and looks that job's cell up -- a scalar binary
// search, a dozen dependent loads -- while the current job's record is on its way; the results do not depend on who
// evaluates a point, so raw stays bit-identical.  (job_counter == NULL: the jobs are dealt round-robin, wave w takes jobs
// w, w + n_waves, ... -- the host's choice for batches with few jobs per wave.)
__device__ __forceinline__ int job_ticket(int* counter, int lane) {   // the number is in lane 0
    int t = 0;
    if (lane == 0) t = atomicAdd(counter, 1);
    return t;
}
__device__ __forceinline__ int job_cell_of(const int* __restrict__ job_prefix, int ncell, int job) {   // last c with job_prefix[c] <= job
    int lo = 0, hi = ncell;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (job_prefix[mid] <= job) lo = mid; else hi = mid;
    }
    return lo;
}

// Persistent waves over jobs; job = up to kJobTiles 32-point tiles of one cell's sorted segment.
//
// What bounds it (measured): on gfx950 the f32 MFMA runs at the vector-ALU rate (64 cycles per
// 32x32x2; 157 TF is also the VALU peak) and inside ONE wave VALU work does not hide under it: tile time = its 96
// MFMAs (6.1k cycles) PLUS its ~540 VALU instructions (~4.7 cycles each; encode_pe_doubling already halved the
// sin/cos share).  Across waves the two pipes do overlap, so the kernel runs TWO waves per SIMD: the cell's matrix
// operands stay in registers for the job (24 float4), biases and head rows are re-read per tile from a wave-private
// LDS slice (as 128 more live registers they cost the second wave and, before that, 69 AGPR read-backs per tile),
// gathers are prefetched three tiles ahead.  History: one wave per SIMD with everything in registers 3.99 ms,
// small section in LDS 3.85 ms, two waves per SIMD 3.4 ms per 27.4 M-sample frame.  That is the issue floor of this
// instruction mix (837 tiles per SIMD x (96 x 64 + 540 x 4) cycles = 3.3 ms at 2.1 GHz): f32 MFMA and VALU share
// the SIMD's issue cycles, so more waves do not help -- a workgroup-shared LDS copy of the whole cell record with
// four waves per SIMD (112 registers) measured 3.5 ms.  Only fewer instructions per tile would.
template <bool INDEXED, bool TABLE>  // directions gathered through dir_index (one per ray) or given per point; TABLE:
                                     // `dirs` is the per-ray encoding table of kilo_dir_table_kernel<0> (INDEXED only)
__global__ __launch_bounds__(256, 2) void kilo_mlp_kernel(const float* __restrict__ packed, int ncell, int job_tiles,
                                                          const int* __restrict__ seg_start,
                                                          const int* __restrict__ job_prefix,
                                                          const int* __restrict__ perm, const float* __restrict__ pts,
                                                          const float* __restrict__ dirs,
                                                          const int* __restrict__ dir_index, float4* __restrict__ raw, int* __restrict__ job_counter) {
    __shared__ __attribute__((aligned(16))) float s_small[4][kKiloSmall];
    const int lane = threadIdx.x & 63;
    const int j = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // uniform: the job's search and pointers become scalar
    const int total_jobs = job_prefix[ncell];
    const bool drawn = job_counter != nullptr;    // (uniform) small batches deal the jobs round-robin instead: kilo_forward_impl
    const int n_waves = gridDim.x * 4;
    int job = drawn ? __builtin_amdgcn_readfirstlane(job_ticket(job_counter, lane)) : int(blockIdx.x) * 4 + wave;
    int ticket = drawn ? job_ticket(job_counter, lane) : 0;   // the job after this one
    int cell = job < total_jobs ? job_cell_of(job_prefix, ncell, job) : 0;
    for (int job_n = 0, cell_n = 0; job < total_jobs; job = job_n, cell = cell_n) {
        const int seg_b = seg_start[cell], seg_e = seg_start[cell + 1];
        const int first = seg_b + (job - job_prefix[cell]) * 32 * job_tiles;
        const float* cw = packed + int64_t(cell) * kKiloCellFloats;
        float4 w[kKiloSteps];
#pragma unroll
        for (int s = 0; s < kKiloSteps; ++s) w[s] = reinterpret_cast<const float4*>(cw)[s * 64 + lane];
        // biases and head rows (288 floats) go to this wave's LDS slice and are re-read per tile: as 128 more live
        // registers they pushed the kernel past 256 VGPRs, every layer's output landed in AGPRs and came back one
        // v_accvgpr_read at a time (69 per tile).  Wave-private region: no barrier, the wave's own LDS order suffices.
        float* sm_job = s_small[wave];
        {
            const float4* src = reinterpret_cast<const float4*>(cw + kKiloSteps * 256);
            const float4 s0 = src[lane], s1 = src[lane < kKiloSmall / 4 - 64 ? 64 + lane : lane];
            // every load of this job's record is issued: the next job (its ticket was drawn a job ago) gets its cell now
            job_n = drawn ? __builtin_amdgcn_readfirstlane(ticket) : job + n_waves;
            if (drawn) ticket = job_ticket(job_counter, lane);
            cell_n = job_n < total_jobs ? job_cell_of(job_prefix, ncell, job_n) : 0;
            reinterpret_cast<float4*>(sm_job)[lane] = s0;
            if (lane < kKiloSmall / 4 - 64) reinterpret_cast<float4*>(sm_job)[64 + lane] = s1;
        }
        const float* bias_job = sm_job + kKOffBias + h * 16;
        const float sb = cw[kKiloSteps * 256 + kKOffSigB], rb0 = cw[kKiloSteps * 256 + kKOffRgbB],
                    rb1 = cw[kKiloSteps * 256 + kKOffRgbB + 1], rb2 = cw[kKiloSteps * 256 + kKOffRgbB + 2];
        auto lds16 = [](const float* p) { return *reinterpret_cast<const f32x16*>(p); };

        // Gather pipeline: perm -> (point, dir index) -> direction is three dependent loads; indices run three
        // tiles ahead, points two, directions one.  Reads past the job's last tile are clamped to the segment's
        // last element (valid memory, results unused).  32-bit element offsets: n_points <= 2^28 (host check).
        const int last = seg_e - 1;
        auto tile_id = [&](int t0) { const int k = t0 + j; return unsigned(perm[unsigned(k < last ? k : last)]); };
        auto dir_of = [&](unsigned id) { return INDEXED ? unsigned(dir_index[id]) : id; };  // no branch in the tile loop
        unsigned id_c = tile_id(first), id_n = tile_id(first + 32), id_nn = tile_id(first + 64);
        float pc[3], pn[3], dc[3] = {0.f, 0.f, 0.f};
        unsigned di_n, di_c = dir_of(id_c);   // (di_c: the current tile's table row, TABLE only)
        {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                pc[a] = pts[id_c * 3u + a];
                if (!TABLE) dc[a] = dirs[di_c * 3u + a];
            }
#pragma unroll
            for (int a = 0; a < 3; ++a) pn[a] = pts[id_n * 3u + a];
            di_n = dir_of(id_n);
        }
        for (int t0 = first; t0 < seg_e && t0 < first + 32 * job_tiles; t0 += 32) {
            const bool valid = t0 + j < seg_e;
            const unsigned id = id_c;
            const float px = pc[0], py = pc[1], pz = pc[2], dx = dc[0], dy = dc[1], dz = dc[2];
            f32x16 x[2], xd;
            if (TABLE) {   // this tile's direction block: the table row of its ray, this lane half's 64 B (used three layers on)
                const float4* row = reinterpret_cast<const float4*>(dirs) + (size_t(di_c) * 2u + unsigned(h)) * 4u;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = row[q];
                    xd[4 * q] = v.x, xd[4 * q + 1] = v.y, xd[4 * q + 2] = v.z, xd[4 * q + 3] = v.w;
                }
            }
            {   // issue the loads of the following tiles, rotate the pipeline registers
                float dn[3] = {0.f, 0.f, 0.f}, pnn[3];
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    if (!TABLE) dn[a] = dirs[di_n * 3u + a];
                    pnn[a] = pts[id_nn * 3u + a];
                }
                const unsigned di_nn = dir_of(id_nn);
                const unsigned id_nnn = tile_id(t0 + 96);
#pragma unroll
                for (int a = 0; a < 3; ++a) pc[a] = pn[a], dc[a] = dn[a], pn[a] = pnn[a];
                id_c = id_n, id_n = id_nn, id_nn = id_nnn, di_c = di_n, di_n = di_nn;
            }
            encode_pe_doubling<10>(px, py, pz, h, x);
            // a run-time zero the compiler cannot see through: the LDS reads below are loop-invariant, and LICM
            // would hoist all of them back into 128 registers held across the tile loop
            int again = 0;
            asm volatile("" : "+v"(again));
            const float* bias_h = bias_job + again;
            const float* sm = sm_job + again;
            x[0] = relu16(tiny_layer<2>(w, lds16(bias_h), x));              // pts_linears 0 (network.py:132-138)
            x[0] = relu16(tiny_layer<1>(w + 8, lds16(bias_h + 32), x));     // pts_linears 1
            float sigma = dot16(lds16(sm + kKOffSigW + h * 16), x[0]);      // sigma_linear (:142-144)
            x[0] = tiny_layer<1>(w + 12, lds16(bias_h + 64), x);            // feature_linear (:147-149), no activation
            if (TABLE) x[1] = xd;
            else encode_pe_doubling<4>(dx, dy, dz, h, x + 1);
            const f32x16 hv = relu16(tiny_layer<2>(w + 16, lds16(bias_h + 96), x));  // view_linears on cat([feature, dirs]) (:152-156)
            // rgb_linear (:159-161)
            float r = dot16(lds16(sm + kKOffRgbW + h * 16), hv), g = dot16(lds16(sm + kKOffRgbW + 32 + h * 16), hv),
                  b = dot16(lds16(sm + kKOffRgbW + 64 + h * 16), hv);
            r = nrf_add_halves(r), g = nrf_add_halves(g), b = nrf_add_halves(b), sigma = nrf_add_halves(sigma);
            if (h == 0 && valid) raw[id] = make_float4(r + rb0, g + rb1, b + rb2, sigma + sb);
        }
    }
}

// ---- the tiny MLPs on the fp16 matrix cores (two fp16 terms per fp32 operand, split2h.h / mlp_fp16x2.hip) -------------
// Same job structure and the same record size: the 24 KiB that hold a cell's matrices as fp32 A operands of 96
// v_mfma_f32_32x32x2_f32 hold them as 12 k16-steps of (hi | lo) fp16 fragments, A operands of 36
// v_mfma_f32_32x32x16_f16 (pts0: 4 steps, pts1: 2, feature: 2, view: 2 feature + 2 direction-encoding steps); the small
// block (biases, head rows) is the fp32 kernel's, plus a status word: 1 if a weight of the cell lies outside fp16.
constexpr int kKH2Steps = 12;
constexpr int kKOffStatus = 264;
static_assert(kKH2Steps * 2 * 1024 == kKiloSteps * 256 * 4 && kKOffStatus >= kKOffRgbB + 3 && kKOffStatus < kKiloSmall,
              "the two formats share the record");

__global__ __launch_bounds__(256) void kilo_pack_h2_kernel(KiloSrc s, int num_cells, float* __restrict__ packed) {
    __shared__ float rec[kRecFloats];
    __shared__ int bad;
    const int cell = blockIdx.x;
    if (threadIdx.x == 0) bad = 0;
    stage_cell_record(s, cell, rec);
    __syncthreads();
    unsigned short* frag = reinterpret_cast<unsigned short*>(packed + int64_t(cell) * kKiloCellFloats);
    for (int idx = threadIdx.x; idx < kKH2Steps * 512; idx += 256) {
        const int jj = idx & 7, lane = (idx >> 3) & 63, step = idx >> 9;
        const int h = lane >> 5, row = lane & 31;
        float v = 0.f;
        if (step < 4) {            // pts0: 63 encoding columns, kernel slot order; k16-step = (k-block, half)
            const int c = pe_col((step >> 1) * 16 + 8 * (step & 1) + jj, h, 10);
            if (c >= 0) v = rec[REC_OFF(0) + row * 63 + c];
        } else if (step < 6) {
            v = rec[REC_OFF(2) + row * 32 + acc_row(8 * (step & 1) + jj, h)];
        } else if (step < 8) {
            v = rec[REC_OFF(4) + row * 32 + acc_row(8 * (step & 1) + jj, h)];
        } else if (step < 10) {    // view layer, feature columns 0..31
            v = rec[REC_OFF(8) + row * 59 + acc_row(8 * (step & 1) + jj, h)];
        } else {                   // view layer, direction encoding columns 32..58
            const int c = pe_col(8 * (step & 1) + jj, h, 4);
            if (c >= 0) v = rec[REC_OFF(8) + row * 59 + 32 + c];
        }
        if (!(fabsf(v) < kF16Limit)) atomicOr(&bad, 1);
        const _Float16 hi = (_Float16)v;
        const _Float16 lo = (_Float16)(__fmul_rn(__fsub_rn(v, (float)hi), kLoScale));
        frag[step * 1024 + lane * 8 + jj] = __builtin_bit_cast(unsigned short, hi);
        frag[step * 1024 + 512 + lane * 8 + jj] = __builtin_bit_cast(unsigned short, lo);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kKiloSmall; i += 256) {   // the fp32 kernel's small block (kilo_pack_kernel)
        float v = 0.f;
        if (i < 128) {
            const int layer = i >> 5, h = (i >> 4) & 1, r = i & 15;
            const int off = layer == 0 ? REC_OFF(1) : layer == 1 ? REC_OFF(3) : layer == 2 ? REC_OFF(5) : REC_OFF(9);
            v = rec[off + acc_row(r, h)];
        } else if (i < 160) {
            const int h = ((i - 128) >> 4) & 1, r = i & 15;
            v = rec[REC_OFF(6) + acc_row(r, h)];
        } else if (i < 256) {
            const int j = i - 160, c = j >> 5, h = (j >> 4) & 1, r = j & 15;
            v = rec[REC_OFF(10) + c * 32 + acc_row(r, h)];
        } else if (i == kKOffSigB) {
            v = rec[REC_OFF(7)];
        } else if (i < kKOffRgbB + 3) {
            v = rec[REC_OFF(11) + (i - kKOffRgbB)];
        } else if (i == kKOffStatus) {
            v = bad ? 1.f : 0.f;
        }
        packed[int64_t(cell) * kKiloCellFloats + kKiloSteps * 256 + i] = v;
    }
}

// One 32x32 output block of a tiny layer on the fp16 pipe: per k16-step the two cross terms into a second accumulator at
// scale 2^11 (started from the constant 0), hi.hi into the main one; the result is main + 2^-11 * second.
template <int NKB>
__device__ __forceinline__ f32x16 tiny_layer_h2(const u32x4 (*w)[2], f32x16 init, const u32x4 (*x)[2][2]) {
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 y = init, yl = zero16;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int st = kb * 2 + s2;
            yl = st == 0 ? NRF_MFMA_F16(w[st][0], x[kb][s2][1], zero16) : NRF_MFMA_F16(w[st][0], x[kb][s2][1], yl);
            yl = NRF_MFMA_F16(w[st][1], x[kb][s2][0], yl);
            y = NRF_MFMA_F16(w[st][0], x[kb][s2][0], y);
        }
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 c = {kLoUnscale, kLoUnscale};
#pragma unroll
    for (int t = 0; t < 16; t += 2) {
        const f32x2 l = {yl[t], yl[t + 1]}, b = {y[t], y[t + 1]};
        const f32x2 r = __builtin_elementwise_fma(l, c, b);
        y[t] = r[0], y[t + 1] = r[1];
    }
    return y;
}

// kilo_mlp_kernel with the cell's matrices as fp16x2 fragments (96 registers, as before).  Points whose operands leave
// fp16's range, and every point of a cell with such a weight, come out as NaN (mlp_fp16x2.hip's guard).
template <bool INDEXED, bool TABLE>   // TABLE: `dirs` is the table of kilo_dir_table_kernel<1>, `dir_guard` its range-guard column
__global__ __launch_bounds__(256, 2) void kilo_mlp_h2_kernel(const float* __restrict__ packed, int ncell, int job_tiles,
                                                             const int* __restrict__ seg_start,
                                                             const int* __restrict__ job_prefix,
                                                             const int* __restrict__ perm, const float* __restrict__ pts,
                                                             const float* __restrict__ dirs,
                                                             const int* __restrict__ dir_index, float4* __restrict__ raw,
                                                             int* __restrict__ job_counter, const float* __restrict__ dir_guard) {
    __shared__ __attribute__((aligned(16))) float s_small[4][kKiloSmall];
    const int lane = threadIdx.x & 63;
    const int j = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // uniform: the job's search and pointers become scalar
    const int total_jobs = job_prefix[ncell];
    const bool drawn = job_counter != nullptr;    // (uniform) small batches deal the jobs round-robin instead: kilo_forward_impl
    const int n_waves = gridDim.x * 4;
    int job = drawn ? __builtin_amdgcn_readfirstlane(job_ticket(job_counter, lane)) : int(blockIdx.x) * 4 + wave;
    int ticket = drawn ? job_ticket(job_counter, lane) : 0;   // the job after this one
    int cell = job < total_jobs ? job_cell_of(job_prefix, ncell, job) : 0;
    for (int job_n = 0, cell_n = 0; job < total_jobs; job = job_n, cell = cell_n) {
        const int seg_b = seg_start[cell], seg_e = seg_start[cell + 1];
        const int first = seg_b + (job - job_prefix[cell]) * 32 * job_tiles;
        const float* cw = packed + int64_t(cell) * kKiloCellFloats;
        u32x4 w[kKH2Steps][2];
#pragma unroll
        for (int s = 0; s < kKH2Steps; ++s) {
            w[s][0] = reinterpret_cast<const u32x4*>(cw)[(2 * s) * 64 + lane];
            w[s][1] = reinterpret_cast<const u32x4*>(cw)[(2 * s + 1) * 64 + lane];
        }
        float* sm_job = s_small[wave];   // wave-private: no barrier (kilo_mlp_kernel)
        {
            const float4* src = reinterpret_cast<const float4*>(cw + kKiloSteps * 256);
            const float4 s0 = src[lane], s1 = src[lane < kKiloSmall / 4 - 64 ? 64 + lane : lane];
            // every load of this job's record is issued: the next job (its ticket was drawn a job ago) gets its cell now
            job_n = drawn ? __builtin_amdgcn_readfirstlane(ticket) : job + n_waves;
            if (drawn) ticket = job_ticket(job_counter, lane);
            cell_n = job_n < total_jobs ? job_cell_of(job_prefix, ncell, job_n) : 0;
            reinterpret_cast<float4*>(sm_job)[lane] = s0;
            if (lane < kKiloSmall / 4 - 64) reinterpret_cast<float4*>(sm_job)[64 + lane] = s1;
        }
        const float* bias_job = sm_job + kKOffBias + h * 16;
        const float sb = cw[kKiloSteps * 256 + kKOffSigB], rb0 = cw[kKiloSteps * 256 + kKOffRgbB],
                    rb1 = cw[kKiloSteps * 256 + kKOffRgbB + 1], rb2 = cw[kKiloSteps * 256 + kKOffRgbB + 2];
        const float poison = cw[kKiloSteps * 256 + kKOffStatus] != 0.f ? __builtin_inff() : 0.f;
        auto lds16 = [](const float* p) { return *reinterpret_cast<const f32x16*>(p); };

        const int last = seg_e - 1;   // gather pipeline: kilo_mlp_kernel
        auto tile_id = [&](int t0) { const int k = t0 + j; return unsigned(perm[unsigned(k < last ? k : last)]); };
        auto dir_of = [&](unsigned id) { return INDEXED ? unsigned(dir_index[id]) : id; };
        unsigned id_c = tile_id(first), id_n = tile_id(first + 32), id_nn = tile_id(first + 64);
        float pc[3], pn[3], dc[3] = {0.f, 0.f, 0.f};
        unsigned di_n, di_c = dir_of(id_c);   // (di_c: the current tile's table row, TABLE only)
        {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                pc[a] = pts[id_c * 3u + a];
                if (!TABLE) dc[a] = dirs[di_c * 3u + a];
            }
#pragma unroll
            for (int a = 0; a < 3; ++a) pn[a] = pts[id_n * 3u + a];
            di_n = dir_of(id_n);
        }
        for (int t0 = first; t0 < seg_e && t0 < first + 32 * job_tiles; t0 += 32) {
            const bool valid = t0 + j < seg_e;
            const unsigned id = id_c;
            const float px = pc[0], py = pc[1], pz = pc[2], dx = dc[0], dy = dc[1], dz = dc[2];
            float ovf = poison, dguard = 0.f;
            f32x16 e[2];
            u32x4 xpe[2][2][2], xh[1][2][2], xv[2][2][2];
            if (TABLE) {   // this tile's direction fragments: the table row of its ray, this lane half's 64 B
                const u32x4* row = reinterpret_cast<const u32x4*>(dirs) + (size_t(di_c) * 2u + unsigned(h)) * 4u;
                xv[1][0][0] = row[0], xv[1][0][1] = row[1], xv[1][1][0] = row[2], xv[1][1][1] = row[3];
                dguard = dir_guard[di_c];   // folded into ovf at the tile's end: a use up here makes the compiler wait for it --
            }                               // the newest load, so vmcnt(0) -- right behind the row loads, every tile
            {
                float dn[3] = {0.f, 0.f, 0.f}, pnn[3];
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    if (!TABLE) dn[a] = dirs[di_n * 3u + a];
                    pnn[a] = pts[id_nn * 3u + a];
                }
                const unsigned di_nn = dir_of(id_nn);
                const unsigned id_nnn = tile_id(t0 + 96);
#pragma unroll
                for (int a = 0; a < 3; ++a) pc[a] = pn[a], dc[a] = dn[a], pn[a] = pnn[a];
                id_c = id_n, id_n = id_nn, id_nn = id_nnn, di_c = di_n, di_n = di_nn;
            }
            encode_pe_doubling<10>(px, py, pz, h, e);
            split_block_f16(e[0], xpe[0], ovf);
            split_block_f16(e[1], xpe[1], ovf);
            int again = 0;   // keeps the LDS reads below inside the tile loop (kilo_mlp_kernel)
            asm volatile("" : "+v"(again));
            const float* bias_h = bias_job + again;
            const float* sm = sm_job + again;
            f32x16 y = relu16(tiny_layer_h2<2>(w, lds16(bias_h), xpe));             // pts_linears 0 (network.py:132-138)
            split_block_f16(y, xh[0], ovf);
            y = relu16(tiny_layer_h2<1>(w + 4, lds16(bias_h + 32), xh));            // pts_linears 1
            float sigma = dot16(lds16(sm + kKOffSigW + h * 16), y);                 // sigma_linear (:142-144)
            split_block_f16(y, xh[0], ovf);
            y = tiny_layer_h2<1>(w + 6, lds16(bias_h + 64), xh);                    // feature_linear (:147-149), no activation
            split_block_f16(y, xv[0], ovf);
            if (!TABLE) {
                encode_pe_doubling<4>(dx, dy, dz, h, e);
                split_block_f16(e[0], xv[1], ovf);
            }
            const f32x16 hv = relu16(tiny_layer_h2<2>(w + 8, lds16(bias_h + 96), xv));   // view_linears on cat([feature, dirs]) (:152-156)
            // rgb_linear (:159-161)
            float r = dot16(lds16(sm + kKOffRgbW + h * 16), hv), g = dot16(lds16(sm + kKOffRgbW + 32 + h * 16), hv),
                  b = dot16(lds16(sm + kKOffRgbW + 64 + h * 16), hv);
            r = nrf_add_halves(r), g = nrf_add_halves(g), b = nrf_add_halves(b), sigma = nrf_add_halves(sigma);
            ovf = fmaxf(fmaxf(ovf, dguard), __shfl_xor(ovf, 32));
            if (h == 0 && valid) {
                float4 o = make_float4(r + rb0, g + rb1, b + rb2, sigma + sb);
                if (!(ovf < kF16Limit)) o = make_float4(__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""));
                raw[id] = o;
            }
        }
    }
}

// ---- the tiny MLPs on the bf16 matrix cores at fp32 accuracy (three bf16 terms per operand, split3.h / mlp_bf16x3.hip) ----
// fp32 w = w0 + w1 + w2 exactly (3 x 8 significand bits), a product keeps the six terms w_i x_j with i + j <= 2, added small
// terms first into ONE fp32 accumulator: 72 v_mfma_f32_32x32x16_bf16 per 32-point tile (2.3 k matrix-pipe cycles) where the
// fp32 form has 96 v_mfma_f32_32x32x2_f32 (6.1 k), at the fp32 form's accuracy and range (no guard, no status word).
// Record: 12 k16-steps of (hi | mid | lo) fragments = 36 KiB + the fp32 kernel's small block.  hi and mid stay in registers
// for the job (96, as the other two kernels); the lo fragments -- one of a step's six MFMAs reads them -- sit in a
// wave-private LDS slice (12 KiB per wave, each lane re-reads only what it wrote: no barrier) so that two waves per SIMD fit.
constexpr int kKX3Steps = 12;
constexpr int kKiloX3MatFloats = kKX3Steps * 3 * 256;                 // 9216
constexpr int kKiloCellFloatsX3 = kKiloX3MatFloats + kKiloSmall;      // 9504


__global__ __launch_bounds__(256) void kilo_pack_x3_kernel(KiloSrc s, int num_cells, float* __restrict__ packed) {
    __shared__ float rec[kRecFloats];
    const int cell = blockIdx.x;
    stage_cell_record(s, cell, rec);
    __syncthreads();
    unsigned short* frag = reinterpret_cast<unsigned short*>(packed + int64_t(cell) * kKiloCellFloatsX3);
    for (int idx = threadIdx.x; idx < kKX3Steps * 512; idx += 256) {   // same (step, lane, element) -> weight map as kilo_pack_h2_kernel
        const int jj = idx & 7, lane = (idx >> 3) & 63, step = idx >> 9;
        const int h = lane >> 5, row = lane & 31;
        float v = 0.f;
        if (step < 4) {
            const int c = pe_col((step >> 1) * 16 + 8 * (step & 1) + jj, h, 10);
            if (c >= 0) v = rec[REC_OFF(0) + row * 63 + c];
        } else if (step < 6) {
            v = rec[REC_OFF(2) + row * 32 + acc_row(8 * (step & 1) + jj, h)];
        } else if (step < 8) {
            v = rec[REC_OFF(4) + row * 32 + acc_row(8 * (step & 1) + jj, h)];
        } else if (step < 10) {
            v = rec[REC_OFF(8) + row * 59 + acc_row(8 * (step & 1) + jj, h)];
        } else {
            const int c = pe_col(8 * (step & 1) + jj, h, 4);
            if (c >= 0) v = rec[REC_OFF(8) + row * 59 + 32 + c];
        }
        const __bf16 b0 = (__bf16)v;
        const float r1 = __fsub_rn(v, (float)b0);
        const __bf16 b1 = (__bf16)r1;
        const __bf16 b2 = (__bf16)__fsub_rn(r1, (float)b1);
        frag[step * 1536 + lane * 8 + jj] = __builtin_bit_cast(unsigned short, b0);
        frag[step * 1536 + 512 + lane * 8 + jj] = __builtin_bit_cast(unsigned short, b1);
        frag[step * 1536 + 1024 + lane * 8 + jj] = __builtin_bit_cast(unsigned short, b2);
    }
    for (int i = threadIdx.x; i < kKiloSmall; i += 256) {   // the fp32 kernel's small block (kilo_pack_kernel)
        float v = 0.f;
        if (i < 128) {
            const int layer = i >> 5, h = (i >> 4) & 1, r = i & 15;
            const int off = layer == 0 ? REC_OFF(1) : layer == 1 ? REC_OFF(3) : layer == 2 ? REC_OFF(5) : REC_OFF(9);
            v = rec[off + acc_row(r, h)];
        } else if (i < 160) {
            const int h = ((i - 128) >> 4) & 1, r = i & 15;
            v = rec[REC_OFF(6) + acc_row(r, h)];
        } else if (i < 256) {
            const int j = i - 160, c = j >> 5, h = (j >> 4) & 1, r = j & 15;
            v = rec[REC_OFF(10) + c * 32 + acc_row(r, h)];
        } else if (i == kKOffSigB) {
            v = rec[REC_OFF(7)];
        } else if (i < kKOffRgbB + 3) {
            v = rec[REC_OFF(11) + (i - kKOffRgbB)];
        }
        packed[int64_t(cell) * kKiloCellFloatsX3 + kKiloX3MatFloats + i] = v;
    }
}

#ifdef KF_STAMP   // diagnostic builds (tools/debug/kilo_fwd_phases.py): s_memtime ticks per phase of kilo_mlp_x3_kernel, summed over all waves
__device__ unsigned long long g_kf_stamps[16];
// (the stamp reads the clock in an asm statement that takes the phase's result as an in-out operand: pure arithmetic has no
// order against a clock read otherwise, and the compiler runs all the reads back to back)
#define KF_T(i, pin)                                                                  \
    do {                                                                              \
        unsigned long long now_;                                                      \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_), "+v"(pin) : : "memory"); \
        kf_acc[i] += now_ - kf_last;   /* (per wave, in scalar registers: one atomic per phase and wave at the kernel's end) */ \
        kf_last = now_;                                                               \
    } while (0)
#else
#define KF_T(i, pin) do {} while (0)
#endif

// One 32x32 output block of a tiny layer on the bf16 pipe, six MFMAs per k16-step into one accumulator, small terms first
// (mlp_bf16x3.hip's order); `wlo`: this lane's lo fragments in LDS, 64 u32x4 per step.
template <int NKB>
__device__ __forceinline__ f32x16 tiny_layer_x3(const u32x4 (*w)[2], const u32x4* wlo, f32x16 init, const u32x4 (*x)[2][3]) {
    f32x16 y = init;
#ifdef KILO_ABL_NOMFMA   // timing-only ablation: the layer without its MFMAs (operands still consumed)
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int st = kb * 2 + s2;
            const u32x4 w2 = wlo[st * 64];
#pragma unroll
            for (int c = 0; c < 4; ++c)
                y[c] += __uint_as_float(w2[c] ^ w[st][0][c] ^ w[st][1][c] ^ x[kb][s2][0][c] ^ x[kb][s2][1][c] ^ x[kb][s2][2][c]);
        }
    return y;
#endif
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int st = kb * 2 + s2;
            const u32x4 w2 = wlo[st * 64];
            y = NRF_MFMA_X3(w2, x[kb][s2][0], y);
            y = NRF_MFMA_X3(w[st][0], x[kb][s2][2], y);
            y = NRF_MFMA_X3(w[st][1], x[kb][s2][1], y);
            y = NRF_MFMA_X3(w[st][1], x[kb][s2][0], y);
            y = NRF_MFMA_X3(w[st][0], x[kb][s2][1], y);
            y = NRF_MFMA_X3(w[st][0], x[kb][s2][0], y);
        }
    return y;
}

template <bool INDEXED, bool TABLE>   // TABLE: `dirs` is the table of kilo_dir_table_kernel<2>
__global__ __launch_bounds__(256, 2) void kilo_mlp_x3_kernel(const float* __restrict__ packed, int ncell, int job_tiles,
                                                             const int* __restrict__ seg_start,
                                                             const int* __restrict__ job_prefix,
                                                             const int* __restrict__ perm, const float* __restrict__ pts,
                                                             const float* __restrict__ dirs,
                                                             const int* __restrict__ dir_index, float4* __restrict__ raw, int* __restrict__ job_counter) {
    __shared__ __attribute__((aligned(16))) float s_small[4][kKiloSmall];
    __shared__ u32x4 s_wlo[4][kKX3Steps][64];
    const SplitSel sel = SplitSel::make();
    const int lane = threadIdx.x & 63;
    const int j = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // uniform: the job's search and pointers become scalar
    const int total_jobs = job_prefix[ncell];
#ifdef KF_STAMP
    unsigned long long kf_last = __builtin_amdgcn_s_memtime();
    int kf_none = 0;   // (a stamp with nothing to pin: the phase ends in memory operations)
    unsigned long long kf_acc[12] = {};
#endif
    const bool drawn = job_counter != nullptr;    // (uniform) small batches deal the jobs round-robin instead: kilo_forward_impl
    const int n_waves = gridDim.x * 4;
    int job = drawn ? __builtin_amdgcn_readfirstlane(job_ticket(job_counter, lane)) : int(blockIdx.x) * 4 + wave;
    int ticket = drawn ? job_ticket(job_counter, lane) : 0;   // the job after this one
    int cell = job < total_jobs ? job_cell_of(job_prefix, ncell, job) : 0;
    for (int job_n = 0, cell_n = 0; job < total_jobs; job = job_n, cell = cell_n) {
        const int seg_b = seg_start[cell], seg_e = seg_start[cell + 1];
        const int first = seg_b + (job - job_prefix[cell]) * 32 * job_tiles;
        const float* cw = packed + int64_t(cell) * kKiloCellFloatsX3;
        u32x4 w[kKX3Steps][2];
        u32x4* wlo_job = &s_wlo[wave][0][lane];   // lane-private slots: the wave's own LDS order suffices
#pragma unroll
        for (int s = 0; s < kKX3Steps; ++s) {
            w[s][0] = reinterpret_cast<const u32x4*>(cw)[(3 * s) * 64 + lane];
            w[s][1] = reinterpret_cast<const u32x4*>(cw)[(3 * s + 1) * 64 + lane];
            wlo_job[s * 64] = reinterpret_cast<const u32x4*>(cw)[(3 * s + 2) * 64 + lane];
        }
        float* sm_job = s_small[wave];   // wave-private: no barrier (kilo_mlp_kernel)
        {
            const float4* src = reinterpret_cast<const float4*>(cw + kKiloX3MatFloats);
            const float4 s0 = src[lane], s1 = src[lane < kKiloSmall / 4 - 64 ? 64 + lane : lane];
            // every load of this job's record is issued: the next job (its ticket was drawn a job ago) gets its cell now
            job_n = drawn ? __builtin_amdgcn_readfirstlane(ticket) : job + n_waves;
            if (drawn) ticket = job_ticket(job_counter, lane);
            cell_n = job_n < total_jobs ? job_cell_of(job_prefix, ncell, job_n) : 0;
            reinterpret_cast<float4*>(sm_job)[lane] = s0;
            if (lane < kKiloSmall / 4 - 64) reinterpret_cast<float4*>(sm_job)[64 + lane] = s1;
        }
        const float* bias_job = sm_job + kKOffBias + h * 16;
        const float sb = cw[kKiloX3MatFloats + kKOffSigB], rb0 = cw[kKiloX3MatFloats + kKOffRgbB],
                    rb1 = cw[kKiloX3MatFloats + kKOffRgbB + 1], rb2 = cw[kKiloX3MatFloats + kKOffRgbB + 2];
        auto lds16 = [](const float* p) { return *reinterpret_cast<const f32x16*>(p); };

        const int last = seg_e - 1;   // gather pipeline: kilo_mlp_kernel
#ifdef KILO_ABL_NOGATHER   // timing-only ablation: points in sorted order (coalesced point loads and raw stores, no perm)
        auto tile_id = [&](int t0) { const int k = t0 + j; return unsigned(k < last ? k : last); };
#else
        auto tile_id = [&](int t0) { const int k = t0 + j; return unsigned(perm[unsigned(k < last ? k : last)]); };
#endif
        auto dir_of = [&](unsigned id) { return INDEXED ? unsigned(dir_index[id]) : id; };
        unsigned id_c = tile_id(first), id_n = tile_id(first + 32), id_nn = tile_id(first + 64);
        float pc[3], pn[3], dc[3] = {0.f, 0.f, 0.f};
        unsigned di_n, di_c = dir_of(id_c);   // (di_c: the current tile's table row, TABLE only)
        {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                pc[a] = pts[id_c * 3u + a];
                if (!TABLE) dc[a] = dirs[di_c * 3u + a];
            }
#pragma unroll
            for (int a = 0; a < 3; ++a) pn[a] = pts[id_n * 3u + a];
            di_n = dir_of(id_n);
        }
        KF_T(0, kf_none);   // job set-up: ticket, record loads issued, small block staged, next job's cell, first gathers issued
        for (int t0 = first; t0 < seg_e && t0 < first + 32 * job_tiles; t0 += 32) {
            const bool valid = t0 + j < seg_e;
            const unsigned id = id_c;
            const float px = pc[0], py = pc[1], pz = pc[2], dx = dc[0], dy = dc[1], dz = dc[2];
            f32x16 e[2];
            u32x4 xpe[2][2][3], xh[1][2][3], xv[2][2][3];
            if (TABLE) {   // this tile's direction fragments: the table row of its ray, this lane half's 96 B
                const u32x4* row = reinterpret_cast<const u32x4*>(dirs) + (size_t(di_c) * 2u + unsigned(h)) * 6u;
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int t = 0; t < 3; ++t)
#ifdef KILO_ABL_NOTABLE   // timing-only ablation: no direction rows
                        xv[1][s2][t] = u32x4{di_c, di_c + t, di_c, unsigned(size_t(row))};
#else
                        xv[1][s2][t] = row[s2 * 3 + t];
#endif
            }
            {
                float dn[3] = {0.f, 0.f, 0.f}, pnn[3];
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    if (!TABLE) dn[a] = dirs[di_n * 3u + a];
                    pnn[a] = pts[id_nn * 3u + a];
                }
                const unsigned di_nn = dir_of(id_nn);
                const unsigned id_nnn = tile_id(t0 + 96);
#pragma unroll
                for (int a = 0; a < 3; ++a) pc[a] = pn[a], dc[a] = dn[a], pn[a] = pnn[a];
                id_c = id_n, id_n = id_nn, id_nn = id_nnn, di_c = di_n, di_n = di_nn;
            }
            encode_pe_doubling<10>(px, py, pz, h, e);
            KF_T(2, e[1]);   // (waits for this tile's point) position encoding
            split_block_x3(sel, e[0], xpe[0]);
            split_block_x3(sel, e[1], xpe[1]);
            KF_T(3, xpe[1][1][2]);   // its split
            int again = 0;   // keeps the LDS reads below inside the tile loop (kilo_mlp_kernel)
            asm volatile("" : "+v"(again));
            const float* bias_h = bias_job + again;
            const float* sm = sm_job + again;
            const u32x4* wlo = wlo_job + again;
            f32x16 y = relu16(tiny_layer_x3<2>(w, wlo, lds16(bias_h), xpe));                        // pts_linears 0 (network.py:132-138)
            KF_T(4, y);
            split_block_x3(sel, y, xh[0]);
            KF_T(5, xh[0][1][2]);
            y = relu16(tiny_layer_x3<1>(w + 4, wlo + 4 * 64, lds16(bias_h + 32), xh));            // pts_linears 1
            KF_T(6, y);
            float sigma = dot16(lds16(sm + kKOffSigW + h * 16), y);                                 // sigma_linear (:142-144)
            split_block_x3(sel, y, xh[0]);
            KF_T(7, xh[0][1][2]);
            y = tiny_layer_x3<1>(w + 6, wlo + 6 * 64, lds16(bias_h + 64), xh);                     // feature_linear (:147-149), no activation
            KF_T(8, y);
            split_block_x3(sel, y, xv[0]);
            KF_T(9, xv[0][1][2]);
            if (!TABLE) {
                encode_pe_doubling<4>(dx, dy, dz, h, e);
                split_block_x3(sel, e[0], xv[1]);
            }
            f32x16 hv = relu16(tiny_layer_x3<2>(w + 8, wlo + 8 * 64, lds16(bias_h + 96), xv));   // view_linears on cat([feature, dirs]) (:152-156)
            KF_T(10, hv);  // (waits for the tile's direction rows)
            // rgb_linear (:159-161)
            float r = dot16(lds16(sm + kKOffRgbW + h * 16), hv), g = dot16(lds16(sm + kKOffRgbW + 32 + h * 16), hv),
                  b = dot16(lds16(sm + kKOffRgbW + 64 + h * 16), hv);
            r = nrf_add_halves(r), g = nrf_add_halves(g), b = nrf_add_halves(b), sigma = nrf_add_halves(sigma);
            if (h == 0 && valid) raw[id] = make_float4(r + rb0, g + rb1, b + rb2, sigma + sb);
            KF_T(11, kf_none);  // heads and store
        }
    }
#ifdef KF_STAMP
    if (lane == 0)
        for (int q = 0; q < 12; ++q) atomicAdd(&g_kf_stamps[q], kf_acc[q]);
#endif
}

// ---- generic exclusive scan of int32 (three small kernels) ---------------------------------
constexpr int kScanBlock = 1024;
__global__ void scan_block_kernel(const int* __restrict__ in, int64_t n, int* __restrict__ out, int* __restrict__ sums) {
    __shared__ int s[kScanBlock];
    const int tid = threadIdx.x;
    const int64_t i = int64_t(blockIdx.x) * kScanBlock + tid;
    const int v = i < n ? in[i] : 0;
    s[tid] = v;
    __syncthreads();
    for (int off = 1; off < kScanBlock; off <<= 1) {
        const int a = tid >= off ? s[tid - off] : 0;
        __syncthreads();
        s[tid] += a;
        __syncthreads();
    }
    if (i < n) out[i] = s[tid] - v;
    if (tid == kScanBlock - 1) sums[blockIdx.x] = s[tid];
}
__global__ void scan_sums_kernel(int* __restrict__ sums, int nblocks, int64_t* __restrict__ total) {
    // single block, sequential over chunks of 1024 block sums
    __shared__ int s[kScanBlock];
    __shared__ long long carry;   // the grand total may pass 2^31 (dense grids, fine steps): the caller must see it
    const int tid = threadIdx.x;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < nblocks; base += kScanBlock) {
        const int i = base + tid;
        const int v = i < nblocks ? sums[i] : 0;
        s[tid] = v;
        __syncthreads();
        for (int off = 1; off < kScanBlock; off <<= 1) {
            const int a = tid >= off ? s[tid - off] : 0;
            __syncthreads();
            s[tid] += a;
            __syncthreads();
        }
        if (i < nblocks) sums[i] = int(carry + s[tid] - v);   // offsets are int32: only meaningful while total < 2^31
        __syncthreads();
        if (tid == kScanBlock - 1) carry += s[tid];
        __syncthreads();
    }
    if (tid == 0) *total = carry;
}
__global__ void scan_add_kernel(int* __restrict__ out, int64_t n, const int* __restrict__ sums) {
    const int64_t i = int64_t(blockIdx.x) * kScanBlock + threadIdx.x;
    if (i < n) out[i] += sums[blockIdx.x];
}

// ---- occupancy-grid ray marching (own specification) ----------------
#ifndef KILO_RAY_STRIDE_BLOCKS
#define KILO_RAY_STRIDE_BLOCKS 32768
#endif
constexpr int64_t kRayStrideBlocks = KILO_RAY_STRIDE_BLOCKS;   // grid of the ray-striding kernels (4 waves each)
struct March {
    Grid g;          // scene box + occupancy resolution
    float t_near, t_far, step;
};

// Interval k of a ray is [t0 + k*step, min(t0 + (k+1)*step, t1)) with [t0, t1] = ray-box overlap
// clipped to [near, far]; it is kept when the occupancy voxel of its mid-point is set.
__device__ __forceinline__ bool ray_box(const March& m, const float o[3], const float d[3], float& t0, float& t1) {
    t0 = m.t_near, t1 = m.t_far;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float inv = 1.0f / d[a];
        float ta = (m.g.mn[a] - o[a]) * inv, tb = (m.g.mx[a] - o[a]) * inv;
        if (ta > tb) { const float t = ta; ta = tb; tb = t; }
        if (d[a] == 0.f) {  // parallel to the slab: inside for all t or never
            if (o[a] < m.g.mn[a] || o[a] > m.g.mx[a]) return false;
            continue;
        }
        t0 = fmaxf(t0, ta), t1 = fminf(t1, tb);
    }
    return t0 < t1;
}

// PINV: every box edge (max - min) is a power of two (the reference scene: [-1, 1]^3), so the division of the voxel
// formula is a multiplication by the exact reciprocal -- same bits, a third fewer VALU instructions per step.
// BRICK: `occ` is the bit-brick copy of the grid (occ_brick_kernel below): one 64-bit word per 4 x 4 x 4 voxels, bit
// (x & 3) * 16 + (y & 3) * 4 + (z & 3).  A ray's 64 consecutive steps cross ~40 voxels -- ~40 cache lines of the byte grid,
// about ten of the brick grid, whose 2 MB (256^3) also stay in every XCD's L2 where the 16.7 MB of bytes do not.
template <bool WRITE, bool PINV, bool BRICK = false>
__global__ __launch_bounds__(256) void kilo_march_kernel(March m, const float* __restrict__ rays_o,
                                                         const float* __restrict__ rays_d, int64_t n_rays,
                                                         const unsigned char* __restrict__ occ, int* __restrict__ counts,
                                                         const int* __restrict__ offsets, float* __restrict__ pts,
                                                         int* __restrict__ ray_idx, float* __restrict__ t_start,
                                                         float* __restrict__ t_end,
                                                         unsigned long long* __restrict__ masks, int max_chunks) {
    const int lane = threadIdx.x & 63;
    const int64_t ray = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (ray >= n_rays) return;
    const float o[3] = {rays_o[ray * 3], rays_o[ray * 3 + 1], rays_o[ray * 3 + 2]};
    const float d[3] = {rays_d[ray * 3], rays_d[ray * 3 + 1], rays_d[ray * 3 + 2]};
    float t0, t1;
    int count = 0;
    unsigned long long held = 0;
    if (ray_box(m, o, d, t0, t1)) {
        const int nsteps = int(ceilf((t1 - t0) / m.step));
        const int base = WRITE ? offsets[ray] : 0;
        for (int k0 = 0; k0 < nsteps; k0 += 64) {
            const int k = k0 + lane;
            const float ts = t0 + float(k) * m.step;
            const float te = fminf(ts + m.step, t1);
            const float tm = 0.5f * (ts + te);
            const float p[3] = {o[0] + d[0] * tm, o[1] + d[1] * tm, o[2] + d[2] * tm};
            bool keep = k < nsteps && ts < t1;
            if (keep) {
                int v[3];
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    const float s = (PINV ? (p[a] - m.g.mn[a]) * m.g.inv[a] : (p[a] - m.g.mn[a]) / (m.g.mx[a] - m.g.mn[a])) *
                                    float(m.g.res[a]);
                    v[a] = int(fminf(fmaxf(s, 0.f), float(m.g.res[a]) - 1.0f));
                }
                if (BRICK) {
                    const int by = (m.g.res[1] + 3) >> 2, bz = (m.g.res[2] + 3) >> 2;
                    const unsigned long long word =
                        reinterpret_cast<const unsigned long long*>(occ)[(int64_t(v[0] >> 2) * by + (v[1] >> 2)) * bz + (v[2] >> 2)];
                    keep = (word >> (((v[0] & 3) << 4) | ((v[1] & 3) << 2) | (v[2] & 3))) & 1ull;
                } else {
                    keep = occ[(int64_t(v[0]) * m.g.res[1] + v[1]) * m.g.res[2] + v[2]] != 0;
                }
            }
            const unsigned long long mask = __ballot(keep);
            if (!WRITE && masks) {   // lane c & 63 holds chunk c's keep mask; flushed 64 chunks (or the tail) at a time
                const int c = k0 >> 6;
                if (lane == (c & 63)) held = mask;
                if ((c & 63) == 63 || k0 + 64 >= nsteps) {
                    if (lane <= (c & 63) && (c & ~63) + lane < max_chunks) masks[ray * max_chunks + (c & ~63) + lane] = held;
                }
            }
            if (WRITE && keep) {
                const int pos = base + count + __popcll(mask & ((1ull << lane) - 1ull));
                pts[int64_t(pos) * 3] = p[0], pts[int64_t(pos) * 3 + 1] = p[1], pts[int64_t(pos) * 3 + 2] = p[2];
                ray_idx[pos] = int(ray);
                t_start[pos] = ts, t_end[pos] = te;
            }
            count += __popcll(mask);
        }
    }
    if (!WRITE && lane == 0) counts[ray] = count;
}

// The byte grid -> its bit bricks: one thread per brick, voxels beyond the grid's edge read as empty.
__global__ __launch_bounds__(256) void occ_brick_kernel(const unsigned char* __restrict__ occ, int rx, int ry, int rz,
                                                        unsigned long long* __restrict__ bricks) {
    const int bx = (rx + 3) >> 2, by = (ry + 3) >> 2, bz = (rz + 3) >> 2;
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= int64_t(bx) * by * bz) return;
    const int z0 = int(i % bz) * 4, y0 = int((i / bz) % by) * 4, x0 = int(i / (int64_t(bz) * by)) * 4;
    unsigned long long word = 0ull;
#pragma unroll
    for (int dx = 0; dx < 4; ++dx)
#pragma unroll
        for (int dy = 0; dy < 4; ++dy)
#pragma unroll
            for (int dz = 0; dz < 4; ++dz) {
                const int x = x0 + dx, y = y0 + dy, z = z0 + dz;
                if (x < rx && y < ry && z < rz && occ[(int64_t(x) * ry + y) * rz + z] != 0)
                    word |= 1ull << ((dx << 4) | (dy << 2) | dz);
            }
    bricks[i] = word;
}

// Pass 2 when pass 1 kept its ballots: no occupancy lookups, rays without samples leave at once and empty chunks
// are skipped, so only kept samples cost arithmetic.  Same expressions for ts / te / p as kilo_march_kernel: the
// two write paths are bit-identical (tests/test_gpu_kilo.py).
__global__ __launch_bounds__(256) void kilo_march_write_masks_kernel(March m, const float* __restrict__ rays_o,
                                                                     const float* __restrict__ rays_d, int64_t n_rays,
                                                                     const int* __restrict__ counts,
                                                                     const int* __restrict__ offsets,
                                                                     const unsigned long long* __restrict__ masks,
                                                                     int max_chunks, float* __restrict__ pts,
                                                                     int* __restrict__ ray_idx, float* __restrict__ t_start,
                                                                     float* __restrict__ t_end) {
    const int lane = threadIdx.x & 63;
    for (int64_t ray = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6); ray < n_rays; ray += int64_t(gridDim.x) * 4) {   // (kilo_composite_kernel)
    const int n = counts[ray];
    if (n == 0) continue;
    const float o[3] = {rays_o[ray * 3], rays_o[ray * 3 + 1], rays_o[ray * 3 + 2]};
    const float d[3] = {rays_d[ray * 3], rays_d[ray * 3 + 1], rays_d[ray * 3 + 2]};
    float t0, t1;
    if (!ray_box(m, o, d, t0, t1)) continue;
    const int nsteps = int(ceilf((t1 - t0) / m.step));
    const int nchunks = min((nsteps + 63) >> 6, max_chunks);
    int pos0 = offsets[ray];
    for (int c0 = 0; c0 < nchunks; c0 += 64) {
        const unsigned long long mine = c0 + lane < nchunks ? masks[ray * max_chunks + c0 + lane] : 0ull;
        unsigned long long nonempty = __ballot(mine != 0ull);
        while (nonempty) {
            const int cl = __ffsll((long long)nonempty) - 1;
            nonempty &= nonempty - 1;
            const unsigned long long mask = (unsigned long long)__shfl((long long)mine, cl);
            if ((mask >> lane) & 1ull) {
                const int k = (c0 + cl) * 64 + lane;
                const float ts = t0 + float(k) * m.step;
                const float te = fminf(ts + m.step, t1);
                const float tm = 0.5f * (ts + te);
                const int pos = pos0 + __popcll(mask & ((1ull << lane) - 1ull));
                pts[int64_t(pos) * 3] = o[0] + d[0] * tm, pts[int64_t(pos) * 3 + 1] = o[1] + d[1] * tm;
                pts[int64_t(pos) * 3 + 2] = o[2] + d[2] * tm;
                ray_idx[pos] = int(ray);
                t_start[pos] = ts, t_end[pos] = te;
            }
            pos0 += __popcll(mask);
        }
    }
    }
}

// ---- packed compositing with early ray termination (own specification) -----------------------
__global__ __launch_bounds__(256) void kilo_composite_kernel(const float4* __restrict__ raw, const float* __restrict__ t_start,
                                                             const float* __restrict__ t_end,
                                                             const int* __restrict__ offsets, const int* __restrict__ counts,
                                                             int64_t n_rays, float ert_eps, int white_bkgd,
                                                             float* __restrict__ rgb_map, float* __restrict__ depth_map,
                                                             float* __restrict__ acc_map) {
    const int lane = threadIdx.x & 63;
    // Waves stride over the rays: most rays of a frame march through no occupied voxel (78 % on the bench's scene), and a
    // wave launched per such ray costs more than the ray -- an empty ray is one load and one store here.
    for (int64_t ray = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6); ray < n_rays; ray += int64_t(gridDim.x) * 4) {
    const int base = offsets[ray], n = counts[ray];
    if (n == 0) {   // what the sums below come to without samples: background, no depth, no opacity
        if (lane == 63) {
            const float bg = white_bkgd ? 1.f : 0.f;
            rgb_map[ray * 3] = bg, rgb_map[ray * 3 + 1] = bg, rgb_map[ray * 3 + 2] = bg;
            depth_map[ray] = 0.f;
            acc_map[ray] = 0.f;
        }
        continue;
    }
    float carry = 1.f, sr = 0.f, sg = 0.f, sb = 0.f, sa = 0.f, sd = 0.f;
    // the next chunk's loads are issued before this chunk's scan (the termination test serialises the chunks of a
    // ray otherwise: load -> scan -> test -> load ...); a chunk fetched behind the termination point is not used
    auto fetch = [&](int c0, float4& rv, float& ts, float& te) {
        const int s = c0 + lane;
        const bool in = s < n;
        rv = in ? raw[base + s] : make_float4(0.f, 0.f, 0.f, 0.f);
        ts = in ? t_start[base + s] : 0.f, te = in ? t_end[base + s] : 0.f;
    };
    float4 rv_n;
    float ts_n, te_n;
    fetch(0, rv_n, ts_n, te_n);
    for (int c0 = 0; c0 < n && carry >= ert_eps; c0 += 64) {  // whole chunks past the threshold are skipped
        const bool in = c0 + lane < n;
        const float4 rv = rv_n;
        const float ts = ts_n, te = te_n;
        fetch(c0 + 64, rv_n, ts_n, te_n);
        const PackedChunk k = packed_chunk(rv, te - ts, in, carry);
        const float w = k.T >= ert_eps ? k.T * k.alpha : 0.f;  // early ray termination: samples behind T < eps are dropped
        sr += w * k.cr, sg += w * k.cg, sb += w * k.cb;
        sa += w;
        sd += w * 0.5f * (ts + te);
    }
    sr = nrf_wave_scan_add(sr), sg = nrf_wave_scan_add(sg), sb = nrf_wave_scan_add(sb), sa = nrf_wave_scan_add(sa),
    sd = nrf_wave_scan_add(sd);
    if (lane == 63) {  // the inclusive sums end in lane 63
        const float bg = white_bkgd ? 1.f - sa : 0.f;  // volumn_renderer.py:116
        rgb_map[ray * 3] = sr + bg, rgb_map[ray * 3 + 1] = sg + bg, rgb_map[ray * 3 + 2] = sb + bg;
        depth_map[ray] = sd / fmaxf(sa, 1.1920929e-07f);
        acc_map[ray] = sa;
    }
    }
}

}  // namespace

NRF_EXPORT int64_t nrf_kilo_packed_floats(int num_cells) {
    return num_cells <= 0 ? NRF_ERR_SHAPE : int64_t(num_cells) * kKiloCellFloats;
}

NRF_EXPORT int nrf_kilo_pack_weights(const float* const* tensors, int num_cells, float* packed, nrf_stream_t stream) {
    if (!tensors || !packed) return NRF_ERR_NULL;
    if (num_cells <= 0) return NRF_ERR_SHAPE;
    KiloSrc s;
    for (int i = 0; i < 12; ++i) {
        if (!tensors[i]) return NRF_ERR_NULL;
        s.t[i] = tensors[i];
    }
    if (!nrf_aligned16(packed)) return NRF_ERR_ALIGN;
    nrf_launch_begin();
    kilo_pack_kernel<<<dim3(unsigned(num_cells)), 256, 0, static_cast<hipStream_t>(stream)>>>(s, num_cells, nullptr, nullptr, packed);
    return nrf_launch_status();
}

NRF_EXPORT int nrf_kilo_pack_weights_cells(const float* const* tensors, int num_cells, const int32_t* cell_mask, float* packed,
                                           float* packed_t, nrf_stream_t stream) {
    if (!tensors || !packed || !cell_mask) return NRF_ERR_NULL;
    if (num_cells <= 0) return NRF_ERR_SHAPE;
    KiloSrc s;
    for (int i = 0; i < 12; ++i) {
        if (!tensors[i]) return NRF_ERR_NULL;
        s.t[i] = tensors[i];
    }
    if (!nrf_aligned16(packed) || (packed_t && !nrf_aligned16(packed_t))) return NRF_ERR_ALIGN;
    nrf_launch_begin();
    kilo_pack_kernel<<<dim3(unsigned(num_cells)), 256, 0, static_cast<hipStream_t>(stream)>>>(s, num_cells, cell_mask, nullptr, packed);
    if (packed_t) kilo_pack_t_launch(s.t, num_cells, cell_mask, nullptr, packed_t, static_cast<hipStream_t>(stream));
    return nrf_launch_status();
}

NRF_EXPORT int nrf_kilo_route(const float* pts, int64_t n, const float* aabb_host, const int* res_host, int32_t* cell,
                              nrf_stream_t stream) {
    if (n < 0) return NRF_ERR_SHAPE;
    if (n == 0) return NRF_OK;
    if (!pts || !aabb_host || !res_host || !cell) return NRF_ERR_NULL;
    const Grid g = make_grid(aabb_host, res_host);
    if (g.res[0] <= 0 || g.res[1] <= 0 || g.res[2] <= 0) return NRF_ERR_SHAPE;
    const int chunk = route_chunk(n);
    const int64_t blocks = (n + chunk - 1) / chunk;   // one contiguous chunk per block
    if (blocks > 0x7fffffffLL) return NRF_ERR_SHAPE;
    nrf_launch_begin();
    kilo_route_kernel<<<dim3(unsigned(blocks)), 256, 0, static_cast<hipStream_t>(stream)>>>(g, pts, n, chunk, cell, nullptr);
    return nrf_launch_status();
}

// workspace layout: kilo_common.h, kilo_sort_ws
NRF_EXPORT int64_t nrf_kilo_forward_workspace_bytes(int64_t n_points, int num_cells) {
    if (n_points < 0 || num_cells <= 0) return NRF_ERR_SHAPE;
    return 2 * a256(n_points * 4) + 4 * a256((int64_t(num_cells) + 1) * 4);
}

// Network.forward (kilonerf/network.py:223-311) for a flat batch: raw[i] = [r, g, b, sigma] of point i
// evaluated by the MLP of its cell.  dirs: (n, 3), or (m, 3) with dir_index (n,) int32 into it.
// `src` (optional): the 12 parameter tensors; the cells the batch routes to (and `done`, if given, does not mark as
// packed already) are packed from them into `packed` (and `packed_t`, if given) between the routing histogram and the
// MLP launch.
static int kilo_forward_impl(int kind, const KiloSrc* src, const int32_t* done, float* packed_t, const float* packed, int num_cells, const float* aabb_host,
                             const int* res_host, const float* pts, const float* dirs, const int32_t* dir_index,
                             int64_t n_points, float* raw, void* workspace, int64_t workspace_bytes, nrf_stream_t stream_,
                             const void* dir_table = nullptr, int64_t n_dirs = 0) {
    if (n_points < 0 || num_cells <= 0 || n_points > (1LL << 28) || kind < 0 || kind > 2) return NRF_ERR_SHAPE;  // 32-bit byte offsets in the kernels
    if (src && kind != 0) return NRF_ERR_SHAPE;   // packing on the way is the training step's, on fp32 records
    if (n_points == 0) return NRF_OK;
    if (!packed || !aabb_host || !res_host || !pts || !(dirs || dir_table) || !raw || !workspace) return NRF_ERR_NULL;
    if (dir_table && (!dir_index || n_dirs <= 0 || n_dirs > (1LL << 24) || !nrf_aligned16(dir_table))) return NRF_ERR_SHAPE;   // 32-bit row offsets
    if (packed_t && !nrf_aligned16(packed_t)) return NRF_ERR_ALIGN;
    if (res_host[0] * res_host[1] * res_host[2] != num_cells || num_cells > 8192) return NRF_ERR_SHAPE;
    if (workspace_bytes < nrf_kilo_forward_workspace_bytes(n_points, num_cells)) return NRF_ERR_WORKSPACE;
    if (!nrf_aligned16(packed) || !nrf_aligned16(raw) || !nrf_aligned16(workspace)) return NRF_ERR_ALIGN;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const KiloSortWs ws = kilo_sort_ws(workspace, n_points, num_cells);
    int *cell = ws.cell, *perm = ws.perm, *hist = ws.hist, *seg_start = ws.seg_start, *cursor = ws.cursor,
        *job_prefix = ws.job_prefix;
    int* job_counter = hist + num_cells;   // the spare word behind the histogram: zeroed with it below, the MLP kernels draw their jobs from it
    const Grid g = make_grid(aabb_host, res_host);
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (cus <= 0) cus = 256;
    nrf_launch_begin();
    (void)hipMemsetAsync(hist, 0, size_t(num_cells + 1) * 4, stream);
    const unsigned rb = unsigned((n_points + kSortChunk - 1) / kSortChunk);   // n_points <= 2^28: at most 65536 blocks
    const int rchunk = route_chunk(n_points);
    kilo_route_kernel<<<dim3(unsigned((n_points + rchunk - 1) / rchunk)), 256, size_t(num_cells) * 4, stream>>>(
        g, pts, n_points, rchunk, cell, hist);
    // tiles of one cell per weight load: up to kJobTiles, fewer for small batches so that every wavefront gets jobs
    int64_t jt = (n_points + 32 * 4 * int64_t(cus) * 4 - 1) / (32 * 4 * int64_t(cus) * 4);
    const int job_tiles = int(jt < 4 ? 4 : jt > kJobTiles ? kJobTiles : jt);
    // A frame has ~25 jobs per wave and the counter evens out their lengths.  A training batch has about one: there every
    // wave's first ticket -- thousands of return-value atomics on one word at the kernel's start -- would be the kernel's
    // time (KiloNeRF training step 0.47 -> 0.55 ms), so few jobs are dealt round-robin, without the counter.
    if (n_points / (32 * int64_t(job_tiles)) + num_cells < 8 * int64_t(2 * cus) * 4) job_counter = nullptr;
    if (src) {   // hist[cell] = samples routed to the cell: pack those cells, nobody reads the others' records
        kilo_pack_kernel<<<dim3(unsigned(num_cells)), 256, 0, stream>>>(*src, num_cells, hist, done, const_cast<float*>(packed));
        if (packed_t) kilo_pack_t_launch(src->t, num_cells, hist, done, packed_t, stream);
    }
    kilo_cell_scan_kernel<<<1, 1024, 0, stream>>>(hist, num_cells, job_tiles, seg_start, cursor, job_prefix);
    kilo_scatter_kernel<<<dim3(rb), 256, size_t(num_cells) * 4, stream>>>(cell, n_points, num_cells, cursor, perm);
#define NRF_KILO_ARGS packed, num_cells, job_tiles, seg_start, job_prefix, perm, pts, dirs, dir_index, reinterpret_cast<float4*>(raw), job_counter
    const dim3 grid(unsigned(2 * cus));
    if (dir_table) {   // per-ray encodings (nrf_kilo_encode_dirs): `dirs` is not read
        const float* tab = static_cast<const float*>(dir_table);
        const float* guard = reinterpret_cast<const float*>(static_cast<const char*>(dir_table) + dir_guard_offset(n_dirs));
        if (kind == 2)
            kilo_mlp_x3_kernel<true, true><<<grid, 256, 0, stream>>>(packed, num_cells, job_tiles, seg_start, job_prefix, perm, pts, tab,
                                                                      dir_index, reinterpret_cast<float4*>(raw), job_counter);
        else if (kind == 1)
            kilo_mlp_h2_kernel<true, true><<<grid, 256, 0, stream>>>(packed, num_cells, job_tiles, seg_start, job_prefix, perm, pts, tab,
                                                                      dir_index, reinterpret_cast<float4*>(raw), job_counter, guard);
        else
            kilo_mlp_kernel<true, true><<<grid, 256, 0, stream>>>(packed, num_cells, job_tiles, seg_start, job_prefix, perm, pts, tab,
                                                                   dir_index, reinterpret_cast<float4*>(raw), job_counter);
    } else if (kind == 2) {   // records from nrf_kilo_pack_weights_bf16x3
        if (dir_index) kilo_mlp_x3_kernel<true, false><<<grid, 256, 0, stream>>>(NRF_KILO_ARGS);
        else kilo_mlp_x3_kernel<false, false><<<grid, 256, 0, stream>>>(NRF_KILO_ARGS);
    } else if (kind == 1) {   // records from nrf_kilo_pack_weights_fp16x2
        if (dir_index) kilo_mlp_h2_kernel<true, false><<<grid, 256, 0, stream>>>(NRF_KILO_ARGS, nullptr);
        else kilo_mlp_h2_kernel<false, false><<<grid, 256, 0, stream>>>(NRF_KILO_ARGS, nullptr);
    } else if (dir_index)
        kilo_mlp_kernel<true, false><<<grid, 256, 0, stream>>>(NRF_KILO_ARGS);
    else
        kilo_mlp_kernel<false, false><<<grid, 256, 0, stream>>>(NRF_KILO_ARGS);
#undef NRF_KILO_ARGS
    return nrf_launch_status();
}

NRF_EXPORT int nrf_kilo_forward(const float* packed, int num_cells, const float* aabb_host, const int* res_host,
                                const float* pts, const float* dirs, const int32_t* dir_index, int64_t n_points,
                                float* raw, void* workspace, int64_t workspace_bytes, nrf_stream_t stream) {
    return kilo_forward_impl(0, nullptr, nullptr, nullptr, packed, num_cells, aabb_host, res_host, pts, dirs, dir_index, n_points, raw,
                             workspace, workspace_bytes, stream);
}

// The same with the cells' matrices on the fp16 matrix cores (two fp16 terms per operand, three MFMAs per product):
// packed_h2 from nrf_kilo_pack_weights_fp16x2 (nrf_kilo_packed_floats(num_cells) floats, like the fp32 records).
NRF_EXPORT int nrf_kilo_forward_fp16x2(const float* packed_h2, int num_cells, const float* aabb_host, const int* res_host,
                                       const float* pts, const float* dirs, const int32_t* dir_index, int64_t n_points,
                                       float* raw, void* workspace, int64_t workspace_bytes, nrf_stream_t stream) {
    return kilo_forward_impl(1, nullptr, nullptr, nullptr, packed_h2, num_cells, aabb_host, res_host, pts, dirs, dir_index, n_points,
                             raw, workspace, workspace_bytes, stream);
}

NRF_EXPORT int64_t nrf_kilo_dir_table_bytes(int64_t n_dirs) { return n_dirs <= 0 ? NRF_ERR_SHAPE : dir_table_bytes(n_dirs); }

NRF_EXPORT int nrf_kilo_encode_dirs(const float* dirs, int64_t n_dirs, int kind, void* table, nrf_stream_t stream) {
    if (n_dirs < 0 || n_dirs > (1LL << 24) || kind < 0 || kind > 2) return NRF_ERR_SHAPE;
    if (n_dirs == 0) return NRF_OK;
    if (!dirs || !table) return NRF_ERR_NULL;
    if (!nrf_aligned16(table)) return NRF_ERR_ALIGN;
    unsigned* tab = static_cast<unsigned*>(table);
    float* guard = reinterpret_cast<float*>(static_cast<char*>(table) + dir_guard_offset(n_dirs));
    const dim3 grid(unsigned((n_dirs + 127) / 128));
    nrf_launch_begin();
    if (kind == 2) kilo_dir_table_kernel<2><<<grid, 256, 0, static_cast<hipStream_t>(stream)>>>(dirs, n_dirs, tab, guard);
    else if (kind == 1) kilo_dir_table_kernel<1><<<grid, 256, 0, static_cast<hipStream_t>(stream)>>>(dirs, n_dirs, tab, guard);
    else kilo_dir_table_kernel<0><<<grid, 256, 0, static_cast<hipStream_t>(stream)>>>(dirs, n_dirs, tab, guard);
    return nrf_launch_status();
}

NRF_EXPORT int nrf_kilo_forward_encoded(const float* packed, int kind, int num_cells, const float* aabb_host, const int* res_host,
                                        const float* pts, const void* dir_table, int64_t n_dirs, const int32_t* dir_index,
                                        int64_t n_points, float* raw, void* workspace, int64_t workspace_bytes,
                                        nrf_stream_t stream) {
    if (!dir_table || !dir_index) return NRF_ERR_NULL;
    return kilo_forward_impl(kind, nullptr, nullptr, nullptr, packed, num_cells, aabb_host, res_host, pts, nullptr, dir_index,
                             n_points, raw, workspace, workspace_bytes, stream, dir_table, n_dirs);
}

// The tiny MLPs at fp32 accuracy on the bf16 matrix cores (three bf16 terms per operand, six MFMAs per product):
// packed_x3 from nrf_kilo_pack_weights_bf16x3 (nrf_kilo_packed_bf16x3_floats(num_cells) floats).
NRF_EXPORT int64_t nrf_kilo_packed_bf16x3_floats(int num_cells) {
    return num_cells <= 0 ? NRF_ERR_SHAPE : int64_t(num_cells) * kKiloCellFloatsX3;
}

NRF_EXPORT int nrf_kilo_pack_weights_bf16x3(const float* const* tensors, int num_cells, float* packed_x3, nrf_stream_t stream) {
    if (!tensors || !packed_x3) return NRF_ERR_NULL;
    if (num_cells <= 0) return NRF_ERR_SHAPE;
    KiloSrc s;
    for (int i = 0; i < 12; ++i) {
        if (!tensors[i]) return NRF_ERR_NULL;
        s.t[i] = tensors[i];
    }
    if (!nrf_aligned16(packed_x3)) return NRF_ERR_ALIGN;
    nrf_launch_begin();
    kilo_pack_x3_kernel<<<dim3(unsigned(num_cells)), 256, 0, static_cast<hipStream_t>(stream)>>>(s, num_cells, packed_x3);
    return nrf_launch_status();
}

NRF_EXPORT int nrf_kilo_forward_bf16x3(const float* packed_x3, int num_cells, const float* aabb_host, const int* res_host,
                                       const float* pts, const float* dirs, const int32_t* dir_index, int64_t n_points,
                                       float* raw, void* workspace, int64_t workspace_bytes, nrf_stream_t stream) {
    return kilo_forward_impl(2, nullptr, nullptr, nullptr, packed_x3, num_cells, aabb_host, res_host, pts, dirs, dir_index, n_points,
                             raw, workspace, workspace_bytes, stream);
}

#ifdef KF_STAMP
NRF_EXPORT int nrf_debug_kf_stamps(unsigned long long* out16, int reset) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_kf_stamps), sizeof(g_kf_stamps)) != hipSuccess) return NRF_ERR_RUNTIME;
    if (reset) {
        unsigned long long z[16] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_kf_stamps), z, sizeof(z)) != hipSuccess) return NRF_ERR_RUNTIME;
    }
    return NRF_OK;
}
#endif

NRF_EXPORT int nrf_kilo_pack_weights_fp16x2(const float* const* tensors, int num_cells, float* packed_h2, nrf_stream_t stream) {
    if (!tensors || !packed_h2) return NRF_ERR_NULL;
    if (num_cells <= 0) return NRF_ERR_SHAPE;
    KiloSrc s;
    for (int i = 0; i < 12; ++i) {
        if (!tensors[i]) return NRF_ERR_NULL;
        s.t[i] = tensors[i];
    }
    if (!nrf_aligned16(packed_h2)) return NRF_ERR_ALIGN;
    nrf_launch_begin();
    kilo_pack_h2_kernel<<<dim3(unsigned(num_cells)), 256, 0, static_cast<hipStream_t>(stream)>>>(s, num_cells, packed_h2);
    return nrf_launch_status();
}

NRF_EXPORT int nrf_kilo_forward_packing(const float* const* tensors, const int32_t* packed_cells, float* packed, float* packed_t,
                                        int num_cells,
                                        const float* aabb_host, const int* res_host, const float* pts, const float* dirs,
                                        const int32_t* dir_index, int64_t n_points, float* raw, void* workspace,
                                        int64_t workspace_bytes, nrf_stream_t stream) {
    if (!tensors) return NRF_ERR_NULL;
    KiloSrc s;
    for (int i = 0; i < 12; ++i) {
        if (!tensors[i]) return NRF_ERR_NULL;
        s.t[i] = tensors[i];
    }
    return kilo_forward_impl(0, &s, packed_cells, packed_t, packed, num_cells, aabb_host, res_host, pts, dirs, dir_index, n_points, raw,
                             workspace, workspace_bytes, stream);
}

// ---- rendering ------------------------------------------------------------------------------
static int check_march(const float* rays_o, const float* rays_d, const unsigned char* occ, const float* aabb,
                       const int* occ_res, float t_near, float t_far, float step) {
    if (!rays_o || !rays_d || !occ || !aabb || !occ_res) return NRF_ERR_NULL;
    if (!(step > 0.f) || !(t_far > t_near) || occ_res[0] <= 0 || occ_res[1] <= 0 || occ_res[2] <= 0) return NRF_ERR_SHAPE;
    return NRF_OK;
}

// workspace: scan block sums (ceil(n/1024) + 1 ints), then the keep masks of pass 1 (one uint64 per ray and 64-step
// chunk; ceil((far - near) / step) steps at most, one chunk of slack for the rounding of the step count)
static int march_max_chunks(float t_near, float t_far, float step) {
    if (!(step > 0.f) || !(t_far > t_near)) return -1;
    const double steps = std::ceil((double(t_far) - double(t_near)) / double(step));
    if (!(steps < double(1 << 22))) return -1;
    return int(steps) / 64 + 2;
}
static int64_t march_scan_bytes(int64_t n_rays) { return a256(((n_rays + kScanBlock - 1) / kScanBlock + 1) * 4); }

NRF_EXPORT int64_t nrf_kilo_march_workspace_bytes(int64_t n_rays, float t_near, float t_far, float step) {
    const int mc = march_max_chunks(t_near, t_far, step);
    if (n_rays < 0 || mc < 0) return NRF_ERR_SHAPE;
    return march_scan_bytes(n_rays) + a256(n_rays * mc * 8);
}

NRF_EXPORT int64_t nrf_occ_bricks_bytes(const int* occ_res_host) {
    if (!occ_res_host) return NRF_ERR_NULL;
    if (occ_res_host[0] <= 0 || occ_res_host[1] <= 0 || occ_res_host[2] <= 0) return NRF_ERR_SHAPE;
    return a256(int64_t((occ_res_host[0] + 3) >> 2) * ((occ_res_host[1] + 3) >> 2) * ((occ_res_host[2] + 3) >> 2) * 8);
}

NRF_EXPORT int nrf_occ_pack_bricks(const uint8_t* occ, const int* occ_res_host, uint64_t* bricks, nrf_stream_t stream) {
    if (!occ || !occ_res_host || !bricks) return NRF_ERR_NULL;
    const int64_t bytes = nrf_occ_bricks_bytes(occ_res_host);
    if (bytes < 0) return int(bytes);
    if ((reinterpret_cast<uintptr_t>(bricks) & 7u) != 0) return NRF_ERR_ALIGN;
    const int64_t n = int64_t((occ_res_host[0] + 3) >> 2) * ((occ_res_host[1] + 3) >> 2) * ((occ_res_host[2] + 3) >> 2);
    if ((n + 255) / 256 > 0x7fffffffLL) return NRF_ERR_SHAPE;
    nrf_launch_begin();
    occ_brick_kernel<<<dim3(unsigned((n + 255) / 256)), 256, 0, static_cast<hipStream_t>(stream)>>>(
        occ, occ_res_host[0], occ_res_host[1], occ_res_host[2], reinterpret_cast<unsigned long long*>(bricks));
    return nrf_launch_status();
}

// Pass 1: per-ray sample counts, their exclusive prefix (offsets) and the total (device int64); the per-chunk keep
// masks stay in the workspace for pass 2.  `bricks`: the grid as nrf_occ_pack_bricks wrote it, `occ` is then not read.
static int march_count_impl(const float* rays_o, const float* rays_d, int64_t n_rays, const uint8_t* occ, const uint64_t* bricks,
                            const int* occ_res_host, const float* aabb_host, float t_near, float t_far,
                            float step, int32_t* counts, int32_t* offsets, int64_t* total, void* workspace,
                            int64_t workspace_bytes, nrf_stream_t stream_) {
    if (n_rays < 0) return NRF_ERR_SHAPE;
    if (n_rays == 0) return NRF_OK;
    if (int st = check_march(rays_o, rays_d, bricks ? reinterpret_cast<const uint8_t*>(bricks) : occ, aabb_host, occ_res_host, t_near,
                             t_far, step))
        return st;
    if (!counts || !offsets || !total || !workspace) return NRF_ERR_NULL;
    const int mc = march_max_chunks(t_near, t_far, step);
    if (mc < 0) return NRF_ERR_SHAPE;
    if (workspace_bytes < nrf_kilo_march_workspace_bytes(n_rays, t_near, t_far, step)) return NRF_ERR_WORKSPACE;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    March m{make_grid(aabb_host, occ_res_host), t_near, t_far, step};
    const bool pinv = m.g.pinv != 0;
    const int64_t blocks = (n_rays + 3) / 4;
    if (blocks > 0x7fffffffLL) return NRF_ERR_SHAPE;
    auto* masks = reinterpret_cast<unsigned long long*>(static_cast<char*>(workspace) + march_scan_bytes(n_rays));
    nrf_launch_begin();
    if (bricks && pinv)
        kilo_march_kernel<false, true, true><<<dim3((unsigned)blocks), 256, 0, stream>>>(
            m, rays_o, rays_d, n_rays, reinterpret_cast<const unsigned char*>(bricks), counts, nullptr, nullptr, nullptr, nullptr,
            nullptr, masks, mc);
    else if (bricks)
        kilo_march_kernel<false, false, true><<<dim3((unsigned)blocks), 256, 0, stream>>>(
            m, rays_o, rays_d, n_rays, reinterpret_cast<const unsigned char*>(bricks), counts, nullptr, nullptr, nullptr, nullptr,
            nullptr, masks, mc);
    else if (pinv)
        kilo_march_kernel<false, true><<<dim3((unsigned)blocks), 256, 0, stream>>>(m, rays_o, rays_d, n_rays, occ, counts, nullptr,
                                                                                   nullptr, nullptr, nullptr, nullptr, masks, mc);
    else
        kilo_march_kernel<false, false><<<dim3((unsigned)blocks), 256, 0, stream>>>(m, rays_o, rays_d, n_rays, occ, counts, nullptr,
                                                                                    nullptr, nullptr, nullptr, nullptr, masks, mc);
    int* sums = static_cast<int*>(workspace);
    const int nb = int((n_rays + kScanBlock - 1) / kScanBlock);
    scan_block_kernel<<<nb, kScanBlock, 0, stream>>>(counts, n_rays, offsets, sums);
    scan_sums_kernel<<<1, kScanBlock, 0, stream>>>(sums, nb, total);
    scan_add_kernel<<<nb, kScanBlock, 0, stream>>>(offsets, n_rays, sums);
    return nrf_launch_status();
}

NRF_EXPORT int nrf_kilo_march_count(const float* rays_o, const float* rays_d, int64_t n_rays, const uint8_t* occ,
                                    const int* occ_res_host, const float* aabb_host, float t_near, float t_far,
                                    float step, int32_t* counts, int32_t* offsets, int64_t* total, void* workspace,
                                    int64_t workspace_bytes, nrf_stream_t stream) {
    return march_count_impl(rays_o, rays_d, n_rays, occ, nullptr, occ_res_host, aabb_host, t_near, t_far, step, counts, offsets,
                            total, workspace, workspace_bytes, stream);
}

NRF_EXPORT int nrf_kilo_march_count_bricks(const float* rays_o, const float* rays_d, int64_t n_rays, const uint64_t* bricks,
                                           const int* occ_res_host, const float* aabb_host, float t_near, float t_far,
                                           float step, int32_t* counts, int32_t* offsets, int64_t* total, void* workspace,
                                           int64_t workspace_bytes, nrf_stream_t stream) {
    if (!bricks) return NRF_ERR_NULL;
    return march_count_impl(rays_o, rays_d, n_rays, nullptr, bricks, occ_res_host, aabb_host, t_near, t_far, step, counts, offsets,
                            total, workspace, workspace_bytes, stream);
}

// Pass 2: write the packed samples (mid-point positions, ray index, interval) in ray order.  With `counts` and the
// workspace pass 1 filled (same rays, same grid, same stream order) the kept steps come from its masks; with
// workspace == NULL the occupancy grid is walked again.
NRF_EXPORT int nrf_kilo_march_write(const float* rays_o, const float* rays_d, int64_t n_rays, const uint8_t* occ,
                                    const int* occ_res_host, const float* aabb_host, float t_near, float t_far,
                                    float step, const int32_t* offsets, float* pts, int32_t* ray_idx, float* t_start,
                                    float* t_end, const int32_t* counts, const void* workspace, int64_t workspace_bytes,
                                    nrf_stream_t stream_) {
    if (n_rays < 0) return NRF_ERR_SHAPE;
    if (n_rays == 0) return NRF_OK;
    if (int st = check_march(rays_o, rays_d, occ, aabb_host, occ_res_host, t_near, t_far, step)) return st;
    if (!offsets || !pts || !ray_idx || !t_start || !t_end) return NRF_ERR_NULL;
    March m{make_grid(aabb_host, occ_res_host), t_near, t_far, step};
    const bool pinv = m.g.pinv != 0;
    const int64_t blocks = (n_rays + 3) / 4;
    if (blocks > 0x7fffffffLL) return NRF_ERR_SHAPE;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (workspace) {
        if (!counts) return NRF_ERR_NULL;
        const int mc = march_max_chunks(t_near, t_far, step);
        if (mc < 0) return NRF_ERR_SHAPE;
        if (workspace_bytes < nrf_kilo_march_workspace_bytes(n_rays, t_near, t_far, step)) return NRF_ERR_WORKSPACE;
        auto* masks = reinterpret_cast<const unsigned long long*>(static_cast<const char*>(workspace) + march_scan_bytes(n_rays));
        nrf_launch_begin();
        kilo_march_write_masks_kernel<<<dim3((unsigned)(blocks < kRayStrideBlocks ? blocks : kRayStrideBlocks)), 256, 0, stream>>>(m, rays_o, rays_d, n_rays, counts, offsets,
                                                                                   masks, mc, pts, ray_idx, t_start, t_end);
        return nrf_launch_status();
    }
    nrf_launch_begin();
    if (pinv)
        kilo_march_kernel<true, true><<<dim3((unsigned)blocks), 256, 0, stream>>>(m, rays_o, rays_d, n_rays, occ, nullptr, offsets,
                                                                                  pts, ray_idx, t_start, t_end, nullptr, 0);
    else
        kilo_march_kernel<true, false><<<dim3((unsigned)blocks), 256, 0, stream>>>(m, rays_o, rays_d, n_rays, occ, nullptr, offsets,
                                                                                   pts, ray_idx, t_start, t_end, nullptr, 0);
    return nrf_launch_status();
}

NRF_EXPORT int nrf_kilo_composite(const float* raw, const float* t_start, const float* t_end, const int32_t* offsets,
                                  const int32_t* counts, int64_t n_rays, float ert_eps, int white_bkgd, float* rgb_map,
                                  float* depth_map, float* acc_map, nrf_stream_t stream) {
    if (n_rays < 0) return NRF_ERR_SHAPE;
    if (n_rays == 0) return NRF_OK;
    if (!offsets || !counts || !rgb_map || !depth_map || !acc_map) return NRF_ERR_NULL;
    const int64_t blocks = (n_rays + 3) / 4;
    if (blocks > 0x7fffffffLL) return NRF_ERR_SHAPE;
    nrf_launch_begin();
    kilo_composite_kernel<<<dim3((unsigned)(blocks < kRayStrideBlocks ? blocks : kRayStrideBlocks)), 256, 0, static_cast<hipStream_t>(stream)>>>(
        reinterpret_cast<const float4*>(raw), t_start, t_end, offsets, counts, n_rays, ert_eps, white_bkgd, rgb_map,
        depth_map, acc_map);
    return nrf_launch_status();
}
