// Backward of the 8x256 NeRF MLP for the training step: parameter gradients from d(raw).
//
// Replaces what autograd does for network.py:49-74 inside `loss.backward()`
// (src/train/trainers/trainer.py:73).  Three kinds of kernels:
//
//   dgrad  register-chained like the forward (mlp_chain.h): d(input of layer l) = W_l^T . dY_l on
//          the fp32 matrix cores with the TRANSPOSED weight stream through the same LDS-DMA ring;
//          relu' comes from the sign bits the training forward saved (4 words per lane and layer);
//          every dY_l is written once to HBM in [tile][row][32 points] layout for the wgrad GEMMs.
//          Inputs (points, directions) need no gradient, so the chain stops at layer 1's input.
//   wgrad  dW_l = dY_l . X_l^T, a (256 x K) x P GEMM with the reduction over P points split across
//          workgroups: each workgroup streams its slice of tiles through LDS (row stride 33: the
//          operand reads have the FEATURE on the lane and are bank-conflict free), keeps the full
//          256 x K partial in registers (4 waves x up to 320 accumulators), writes it once.
//          Bias gradients (row sums of dY) ride along.
//   reduce sums the per-slice partials and scatters into the flat gradient buffer in the
//          reference's (out, in) state-dict layout, undoing the kernel's encoding-slot order.
#include "mlp_chain.h"
#include "split3.h"
#include "split2h.h"

namespace {

// ------------------------------------------------------------------------------------
// dgrad
// ------------------------------------------------------------------------------------
struct OpBwdCopy {  // no activation between feature_linear and views_linears: dY8 = d(feature)
    float* dst;
    __device__ __forceinline__ f32x16 operator()(int i, f32x16 y) {
        store_block(dst, i, y);
        return y;
    }
};

constexpr int kBwdChunks = kBwdSteps / kChunkSteps;  // 136
static_assert(kBwdSteps % kChunkSteps == 0, "backward layers must be chunk aligned");

__global__ __launch_bounds__(256, 1) void mlp_dgrad_kernel(const float* __restrict__ packed,    // forward pack: heads
                                                           const float* __restrict__ packed_t,  // transposed stream
                                                           const float4* __restrict__ graw,     // (n_points, 4)
                                                           const unsigned* __restrict__ masks, int64_t n_points,
                                                           int64_t n_groups, float* __restrict__ dys) {
    __shared__ __attribute__((aligned(16))) char smem[kLdsBytes];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    {
        const float4* src = reinterpret_cast<const float4*>(packed + kOffBias);
        float4* dst = reinterpret_cast<float4*>(smem + kLdsConstOff);
        for (int i = threadIdx.x; i < kSmallFloats / 4; i += 256) dst[i] = src[i];
    }
    __syncthreads();
    const float* cst = reinterpret_cast<const float*>(smem + kLdsConstOff);
    const float* aw = cst + (kOffAlphaW - kOffBias) + h * 16;
    const float* rw = cst + (kOffRgbW - kOffBias) + h * 16;

    WeightStream st;
    ws_start(st, reinterpret_cast<const char*>(packed_t), kBwdChunks, smem, wave, lane);

    for (int64_t tg = blockIdx.x; tg < n_groups; tg += gridDim.x) {
        const int64_t tile = tg * 4 + wave;
        int64_t pidx = tile * 32 + j;
        const bool valid = pidx < n_points;
        if (!valid) pidx = n_points - 1;
        float4 g = graw[pidx];
        if (!valid) g = make_float4(0.f, 0.f, 0.f, 0.f);
        const unsigned* mrow = masks + tile * (kMaskLayers * 4 * 64) + lane;
        float* drow = dys + tile * (int64_t(kDyRows) * 32) + 4 * h * 32 + j;
        if (h == 0) {  // d(raw) as 4 rows for the head wgrad jobs
            float* r = dys + tile * (int64_t(kDyRows) * 32) + kDyRaw * 32 + j;
            r[0] = g.x, r[32] = g.y, r[64] = g.z, r[96] = g.w;
        }

        // rgb_linear^T on the VALU: dYv = relu'(hv) * (W_rgb^T . d rgb)
        f32x16 P[4], X[8], Y[8];
        {
            const unsigned m0 = mrow[(8 * 4 + 0) * 64], m1 = mrow[(8 * 4 + 1) * 64];
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                const f32x16 w0 = *reinterpret_cast<const f32x16*>(rw + kb * 32);
                const f32x16 w1 = *reinterpret_cast<const f32x16*>(rw + 128 + kb * 32);
                const f32x16 w2 = *reinterpret_cast<const f32x16*>(rw + 256 + kb * 32);
                f32x16 v;
#pragma unroll
                for (int t = 0; t < 16; ++t) v[t] = fmaf(w2[t], g.z, fmaf(w1[t], g.y, w0[t] * g.x));
                P[kb] = mask16(v, ((kb < 2 ? m0 : m1) >> (16 * (kb & 1))) & 0xffffu);
                store_block(drow + kDyV * 32, kb, P[kb]);
            }
        }
        OpCopy op_none;
        // d feature = W_views[:, :256]^T . dYv
        mfma_layer<8, 4, 0, 0, INIT_ZERO>(st, lane, nullptr, P, X, nullptr, Y, op_none);
        {   // d h7 (through feature_linear) = W_feature^T . d feature
            OpBwdCopy op{drow + kDyFeat * 32};
            mfma_layer<8, 0, 8, 0, INIT_ZERO>(st, lane, nullptr, nullptr, X, nullptr, Y, op);
        }
        for (int l = 7; l >= 1; --l) {  // dY_l = relu'(h_l) * d h_l;  d h_{l-1} = W_l^T . dY_l
            OpBwdMask op;
            op.dst = drow + l * 256 * 32;
#pragma unroll
            for (int w = 0; w < 4; ++w) op.m[w] = mrow[(l * 4 + w) * 64];
            op.aw = aw;
            op.dsig = l == 7 ? g.w : 0.f;  // h7 also feeds alpha_linear
            mfma_layer<8, 0, 8, 0, INIT_ZERO>(st, lane, nullptr, nullptr, X, nullptr, Y, op);
        }
        {   // dY_0 = relu'(h_0) * d h_0 (the encoding needs no gradient: chain ends)
            OpBwdMask op;
            op.dst = drow;
#pragma unroll
            for (int w = 0; w < 4; ++w) op.m[w] = mrow[w * 64];
            op.aw = aw;
            op.dsig = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) (void)op(i, Y[i]);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ------------------------------------------------------------------------------------
// wgrad
// ------------------------------------------------------------------------------------
struct WgradJob {
    const float* a;      // dys: [tile][a_stride rows][32]
    const float* x;      // acts: [tile][x_stride rows][32]
    int a_stride, x_stride;
    int a_row0, a_rows;  // first dY row, valid rows (zero-filled up to M)
    int x_row0[2], x_rows[2];  // up to two row segments of X (K = sum, each a multiple of 32)
    int64_t n_tiles;
    int tiles_per_slice;
    float* partial;      // [slice][M][K]
    float* pbias;        // [slice][M]
    const unsigned* scale_words;   // fp16x2 GEMMs: the chain's scale block (mlp_fp16x2.hip: kGmax*), else nullptr
    int scale_word;                // ... and the word of this job's dY row set
};

constexpr int kLdsRow = 33;  // padded tile row: feature-on-lane operand reads hit 32 distinct banks

// One wave per 32-row block of dY (M = 32 * waves; 8 waves = 512 threads for the 256-row layers, so
// two waves share each SIMD and cover each other's LDS latency), K = 32 * KB columns per wave.
// HEAD: a single (zero-padded) row block, the 4 waves split the column blocks instead.
template <int NW, int KB, bool HEAD>
__global__ __launch_bounds__(NW * 64, NW / 4) void wgrad_kernel(WgradJob job) {
    constexpr int NT = NW * 64;
    constexpr int M = HEAD ? 32 : 32 * NW;
    constexpr int K = HEAD ? 32 * NW * KB : 32 * KB;
    __shared__ float sA[M * kLdsRow];
    __shared__ float sB[K * kLdsRow];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, kh = lane >> 5;
    const int slice = blockIdx.x;
    const int64_t t0 = int64_t(slice) * job.tiles_per_slice;
    int64_t t1 = t0 + job.tiles_per_slice;
    if (t1 > job.n_tiles) t1 = job.n_tiles;

    f32x16 acc[KB];
#pragma unroll
    for (int a = 0; a < KB; ++a)
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[a][t] = 0.f;
    float bsum = 0.f;  // thread tid < M: row sum of dY row tid

    constexpr int A4 = M * 8, B4 = K * 8;  // float4 pieces per tile (32 floats = 8 float4 per row)
    constexpr int NA = (A4 + NT - 1) / NT, NB = (B4 + NT - 1) / NT;
    float4 ra[NA], rb[NB];

    auto load_tile = [&](int64_t tile) {
        const float4* pa = reinterpret_cast<const float4*>(job.a + (tile * job.a_stride + job.a_row0) * 32);
#pragma unroll
        for (int n = 0; n < NA; ++n) {
            const int p = tid + NT * n;
            ra[n] = (p < A4 && (p >> 3) < job.a_rows) ? pa[p] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const float* xb = job.x + tile * job.x_stride * 32;
#pragma unroll
        for (int n = 0; n < NB; ++n) {
            const int p = tid + NT * n;
            const int row = p >> 3;
            if (p < B4) {
                const int src_row = row < job.x_rows[0] ? job.x_row0[0] + row : job.x_row0[1] + row - job.x_rows[0];
                rb[n] = reinterpret_cast<const float4*>(xb + src_row * 32)[p & 7];
            }
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int n = 0; n < NA; ++n) {
            const int p = tid + NT * n;
            if (p < A4) {
                float* d = sA + (p >> 3) * kLdsRow + (p & 7) * 4;
                d[0] = ra[n].x, d[1] = ra[n].y, d[2] = ra[n].z, d[3] = ra[n].w;
            }
        }
#pragma unroll
        for (int n = 0; n < NB; ++n) {
            const int p = tid + NT * n;
            if (p < B4) {
                float* d = sB + (p >> 3) * kLdsRow + (p & 7) * 4;
                d[0] = rb[n].x, d[1] = rb[n].y, d[2] = rb[n].z, d[3] = rb[n].w;
            }
        }
    };

    if (t0 < t1) load_tile(t0);
    for (int64_t tile = t0; tile < t1; ++tile) {
        __syncthreads();  // everyone is done reading the previous tile
        store_tile();
        __syncthreads();
        if (tile + 1 < t1) load_tile(tile + 1);  // in flight under the MFMAs below
        if (tid < M) {
            const float* r = sA + tid * kLdsRow;
#pragma unroll
            for (int p = 0; p < 32; ++p) bsum += r[p];
        }
        const float* pa = sA + ((HEAD ? 0 : wave * 32) + i) * kLdsRow + kh;
        const float* pb = sB + ((HEAD ? wave * KB * 32 : 0) + i) * kLdsRow + kh;
#pragma unroll 4
        for (int s = 0; s < 16; ++s) {  // 2 points per MFMA
            const float av = pa[2 * s];
            float bv[KB];
#pragma unroll
            for (int k = 0; k < KB; ++k) bv[k] = pb[k * 32 * kLdsRow + 2 * s];
#pragma unroll
            for (int k = 0; k < KB; ++k) acc[k] = NRF_MFMA(av, bv[k], acc[k]);
        }
    }
    // partial[slice][row][col]: 128 B contiguous per half-wave store
    float* out = job.partial + int64_t(slice) * M * K;
#pragma unroll
    for (int k = 0; k < KB; ++k) {
        const int rb0 = HEAD ? 0 : wave * 32;
        const int cb0 = HEAD ? (wave * KB + k) * 32 : k * 32;
#pragma unroll
        for (int t = 0; t < 16; ++t) out[int64_t(rb0 + acc_row(t, kh)) * K + cb0 + i] = acc[k][t];
    }
    if (tid < M) job.pbias[int64_t(slice) * M + tid] = bsum;
}

// The weight-gradient GEMM dW += dY . X^T on the bf16 / fp16 matrix pipe with split operands, pipelined: a tile's 32
// points are processed in two 16-point halves (one k16-step each) through two LDS stages ([stage][plane][row][16 points],
// 48-byte rows: conflict free for the fragments' ds_read_b128), and the operand split of half s + 1 runs in the MFMA gaps of
// half s -- a few VALU instructions per gap hide completely under a 32x32x16 MFMA -- instead of in a phase of its own
// between two barriers.  A thread owns the same float4 pieces of every tile (row (tid >> 2) + NT/4 * j, points
// 16 h + 4 (tid & 3) ..).  One barrier per half.  Bias gradients (row sums of dY) are summed in fp32 from the load registers.
//
// X3 (wgrad_x3p_kernel, the route that carries the reference's 24-bit operands): three bf16 terms per operand
// (split3.h), six v_mfma_f32_32x32x16_bf16 per product block, planes [dY 0 | 1 | 2 | X 0 | 1 | 2].
//
// !X3 (wgrad_h2p_kernel): two fp16 terms per operand (split2h.h) and THREE MFMAs per product into ONE accumulator.  The
// second accumulator of the forward kernels would cost KB more blocks here; instead the dY operand, whose scale is ours
// to choose, is multiplied by 2^T so that its row set's largest entry (left by mlp_dgrad_h2_kernel in the chain's scale
// block) lands in [2048, 4096), and its planes are hi, hi * 2^-11 and the UNSCALED low term: a_hi.b_hi, (a_hi 2^-11).b_lo'
// and a_lo.b_hi then share one scale.  The two small planes go subnormal for entries below 2^-15 of the maximum, whose
// absolute error (2^-25 of the scaled operand) is 2^-37 of the largest term; X keeps the forward's hi | lo * 2^11.  The
// sums are unscaled (exactly) when the partials are written.  Planes [dY hi | hs | lo | X hi | lo'].
//
// Both forms share the loader: operand halves are requested kWgradDepth - 1 half-steps ahead by inline-asm loads and
// waited for by counted s_waitcnt (round 2 built this for the fp16 form; round 3 gave the bf16x3 form the same loader --
// it had run on compiler-placed vmcnt(0) waits, i.e. without any prefetch).
constexpr int kX3HalfRow = 48;
constexpr int kWgradDepth = 3;   // register sets of loaded halves (2: 4 % slower; 4: spills, 2.5 % slower)

template <int NW, int KB, bool X3>
__device__ __forceinline__ void wgrad_p_body(const WgradJob& job) {
    constexpr int NT = NW * 64;
    constexpr int M = 32 * NW, K = 32 * KB;
    constexpr int kPA = M * kX3HalfRow, kPB = K * kX3HalfRow, kStage = 3 * kPA + (X3 ? 3 : 2) * kPB;
    extern __shared__ __attribute__((aligned(16))) char s_x3[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, kh = lane >> 5, q = tid & 3, r0row = tid >> 2;
    const int slice = blockIdx.x;
    const int64_t t0 = int64_t(slice) * job.tiles_per_slice;
    int64_t t1 = t0 + job.tiles_per_slice;
    if (t1 > job.n_tiles) t1 = job.n_tiles;
    const int64_t n_half = t1 > t0 ? 2 * (t1 - t0) : 0;   // half-steps of this slice
    // stored dY = chain value * 2^-S; chain maximum of the row set in scale_words[scale_word]
    int T = 0;
    if constexpr (!X3) {
        const unsigned row_max = job.scale_words[job.scale_word];
        if (((row_max >> 23) & 0xffu) != 0)
            T = scale_exponent(job.scale_words[0], kChainScaleTarget) + scale_exponent(row_max, kWgradScaleTarget);
        T = T > 120 ? 120 : T < -120 ? -120 : T;
    }
    const float sc = exp2i(T), unsc = exp2i(-T);

    f32x16 acc[KB];
#pragma unroll
    for (int a = 0; a < KB; ++a)
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[a][t] = 0.f;

    constexpr int NAh = M * 4 / NT;                 // float4 pieces of dY per thread and half
    constexpr int NBh = (K * 4 + NT - 1) / NT;      // ... of X
    static_assert(M * 4 % NT == 0 && NAh >= 1, "whole dY pieces per thread");
    constexpr int NI = NAh + NBh;                   // items per half
    constexpr int D = kWgradDepth;                  // register sets: a half is loaded D - 1 half-steps before its split
    // [half mod D][item]: dY items first.  The loads are inline asm and so are their waits: left to the compiler, every
    // consumption of a set came out as s_waitcnt vmcnt(0) or (1) -- its counter model gives up on the predicated loads
    // and the unrolled loop's back edge -- i.e. each half-step waited for the loads it had just issued for three
    // half-steps later, and the kernel spent half its wave time in s_waitcnt.  In issue order the sets younger than the
    // one about to be split hold 2 NI loads: the wait is vmcnt(2 NI), and names the set's registers so that no read of
    // them can be scheduled ahead of it.
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    f32x4v reg[D][NI];
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
        for (int n = 0; n < NI; ++n) reg[d][n] = f32x4v{0.f, 0.f, 0.f, 0.f};
    // in-out: the load lands in the register its wait and its readers name.  One wave per SIMD (NW = 4: 9 accumulator
    // blocks + 21 loaded float4 do not fit 256 VGPRs): the sets live in accumulation registers -- as "v" operands the
    // allocator parked them there itself, by copies placed between a load and its wait.
    constexpr bool kSetsInAgprs = NW == 4;
    float bsum[NAh];   // lanes with q == 0: row sum of dY row r0row + (NT / 4) * j
#pragma unroll
    for (int n = 0; n < NAh; ++n) bsum[n] = 0.f;
    // address = uniform base (SGPR pair: tile, row group, half) + this lane's 32-bit byte offset: per-lane 64-bit
    // pointers cost the 256 x 256 kernel the registers it spilled (and a scratch reload's wait drains the queue)
    auto gload = [](f32x4v& dst, const float* base, unsigned voff) {
        if constexpr (kSetsInAgprs) asm volatile("global_load_dwordx4 %0, %1, %2" : "+a"(dst) : "v"(voff), "s"(base) : "memory");
        else asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(dst) : "v"(voff), "s"(base) : "memory");
    };
    // lane offsets inside a group of NT / 4 rows: row r0row, points 4 q ..; the last X group may run past row K - 1:
    // those lanes load the last live row and store nothing (split_sub)
    const unsigned voff = unsigned(r0row * 32 + q * 4) * 4u;
    unsigned voff_last = voff;
    if constexpr ((NT / 4) * NBh > K) {   // (no tail when the groups tile K exactly: one register less in the 256 x 256 kernel)
        const int row = r0row + (NT / 4) * (NBh - 1);
        if (row >= K) voff_last = unsigned((K - 1 - (NT / 4) * (NBh - 1)) * 32 + q * 4) * 4u;
    }
    static_assert(K - 1 - (NT / 4) * (NBh - 1) >= 0, "the last X group starts at a live row");
    // hs = half-step index inside the slice.  Past the slice's end the last half is loaded again and never used: every
    // half-step then issues NI loads, "two sets are younger" holds from the prologue to the last half-step, and ONE
    // wait statement serves them all (a second form for the tail sat in another branch, and the allocator bridged the
    // two with copies of the set placed BEFORE that wait: stale data in a slice's last half-steps).
    auto load_half = [&](int64_t hs, f32x4v (&r)[NI]) {
        hs = hs < n_half ? hs : n_half - 1;
        const int64_t tile = t0 + (hs >> 1);
        const int h = int(hs & 1);
        const float* pa = job.a + (tile * job.a_stride + job.a_row0) * 32 + 16 * h;
#pragma unroll
        for (int j = 0; j < NAh; ++j) gload(r[j], pa + (NT / 4) * j * 32, voff);   // all M rows are live (launcher)
        const float* xb = job.x + tile * job.x_stride * 32 + 16 * h;
#pragma unroll
        for (int j = 0; j < NBh; ++j) {
            const int row0 = (NT / 4) * j;   // the group's first row: segments are whole groups (launcher)
            const int src_row0 = row0 < job.x_rows[0] ? job.x_row0[0] + row0 : job.x_row0[1] + row0 - job.x_rows[0];
            gload(r[NAh + j], xb + src_row0 * 32, j == NBh - 1 ? voff_last : voff);
        }
    };
    // the set `r` is complete once at most the two younger sets' loads are in flight
    auto wait_set = [&](f32x4v (&r)[NI]) {
        static_assert(NI == 3 || NI == 4 || NI == 7, "one operand list per NI");
        static_assert(D == 3, "vmcnt((D - 1) NI)");
        static_assert(kSetsInAgprs == (NI == 7), "the NI = 7 operand list is the accumulation-register one");
        if constexpr (NI == 3) asm volatile("s_waitcnt vmcnt(6)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]) : : "memory");
        else if constexpr (NI == 4)
            asm volatile("s_waitcnt vmcnt(8)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : : "memory");
        else
            asm volatile("s_waitcnt vmcnt(14)"
                         : "+a"(r[0]), "+a"(r[1]), "+a"(r[2]), "+a"(r[3]), "+a"(r[4]), "+a"(r[5]), "+a"(r[6])
                         :
                         : "memory");
    };
    // sub-slice w of the split of one half (items x 2 pairs x 3 phases); the item's 8-byte stores ride on its last one
    unsigned p1[3], p2[3];   // the planes' words of the item's first / second pair
    float sx, sy, r0, r1;
    auto pair_phase = [&](bool is_a, float x, float y, int phase, unsigned (&p)[3]) {
        if constexpr (X3) {   // one third of split_pair (split3.h): a pair's 15 instructions spread over three sub-slices
            if (phase == 0) {
                p[0] = pk_bf16(x, y);
            } else if (phase == 1) {
                r0 = __fsub_rn(x, bf_lo(p[0])), r1 = __fsub_rn(y, bf_hi(p[0]));
                p[1] = pk_bf16(r0, r1);
            } else {
                p[2] = pk_bf16(__fsub_rn(r0, bf_lo(p[1])), __fsub_rn(r1, bf_hi(p[1])));
            }
        } else if (phase == 0) {
            sx = is_a ? x * sc : x, sy = is_a ? y * sc : y;
            p[0] = pk_f16(sx, sy);
        } else if (phase == 1) {
            r0 = f16_residual_lo(sx, p[0]), r1 = f16_residual_hi(sy, p[0]);
            if (is_a) {
                const f16x2 hs = __builtin_bit_cast(f16x2, p[0]) * (f16x2){(_Float16)kLoUnscale, (_Float16)kLoUnscale};
                p[1] = __builtin_bit_cast(unsigned, hs);
            }
        } else {
            if (is_a) p[2] = pk_f16(r0, r1);
            else p[1] = pk_f16_scaled(r0, r1);
        }
    };
    auto split_sub = [&](char* stage, f32x4v (&r)[NI], int w) {
        const int item = w / 6, ph = w % 6;
        const bool is_a = item < NAh;
        const int row = r0row + (NT / 4) * (is_a ? item : item - NAh);
        if (!is_a && row >= K) return;
        const f32x4v v = r[item];
        char* d = stage + (is_a ? 0 : 3 * kPA) + row * kX3HalfRow + q * 8;
        const int plane = is_a ? kPA : kPB;
        if constexpr (X3) {
            // each pair's three plane words are stored as soon as the pair is split (4-byte stores): one set of plane
            // words live instead of two -- the three registers the 256 x 256 kernel did not have
            if (ph < 3) pair_phase(is_a, v.x, v.y, ph, p1);
            else pair_phase(is_a, v.z, v.w, ph - 3, p1);
            if (ph == 2 || ph == 5) {
                char* dp = d + (ph == 5 ? 4 : 0);
                *reinterpret_cast<unsigned*>(dp) = p1[0];
                *reinterpret_cast<unsigned*>(dp + plane) = p1[1];
                *reinterpret_cast<unsigned*>(dp + 2 * plane) = p1[2];
            }
        } else {
            if (ph < 3) pair_phase(is_a, v.x, v.y, ph, p1);
            else pair_phase(is_a, v.z, v.w, ph - 3, p2);
        }
        if (ph == 5) {
            if constexpr (!X3) {
                *reinterpret_cast<uint2*>(d) = make_uint2(p1[0], p2[0]);
                *reinterpret_cast<uint2*>(d + plane) = make_uint2(p1[1], p2[1]);
                if (is_a) *reinterpret_cast<uint2*>(d + 2 * plane) = make_uint2(p1[2], p2[2]);
            }
            if (is_a) {
                // bias gradient: the row's 16 points of this half = the four lanes of a quad
                float t = (v.x + v.y) + (v.z + v.w);
                t += nrf_dpp<0xB1>(0.f, t);   // quad_perm [1, 0, 3, 2]
                t += nrf_dpp<0x4E>(0.f, t);   // quad_perm [2, 3, 0, 1]
                bsum[item] += t;
            }
        }
    };
    auto drain_sets = [&] {   // everything asked for so far has landed
        if constexpr (NI == 3)
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(reg[0][0]), "+v"(reg[0][1]), "+v"(reg[0][2]), "+v"(reg[1][0]), "+v"(reg[1][1]),
                         "+v"(reg[1][2]), "+v"(reg[2][0]), "+v"(reg[2][1]), "+v"(reg[2][2]) : : "memory");
        else if constexpr (NI == 4)
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(reg[0][0]), "+v"(reg[0][1]), "+v"(reg[0][2]), "+v"(reg[0][3]), "+v"(reg[1][0]),
                         "+v"(reg[1][1]), "+v"(reg[1][2]), "+v"(reg[1][3]), "+v"(reg[2][0]), "+v"(reg[2][1]), "+v"(reg[2][2]),
                         "+v"(reg[2][3]) : : "memory");
        else   // (a statement takes at most 30 operands and an in-out one counts twice: set by set)
            for (int d = 0; d < D; ++d)
                asm volatile("s_waitcnt vmcnt(0)" : "+a"(reg[d][0]), "+a"(reg[d][1]), "+a"(reg[d][2]), "+a"(reg[d][3]), "+a"(reg[d][4]),
                             "+a"(reg[d][5]), "+a"(reg[d][6]) : : "memory");
    };
    constexpr int W = NI * 6;                    // sub-slices per half
    constexpr int G = X3 ? 6 : 3;                // MFMAs (= gaps) per column block
    constexpr int S = G * KB;                    // MFMA gaps per half

    // one half-step: MFMAs on stage `cur` (KB groups of G), the split of the next half into the other stage in the gaps
    auto half_step = [&](const char* cur, char* nxt, f32x4v (&rn)[NI], bool do_split) {
        const char* pa = cur + (wave * 32 + i) * kX3HalfRow + 16 * kh;
        const char* pb = cur + 3 * kPA + i * kX3HalfRow + 16 * kh;
        auto frag_b = [&](int k, int t) { return *reinterpret_cast<const u32x4*>(pb + t * kPB + k * 32 * kX3HalfRow); };
        // fp16 form: both B planes double-buffered over the column blocks.  bf16x3 form: only plane 0, which the block's
        // first and last products read; planes 1 and 2 are dead after the block's gaps 3 and 0, so the next block's
        // fragments land in the same registers (24 -> 16 fragment registers: the 256 x 256 kernel spilled without it).
        u32x4 a[3], b[2][2], b2x;
#pragma unroll
        for (int t = 0; t < 3; ++t) a[t] = *reinterpret_cast<const u32x4*>(pa + t * kPA);
        b[0][0] = frag_b(0, 0), b[0][1] = frag_b(0, 1);
        if constexpr (X3) b2x = frag_b(0, 2);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < KB; ++k) {
            const int c = k & 1;
            const bool more = k + 1 < KB;
#define NRF_WGP_GAP(MFMA, m, A, B, RD)                                                  \
    acc[k] = MFMA(A, B, acc[k]);                                                        \
    RD;                                                                                 \
    if (do_split) {                                                                     \
        _Pragma("unroll") for (int w = ((k * G + m) * W) / S; w < ((k * G + m + 1) * W) / S; ++w) split_sub(nxt, rn, w); \
    }                                                                                   \
    __builtin_amdgcn_sched_barrier(0);
            if constexpr (X3) {   // six products with i + j <= 2, small terms first (split3.h); plane 1 lives in b[0][1]
                NRF_WGP_GAP(NRF_MFMA_X3, 0, a[0], b2x, if (more) b[c ^ 1][0] = frag_b(k + 1, 0))
                NRF_WGP_GAP(NRF_MFMA_X3, 1, a[2], b[c][0], if (more) b2x = frag_b(k + 1, 2))
                NRF_WGP_GAP(NRF_MFMA_X3, 2, a[1], b[0][1], (void)0)
                NRF_WGP_GAP(NRF_MFMA_X3, 3, a[0], b[0][1], (void)0)
                NRF_WGP_GAP(NRF_MFMA_X3, 4, a[1], b[c][0], if (more) b[0][1] = frag_b(k + 1, 1))
                NRF_WGP_GAP(NRF_MFMA_X3, 5, a[0], b[c][0], (void)0)
            } else {
                NRF_WGP_GAP(NRF_MFMA_F16, 0, a[2], b[c][0], if (more) b[c ^ 1][0] = frag_b(k + 1, 0))   // small terms first: a_lo . b_hi
                NRF_WGP_GAP(NRF_MFMA_F16, 1, a[1], b[c][1], if (more) b[c ^ 1][1] = frag_b(k + 1, 1))   // (a_hi 2^-11) . b_lo'
                NRF_WGP_GAP(NRF_MFMA_F16, 2, a[0], b[c][0], (void)0)                                    // a_hi . b_hi
            }
#undef NRF_WGP_GAP
        }
    };

    if (n_half > 0) {
        char* st0 = s_x3;
        char* st1 = s_x3 + kStage;
#pragma unroll
        for (int d = 0; d < D; ++d)
            load_half(d, reg[d]);
        wait_set(reg[0]);
#pragma unroll
        for (int w = 0; w < W; ++w) split_sub(st0, reg[0], w);   // the first half: nothing to hide it under
        __syncthreads();
        // half-step hs: MFMAs on stage hs & 1, the split of half hs + 1 (set (hs + 1) % D) into the other stage; set
        // hs % D was split one half-step ago and is refilled with half hs + D.  (n_half is even.)
        constexpr int U = D % 2 ? 2 * D : D;   // unrolled so that set (hs % D) and stage (hs & 1) are compile-time
        // Half-step hs + d, d a literal: loads for half hs + d + D into set d % D, the wait for set (d + 1) % D, the
        // MFMAs of stage d & 1 with the split of half hs + d + 1 in their gaps.  The main loop runs whole groups of U
        // half-steps without a test inside; the remaining 0 / 2 / 4 (n_half is even) are NESTED ifs -- side by side they
        // would give the control-flow graph paths that skip one half-step and run the next, never taken, but on them
        // the counted wait would be wrong and tests/test_asm_lint.py cannot know that they are not.
#ifdef WG_ABL_NOLOAD   // (timing-only ablations, tools/debug/lib_ab.sh: wrong results)
#define NRF_WG_LOAD(d)
#else   // (asking for both halves of a tile together, whole 128-byte rows per burst, measured 5 % slower)
#define NRF_WG_LOAD(d) load_half(hs + (d) + D, reg[(d) % D]);
#endif
#ifdef WG_ABL_NOSPLIT
#define NRF_WG_SPLIT(d) false
#else
#define NRF_WG_SPLIT(d) hs + (d) + 1 < n_half
#endif
#ifdef WG_ABL_NOBAR
#define NRF_WG_SYNC
#else
#define NRF_WG_SYNC __syncthreads();
#endif
#define NRF_WG_HALF(d)                                                                                  \
    {                                                                                                   \
        NRF_WG_LOAD(d)                                                                                  \
        wait_set(reg[((d) + 1) % D]);                                                                   \
        half_step((d) & 1 ? st1 : st0, (d) & 1 ? st0 : st1, reg[((d) + 1) % D], NRF_WG_SPLIT(d));       \
        NRF_WG_SYNC                                                                                     \
    }
        static_assert(U == 6, "six half-steps written out below");
        int64_t hs = 0;
        for (; hs + U <= n_half; hs += U) {
            NRF_WG_HALF(0) NRF_WG_HALF(1) NRF_WG_HALF(2) NRF_WG_HALF(3) NRF_WG_HALF(4) NRF_WG_HALF(5)
        }
        if (hs + 2 <= n_half) {
            // the slice's last 2 or 4 half-steps: their halves hs + 1, hs + 2 are in sets 1, 2 (possibly still in flight);
            // half hs + 3 goes into set 0 now, and ONE wait for everything, naming all three sets, ends the loading: a
            // load past the end issued here would be dead on arrival and its registers handed out under it
            load_half(hs + 3, reg[0]);
            drain_sets();
            half_step(st0, st1, reg[1], true);
            NRF_WG_SYNC
            half_step(st1, st0, reg[2], hs + 2 < n_half);
            NRF_WG_SYNC
            if (hs + 4 <= n_half) {
                half_step(st0, st1, reg[0], true);
                NRF_WG_SYNC
                half_step(st1, st0, reg[0], false);
                NRF_WG_SYNC
            }
        }
#undef NRF_WG_HALF
#undef NRF_WG_LOAD
#undef NRF_WG_SPLIT
#undef NRF_WG_SYNC
    }
    // The loads past the slice's end are still in flight and their registers dead: nothing may be given those registers
    // before the loads have landed (tests/test_asm_lint.py checks that nothing is).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float* out = job.partial + int64_t(slice) * M * K;
#pragma unroll
    for (int k = 0; k < KB; ++k) {
#pragma unroll
        for (int t = 0; t < 16; ++t) out[int64_t(wave * 32 + acc_row(t, kh)) * K + k * 32 + i] = X3 ? acc[k][t] : acc[k][t] * unsc;
    }
    if (q == 0) {
#pragma unroll
        for (int n = 0; n < NAh; ++n) job.pbias[int64_t(slice) * M + r0row + (NT / 4) * n] = bsum[n];
    }
}

template <int NW, int KB>
__global__ __launch_bounds__(NW * 64, 1) void wgrad_h2p_kernel(WgradJob job) { wgrad_p_body<NW, KB, false>(job); }

template <int NW, int KB>
__global__ __launch_bounds__(NW * 64, 1) void wgrad_x3p_kernel(WgradJob job) { wgrad_p_body<NW, KB, true>(job); }

template <int NW, int KB>
static int launch_wgrad_x3p(const WgradJob& job, unsigned slices, hipStream_t stream) {
    constexpr int bytes = 2 * 3 * (32 * NW + 32 * KB) * kX3HalfRow;
    static_assert(bytes <= 160 * 1024, "LDS budget");
    static bool attr_set[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_x3p_kernel<NW, KB>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                bytes) != hipSuccess)
            return NRF_ERR_LAUNCH;
        attr_set[dev] = true;
    }
    if (job.a_rows != 32 * NW) return NRF_ERR_SHAPE;   // the kernel's dY loads are unconditional
    // ... and a group of NW * 16 X rows lies in one segment (the clamped tail of the last group aside)
    if (job.x_rows[0] < 32 * KB && job.x_rows[0] % (NW * 16) != 0) return NRF_ERR_SHAPE;
    wgrad_x3p_kernel<NW, KB><<<dim3(slices), NW * 64, bytes, stream>>>(job);
    return NRF_OK;
}

template <int NW, int KB>
static int launch_wgrad_h2p(const WgradJob& job, unsigned slices, hipStream_t stream) {
    constexpr int bytes = 2 * (3 * 32 * NW + 2 * 32 * KB) * kX3HalfRow;
    static_assert(bytes <= 160 * 1024, "LDS budget");
    static bool attr_set[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_h2p_kernel<NW, KB>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                bytes) != hipSuccess)
            return NRF_ERR_LAUNCH;
        attr_set[dev] = true;
    }
    if (job.a_rows != 32 * NW) return NRF_ERR_SHAPE;   // the kernel's dY loads are unconditional
    // ... and a group of NW * 16 X rows lies in one segment (the clamped tail of the last group aside)
    if (job.x_rows[0] < 32 * KB && job.x_rows[0] % (NW * 16) != 0) return NRF_ERR_SHAPE;
    wgrad_h2p_kernel<NW, KB><<<dim3(slices), NW * 64, bytes, stream>>>(job);
    return NRF_OK;
}

struct ColMap {
    int seg_cols[2];   // columns of each X segment
    int seg_kind[2];   // 0: hidden (identity), L > 0: encoding with L frequencies in kernel slot order
    int seg_col0[2];   // first reference column of the segment
};

__device__ __forceinline__ int map_col(const ColMap& cm, int c) {
    int seg = 0;
    if (c >= cm.seg_cols[0]) { c -= cm.seg_cols[0]; seg = 1; }
    if (cm.seg_kind[seg] == 0) return cm.seg_col0[seg] + c;
    // encoding rows are saved at row 32*kb + acc_row(t, h): invert to slot n = 16*kb + t and half h
    const int kb = c >> 5, r = c & 31;
    const int h = (r >> 2) & 1, t = (r & 3) + 4 * (r >> 3);
    const int col = pe_col(kb * 16 + t, h, cm.seg_kind[seg]);
    return col < 0 ? -1 : cm.seg_col0[seg] + col;
}

// alpha_linear / rgb_linear weight gradients: dW[a][k] = sum_p dA[a][p] * X[k][p] with NA <= 3 rows of d raw -- a job for
// HBM (X is read once, K * 128 B per tile) and the VALU, not for a 32-row MFMA tile with one or three live rows
// (wgrad_kernel<4, *, true>: 2.1x the time).  A thread owns one float4 (4 points) of row r of each 32-row pass; the
// eight threads of a row are summed at the end.  partial: [slice][NA][K], pbias: [slice][NA]; fp32 throughout.
template <int NA, int K>
__global__ __launch_bounds__(256) void head_wgrad_kernel(WgradJob job) {
    constexpr int P = K / 32;
    const int tid = threadIdx.x, q = tid & 7, r = tid >> 3;
    const int slice = blockIdx.x;
    const int64_t t0 = int64_t(slice) * job.tiles_per_slice;
    int64_t t1 = t0 + job.tiles_per_slice;
    if (t1 > job.n_tiles) t1 = job.n_tiles;
    float acc[P][NA], bsum[NA];
#pragma unroll
    for (int a = 0; a < NA; ++a) {
        bsum[a] = 0.f;
#pragma unroll
        for (int p = 0; p < P; ++p) acc[p][a] = 0.f;
    }
    for (int64_t tile = t0; tile < t1; ++tile) {
        const float* pa = job.a + (tile * job.a_stride + job.a_row0) * 32 + q * 4;
        const float* xb = job.x + (tile * job.x_stride + job.x_row0[0] + r) * 32 + q * 4;
        float4 da[NA], xv[P];
#pragma unroll
        for (int p = 0; p < P; ++p) xv[p] = *reinterpret_cast<const float4*>(xb + p * 32 * 32);
#pragma unroll
        for (int a = 0; a < NA; ++a) da[a] = *reinterpret_cast<const float4*>(pa + a * 32);
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            bsum[a] += (da[a].x + da[a].y) + (da[a].z + da[a].w);
#pragma unroll
            for (int p = 0; p < P; ++p)
                acc[p][a] = fmaf(xv[p].w, da[a].w, fmaf(xv[p].z, da[a].z, fmaf(xv[p].y, da[a].y, fmaf(xv[p].x, da[a].x, acc[p][a]))));
        }
    }
    auto sum8 = [](float v) {   // over the eight consecutive lanes of a row
        v += __shfl_xor(v, 1), v += __shfl_xor(v, 2), v += __shfl_xor(v, 4);
        return v;
    };
    float* out = job.partial + int64_t(slice) * NA * K;
#pragma unroll
    for (int a = 0; a < NA; ++a) {
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const float v = sum8(acc[p][a]);
            if (q == 0) out[a * K + p * 32 + r] = v;
        }
        const float b = sum8(bsum[a]);
        if (tid == 0) job.pbias[int64_t(slice) * NA + a] = b;
    }
}

// Sum of the split-P partials, scattered into the reference's (out, in) layout.  A block is 256 columns x kRedY
// slice groups: the 256 slices of one element are summed by kRedY threads (16 independent loads in flight each) and
// combined through LDS -- with one thread per element the 12 reduces of a backward pass ran at 52 GB/s.
// (the summation order of a bias element changes with kRedY; the weight elements' order does too: fp32 sums)
constexpr int kRedY = 16, kRedX = 64;
__global__ __launch_bounds__(kRedX * kRedY) void wgrad_reduce_kernel(const float* __restrict__ partial,
                                                                  const float* __restrict__ pbias, int n_slices, int M,
                                                                  int K, int m_valid, ColMap cm, float* __restrict__ gw,
                                                                  int ldw, float* __restrict__ gb) {
    __shared__ float s_part[kRedY][kRedX];
    const int idx = blockIdx.x * kRedX + threadIdx.x, y = threadIdx.y;
    const bool in = idx < M * K;
    float s = 0.f;
    if (in) {
#pragma unroll 16
        for (int sl = y; sl < n_slices; sl += kRedY) s += partial[int64_t(sl) * M * K + idx];
    }
    s_part[y][threadIdx.x] = s;
    __syncthreads();
    if (y == 0 && in) {
        const int m = idx / K, c = idx - m * K;
        const int col = map_col(cm, c);
        if (m < m_valid && col >= 0) {
            float t = s_part[0][threadIdx.x];
#pragma unroll
            for (int k = 1; k < kRedY; ++k) t += s_part[k][threadIdx.x];
            gw[int64_t(m) * ldw + col] = t;
        }
    }
    // bias: the same split over the slice groups (one thread walking all 256 slices was a 256-deep load chain:
    // 60 of the kernel's 65 microseconds)
    __syncthreads();
    const bool bin = idx < m_valid;
    float bsum = 0.f;
    if (bin) {
#pragma unroll 16
        for (int sl = y; sl < n_slices; sl += kRedY) bsum += pbias[int64_t(sl) * M + idx];
    }
    s_part[y][threadIdx.x] = bsum;
    __syncthreads();
    if (y == 0 && bin) {
        float t = s_part[0][threadIdx.x];
#pragma unroll
        for (int k = 1; k < kRedY; ++k) t += s_part[k][threadIdx.x];
        gb[idx] = t;
    }
}

}  // namespace

// flat gradient layout of one model: the reference's state-dict order (include/nerf_amd.h layer
// order), each layer weight (out, in) row-major followed by its bias
static const int kLayerOut[12] = {256, 256, 256, 256, 256, 256, 256, 256, 256, 1, 128, 3};
static const int kLayerIn[12] = {63, 256, 256, 256, 256, 319, 256, 256, 256, 256, 283, 128};

NRF_EXPORT int64_t nrf_model_param_floats(void) {
    int64_t n = 0;
    for (int l = 0; l < 12; ++l) n += int64_t(kLayerOut[l]) * kLayerIn[l] + kLayerOut[l];
    return n;  // 595,844
}

NRF_EXPORT int64_t nrf_mlp_dys_bytes(int64_t n_rays, int n_samples) {
    const int64_t t = nrf_mlp_train_tiles(n_rays, n_samples);
    return t < 0 ? t : t * int64_t(kDyRows) * 32 * 4;
}

static int wgrad_slices(int64_t n_tiles) { return int(n_tiles < 256 ? n_tiles : 256); }

NRF_EXPORT int64_t nrf_mlp_backward_workspace_bytes(int64_t n_rays, int n_samples) {
    const int64_t t = nrf_mlp_train_tiles(n_rays, n_samples);
    if (t < 0) return t;
    const int64_t slices = wgrad_slices(t);
    return nrf_mlp_dys_bytes(n_rays, n_samples) + slices * (256 * 320 + 256) * 4 + 256;   // + the fp16x2 chain's scale word
}

// packed_t_x3 != nullptr: the data-gradient chain on the three-term bf16 kernel (mlp_bf16x3.hip) instead of mlp_dgrad_kernel
// packed_t_h2 != nullptr: the chain on the two-term fp16 kernel (mlp_fp16x2.hip) and the weight-gradient GEMMs on wgrad_h2p_kernel
static int mlp_backward_impl(const float* packed, const float* packed_t, const void* packed_t_x3, const float* grad_raw,
                             const float* acts, const uint32_t* masks, int64_t n_rays, int n_samples, void* workspace,
                             int64_t workspace_bytes, float* grad_flat, nrf_stream_t stream_, const void* packed_t_h2 = nullptr) {
    if (n_rays < 0 || n_samples <= 0 || n_samples % 32 != 0) return NRF_ERR_SHAPE;
    if (!packed || !packed_t || !grad_raw || !acts || !masks || !workspace || !grad_flat) return NRF_ERR_NULL;
    if (!nrf_aligned16(packed) || !nrf_aligned16(packed_t) || !nrf_aligned16(grad_raw) || !nrf_aligned16(workspace) ||
        !nrf_aligned16(acts))
        return NRF_ERR_ALIGN;
    if (workspace_bytes < nrf_mlp_backward_workspace_bytes(n_rays, n_samples)) return NRF_ERR_WORKSPACE;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int64_t n_points = n_rays * n_samples;
    const int64_t n_tiles = n_points / 32;
    if (n_tiles == 0) return NRF_OK;
    const int64_t groups = (n_tiles + 3) / 4;
    float* dys = static_cast<float*>(workspace);
    float* partial = dys + nrf_mlp_dys_bytes(n_rays, n_samples) / 4;
    const int slices = wgrad_slices(n_tiles);
    float* pbias = partial + int64_t(slices) * 256 * 320;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (cus <= 0) cus = 256;
    nrf_launch_begin();
    unsigned* gmax = reinterpret_cast<unsigned*>(pbias + int64_t(slices) * 256);   // the fp16x2 chain's scale block
    if (packed_t_h2) {
        if (const int rc = launch_dgrad_h2(packed, packed_t_h2, grad_raw, masks, n_points, dys, gmax, stream)) return rc;
    } else if (packed_t_x3) {
        if (const int rc = launch_dgrad_x3(packed, packed_t_x3, grad_raw, masks, n_points, dys, stream)) return rc;
    } else {
        mlp_dgrad_kernel<<<dim3(unsigned(groups < cus ? groups : cus)), 256, 0, stream>>>(
            packed, packed_t, reinterpret_cast<const float4*>(grad_raw), masks, n_points, groups, dys);
    }

    // offsets of each layer inside grad_flat
    int64_t woff[12], boff[12], o = 0;
    for (int l = 0; l < 12; ++l) {
        woff[l] = o;
        o += int64_t(kLayerOut[l]) * kLayerIn[l];
        boff[l] = o;
        o += kLayerOut[l];
    }
    const int tps = int((n_tiles + slices - 1) / slices);
    auto job = [&](int a_row0, int a_rows, int xr0, int xn0, int xr1, int xn1, int scale_word = 0) {
        WgradJob j;
        j.scale_words = gmax, j.scale_word = scale_word;
        j.a = dys, j.x = acts, j.a_stride = kDyRows, j.x_stride = kActRows;
        j.a_row0 = a_row0, j.a_rows = a_rows;
        j.x_row0[0] = xr0, j.x_rows[0] = xn0, j.x_row0[1] = xr1, j.x_rows[1] = xn1;
        j.n_tiles = n_tiles, j.tiles_per_slice = tps, j.partial = partial, j.pbias = pbias;
        return j;
    };
    auto reduce = [&](int M, int K, int m_valid, ColMap cm, int layer) {
        const int total = M * K;
        wgrad_reduce_kernel<<<(total + kRedX - 1) / kRedX, dim3(kRedX, kRedY), 0, stream>>>(partial, pbias, slices, M, K, m_valid, cm,
                                                                     grad_flat + woff[layer], kLayerIn[layer],
                                                                     grad_flat + boff[layer]);
    };
    const dim3 grid(slices);
    // pts_linears.0: dY0 x encoding
    const bool h2 = packed_t_h2 != nullptr;   // the fp16x2 route: the four wide GEMMs on wgrad_h2p_kernel, scaled per row set
    const bool x3 = packed_t_x3 != nullptr;   // the bf16x3 route: these GEMMs on the bf16 pipe with three-term operands
    if (h2) { if (const int rc = launch_wgrad_h2p<8, 2>(job(0, 256, kActPe, 64, 0, 0, kGmaxLayer0), slices, stream)) return rc; }
    else if (x3) { if (const int rc = launch_wgrad_x3p<8, 2>(job(0, 256, kActPe, 64, 0, 0), slices, stream)) return rc; }
    else wgrad_kernel<8, 2, false><<<grid, 512, 0, stream>>>(job(0, 256, kActPe, 64, 0, 0));
    reduce(256, 64, 256, ColMap{{64, 0}, {10, 0}, {0, 0}}, 0);
    for (int l = 1; l <= 7; ++l) {
        if (l == 5) {  // cat([encoding, h4]) (network.py:58)
            if (h2) {
                // 320 columns as the 256 hidden ones + the 64 of the encoding, two GEMMs over the same dY rows (480 against
                // 570 us for the one-pass bf16x3 kernel; wgrad_h2p<8, 10> itself needs 96 B of scratch per lane and was slower still)
                if (const int rc = launch_wgrad_h2p<8, 8>(job(5 * 256, 256, kActH + 4 * 256, 256, 0, 0, kGmaxLayer0 + 5), slices, stream)) return rc;
                reduce(256, 256, 256, ColMap{{256, 0}, {0, 0}, {63, 0}}, 5);
                if (const int rc = launch_wgrad_h2p<8, 2>(job(5 * 256, 256, kActPe, 64, 0, 0, kGmaxLayer0 + 5), slices, stream)) return rc;
                reduce(256, 64, 256, ColMap{{64, 0}, {10, 0}, {0, 0}}, 5);
                continue;
            }
            if (x3) {   // the same two GEMMs on the three-term kernel
                if (const int rc = launch_wgrad_x3p<8, 8>(job(5 * 256, 256, kActH + 4 * 256, 256, 0, 0), slices, stream)) return rc;
                reduce(256, 256, 256, ColMap{{256, 0}, {0, 0}, {63, 0}}, 5);
                if (const int rc = launch_wgrad_x3p<8, 2>(job(5 * 256, 256, kActPe, 64, 0, 0), slices, stream)) return rc;
                reduce(256, 64, 256, ColMap{{64, 0}, {10, 0}, {0, 0}}, 5);
                continue;
            }
            wgrad_kernel<8, 10, false><<<grid, 512, 0, stream>>>(job(5 * 256, 256, kActPe, 64, kActH + 4 * 256, 256));
            reduce(256, 320, 256, ColMap{{64, 256}, {10, 0}, {0, 63}}, 5);
        } else {
            if (h2) { if (const int rc = launch_wgrad_h2p<8, 8>(job(l * 256, 256, kActH + (l - 1) * 256, 256, 0, 0, kGmaxLayer0 + l), slices, stream)) return rc; }
            else if (x3) { if (const int rc = launch_wgrad_x3p<8, 8>(job(l * 256, 256, kActH + (l - 1) * 256, 256, 0, 0), slices, stream)) return rc; }
            else wgrad_kernel<8, 8, false><<<grid, 512, 0, stream>>>(job(l * 256, 256, kActH + (l - 1) * 256, 256, 0, 0));
            reduce(256, 256, 256, ColMap{{256, 0}, {0, 0}, {0, 0}}, l);
        }
    }
    // feature_linear: d feature x h7
    if (h2) { if (const int rc = launch_wgrad_h2p<8, 8>(job(kDyFeat, 256, kActH + 7 * 256, 256, 0, 0, kGmaxFeat), slices, stream)) return rc; }
    else if (x3) { if (const int rc = launch_wgrad_x3p<8, 8>(job(kDyFeat, 256, kActH + 7 * 256, 256, 0, 0), slices, stream)) return rc; }
    else wgrad_kernel<8, 8, false><<<grid, 512, 0, stream>>>(job(kDyFeat, 256, kActH + 7 * 256, 256, 0, 0));
    reduce(256, 256, 256, ColMap{{256, 0}, {0, 0}, {0, 0}}, 8);
    // views_linears.0: dYv x cat([feature, dir encoding]) (network.py:63)
    if (h2) { if (const int rc = launch_wgrad_h2p<4, 9>(job(kDyV, 128, kActFeat, 256, kActDpe, 32, kGmaxViews), slices, stream)) return rc; }
    else if (x3) { if (const int rc = launch_wgrad_x3p<4, 9>(job(kDyV, 128, kActFeat, 256, kActDpe, 32), slices, stream)) return rc; }
    else wgrad_kernel<4, 9, false><<<grid, 256, 0, stream>>>(job(kDyV, 128, kActFeat, 256, kActDpe, 32));
    reduce(128, 288, 128, ColMap{{256, 32}, {0, 4}, {0, 256}}, 10);
    // alpha_linear: d sigma (raw row 3) x h7;  rgb_linear: d rgb (raw rows 0..2) x hv
    // (four times the slices of the wide GEMMs: a workgroup is 256 threads with 32 KB of loads in flight, several fit a CU;
    // [slices][3][256] floats fit the partial buffer many times over)
    {
        const int head_slices = int(n_tiles < 1024 ? n_tiles : 1024);
        const int head_tps = int((n_tiles + head_slices - 1) / head_slices);
        auto head_job = [&](int a_row0, int a_rows, int xr0, int xn0) {
            WgradJob j = job(a_row0, a_rows, xr0, xn0, 0, 0);
            j.tiles_per_slice = head_tps;
            return j;
        };
        auto head_reduce = [&](int M, int K, int layer) {
            wgrad_reduce_kernel<<<(M * K + kRedX - 1) / kRedX, dim3(kRedX, kRedY), 0, stream>>>(
                partial, pbias, head_slices, M, K, M, ColMap{{K, 0}, {0, 0}, {0, 0}}, grad_flat + woff[layer], kLayerIn[layer],
                grad_flat + boff[layer]);
        };
        head_wgrad_kernel<1, 256><<<dim3(head_slices), 256, 0, stream>>>(head_job(kDyRaw + 3, 1, kActH + 7 * 256, 256));
        head_reduce(1, 256, 9);
        head_wgrad_kernel<3, 128><<<dim3(head_slices), 256, 0, stream>>>(head_job(kDyRaw, 3, kActHv, 128));
        head_reduce(3, 128, 11);
    }
    return nrf_launch_status();
}

NRF_EXPORT int nrf_mlp_backward(const float* packed, const float* packed_t, const float* grad_raw, const float* acts,
                                const uint32_t* masks, int64_t n_rays, int n_samples, void* workspace,
                                int64_t workspace_bytes, float* grad_flat, nrf_stream_t stream) {
    return mlp_backward_impl(packed, packed_t, nullptr, grad_raw, acts, masks, n_rays, n_samples, workspace, workspace_bytes,
                             grad_flat, stream);
}

NRF_EXPORT int nrf_mlp_backward_bf16x3(const float* packed, const void* packed_t_x3, const float* grad_raw, const float* acts,
                                       const uint32_t* masks, int64_t n_rays, int n_samples, void* workspace,
                                       int64_t workspace_bytes, float* grad_flat, nrf_stream_t stream) {
    if (!packed_t_x3) return NRF_ERR_NULL;
    if (!nrf_aligned16(packed_t_x3)) return NRF_ERR_ALIGN;
    // (the fp32 transposed stream is not read on this route: the forward pack stands in for its pointer checks)
    return mlp_backward_impl(packed, packed, packed_t_x3, grad_raw, acts, masks, n_rays, n_samples, workspace, workspace_bytes,
                             grad_flat, stream);
}

// nrf_mlp_backward with the data-gradient chain on the two-term fp16 kernel at a loss scale chosen on the device
// (mlp_fp16x2.hip: mlp_dgrad_h2_kernel) and the weight-gradient GEMMs on the fp16 pipe as well (wgrad_h2p_kernel: dY scaled
// per row set from the chain's maxima; the two head layers on head_wgrad_kernel).
NRF_EXPORT int nrf_mlp_backward_fp16x2(const float* packed, const void* packed_t_h2, const float* grad_raw, const float* acts,
                                       const uint32_t* masks, int64_t n_rays, int n_samples, void* workspace,
                                       int64_t workspace_bytes, float* grad_flat, nrf_stream_t stream) {
    if (!packed_t_h2) return NRF_ERR_NULL;
    if (!nrf_aligned16(packed_t_h2)) return NRF_ERR_ALIGN;
    return mlp_backward_impl(packed, packed, nullptr, grad_raw, acts, masks, n_rays, n_samples, workspace, workspace_bytes,
                             grad_flat, stream, packed_t_h2);
}
