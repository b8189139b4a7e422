// fp32-class 8x256 NeRF MLP on the fp16 matrix cores: every fp32 operand is split into TWO fp16 terms
// (a ~ a_hi + a_lo / 2048, split2h.h: 22 significand bits, round to nearest even twice) and every product W.x becomes
// THREE v_mfma_f32_32x32x16_f16: w_hi.x_lo and w_lo.x_hi into a second accumulator block that lives at scale 2^11,
// w_hi.x_hi into the main one; a finished block is acc + 2^-11 * acc_lo.  Products of fp16 numbers are exact in the
// MFMA's fp32 accumulate, the dropped lo.lo term and the representation error are 2^-22 of a product -- four times
// fp32's rounding unit per operand, and still below what the reference's own fp32 accumulation of a 256-term sum
// loses: the measured distance from a float64 evaluation stays at the fp32 reference's own (oracle restatement:
// oracle/nerf_oracle.py:mlp_forward(split2_f16=True); tests/test_gpu_fp16x2.py holds the fp32 kernel's bars).  Three
// MFMAs per fp32 product instead of mlp_bf16x3.hip's six (or eight fp32 MFMAs at 1/16 of the rate).  Range: operands
// must lie inside fp16's (|a| < 65520: a trained NeRF's activations are O(10)), and the kernel says so when they do
// not: a point whose activations (or a network whose weights) leave the range comes out as NaN, never as a finite wrong
// number.  The backward chain (mlp_dgrad_h2_kernel below) runs the same machinery at a power-of-two loss scale.
// Same reference path as mlp_fp32.hip (volume_renderer.py:227-231, 393-423, network.py:49-74, 171-195,
// encoding/freq.py:31-32).
//
// Shape: mlp_bf16x3.hip's chain -- one wave = 32 points x 256 features, all 8 accumulator blocks Y of a layer live, the
// next layer's input X made from Y just in time under the first two output blocks -- with X held as two fp16
// B-fragment sets (128 registers), the weight stream in 2-KiB steps (hi | lo fragment of one k16-step of one 32-row
// block) through the same LDS-DMA ring (8-step = 16-KiB chunks, 6 slots, one raw barrier per chunk), and the second
// accumulator ping-ponging between two blocks so that a block's fold-in runs in the MFMA gaps of the next block.
// Heads (alpha 256->1, rgb 128->3) are fp32 VALU dot products on the fp32 accumulators, as in the fp32 kernel.
#include "split_stream.h"
#include "split2h.h"

namespace {

constexpr int kHStepBytes = 2 * 1024;                 // hi | lo A fragments of one k16-step
constexpr int kHChunkBytes = kXChunkSteps * kHStepBytes;   // 16 KiB
#ifndef H2_RING
#define H2_RING 6
#endif
constexpr int kHRing = H2_RING;
constexpr int kHPieces = kHChunkBytes / 1024 / 4;     // LDS-DMA pieces per wave per chunk: 4
#ifndef H2_LOOK
#define H2_LOOK 2
#endif
#ifndef H2_LEAD
#define H2_LEAD 2
#endif
constexpr int kHLead = H2_LEAD;                       // output blocks a hidden-type layer opens with (h2_hidden_layer): 2 or 3,
                                                      // measured alike (37.4 | 37.7 ms per 65,536 x 192 points, one process)
constexpr int kHLook = H2_LOOK;                       // LDS -> register look-ahead of the weight fragments, in steps (< 4)
constexpr int64_t kHSmallFloatOff = int64_t(kXStreamSteps) * kHStepBytes / 4;   // fp32 small block behind the stream
constexpr int64_t kHStatusOff = kHSmallFloatOff * 4 + int64_t(kSmallFloats) * 4;   // bytes: one status word behind the small block
constexpr int kHLdsConstOff = kHRing * kHChunkBytes;                             // 96 KiB
constexpr int kHLdsBytes = kHLdsConstOff + kSmallFloats * 4;                     // + 12.1 KiB
static_assert(kHLdsBytes <= 160 * 1024, "LDS budget");
static_assert(kHPieces * (kHRing - 2) + kHLook < 64, "vmcnt is a 6-bit counter");

__device__ __forceinline__ void split_tile_h(const f32x16& v, u32x4 (&out)[2][2]) { split_block_f16(v, out); }

// ---- weight packing ------------------------------------------------------------------------
// all layers of a stream in one launch (blockIdx.y = layer; mlp_fp32.hip: pack_layers_kernel)
constexpr int kHPackBatch = 11;
struct HPackBatch { XPackLayer l[kHPackBatch]; };
__global__ void hpack_layers_kernel(HPackBatch b, unsigned* status) {   // one thread writes the two terms of one weight
    const XPackLayer& p = b.l[blockIdx.y];
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= xpack_total(p)) return;
    int step, lane, jj;
    const float v = xpack_fetch(p, idx, step, lane, jj);
    if (status && !(fabsf(v) < kF16Limit)) atomicOr(status, 1u);   // outside fp16 (or not a number): the kernel will say so
    const _Float16 hi = (_Float16)v;                                   // round to nearest even
    const _Float16 lo = (_Float16)(__fmul_rn(__fsub_rn(v, (float)hi), kLoScale));
    unsigned short* d = p.dst + int64_t(step) * (kHStepBytes / 2) + lane * 8 + jj;
    d[0] = __builtin_bit_cast(unsigned short, hi), d[512] = __builtin_bit_cast(unsigned short, lo);
}

// ---- LDS ring of the 2-term stream (mlp_bf16x3.hip's XRing with 2-KiB steps) -----------------
struct HRing {
    const char* gnext;   // per lane: this lane's 16 B of piece 4w of the next chunk to prefetch
    const char* gbegin;
    const char* gend;
    unsigned lds_base;   // LDS byte address of smem + wave * 4 KiB (this wave fills pieces 4w..4w+3)
    char* smem;
    int pf_slot, rd_slot;
    const u32x4* rd;     // this lane's 16 B of term 0 of step 0 of the chunk being read
    u32x4 wq[4][2];      // the two weight fragments of steps s .. s + kHLook - 1 (slot = step & 3), read from LDS
                         // kHLook steps ahead of their MFMAs
};
// piece j of the wave's share of the chunk being prefetched: the j KiB go into the instruction's offset field (no 64-bit
// add per piece; the LDS-DMA adds the same offset to its LDS address, so M0 = the chunk's slot for all four pieces;
// mlp_chain.h: glds16 on why this is inline asm)
template <int J>
__device__ __forceinline__ void hr_piece_imm(const HRing& st) {
#ifdef H2_ABL_NODMA
    return;
#endif
    asm volatile(
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, off offset:%2"
        :
        : "v"(st.gnext), "s"(__builtin_amdgcn_readfirstlane(st.lds_base + unsigned(st.pf_slot) * kHChunkBytes)), "n"(J * 1024)
        : "memory");
}
__device__ __forceinline__ void hr_piece(const HRing& st, int j) {
    static_assert(kHPieces == 4, "one case per piece");
    if (j == 0) hr_piece_imm<0>(st);
    else if (j == 1) hr_piece_imm<1>(st);
    else if (j == 2) hr_piece_imm<2>(st);
    else hr_piece_imm<3>(st);
}
__device__ __forceinline__ void hr_advance(HRing& st) {
    const char* nx = st.gnext + kHChunkBytes;
    st.gnext = nx == st.gend ? st.gbegin : nx;
    asm volatile("" : "+v"(st.gnext));  // keep it a run-time value (see ws_prefetch_advance)
    st.pf_slot = st.pf_slot + 1 == kHRing ? 0 : st.pf_slot + 1;
}
__device__ __forceinline__ void hr_acquire(HRing& st, int lane) {
#ifndef H2_ABL_NOBARRIER   // H2_ABL_*: timing-only ablation builds (tools/debug/h2_ab.sh): results are wrong
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)" ::"n"(kHPieces * (kHRing - 2)) : "memory");
    __builtin_amdgcn_s_barrier();
#endif
    st.rd_slot = st.rd_slot + 1 == kHRing ? 0 : st.rd_slot + 1;
    st.rd = reinterpret_cast<const u32x4*>(st.smem + st.rd_slot * kHChunkBytes) + lane;
}
__device__ __forceinline__ void hr_read1(HRing& st, int slot, int step_in_chunk, int t) {
#ifdef H2_ABL_NOLDS
    return;
#endif
    st.wq[slot][t] = st.rd[(step_in_chunk * 2 + t) * 64];
}
__device__ __forceinline__ void hr_start(HRing& st, const char* stream_base, char* smem, int wave, int lane,
                                         int n_chunks = kXChunks) {
    st.gbegin = stream_base + wave * (kHPieces * 1024) + lane * 16;
    st.gend = st.gbegin + size_t(n_chunks) * kHChunkBytes;
    st.gnext = st.gbegin;
    st.smem = smem;
    st.lds_base = __builtin_amdgcn_readfirstlane(unsigned(reinterpret_cast<uintptr_t>((lds_vptr)smem)) + unsigned(wave) * unsigned(kHPieces * 1024));
    st.pf_slot = 0;
    // the state of a stream in flight at step 0: chunks 0 .. kHRing-2 on their way and the last slot's refill at piece
    // kHLook (h2_step issues piece (s + kHLook) % 8 of the current refill on step s)
#pragma unroll
    for (int i = 0; i < kHRing; ++i) {
#pragma unroll
        for (int jj = 0; jj < (i + 1 < kHRing ? kHPieces : kHLook); ++jj) hr_piece(st, jj);
        if (i + 1 < kHRing) hr_advance(st);
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kHPieces * (kHRing - 2) + kHLook) : "memory");
    __builtin_amdgcn_s_barrier();
    st.rd_slot = 0;
    st.rd = reinterpret_cast<const u32x4*>(smem) + lane;
#pragma unroll
    for (int i = 0; i < kHLook; ++i) {
        hr_read1(st, i, i, 0);
        hr_read1(st, i, i, 1);
    }
}

// Input op of a hidden-type layer: X[kb] (two-term B fragments) from Y[kb] of the previous layer, one register pair
// at a time in three sub-slices (activation + hi, the two residuals, lo: 3 + 2 + 2 VALU instructions).
struct OpInH : OpInBase {
    float sa, sb, ra, rb;
    unsigned p0;
    float ovf;   // range guard: the largest |operand| this point has fed to a split so far
    template <int MODE>
    __device__ __forceinline__ void sub(const f32x16& y, u32x4 (&x)[2][2], int blk, int q, int phase) {
#ifdef H2_ABL_NOSPLIT
        return;
#endif
        if (phase == 0) {
            input<MODE, MODE == kOpBwd>(y, blk, q, sa, sb);
            p0 = pk_f16(sa, sb);
            x[q / 4][0][q % 4] = p0;
        } else if (phase == 1) {
#ifndef H2_ABL_NOLO
            ra = f16_residual_lo(sa, p0), rb = f16_residual_hi(sb, p0);
#endif
#ifndef H2_ABL_NOTRACK
            track_max(ovf, sa, sb);
#endif
        } else {
#ifdef H2_ABL_NOLO
            return;
#endif
            x[q / 4][1][q % 4] = pk_f16_scaled(ra, rb);
        }
    }
    template <int MODE>
    __device__ __forceinline__ void pair(const f32x16& y, u32x4 (&x)[2][2], int blk, int q) {
        float a, b;
        input<MODE, MODE == kOpBwd>(y, blk, q, a, b);
        unsigned hi, lo;
        split_pair_f16(a, b, hi, lo);
        track_max(ovf, a, b);
        x[q / 4][0][q % 4] = hi, x[q / 4][1][q % 4] = lo;
    }
};

// y[t0 .. t1) = base[t0 .. t1) + 2^-11 * yl[t0 .. t1): a slice of a finished block's fold-in
__device__ __forceinline__ void fold_slice(f32x16& y, const f32x16& base, const f32x16& yl, int t0, int t1) {
#ifdef H2_ABL_NOFOLD
    return;
#endif
    f32x16 v = y;   // whole-vector read / write: element stores into the block array keep it out of registers
#ifdef H2_FOLD_SCALAR
#pragma unroll
    for (int t = 0; t < 16; ++t)
        if (t >= t0 && t < t1) v[t] = fmaf(yl[t], kLoUnscale, base[t]);
#else
    // two elements per instruction (v_pk_fma_f32); every slice is a whole number of even-aligned pairs
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 c = {kLoUnscale, kLoUnscale};
#pragma unroll
    for (int t = 0; t < 16; t += 2)
        if (t >= t0 && t < t1) {
            const f32x2 l = {yl[t], yl[t + 1]}, b = {base[t], base[t + 1]};
            const f32x2 r = __builtin_elementwise_fma(l, c, b);
            v[t] = r[0], v[t + 1] = r[1];
        }
#endif
    y = v;
}

// One k16-step of the stream: the three MFMAs of one (output block, k16) pair -- the two cross terms into `yl` (which
// the block's FIRST step starts from zero), hi.hi into `y`; `hook(m)` runs after MFMA m.  The ring hand-over (acquire;
// the wave's 4 refill pieces on this and the next three steps) rides on the step kHLook before a chunk's end.  Every
// MFMA gap is its own scheduling region (mlp_bf16x3.hip: x3_step).
template <bool FIRST, class Hook>
__device__ __forceinline__ void h2_step(HRing& st, int lane, int s /* stream step inside the layer */, f32x16& y, f32x16& yl,
                                        const u32x4* x, Hook&& hook) {
    const int piece = (s + kHLook) % kXChunkSteps;
    const bool hand = piece == 0;
    const u32x4 wh = st.wq[s & 3][0], wl = st.wq[s & 3][1];
    if (hand) hr_acquire(st, lane);
    const int rs = (s + kHLook) & 3, rc = (s + kHLook) % kXChunkSteps;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_sched_barrier(0);
    yl = FIRST ? NRF_MFMA_F16(wh, x[1], zero16) : NRF_MFMA_F16(wh, x[1], yl);
    hr_read1(st, rs, rc, 0);
    hook(0);
    __builtin_amdgcn_sched_barrier(0);
    yl = NRF_MFMA_F16(wl, x[0], yl);
    hr_read1(st, rs, rc, 1);
    hook(1);
    __builtin_amdgcn_sched_barrier(0);
    y = NRF_MFMA_F16(wh, x[0], y);
    if (piece < kHPieces) {
        hr_piece(st, piece);
        if (piece == kHPieces - 1) hr_advance(st);
    }
    hook(2);
    __builtin_amdgcn_sched_barrier(0);
}

// Encoding-only layer (pts_linears.0, and the skip layer's PE pass which ACCUMULATES onto pts_linears.5's hidden part):
// Y[ob] = (accum ? Y[ob] : bias) + W . pe.  `accum` is a run-time flag so both uses share one body.  Block ob's second
// accumulator is YL[ob & 1]; it is folded in under block ob + 1 (from the third MFMA on: the result is out by then).
__device__ __forceinline__ void h2_pe_layer(HRing& st, int lane, const float* bias_h, bool accum, const u32x4 (*pe)[2][2],
                                            f32x16* Y, f32x16* YL) {
#pragma unroll
    for (int ob = 0; ob < 8; ++ob) {
        const f32x16 bias = *reinterpret_cast<const f32x16*>(bias_h + ob * 32);
#pragma unroll
        for (int t = 0; t < 16; ++t) Y[ob][t] = accum ? Y[ob][t] : bias[t];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                auto hook = [&](int m) {
                    const int n = (kb * 2 + s2) * 3 + m;   // 12 gaps
                    if (ob > 0 && n >= 2 && n < 10) fold_slice(Y[ob - 1], Y[ob - 1], YL[(ob - 1) & 1], 2 * (n - 2), 2 * (n - 1));
                };
                if (kb == 0 && s2 == 0) h2_step<true>(st, lane, (ob * 2 + kb) * 2 + s2, Y[ob], YL[ob & 1], pe[kb][s2], hook);
                else h2_step<false>(st, lane, (ob * 2 + kb) * 2 + s2, Y[ob], YL[ob & 1], pe[kb][s2], hook);
            }
    }
    fold_slice(Y[7], Y[7], YL[1], 0, 16);
}

// Hidden-type layer: Y[ob] = bias + W . [X (NHID k-blocks), post], X[kb] = op(Y[kb] of the previous layer).
// Making one X[kb] costs about as much VALU time as twelve MFMAs give, so the layer opens with its first LEAD output
// blocks accumulated together (stream order [kb][ob < LEAD][s2]): X[kb+1] is made in 24 sub-slices spread over the
// 6 LEAD MFMA gaps of k-block kb.  Blocks 1 .. LEAD-1 accumulate into the spare blocks Z because the previous layer's
// Y[1 ..] are still needed (Y[b] becomes X[b] during k-block b - 1).  Second accumulators: YL[b] for lead block b --
// block 0's is folded in under the last lead steps, the others' under the first block after the lead, together with the
// moves Z -> Y -- then two of them in turn, each folded in under the next block; the last block's fold-in is the only
// one that waits for its MFMAs.
template <int NOB, int NPOST, int MODE, int NHID = 8, int LEAD = kHLead>
__device__ __forceinline__ void h2_hidden_layer(HRing& st, int lane, const float* bias_h, u32x4 (*X)[2][2],
                                                const u32x4 (*post)[2][2], f32x16* Y, f32x16* Z, f32x16* YL, OpInH& op) {
    constexpr int NKB = NHID + NPOST;
    static_assert(LEAD >= 2 && LEAD <= 3 && NOB > LEAD, "the fold-in schedule assumes a block after the lead");
    static_assert(8 * LEAD - 6 <= 2 * NKB * 3, "block LEAD must have gaps for the lead blocks' fold-ins");
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto bias = [&](int ob) { return MODE == kOpBwd ? zero16 : *reinterpret_cast<const f32x16*>(bias_h + ob * 32); };
    static_assert((NOB * NKB * 2) % kXChunkSteps == 0, "layer must be a whole number of chunks");
#pragma unroll
    for (int q = 0; q < 8; ++q) op.template pair<MODE>(Y[0], X[0], 0, q);
    Y[0] = bias(0);
#pragma unroll
    for (int b = 1; b < LEAD; ++b) Z[b - 1] = bias(b);
    constexpr int G = 6 * LEAD;   // MFMA gaps per lead k-block
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        const bool make = kb + 1 < NHID;
#pragma unroll
        for (int b = 0; b < LEAD; ++b)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const u32x4* x = kb < NHID ? X[kb][s2] : post[kb - NHID][s2];
                const int g = b * 2 + s2;                       // which of the k-block's 2 LEAD steps
                auto hook = [&](int m) {                        // sub-slices [24 n / G, 24 (n + 1) / G) after MFMA n
                    const int n = g * 3 + m;
                    if (make) {
#pragma unroll
                        for (int w = (24 * n) / G; w < (24 * (n + 1)) / G; ++w)
                            op.template sub<MODE>(Y[kb + 1], X[kb + 1], kb + 1, w / 3, w % 3);
                    }
                    if (kb == NKB - 1) {
                        // block 0 finished with step g = 1: its fold-in goes under the other lead blocks' last steps
                        if (n >= 8 && n < 12) fold_slice(Y[0], Y[0], YL[0], 4 * (n - 8), 4 * (n - 7));
                        // the next block's bias is read one step ahead of its first MFMA (Y[LEAD] is free: X[LEAD] is made)
                        if (g == 2 * LEAD - 1 && m == 0) Y[LEAD] = bias(LEAD);
                    }
                };
                if (kb == 0 && s2 == 0) h2_step<true>(st, lane, kb * 2 * LEAD + g, b == 0 ? Y[0] : Z[b - 1], YL[b], x, hook);
                else h2_step<false>(st, lane, kb * 2 * LEAD + g, b == 0 ? Y[0] : Z[b - 1], YL[b], x, hook);
            }
    }
    // second accumulator of a block after the lead: block LEAD takes YL[0] (folded above), then they alternate
    auto yl_of = [](int ob) { return ob < LEAD ? ob : (ob - LEAD) & 1; };
#pragma unroll
    for (int ob = LEAD; ob < NOB; ++ob) {
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                auto hook = [&](int m) {
                    const int n = (kb * 2 + s2) * 3 + m;
                    if (ob == LEAD) {         // the lead blocks 1 .. LEAD-1: out of Z, eight gaps each
#pragma unroll
                        for (int b = 1; b < LEAD; ++b)
                            if (n >= 8 * b - 6 && n < 8 * b + 2) fold_slice(Y[b], Z[b - 1], YL[b], 2 * (n - 8 * b + 6), 2 * (n - 8 * b + 7));
                    } else if (n >= 2 && n < 10) {   // the previous block's fold-in
                        fold_slice(Y[ob - 1], Y[ob - 1], YL[yl_of(ob - 1)], 2 * (n - 2), 2 * (n - 1));
                    }
                    if (kb == NKB - 1 && s2 == 1 && m == 0 && ob + 1 < NOB) Y[ob + 1] = bias(ob + 1);
                };
                const u32x4* x = kb < NHID ? X[kb][s2] : post[kb - NHID][s2];
                if (kb == 0 && s2 == 0) h2_step<true>(st, lane, (ob * NKB + kb) * 2 + s2, Y[ob], YL[yl_of(ob)], x, hook);
                else h2_step<false>(st, lane, (ob * NKB + kb) * 2 + s2, Y[ob], YL[yl_of(ob)], x, hook);
            }
    }
    fold_slice(Y[NOB - 1], Y[NOB - 1], YL[yl_of(NOB - 1)], 0, 16);
}

// Persistent, one 4-wave workgroup per CU, one wave per SIMD, a wave's tile = 32 points (mlp_fp32.hip's tiling).
// SAVE (training forward): also writes every layer's activated input and relu sign bits in mlp_fwd_kernel<true>'s
// layout, for the backward kernels of mlp_bwd.hip.
template <bool SAVE>
__global__ __launch_bounds__(256, 1) void mlp_fp16x2_kernel(const char* __restrict__ packed, const float* __restrict__ a0,
                                                            const float* __restrict__ a1, const float* __restrict__ zbuf,
                                                            int64_t z_stride, int tiles_per_ray, int rays_mode,
                                                            int64_t n_points, int64_t n_groups, float4* __restrict__ raw,
                                                            float* __restrict__ acts, unsigned* __restrict__ masks) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    {
        const float4* src = reinterpret_cast<const float4*>(packed + kHSmallFloatOff * 4);
        float4* dst = reinterpret_cast<float4*>(smem + kHLdsConstOff);
        for (int i = threadIdx.x; i < kSmallFloats / 4; i += 256) dst[i] = src[i];
    }
    // a weight outside fp16's range was seen by the packer: every output is NaN (nrf_packed_fp16x2_status says why)
    const float poison = *reinterpret_cast<const unsigned*>(packed + kHStatusOff) ? __builtin_inff() : 0.f;
    __syncthreads();
    const float* cst = reinterpret_cast<const float*>(smem + kHLdsConstOff);
    const float* bias_h = cst + h * 16;

    HRing st;
    hr_start(st, packed, smem, wave, lane);

    // A tile's per-lane inputs (rays: its sample depth; flat points: position and direction) are loaded ONE TILE AHEAD by
    // loads the compiler does not see (tile_load).  A load it does see is waited for with a vmcnt that knows nothing of
    // the ring's LDS-DMA pieces in flight, i.e. by draining the ring at the top of every tile.  These are 580 VMEM
    // instructions old when they are used, and the counted wait at the end of a tile (<= 40 outstanding; the ring keeps
    // at most 18) proves them complete.
    struct TileIn { float v[7]; };   // rays: origin, direction, depth; flat points: position, direction
    auto tile_load = [&](int64_t tg_) {
        TileIn in;
        in.v[6] = 0.f;
        const int64_t tile_ = tg_ * 4 + wave;
        int64_t p_ = tile_ * 32 + j;
        const bool ok = p_ < n_points;
        if (!ok) p_ = n_points - 1;
        auto ld = [](const float* q) {
            float r;
            asm volatile("global_load_dword %0, %1, off" : "=a"(r) : "v"(q) : "memory");   // into an accumulation register: nothing touches it before the next tile reads it
            return r;
        };
        if (rays_mode) {
            const int64_t tile_c = p_ / 32;
            const int64_t ray = tile_c / tiles_per_ray;
            const int s0 = int(tile_c - ray * tiles_per_ray) * 32;
#pragma unroll
            for (int k = 0; k < 3; ++k) in.v[k] = ld(a0 + ray * 3 + k), in.v[3 + k] = ld(a1 + ray * 3 + k);
            in.v[6] = ld(zbuf + ray * z_stride + s0 + (ok ? j : int(p_ & 31)));
        } else {
#pragma unroll
            for (int k = 0; k < 3; ++k) in.v[k] = ld(a0 + p_ * 3 + k), in.v[3 + k] = ld(a1 + p_ * 3 + k);
        }
        return in;
    };
    // the waits carry the loaded registers as operands: what reads them afterwards cannot be scheduled ahead of the wait
#define NRF_TILE_WAIT(N, in)                                                                                             \
    asm volatile("s_waitcnt vmcnt(" #N ")"                                                                               \
                 : "+a"(in.v[0]), "+a"(in.v[1]), "+a"(in.v[2]), "+a"(in.v[3]), "+a"(in.v[4]), "+a"(in.v[5]), "+a"(in.v[6]) \
                 :                                                                                                       \
                 : "memory")
    TileIn nxt = tile_load(blockIdx.x < n_groups ? int64_t(blockIdx.x) : 0);
    NRF_TILE_WAIT(0, nxt);

    for (int64_t tg = blockIdx.x; tg < n_groups; tg += gridDim.x) {
        const int64_t tile = tg * 4 + wave;
        int64_t pidx = tile * 32 + j;
        const bool valid = pidx < n_points;
        if (!valid) pidx = n_points - 1;
        const TileIn cur = nxt;
        if (tg + gridDim.x < n_groups) nxt = tile_load(tg + gridDim.x);
        float px, py, pz, dx, dy, dz;
        if (rays_mode) {
            const float ox = cur.v[0], oy = cur.v[1], oz = cur.v[2];
            dx = cur.v[3], dy = cur.v[4], dz = cur.v[5];
            const float nrm = sqrtf(fmaf(dz, dz, fmaf(dy, dy, dx * dx)));  // volume_renderer.py:230 (torch's 3-element norm)
            dx = dx / nrm, dy = dy / nrm, dz = dz / nrm;
            const float zz = cur.v[6];
            px = __fadd_rn(ox, __fmul_rn(zz, dx));  // :231
            py = __fadd_rn(oy, __fmul_rn(zz, dy));
            pz = __fadd_rn(oz, __fmul_rn(zz, dz));
        } else {
            px = cur.v[0], py = cur.v[1], pz = cur.v[2];
            dx = cur.v[3], dy = cur.v[4], dz = cur.v[5];
        }
        u32x4 pe[2][2][2], dpe[1][2][2];
        float* arow = nullptr;     // this lane's column of the tile's saved rows (mlp_fp32.hip)
        unsigned* mrow = nullptr;
        if (SAVE) {
            arow = acts + tile * (int64_t(kActRows) * 32) + 4 * h * 32 + j;
            mrow = masks + tile * (kMaskLayers * 4 * 64) + lane;
        }
        {
            const f32x16 e0 = encode_block<10, 0>(px, py, pz, h), e1 = encode_block<10, 1>(px, py, pz, h);
            const f32x16 ed = encode_block<4, 0>(dx, dy, dz, h);
            if (SAVE) {
                store_block(arow + kActPe * 32, 0, e0);
                store_block(arow + kActPe * 32, 1, e1);
                store_block(arow + kActDpe * 32, 0, ed);
            }
            split_tile_h(e0, pe[0]);
            split_tile_h(e1, pe[1]);
            split_tile_h(ed, dpe[0]);
        }

        u32x4 X[8][2][2];
        f32x16 Y[8], Z[kHLead - 1], YL[kHLead];
        OpInH op{op_in(kActRelu, cst + (kOffAlphaW - kOffBias) + h * 16), 0.f, 0.f, 0.f, 0.f, 0u, poison};
        track_max(op.ovf, px, py);   // the raw coordinates are operands too (the encoding's sin / cos cannot overflow)
        track_max(op.ovf, pz, pz);
        // pts_linears.0 .. 7 and feature_linear (li = 8) as ONE hidden-layer body in a run-time loop; the encoding-only
        // layer runs before li = 1 (pts_linears.0) and before li = 6 (the PE columns of the skip layer pts_linears.5,
        // network.py:58 cat([pe, h]), accumulated onto its hidden part)
        for (int li = 1; li <= 8; ++li) {
            if (li == 1 || li == 6) h2_pe_layer(st, lane, bias_h, li == 6, pe, Y, YL);
            if (li == 8) op.alpha_head(Y);                            // feature_linear's input relu(h7) feeds alpha_linear
            if (SAVE) {   // the layer's input h_{li-1} = relu(Y) and its signs (mask li - 1)
                op.save = arow + (kActH + (li - 1) * 256) * 32;
                op.msave = mrow + (li - 1) * 4 * 64;
            }
            h2_hidden_layer<8, 0, SAVE ? kOpFwdSave : kOpFwd>(st, lane, bias_h + li * 256, X, nullptr, Y, Z, YL, op);
        }
        float sigma = op.sigma;
        asm volatile("" : "+v"(sigma));   // keep the head where it is (see mlp_bf16.hip)
        // feature -> views: no activation (saved as it is, no mask)
        OpInH op_views{op_in(kActNone, nullptr, SAVE ? arow + kActFeat * 32 : nullptr), 0.f, 0.f, 0.f, 0.f, 0u, op.ovf};
        h2_hidden_layer<4, 1, SAVE ? kOpFwdSave : kOpFwd, 8, kHLead>(st, lane, bias_h + 9 * 256, X, dpe, Y, Z, YL, op_views);   // views_linears.0
        float r = 0.f, g = 0.f, b = 0.f;
        {
            const float* rw = cst + (kOffRgbW - kOffBias) + h * 16;
            OpFwdSave<true> ops{SAVE ? arow + kActHv * 32 : nullptr, SAVE ? mrow + 8 * 4 * 64 : nullptr, 0u};
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                const f32x16 hv = SAVE ? ops(kb, Y[kb]) : relu16(Y[kb]);
                const f32x16 w0 = *reinterpret_cast<const f32x16*>(rw + kb * 32);
                const f32x16 w1 = *reinterpret_cast<const f32x16*>(rw + 128 + kb * 32);
                const f32x16 w2 = *reinterpret_cast<const f32x16*>(rw + 256 + kb * 32);
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    r = fmaf(w0[t], hv[t], r);
                    g = fmaf(w1[t], hv[t], g);
                    b = fmaf(w2[t], hv[t], b);
                }
            }
        }
        r += __shfl_xor(r, 32);
        g += __shfl_xor(g, 32);
        b += __shfl_xor(b, 32);
        sigma += __shfl_xor(sigma, 32);
        float ovf = op_views.ovf;
        ovf = fmaxf(ovf, __shfl_xor(ovf, 32));
        if (h == 0 && valid) {
            const float* rb = cst + (kOffRgbB - kOffBias);
            float4 o = make_float4(r + rb[0], g + rb[1], b + rb[2], sigma + cst[kOffAlphaB - kOffBias]);
            // an operand of this point left fp16's range somewhere on the way: refuse it
            if (!(ovf < kF16Limit)) o = make_float4(__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""));
            raw[pidx] = o;
        }
        NRF_TILE_WAIT(40, nxt);   // the next tile's inputs (issued ~580 VMEM instructions ago) are in
    }
#undef NRF_TILE_WAIT
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA may outlive the workgroup
}

// ---- backward data-gradient chain on the same machinery (mlp_bf16x3.hip: mlp_dgrad_x3_kernel is the bf16x3 form) -------
// d(input of layer l) = W_l^T . dY_l with the transposed weights as the two-term stream (views^T, feature^T, pts7^T ..
// pts1^T; nrf_pack_weights_transposed_fp16x2).  Gradients are O(1e-6): far below fp16's normal range, where the split
// would lose its bits.  So the chain runs at a LOSS SCALE 2^S, S chosen on the device from max |d raw| of the batch
// (grad_max_kernel; 2^S max = 16: twelve binades of head room above, fourteen of normal range below) -- d raw is scaled
// on load, every dY row is unscaled on its way to HBM (both exact: powers of two), so the wgrad GEMMs and everything
// else see ordinary fp32 gradients.  A point whose scaled gradients still leave fp16's range gets NaN rows.
__global__ void grad_max_kernel(const float4* __restrict__ g, int64_t n, unsigned* __restrict__ out) {
    float m = 0.f;
    for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) {
        const float4 v = g[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));   // non-negative floats order like their bits
}

__global__ __launch_bounds__(256, 1) void mlp_dgrad_h2_kernel(const float* __restrict__ packed,   // fp32 pack: head rows
                                                              const char* __restrict__ packed_t,  // two-term W^T stream
                                                              const float4* __restrict__ graw, const unsigned* __restrict__ masks,
                                                              int64_t n_points, int64_t n_groups, float* __restrict__ dys,
                                                              unsigned* gmax_bits) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    {
        const float4* src = reinterpret_cast<const float4*>(packed + kOffBias);
        float4* dst = reinterpret_cast<float4*>(smem + kHLdsConstOff);
        for (int i = threadIdx.x; i < kSmallFloats / 4; i += 256) dst[i] = src[i];
    }
    // loss scale 2^S with 2^S * max |d raw| in [16, 32); max = 0, subnormal or not finite: S = 0
    const int S = scale_exponent(*gmax_bits, kChainScaleTarget);
    const float scale = exp2i(S), unscale = exp2i(-S);
    __syncthreads();
    const float* cst = reinterpret_cast<const float*>(smem + kHLdsConstOff);
    const float* aw = cst + (kOffAlphaW - kOffBias) + h * 16;
    const float* rw = cst + (kOffRgbW - kOffBias) + h * 16;

    HRing st;
    hr_start(st, packed_t, smem, wave, lane, kXBwdChunks);

    // d raw and the relu sign words are loaded one stage ahead by loads the compiler does not see (into accumulation
    // registers; mlp_fp16x2_kernel: tile_load): loaded in place they cost a ring drain at the top of the tile and one per
    // layer.  Stages of a tile: views (d raw + words 32, 33), layers 7 .. 1 (4 words each), layer 0 (4 words); every
    // stage issues well over 40 vector-memory instructions, which is what the counted waits rely on.
    float pg[4];
    unsigned pm[4];
    const int64_t last_tile = (n_points - 1) / 32;
    const unsigned lane4 = unsigned(lane) * 4u;
    auto clamp_tile = [&](int64_t tile_) { return tile_ > last_tile ? last_tile : tile_; };   // a wave past the last tile
    auto mask_base = [&](int64_t tile_) {   // reads the last tile's inputs; its rows are zero
        return masks + clamp_tile(tile_) * (kMaskLayers * 4 * 64);
    };
    // addresses as uniform base + 32-bit lane offset + immediate: one address register for the loads of a stage; the
    // destination is an in-out operand, so a load lands in the register its wait and its reader name (as a plain
    // output the compiler gave it a new register and copied that, unwaited, into the loop-carried one)
#define NRF_BWD_LOAD(dst, voff, base, imm) \
    asm volatile("global_load_dword %0, %1, %2 offset:" #imm : "+a"(dst) : "v"(voff), "s"(base) : "memory")
    auto issue_words4 = [&](const unsigned* p) {
        NRF_BWD_LOAD(pm[0], lane4, p, 0);
        NRF_BWD_LOAD(pm[1], lane4, p, 256);
        NRF_BWD_LOAD(pm[2], lane4, p, 512);
        NRF_BWD_LOAD(pm[3], lane4, p, 768);
    };
    auto issue_tile = [&](int64_t tg_) {
        const int64_t tile_ = clamp_tile(tg_ * 4 + wave);
        const int64_t rows = n_points - tile_ * 32;   // >= 1
        const unsigned goff = unsigned(int64_t(j) < rows ? j : int(rows) - 1) * 16u;
        const float4* gp = graw + tile_ * 32;
        NRF_BWD_LOAD(pg[0], goff, gp, 0);
        NRF_BWD_LOAD(pg[1], goff, gp, 4);
        NRF_BWD_LOAD(pg[2], goff, gp, 8);
        NRF_BWD_LOAD(pg[3], goff, gp, 12);
        const unsigned* p = mask_base(tile_) + 8 * 4 * 64;
        NRF_BWD_LOAD(pm[0], lane4, p, 0);
        NRF_BWD_LOAD(pm[1], lane4, p, 256);
    };
    // the waits carry the loaded registers as operands: their readers stay behind them
#define NRF_BWD_WAIT(N)                                                                                              \
    asm volatile("s_waitcnt vmcnt(" #N ")"                                                                           \
                 : "+a"(pg[0]), "+a"(pg[1]), "+a"(pg[2]), "+a"(pg[3]), "+a"(pm[0]), "+a"(pm[1]), "+a"(pm[2]), "+a"(pm[3]) \
                 :                                                                                                   \
                 : "memory")
    // a stage's values are taken out by a statement that stays ahead of the next stage's loads (volatile asm keeps its
    // order): the old value is dead when the load's in-out operand is set up, so no copy of it -- and no copy back of
    // the unwaited new one -- is needed.  tests/test_asm_lint.py checks the built code for exactly that.
#define NRF_BWD_TAKE(T, src)                                                          \
    [&] {                                                                             \
        T r_;                                                                         \
        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(r_) : "a"(src));              \
        return r_;                                                                    \
    }()
    pg[0] = pg[1] = pg[2] = pg[3] = 0.f, pm[0] = pm[1] = pm[2] = pm[3] = 0u;
    issue_tile(blockIdx.x < n_groups ? int64_t(blockIdx.x) : 0);
    NRF_BWD_WAIT(0);

    for (int64_t tg = blockIdx.x; tg < n_groups; tg += gridDim.x) {
        const int64_t tile = tg * 4 + wave;
        const bool valid = tile * 32 + j < n_points;
        float4 g = make_float4(NRF_BWD_TAKE(float, pg[0]), NRF_BWD_TAKE(float, pg[1]), NRF_BWD_TAKE(float, pg[2]),
                               NRF_BWD_TAKE(float, pg[3]));
        if (!valid) g = make_float4(0.f, 0.f, 0.f, 0.f);
        const unsigned mv0 = NRF_BWD_TAKE(unsigned, pm[0]), mv1 = NRF_BWD_TAKE(unsigned, pm[1]);
        const unsigned* mrow = mask_base(tile);
        issue_words4(mrow + 7 * 4 * 64);   // layer 7's words, for li = 1
        float* drow = dys + tile * (int64_t(kDyRows) * 32) + 4 * h * 32 + j;
        if (h == 0) {  // d(raw) as 4 rows for the head wgrad jobs (unscaled)
            float* r = dys + tile * (int64_t(kDyRows) * 32) + kDyRaw * 32 + j;
            r[0] = g.x, r[32] = g.y, r[64] = g.z, r[96] = g.w;
        }
        g.x *= scale, g.y *= scale, g.z *= scale, g.w *= scale;
        u32x4 X[8][2][2];
        f32x16 Y[8], Z[kHLead - 1], YL[kHLead];
        // rgb_linear^T on the VALU (fp32): dYv = relu'(hv) * (W_rgb^T . d rgb), stored, and the input of views^T
        {
            const unsigned m0 = mv0, m1 = mv1;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                const f32x16 w0 = *reinterpret_cast<const f32x16*>(rw + kb * 32);
                const f32x16 w1 = *reinterpret_cast<const f32x16*>(rw + 128 + kb * 32);
                const f32x16 w2 = *reinterpret_cast<const f32x16*>(rw + 256 + kb * 32);
                f32x16 v;
#pragma unroll
                for (int t = 0; t < 16; ++t) v[t] = fmaf(w2[t], g.z, fmaf(w1[t], g.y, w0[t] * g.x));
                Y[kb] = mask16(v, ((kb < 2 ? m0 : m1) >> (16 * (kb & 1))) & 0xffffu);
            }
        }
        // d feature = W_views[:, :256]^T . dYv: four k-blocks in (masked above: all-ones mask here; the split stores them
        // as the dY rows of views_linears.0), eight blocks out
        OpInH op{op_in(kActNone, aw, drow + kDyV * 32), 0.f, 0.f, 0.f, 0.f, 0u, 0.f};
        op.kscale = unscale;
#pragma unroll
        for (int w = 0; w < 4; ++w) op.m[w] = ~0u;
        // op.ovf is the largest |scaled dY| of the rows the layer just stored: the range guard's input, and -- reduced
        // over the batch into word 1 + (row set) -- what the fp16x2 weight-gradient GEMMs scale that row set by
        float ovf_all = 0.f;
        auto layer_max = [&](int word) {
            float m = op.ovf;
            ovf_all = fmaxf(ovf_all, m);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
            if (lane == 0 && m < kF16Limit) atomicMax(gmax_bits + word, __float_as_uint(m));
            op.ovf = 0.f;
        };
        h2_hidden_layer<8, 0, kOpBwd, 4>(st, lane, nullptr, X, nullptr, Y, Z, YL, op);
        layer_max(kGmaxViews);
        // li = 0: d h7 = W_feature^T . d feature (no activation between feature_linear and views_linears: all-ones mask,
        // stored as the dY rows of feature_linear); li = 1 .. 7: layer l = 8 - li, dY_l = relu'(h_l) * d h_l (+ the
        // alpha_linear term for h7), d h_{l-1} = W_l^T . dY_l.  One body, run-time loop.
        for (int li = 0; li <= 7; ++li) {
            const int l = 8 - li;
            op.save = drow + (li == 0 ? kDyFeat : l * 256) * 32;
#pragma unroll
            for (int w = 0; w < 4; ++w) op.m[w] = 0xffffffffu;
            if (li > 0) {
                NRF_BWD_WAIT(40);
#pragma unroll
                for (int w = 0; w < 4; ++w) op.m[w] = NRF_BWD_TAKE(unsigned, pm[w]);
                issue_words4(mrow + (l - 1) * 4 * 64);   // the next stage's words
            }
            op.dsig = li == 1 ? g.w : 0.f;                       // h7 also feeds alpha_linear
            h2_hidden_layer<8, 0, kOpBwd>(st, lane, nullptr, X, nullptr, Y, Z, YL, op);
            layer_max(li == 0 ? kGmaxFeat : kGmaxLayer0 + l);
        }
        {   // dY_0 = relu'(h_0) * d h_0 (the encoding needs no gradient: the chain ends here)
            float ovf = ovf_all;
            ovf = fmaxf(ovf, __shfl_xor(ovf, 32));
            const float out_scale = ovf < kF16Limit ? unscale : __builtin_nanf("");   // a point that left the range says so
            OpBwdMask last;
            last.dst = drow;
            NRF_BWD_WAIT(40);
#pragma unroll
            for (int w = 0; w < 4; ++w) last.m[w] = NRF_BWD_TAKE(unsigned, pm[w]);
            if (tg + gridDim.x < n_groups) issue_tile(tg + gridDim.x);
            last.aw = aw;
            last.dsig = 0.f;
            float m0 = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const f32x16 x = mask16(Y[i], (last.m[i >> 1] >> (16 * (i & 1))) & 0xffffu);
#pragma unroll
                for (int t = 0; t < 16; t += 2) track_max(m0, x[t], x[t + 1]);
                f32x16 y = Y[i];
#pragma unroll
                for (int t = 0; t < 16; ++t) y[t] *= out_scale;
                (void)last(i, y);
            }
            op.ovf = m0;
            layer_max(kGmaxLayer0);
        }
        NRF_BWD_WAIT(40);   // the next tile's inputs, 128 stores old, are in
    }
#undef NRF_BWD_WAIT
#undef NRF_BWD_LOAD
#undef NRF_BWD_TAKE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA may outlive the workgroup
}

int launch_h2(const void* packed, const float* a0, const float* a1, const float* z, int64_t z_stride, int tiles_per_ray,
              int rays_mode, int64_t n_points, float* raw, hipStream_t stream, float* acts = nullptr, unsigned* masks = nullptr) {
    const int64_t tiles = (n_points + 31) / 32;
    const int64_t groups = (tiles + 3) / 4;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (cus <= 0) cus = 256;
    static bool attr_set[64] = {};
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_fp16x2_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                kHLdsBytes) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_fp16x2_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                kHLdsBytes) != hipSuccess)
            return NRF_ERR_LAUNCH;
        attr_set[dev] = true;
    }
    const unsigned grid = unsigned(groups < cus ? groups : cus);
    nrf_launch_begin();
    if (acts)
        mlp_fp16x2_kernel<true><<<dim3(grid), 256, kHLdsBytes, stream>>>(static_cast<const char*>(packed), a0, a1, z, z_stride,
                                                                         tiles_per_ray, rays_mode, n_points, groups,
                                                                         reinterpret_cast<float4*>(raw), acts, masks);
    else
        mlp_fp16x2_kernel<false><<<dim3(grid), 256, kHLdsBytes, stream>>>(static_cast<const char*>(packed), a0, a1, z, z_stride,
                                                                          tiles_per_ray, rays_mode, n_points, groups,
                                                                          reinterpret_cast<float4*>(raw), nullptr, nullptr);
    return nrf_launch_status();
}

}  // namespace

// nrf_mlp_backward_fp16x2 (mlp_bwd.hip) calls this in place of mlp_dgrad_kernel; gmax: one zeroed-here word of workspace
int launch_dgrad_h2(const float* packed, const void* packed_t, const float* grad_raw, const unsigned* masks, int64_t n_points,
                    float* dys, unsigned* gmax, hipStream_t stream) {
    const int64_t groups = (n_points / 32 + 3) / 4;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (cus <= 0) cus = 256;
    static bool attr_set[64] = {};
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_dgrad_h2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                kHLdsBytes) != hipSuccess)
            return NRF_ERR_LAUNCH;
        attr_set[dev] = true;
    }
    if (hipMemsetAsync(gmax, 0, 256, stream) != hipSuccess) return NRF_ERR_RUNTIME;
    grad_max_kernel<<<dim3(unsigned(cus)), 256, 0, stream>>>(reinterpret_cast<const float4*>(grad_raw), n_points, gmax);
    mlp_dgrad_h2_kernel<<<dim3(unsigned(groups < cus ? groups : cus)), 256, kHLdsBytes, stream>>>(
        packed, static_cast<const char*>(packed_t), reinterpret_cast<const float4*>(grad_raw), masks, n_points, groups, dys, gmax);
    return NRF_OK;
}

NRF_EXPORT int64_t nrf_packed_bwd_fp16x2_bytes(void) { return int64_t(kXBwdSteps) * kHStepBytes; }

// The backward chain's stream in the order it is consumed (mlp_bf16x3.hip: nrf_pack_weights_transposed_bf16x3), two fp16
// terms per weight.  (Out-of-range weights are reported by the forward packer, nrf_packed_fp16x2_status.)
NRF_EXPORT int nrf_pack_weights_transposed_fp16x2(const float* const* w, void* packed_t, nrf_stream_t stream_) {
    if (!w || !packed_t) return NRF_ERR_NULL;
    for (int i = 0; i < NRF_NUM_LAYERS; ++i)
        if (!w[i]) return NRF_ERR_NULL;
    if (!nrf_aligned16(packed_t)) return NRF_ERR_ALIGN;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    unsigned short* base = static_cast<unsigned short*>(packed_t);
    nrf_launch_begin();
    int step = 0;
    HPackBatch batch;
    int n_batch = 0, max_total = 0;
    auto launch = [&](const float* W, int ldw, int in_blocks, int col0) {   // in_blocks = k-blocks = rows of W / 32
        batch.l[n_batch++] = XPackLayer{W, base + int64_t(step) * (kHStepBytes / 2), ldw, 8, 0, in_blocks, 0, 0, 0, 0, col0, 0, kHLead, 1};
        const int total = 8 * in_blocks * 2 * 512;
        max_total = total > max_total ? total : max_total;
        step += 8 * in_blocks * 2;
    };
    launch(w[10], 283, 4, 0);   // views^T: 128 outputs of views_linears.0 contracted, 256 feature columns out
    launch(w[8], 256, 8, 0);    // feature^T
    for (int l = 7; l >= 1; --l) launch(w[l], l == 5 ? 319 : 256, 8, l == 5 ? 63 : 0);
    hpack_layers_kernel<<<dim3((max_total + 255) / 256, n_batch), 256, 0, stream>>>(batch, nullptr);
    return nrf_launch_status();
}

NRF_EXPORT int64_t nrf_packed_fp16x2_bytes(void) { return kHStatusOff + 16; }

NRF_EXPORT int nrf_pack_weights_fp16x2(const float* const* w, const float* const* b, void* packed, nrf_stream_t stream_) {
    if (!w || !b || !packed) return NRF_ERR_NULL;
    for (int i = 0; i < NRF_NUM_LAYERS; ++i)
        if (!w[i] || !b[i]) return NRF_ERR_NULL;
    if (!nrf_aligned16(packed)) return NRF_ERR_ALIGN;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    unsigned short* base = static_cast<unsigned short*>(packed);
    unsigned* status = reinterpret_cast<unsigned*>(static_cast<char*>(packed) + kHStatusOff);
    if (hipMemsetAsync(status, 0, 16, stream) != hipSuccess) return NRF_ERR_RUNTIME;
    nrf_launch_begin();
    HPackBatch batch;
    int n_batch = 0, max_total = 0;
    auto launch = [&](XPackLayer p, int step0) {
        p.dst = base + int64_t(step0) * (kHStepBytes / 2);
        const int total = p.nob * (p.pre_kb + p.hid_kb + p.post_kb) * 2 * 512;
        batch.l[n_batch++] = p;
        max_total = total > max_total ? total : max_total;
    };
    xpack_forward_layers(w, launch, kHLead);
    hpack_layers_kernel<<<dim3((max_total + 255) / 256, n_batch), 256, 0, stream>>>(batch, status);
    const PackSmall s = xpack_small(w, b, packed, kHSmallFloatOff);
    pack_small_kernel<<<(kSmallFloats + 255) / 256, 256, 0, stream>>>(s, 0, 0);
    return nrf_launch_status();
}

// NRF_OK, or NRF_ERR_RANGE when the packer met a weight outside fp16's range (the forward kernels then return NaN).
// Waits for the stream.
NRF_EXPORT int nrf_packed_fp16x2_status(const void* packed, nrf_stream_t stream_) {
    if (!packed) return NRF_ERR_NULL;
    unsigned flag = 0;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (hipMemcpyAsync(&flag, static_cast<const char*>(packed) + kHStatusOff, 4, hipMemcpyDeviceToHost, stream) != hipSuccess ||
        hipStreamSynchronize(stream) != hipSuccess)
        return NRF_ERR_RUNTIME;
    return flag ? NRF_ERR_RANGE : NRF_OK;
}

NRF_EXPORT int nrf_mlp_forward_rays_fp16x2(const void* packed, const float* rays_o, const float* rays_d, const float* z,
                                           int64_t z_stride, int64_t n_rays, int n_samples, float* raw, nrf_stream_t stream) {
    if (n_rays < 0 || n_samples <= 0 || n_samples % 32 != 0 || z_stride < 0) return NRF_ERR_SHAPE;
    if (z_stride != 0 && z_stride < n_samples) return NRF_ERR_SHAPE;
    if (n_rays == 0) return NRF_OK;
    if (!packed || !rays_o || !rays_d || !z || !raw) return NRF_ERR_NULL;
    if (!nrf_aligned16(packed) || !nrf_aligned16(raw)) return NRF_ERR_ALIGN;
    if (n_rays > (int64_t(1) << 40) / n_samples) return NRF_ERR_SHAPE;
    return launch_h2(packed, rays_o, rays_d, z, z_stride, n_samples / 32, 1, n_rays * n_samples, raw,
                     static_cast<hipStream_t>(stream));
}

NRF_EXPORT int nrf_mlp_forward_rays_train_fp16x2(const void* packed, const float* rays_o, const float* rays_d, const float* z,
                                                 int64_t z_stride, int64_t n_rays, int n_samples, float* raw, float* acts,
                                                 uint32_t* masks, nrf_stream_t stream) {
    if (n_rays < 0 || n_samples <= 0 || n_samples % 32 != 0 || z_stride < 0) return NRF_ERR_SHAPE;
    if (z_stride != 0 && z_stride < n_samples) return NRF_ERR_SHAPE;
    if (n_rays == 0) return NRF_OK;
    if (!packed || !rays_o || !rays_d || !z || !raw || !acts || !masks) return NRF_ERR_NULL;
    if (!nrf_aligned16(packed) || !nrf_aligned16(raw)) return NRF_ERR_ALIGN;
    if (n_rays > (int64_t(1) << 40) / n_samples) return NRF_ERR_SHAPE;
    return launch_h2(packed, rays_o, rays_d, z, z_stride, n_samples / 32, 1, n_rays * n_samples, raw,
                     static_cast<hipStream_t>(stream), acts, masks);
}

NRF_EXPORT int nrf_mlp_forward_points_fp16x2(const void* packed, const float* pts, const float* dirs, int64_t n_points,
                                             float* raw, nrf_stream_t stream) {
    if (n_points < 0 || n_points > (int64_t(1) << 40)) return NRF_ERR_SHAPE;
    if (n_points == 0) return NRF_OK;
    if (!packed || !pts || !dirs || !raw) return NRF_ERR_NULL;
    if (!nrf_aligned16(packed) || !nrf_aligned16(raw)) return NRF_ERR_ALIGN;
    return launch_h2(packed, pts, dirs, nullptr, 0, 1, 0, n_points, raw, static_cast<hipStream_t>(stream));
}
