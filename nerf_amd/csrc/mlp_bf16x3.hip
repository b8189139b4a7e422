// fp32-class 8x256 NeRF MLP on the bf16 matrix cores: every fp32 operand is split into THREE bf16 terms
// (a = a0 + a1 + a2, a0 = bf16(a), a1 = bf16(a - a0), a2 = bf16(a - a0 - a1): 24 significand bits) and every product
// W.x becomes the six MFMAs with i + j <= 2 (w2.x0, w0.x2, w1.x1, w1.x0, w0.x1, w0.x0; fp32 accumulate).  The dropped
// terms are 2^-24 of the product, fp32's own rounding unit, so results are fp32-class (oracle restatement:
// oracle/nerf_oracle.py:mlp_forward(split3=True); measured distance from a float64 evaluation <= the fp32
// reference's own), while six v_mfma_f32_32x32x16_bf16 (192 cycles per 16 k) replace eight
// v_mfma_f32_32x32x2_f32 (512 cycles): 2.67x the fp32 MFMA rate on paper.  Same reference path as mlp_fp32.hip
// (volume_renderer.py:227-231, 393-423, network.py:49-74, 171-195, encoding/freq.py:31-32).
//
// Two forms of the same chain live here: the forward kernels (inference and training forward) issue
// v_mfma_f32_16x16x32_bf16 -- mlp_bf16x3_m16_kernel, further down: the chip holds a higher clock on that shape, DESIGN 3.9 --
// and the data-gradient chain (mlp_dgrad_x3_kernel) v_mfma_f32_32x32x16_bf16, the form everything was first written in
// (mlp_bf16x3_kernel: kept as an A/B build of the forward, -DX3_FWD_32X32).  The description that follows is the 32x32x16
// form; the 16x16x32 section says what changes.
//
// Shape: the fp32 kernel's chain (mlp_chain.h: one wave = 32 points x 256 features, all 8 accumulator blocks Y of a
// layer live, the next layer's input X made from Y just in time while output block 0 accumulates), with X held as
// three bf16 B-fragment sets (192 registers) instead of fp32 (128), and the weight stream in 3-KiB steps
// (hi | mid | lo fragment of one k16-step of one 32-row block) through the same LDS-DMA ring: 8-step = 24-KiB chunks,
// 5 slots, one raw barrier per chunk, the 6 refill pieces of a wave in the 6 MFMA gaps of the hand-over step.
// Heads (alpha 256->1, rgb 128->3) are fp32 VALU dot products on the fp32 accumulators, as in the fp32 kernel.
#include "split_stream.h"
#include "split3.h"


namespace {

constexpr int kXStepBytes = 3 * 1024;                 // hi | mid | lo A fragments of one k16-step
constexpr int kXChunkBytes = kXChunkSteps * kXStepBytes;   // 24 KiB
constexpr int kXRing = 5;
constexpr int kXPieces = kXChunkBytes / 1024 / 4;     // LDS-DMA pieces per wave per chunk: 6
#ifndef X3_LOOK
#define X3_LOOK 2
#endif
constexpr int kXLook = X3_LOOK;                       // LDS -> register look-ahead of the weight fragments, in steps (< 4)
constexpr int64_t kXSmallFloatOff = int64_t(kXStreamSteps) * kXStepBytes / 4;   // fp32 small block behind the stream
constexpr int kXLdsConstOff = kXRing * kXChunkBytes;                             // 120 KiB
constexpr int kXLdsBytes = kXLdsConstOff + kSmallFloats * 4;                     // + 12.1 KiB

// ---- weight packing ------------------------------------------------------------------------
__device__ __forceinline__ unsigned short bf_bits(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
__device__ __forceinline__ float bf_val(unsigned short b) { return __uint_as_float(unsigned(b) << 16); }

// all layers of a stream in one launch (blockIdx.y = layer; mlp_fp16x2.hip: hpack_layers_kernel)
constexpr int kXPackBatch = 11;
struct XPackBatch { XPackLayer l[kXPackBatch]; };
__global__ void xpack_layers_kernel(XPackBatch b) {   // one thread writes the three terms of one weight
    const XPackLayer& p = b.l[blockIdx.y];
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= xpack_total(p)) return;
    int step, lane, jj;
    const float v = xpack_fetch(p, idx, step, lane, jj);
    const unsigned short b0 = bf_bits(v);
    const float r1 = v - bf_val(b0);
    const unsigned short b1 = bf_bits(r1);
    const unsigned short b2 = bf_bits(r1 - bf_val(b1));
    unsigned short* d = p.dst + int64_t(step) * (kXStepBytes / 2) + lane * 8 + jj;
    d[0] = b0, d[512] = b1, d[1024] = b2;
}

// ---- LDS ring of the 3-term stream (mlp_chain.h's WeightStream with 3-KiB steps) -----------------
struct XRing {
    const char* gnext;   // per lane: this lane's 16 B of piece 6w of the next chunk to prefetch
    const char* gbegin;
    const char* gend;
    unsigned lds_base;   // LDS byte address of smem + wave * 6 KiB (this wave fills pieces 6w..6w+5)
    char* smem;
    int pf_slot, rd_slot;
    const u32x4* rd;     // this lane's 16 B of term 0 of step 0 of the chunk being read
    u32x4 wq[4][3];      // the three weight fragments of steps s .. s + kXLook - 1 (slot = step & 3), read from LDS
                         // kXLook steps ahead of their MFMAs
};
__device__ __forceinline__ void xr_piece(const XRing& st, int j) {
#ifdef X3_ABL_NODMA   // timing-only ablation builds (tools/debug/x3_ablate.py): results are wrong
    return;
#endif
    glds16(st.gnext + j * 1024, __builtin_amdgcn_readfirstlane(st.lds_base + unsigned(st.pf_slot) * kXChunkBytes + j * 1024));
}
__device__ __forceinline__ void xr_advance(XRing& st) {
    const char* nx = st.gnext + kXChunkBytes;
    st.gnext = nx == st.gend ? st.gbegin : nx;
    asm volatile("" : "+v"(st.gnext));  // keep it a run-time value (see ws_prefetch_advance)
    st.pf_slot = st.pf_slot + 1 == kXRing ? 0 : st.pf_slot + 1;
}
__device__ __forceinline__ void xr_acquire(XRing& st, int lane) {
#ifndef X3_ABL_NOBARRIER
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)" ::"n"(kXPieces * (kXRing - 2)) : "memory");
    __builtin_amdgcn_s_barrier();
#endif
    st.rd_slot = st.rd_slot + 1 == kXRing ? 0 : st.rd_slot + 1;
    st.rd = reinterpret_cast<const u32x4*>(st.smem + st.rd_slot * kXChunkBytes) + lane;
}
__device__ __forceinline__ void xr_read1(XRing& st, int slot, int step_in_chunk, int t) {
#ifdef X3_ABL_NOLDS     // timing-only: operands stay whatever the registers hold
    return;
#endif
    st.wq[slot][t] = st.rd[(step_in_chunk * 3 + t) * 64];
}
__device__ __forceinline__ void xr_read(XRing& st, int slot, int step_in_chunk) {
#pragma unroll
    for (int t = 0; t < 3; ++t) xr_read1(st, slot, step_in_chunk, t);
}
__device__ __forceinline__ void xr_start(XRing& st, const char* stream_base, char* smem, int wave, int lane,
                                         int n_chunks = kXChunks) {
    st.gbegin = stream_base + wave * (kXPieces * 1024) + lane * 16;
    st.gend = st.gbegin + size_t(n_chunks) * kXChunkBytes;
    st.gnext = st.gbegin;
    st.smem = smem;
    st.lds_base = __builtin_amdgcn_readfirstlane(unsigned(reinterpret_cast<uintptr_t>((lds_vptr)smem)) + unsigned(wave) * unsigned(kXPieces * 1024));
    st.pf_slot = 0;
    // the state of a stream in flight at step 0: chunks 0 .. kXRing-2 on their way and the last slot's refill at piece
    // kXLook (x3_step issues piece (s + kXLook) % 8 of the current refill on step s)
#pragma unroll
    for (int i = 0; i < kXRing; ++i) {
#pragma unroll
        for (int jj = 0; jj < (i + 1 < kXRing ? kXPieces : kXLook); ++jj) xr_piece(st, jj);
        if (i + 1 < kXRing) xr_advance(st);
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kXPieces * (kXRing - 2) + kXLook) : "memory");
    __builtin_amdgcn_s_barrier();
    st.rd_slot = 0;
    st.rd = reinterpret_cast<const u32x4*>(smem) + lane;
#pragma unroll
    for (int i = 0; i < kXLook; ++i) xr_read(st, i, i);
}


// The same ring with the stream addressed as a UNIFORM base (an SGPR pair that walks the chunks) + one 32-bit lane offset
// (global_load_lds_dwordx4 voff, s[base]) instead of three per-lane 64-bit pointers: five registers fewer, and no
// vector arithmetic on addresses (the 16x16x32 kernel below: one of those pointers lived in scratch, and every reload's
// wait drained the ring).
struct XRingS {
    const char* snext;   // uniform: this wave's piece 0 of the next chunk to prefetch
    const char* sbegin;
    const char* send;
    unsigned voff;       // lane * 16
    unsigned lds_base;
    char* smem;
    int pf_slot, rd_slot;
    const u32x4* rd;
    u32x4 wq[4][3];
};
__device__ __forceinline__ void xr_piece(const XRingS& st, int j) {
    asm volatile(
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, %1"
        :
        : "v"(st.voff), "s"(st.snext + j * 1024),
          "s"(__builtin_amdgcn_readfirstlane(st.lds_base + unsigned(st.pf_slot) * kXChunkBytes + j * 1024))
        : "memory");
}
__device__ __forceinline__ void xr_advance(XRingS& st) {
    const char* nx = st.snext + kXChunkBytes;
    st.snext = nx == st.send ? st.sbegin : nx;
    asm volatile("" : "+s"(st.snext));  // keep it a run-time value (see ws_prefetch_advance)
    st.pf_slot = st.pf_slot + 1 == kXRing ? 0 : st.pf_slot + 1;
}
__device__ __forceinline__ void xr_acquire(XRingS& st, int lane) {
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)" ::"n"(kXPieces * (kXRing - 2)) : "memory");
    __builtin_amdgcn_s_barrier();
    st.rd_slot = st.rd_slot + 1 == kXRing ? 0 : st.rd_slot + 1;
    st.rd = reinterpret_cast<const u32x4*>(st.smem + st.rd_slot * kXChunkBytes) + lane;
}
__device__ __forceinline__ void xr_read1(XRingS& st, int slot, int step_in_chunk, int t) {
    st.wq[slot][t] = st.rd[(step_in_chunk * 3 + t) * 64];
}
__device__ __forceinline__ void xr_start(XRingS& st, const char* stream_base, char* smem, int wave, int lane) {
    st.sbegin = stream_base + wave * (kXPieces * 1024);
    st.send = st.sbegin + size_t(kXChunks) * kXChunkBytes;
    st.snext = st.sbegin;
    st.voff = unsigned(lane) * 16u;
    st.smem = smem;
    st.lds_base = __builtin_amdgcn_readfirstlane(unsigned(reinterpret_cast<uintptr_t>((lds_vptr)smem)) + unsigned(wave) * unsigned(kXPieces * 1024));
    st.pf_slot = 0;
#pragma unroll
    for (int i = 0; i < kXRing; ++i) {
#pragma unroll
        for (int jj = 0; jj < (i + 1 < kXRing ? kXPieces : kXLook); ++jj) xr_piece(st, jj);
        if (i + 1 < kXRing) xr_advance(st);
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kXPieces * (kXRing - 2) + kXLook) : "memory");
    __builtin_amdgcn_s_barrier();
    st.rd_slot = 0;
    st.rd = reinterpret_cast<const u32x4*>(smem) + lane;
#pragma unroll
    for (int i = 0; i < kXLook; ++i)
#pragma unroll
        for (int t = 0; t < 3; ++t) xr_read1(st, i, i, t);
}

// Input op of a hidden-type layer: X[kb] (three-term B fragments) from Y[kb] of the previous layer, in eight slices of
// one register pair each (15 VALU instructions per slice behind OpInBase's activation).
// backward layers whose output did not feed alpha_linear (all but h7): kOpBwd without the alpha term -- no head-row read
// from LDS (and its wait) in every third MFMA gap, no multiply-add by a d sigma that is zero.  Same bits (the term was
// fma(w, 0, y)); chain 5.52 -> 5.36 ms per step on one box.  The h7 layer is peeled out of the layer loop: as an if / else
// of two instantiations inside the loop the kernel needed 300 B of scratch per lane and ran 24 % slower.
#ifdef X3_BWD_ALPHA_ALWAYS   // (A/B switch)
constexpr int kOpBwdPlain = kOpBwd;
#else
constexpr int kOpBwdPlain = 3;
#endif
// ... and the two layers whose input has no activation in front of it (views^T: dYv is masked where it is made; feature^T:
// no relu between feature_linear and views_linears): no sign word either
#ifdef X3_BWD_MASK_ALWAYS    // (A/B switch)
constexpr int kOpBwdOpen = kOpBwdPlain;
#else
constexpr int kOpBwdOpen = 4;
#endif
constexpr bool op_is_bwd(int mode) { return mode == kOpBwd || mode == kOpBwdPlain || mode == kOpBwdOpen; }
struct OpIn : OpInBase {
    // the same pair in three sub-slices of ~5 VALU instructions, one per MFMA gap (15 in one gap hold the wave's
    // in-order MFMA issue for 60 cycles: the matrix pipe runs dry behind them)
    float sa, sb, ra, rb;
    unsigned p0, p1;
    SplitSel k;   // the residuals by v_dot2c_f32_bf16 (split3.h): 3 + 3 + 3 instructions per pair behind the activation
    template <int MODE, class YT>
    __device__ __forceinline__ void input(const YT& y, int blk, int q, float& a, float& b) {
        if (MODE == kOpBwdOpen && kOpBwdOpen != kOpBwdPlain) {
            a = y[2 * q], b = y[2 * q + 1];
        } else if (MODE == kOpBwdPlain && kOpBwdPlain != kOpBwd) {
            const int word = int(m[blk >> 1]), at = 16 * (blk & 1) + 2 * q;
            a = __int_as_float(__float_as_int(y[2 * q]) & __builtin_amdgcn_sbfe(word, at, 1));
            b = __int_as_float(__float_as_int(y[2 * q + 1]) & __builtin_amdgcn_sbfe(word, at + 1, 1));
        } else if (MODE == kOpBwd) {
            const float2 w = *reinterpret_cast<const float2*>(aw + blk * 32 + 2 * q);
            const int word = int(m[blk >> 1]), at = 16 * (blk & 1) + 2 * q;   // (split_stream.h: OpInBase::input)
            a = __int_as_float(__float_as_int(fmaf(w.x, dsig, y[2 * q])) & __builtin_amdgcn_sbfe(word, at, 1));
            b = __int_as_float(__float_as_int(fmaf(w.y, dsig, y[2 * q + 1])) & __builtin_amdgcn_sbfe(word, at + 1, 1));
        } else {
            const int b0 = __float_as_int(y[2 * q]), b1 = __float_as_int(y[2 * q + 1]);
            a = __int_as_float(b0 > lim ? b0 : lim), b = __int_as_float(b1 > lim ? b1 : lim);
        }
        keep<MODE, false, is_blk16<YT>::value>(blk, q, a, b);
    }
    template <int MODE, class YT>
    __device__ __forceinline__ void sub(const YT& y, u32x4 (&x)[2][3], int blk, int q, int phase) {
        if (phase == 0) {
            input<MODE>(y, blk, q, sa, sb);
            p0 = pk_bf16(sa, sb);
            x[q / 4][0][q % 4] = p0;
        } else if (op_is_bwd(MODE)) {   // the backward chain keeps the shift / mask / subtract form: measured 1.5 % faster there
            if (phase == 1) {          // (tools/debug/lib_ab_prof.sh, one box: chain 5.70 against 5.78 ms per step)
                ra = __fsub_rn(sa, bf_lo(p0)), rb = __fsub_rn(sb, bf_hi(p0));
                p1 = pk_bf16(ra, rb);
                x[q / 4][1][q % 4] = p1;
            } else {
                x[q / 4][2][q % 4] = pk_bf16(__fsub_rn(ra, bf_lo(p1)), __fsub_rn(rb, bf_hi(p1)));
            }
        } else if (phase == 1) {
            ra = minus_bf(sa, p0, k.lo), rb = minus_bf(sb, p0, k.hi);
            p1 = pk_bf16(ra, rb);
            x[q / 4][1][q % 4] = p1;
        } else {
            x[q / 4][2][q % 4] = pk_bf16(minus_bf(ra, p1, k.lo), minus_bf(rb, p1, k.hi));
        }
    }
    template <int MODE, class YT>
    __device__ __forceinline__ void pair(const YT& y, u32x4 (&x)[2][3], int blk, int q) {
        float a, b;
        input<MODE>(y, blk, q, a, b);
        unsigned p0, p1, p2;
        if (op_is_bwd(MODE)) split_pair(a, b, p0, p1, p2);
        else split_pair_dot2(k, a, b, p0, p1, p2);
        x[q / 4][0][q % 4] = p0, x[q / 4][1][q % 4] = p1, x[q / 4][2][q % 4] = p2;
    }
};

// One k16-step of the stream: the six MFMAs of one (output block, k16) pair into `y`; `hook(m)` runs after MFMA m.
// The ring hand-over (acquire; the wave's 6 refill pieces on this and the next five steps) rides on the step kXLook before a chunk's end.
template <class Hook>
__device__ __forceinline__ void x3_step(XRing& st, int lane, int s /* stream step inside the layer */, f32x16& y,
                                        const u32x4* x, Hook&& hook) {
    // the hand-over to the next chunk comes with the first read that reaches into it, kXLook steps before the chunk's end
    // the wave's six refill pieces follow one per step (the four waves leave the barrier together: six pieces each in
    // six consecutive gaps are 128 B/clk asked of a 64 B/clk address path, and the MFMA issue waits behind them)
    const int piece = (s + kXLook) % kXChunkSteps;
    const bool hand = piece == 0;
    const u32x4 w0 = st.wq[s & 3][0], w1 = st.wq[s & 3][1], w2 = st.wq[s & 3][2];
    if (hand) xr_acquire(st, lane);
    // one ds_read_b128 per MFMA gap (three in one gap saturate the LDS array when the four waves coincide)
    const int rs = (s + kXLook) & 3, rc = (s + kXLook) % kXChunkSteps;
    // small terms first; all six accumulate into the same block (a dependent chain issues at full rate)
    // Every MFMA gap is its own scheduling region: left alone, the compiler (at ~420 live registers) sinks the
    // look-ahead reads down to their first use and reorders the six MFMAs, and every step then waits out the LDS latency.
#define X3_GAP(m, wa, xb, RD)                                                                            \
    y = NRF_MFMA_X3(wa, xb, y);                                                                          \
    RD;                                                                                                  \
    if (m == 4 && piece < kXPieces) { xr_piece(st, piece); if (piece == kXPieces - 1) xr_advance(st); }  \
    hook(m);                                                                                             \
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_sched_barrier(0);
    X3_GAP(0, w2, x[0], xr_read1(st, rs, rc, 0))
    X3_GAP(1, w0, x[2], xr_read1(st, rs, rc, 1))
    X3_GAP(2, w1, x[1], xr_read1(st, rs, rc, 2))
    X3_GAP(3, w1, x[0], (void)0)
    X3_GAP(4, w0, x[1], (void)0)
    X3_GAP(5, w0, x[0], (void)0)
#undef X3_GAP
}

// Encoding-only layer (pts_linears.0, and the skip layer's PE pass which ACCUMULATES onto pts_linears.5's hidden part):
// Y[ob] = (accum ? Y[ob] : bias) + W . pe.  `accum` is a run-time flag so both uses share one body.
__device__ __forceinline__ void x3_pe_layer(XRing& st, int lane, const float* bias_h, bool accum, const u32x4 (*pe)[2][3],
                                            f32x16* Y) {
    static_assert(kXStepsL0 % kXChunkSteps == 0, "layer must be a whole number of chunks");
    auto none = [](int) {};
#pragma unroll
    for (int ob = 0; ob < 8; ++ob) {
        const f32x16 bias = *reinterpret_cast<const f32x16*>(bias_h + ob * 32);
#pragma unroll
        for (int t = 0; t < 16; ++t) Y[ob][t] = accum ? Y[ob][t] : bias[t];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) x3_step(st, lane, (ob * 2 + kb) * 2 + s2, Y[ob], pe[kb][s2], none);
    }
}

// Hidden-type layer: Y[ob] = bias + W . [X (8 k-blocks), post], X[kb] = op(Y[kb] of the previous layer).
// Making one X[kb] is 120 VALU instructions, more than the 12 MFMAs of one (block, k-block) can cover, so the layer
// opens with output blocks 0 AND 1 accumulated together (stream order [kb][ob in {0,1}][s2] for that part, see
// xpack_layers_kernel): X[kb+1] is made in eight slices under the 24 MFMAs of k-block kb.  Block 1 accumulates into the
// spare block Z because the previous layer's Y[1] is still needed (it becomes X[1] during k-block 0).
template <int NOB, int NPOST, int MODE, int NHID = 8>
__device__ __forceinline__ void x3_hidden_layer(XRing& st, int lane, const float* bias_h, u32x4 (*X)[2][3],
                                                const u32x4 (*post)[2][3], f32x16* Y, f32x16& Z, OpIn& op) {
    constexpr int NKB = NHID + NPOST;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto bias = [&](int ob) { return op_is_bwd(MODE) ? zero16 : *reinterpret_cast<const f32x16*>(bias_h + ob * 32); };
    static_assert((NOB * NKB * 2) % kXChunkSteps == 0, "layer must be a whole number of chunks");
#pragma unroll
    for (int q = 0; q < 8; ++q) op.template pair<MODE>(Y[0], X[0], 0, q);
    Y[0] = bias(0);
    Z = bias(1);
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        const bool make = kb + 1 < NHID;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const u32x4* x = kb < NHID ? X[kb][s2] : post[kb - NHID][s2];
                const int g = b * 2 + s2;                       // which of the k-block's four steps
                auto hook = [&](int m) {                        // 24 MFMAs: pair n / 3, sub-slice n % 3 after MFMA n
                    const int n = g * 6 + m;
                    if (make) op.template sub<MODE>(Y[kb + 1], X[kb + 1], kb + 1, n / 3, n % 3);
                    // the next block's bias is read one step ahead of its first MFMA (Y[2] is free: X[2] is made)
                    if (kb == NKB - 1 && g == 3 && m == 0 && NOB > 2) Y[2] = bias(2);
                };
                x3_step(st, lane, kb * 4 + g, b == 0 ? Y[0] : Z, x, hook);
            }
    }
    Y[1] = Z;
#pragma unroll
    for (int ob = 2; ob < NOB; ++ob) {
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                auto hook = [&](int m) {
                    if (kb == NKB - 1 && s2 == 1 && m == 0 && ob + 1 < NOB) Y[ob + 1] = bias(ob + 1);
                };
                x3_step(st, lane, (ob * NKB + kb) * 2 + s2, Y[ob], kb < NHID ? X[kb][s2] : post[kb - NHID][s2], hook);
            }
    }
}

// Persistent, one 4-wave workgroup per CU, one wave per SIMD, a wave's tile = 32 points (mlp_fp32.hip's tiling).
// SAVE (training forward): also writes every layer's activated input and relu sign bits in mlp_fwd_kernel<true>'s
// layout, for the fp32 backward kernels of mlp_bwd.hip.
template <bool SAVE>
__global__ __launch_bounds__(256, 1) void mlp_bf16x3_kernel(const char* __restrict__ packed, const float* __restrict__ a0,
                                                            const float* __restrict__ a1, const float* __restrict__ zbuf,
                                                            int64_t z_stride, int tiles_per_ray, int rays_mode,
                                                            int64_t n_points, int64_t n_groups, float4* __restrict__ raw,
                                                            float* __restrict__ acts, unsigned* __restrict__ masks) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    {
        const float4* src = reinterpret_cast<const float4*>(packed + kXSmallFloatOff * 4);
        float4* dst = reinterpret_cast<float4*>(smem + kXLdsConstOff);
        for (int i = threadIdx.x; i < kSmallFloats / 4; i += 256) dst[i] = src[i];
    }
    __syncthreads();
    const float* cst = reinterpret_cast<const float*>(smem + kXLdsConstOff);
    const float* bias_h = cst + h * 16;

    XRing st;
    xr_start(st, packed, smem, wave, lane);
    const SplitSel sel = SplitSel::make();

    for (int64_t tg = blockIdx.x; tg < n_groups; tg += gridDim.x) {
        const int64_t tile = tg * 4 + wave;
        int64_t pidx = tile * 32 + j;
        const bool valid = pidx < n_points;
        if (!valid) pidx = n_points - 1;
        float px, py, pz, dx, dy, dz;
        if (rays_mode) {
            const int64_t tile_c = pidx / 32;
            const int64_t ray = tile_c / tiles_per_ray;
            const int s0 = int(tile_c - ray * tiles_per_ray) * 32;
            const float ox = a0[ray * 3 + 0], oy = a0[ray * 3 + 1], oz = a0[ray * 3 + 2];
            dx = a1[ray * 3 + 0], dy = a1[ray * 3 + 1], dz = a1[ray * 3 + 2];
            const float nrm = sqrtf(fmaf(dz, dz, fmaf(dy, dy, dx * dx)));  // volume_renderer.py:230 (torch's 3-element norm)
            dx = dx / nrm, dy = dy / nrm, dz = dz / nrm;
            const float zz = zbuf[ray * z_stride + s0 + (valid ? j : int(pidx & 31))];
            px = __fadd_rn(ox, __fmul_rn(zz, dx));  // :231
            py = __fadd_rn(oy, __fmul_rn(zz, dy));
            pz = __fadd_rn(oz, __fmul_rn(zz, dz));
        } else {
            px = a0[pidx * 3 + 0], py = a0[pidx * 3 + 1], pz = a0[pidx * 3 + 2];
            dx = a1[pidx * 3 + 0], dy = a1[pidx * 3 + 1], dz = a1[pidx * 3 + 2];
        }
        u32x4 pe[2][2][3], dpe[1][2][3];
        float* arow = nullptr;     // this lane's column of the tile's saved rows (mlp_fp32.hip)
        unsigned* mrow = nullptr;
        if (SAVE) {
#ifdef NRF_ABL_L2WRAP   // timing only (tools/debug/lib_ab.sh): tiles share NRF_ABL_L2WRAP + 1 slots (7: L2-resident, 255: 83 MB)
            arow = acts + (tile & NRF_ABL_L2WRAP) * (int64_t(kActRows) * 32) + 4 * h * 32 + j;
            mrow = masks + (tile & NRF_ABL_L2WRAP) * (kMaskLayers * 4 * 64) + lane;
#else
            arow = acts + tile * (int64_t(kActRows) * 32) + 4 * h * 32 + j;
            mrow = masks + tile * (kMaskLayers * 4 * 64) + lane;
#endif
        }
        {
            const f32x16 e0 = encode_block<10, 0>(px, py, pz, h), e1 = encode_block<10, 1>(px, py, pz, h);
            const f32x16 ed = encode_block<4, 0>(dx, dy, dz, h);
            if (SAVE) {
                store_block(arow + kActPe * 32, 0, e0);
                store_block(arow + kActPe * 32, 1, e1);
                store_block(arow + kActDpe * 32, 0, ed);
            }
            split_block_x3(sel, e0, pe[0]);
            split_block_x3(sel, e1, pe[1]);
            split_block_x3(sel, ed, dpe[0]);
        }

        u32x4 X[8][2][3];
        f32x16 Y[8], Z;
        OpIn op{op_in(kActRelu, cst + (kOffAlphaW - kOffBias) + h * 16), 0.f, 0.f, 0.f, 0.f, 0u, 0u, sel};
        // pts_linears.0 .. 7 and feature_linear (li = 8) as ONE hidden-layer body in a run-time loop; the encoding-only
        // layer runs before li = 1 (pts_linears.0) and before li = 6 (the PE columns of the skip layer pts_linears.5,
        // network.py:58 cat([pe, h]), accumulated onto its hidden part)
        for (int li = 1; li <= 8; ++li) {
            if (li == 1 || li == 6) x3_pe_layer(st, lane, bias_h, li == 6, pe, Y);
            if (li == 8) op.alpha_head(Y);                            // feature_linear's input relu(h7) feeds alpha_linear
            if (SAVE) {   // the layer's input h_{li-1} = relu(Y) and its signs (mask li - 1)
                op.save = arow + (kActH + (li - 1) * 256) * 32;
                op.msave = mrow + (li - 1) * 4 * 64;
            }
            x3_hidden_layer<8, 0, SAVE ? kOpFwdSave : kOpFwd>(st, lane, bias_h + li * 256, X, nullptr, Y, Z, op);
        }
        float sigma = op.sigma;
        asm volatile("" : "+v"(sigma));   // keep the head where it is (see mlp_bf16.hip)
        // feature -> views: no activation (saved as it is, no mask)
        OpIn op_views{op_in(kActNone, nullptr, SAVE ? arow + kActFeat * 32 : nullptr), 0.f, 0.f, 0.f, 0.f, 0u, 0u, sel};
        x3_hidden_layer<4, 1, SAVE ? kOpFwdSave : kOpFwd>(st, lane, bias_h + 9 * 256, X, dpe, Y, Z, op_views);        // views_linears.0
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
        if (h == 0 && valid) {
            const float* rb = cst + (kOffRgbB - kOffBias);
            raw[pidx] = make_float4(r + rb[0], g + rb[1], b + rb[2], sigma + cst[kOffAlphaB - kOffBias]);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA may outlive the workgroup
}

// ---- the inference forward on v_mfma_f32_16x16x32_bf16 --------------------------------------------------------------------
// Same chain, same stream positions, same VALU work; the matrix shape is the change.  Under this kernel's load the chip
// holds its clock by the energy a FLOP costs, and the 16x16x32 shape costs less than the 32x32x16 one at the same cycles
// per FLOP (MI355X_MICROARCH.md, DVFS give-back item 7; tools/ubench/mfma_shape_clock.hip models this kernel's step:
// 1.66 -> 1.91 GHz; a timing-only build of the kernel above with every MFMA replaced by two of the other shape: -7 %).
// A step at stream position (ob, kb, s2) is the three weight terms of rows 32 ob + 16 s2 .. + 15 against ALL 32 features of
// k-block kb (the 32x32 form: 32 rows against the 16 features of half s2), multiplied into two 16-point accumulators.
// Layout (split_stream.h: m16_slot): lane (c = lane & 15, g = lane >> 4); a block's 16 registers are four 16 x 16
// accumulators q[2 nb + mh]; register 8 nb + 4 mh + r is feature 16 mh + 4 g + r of point 16 nb + c -- so the eight
// registers of one nb are, in order, the lane's eight k-values of the next layer's B fragment, and the input op, the
// operand split and the fragment arrays X[kb][nb][term] of the 32x32 form are used unchanged (its s2 index is nb here).
#define NRF_MFMA_X16(a, b, c) \
    __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, (a)), __builtin_bit_cast(bf16x8, (b)), (c), 0, 0, 0)

// The pack holds the 16x16x32 stream and its small block; an A/B build with -DX3_FWD_32X32 runs the 32x32x16 kernels and puts
// their stream and small block in front.
#ifdef X3_FWD_32X32
constexpr int64_t kX16StreamOff = (kXSmallFloatOff + kSmallFloats) * 4;                  // bytes from the start of the pack
#else
constexpr int64_t kX16StreamOff = 0;
#endif
constexpr int64_t kX16SmallOff = kX16StreamOff + int64_t(kXStreamSteps) * kXStepBytes;
constexpr int64_t kXPackedBytes = kX16SmallOff + int64_t(kS16Floats) * 4;
static_assert(kX16StreamOff % 16 == 0 && kX16SmallOff % 16 == 0, "LDS-DMA and float4 alignment");
constexpr int kX16LdsBytes = kXLdsConstOff + kS16Floats * 4;                             // 120 KiB + 21.6 KiB
static_assert(kX16LdsBytes <= 160 * 1024, "LDS budget");

template <class Hook>
__device__ __forceinline__ void x16_step(XRingS& st, int lane, int s, Blk16& y, int mh, const u32x4* xk /* [nb][term] */,
                                         Hook&& hook) {
    f32x4 ya, yb;
    if (mh) blk16_get<1>(y, ya, yb);
    else blk16_get<0>(y, ya, yb);
    const int piece = (s + kXLook) % kXChunkSteps;   // ring hand-over and refill exactly as x3_step
    const bool hand = piece == 0;
    const u32x4 w0 = st.wq[s & 3][0], w1 = st.wq[s & 3][1], w2 = st.wq[s & 3][2];
    if (hand) xr_acquire(st, lane);
    const int rs = (s + kXLook) & 3, rc = (s + kXLook) % kXChunkSteps;
    // a gap = the same product for both point halves (2 x 16 matrix cycles); small terms first, as x3_step.  (Measured and
    // not kept, one box: the gap's VALU work between its two MFMAs +1.3 %; look-ahead of 1 / 3 steps +0.3 / +2 %.)
#define X16_GAP(m, wa, t, RD)                                                                            \
    ya = NRF_MFMA_X16(wa, xk[t], ya);                                                                    \
    yb = NRF_MFMA_X16(wa, xk[3 + t], yb);                                                                \
    RD;                                                                                                  \
    if (m == 4 && piece < kXPieces) { xr_piece(st, piece); if (piece == kXPieces - 1) xr_advance(st); }  \
    hook(m);                                                                                             \
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_sched_barrier(0);
    X16_GAP(0, w2, 0, xr_read1(st, rs, rc, 0))
    X16_GAP(1, w0, 2, xr_read1(st, rs, rc, 1))
    X16_GAP(2, w1, 1, xr_read1(st, rs, rc, 2))
    X16_GAP(3, w1, 0, (void)0)
    X16_GAP(4, w0, 1, (void)0)
    X16_GAP(5, w0, 0, (void)0)
#undef X16_GAP
    if (mh) blk16_put<1>(y, ya, yb);
    else blk16_put<0>(y, ya, yb);
}

__device__ __forceinline__ Blk16 bias16(const float* bias_g, int ob) {
    return Blk16{*reinterpret_cast<const f32x16*>(bias_g + ob * 64)};
}

__device__ __forceinline__ void x16_pe_layer(XRingS& st, int lane, const float* bias_g, bool accum, const u32x4 (*pe)[2][3], Blk16* Y) {
    auto none = [](int) {};
#pragma unroll
    for (int ob = 0; ob < 8; ++ob) {
        // (a uniform branch around the block's four LDS reads, not sixteen selects: the flag is a scalar, and the 256 selects
        // per tile were instructions that nothing hid)
        if (!accum) Y[ob] = bias16(bias_g, ob);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
                x16_step(st, lane, (ob * 2 + kb) * 2 + s2, Y[ob], s2, pe[kb][0], none);
    }
}

// x3_hidden_layer in the 16x16x32 form
template <int NOB, int NPOST, int MODE, int NHID = 8>
__device__ __forceinline__ void x16_hidden_layer(XRingS& st, int lane, const float* bias_g, u32x4 (*X)[2][3],
                                                 const u32x4 (*post)[2][3], Blk16* Y, Blk16& Z, OpIn& op) {
    constexpr int NKB = NHID + NPOST;
    static_assert((NOB * NKB * 2) % kXChunkSteps == 0, "layer must be a whole number of chunks");
#pragma unroll
    for (int q = 0; q < 8; ++q) op.template pair<MODE>(Y[0], X[0], 0, q);
    Y[0] = bias16(bias_g, 0);
    Z = bias16(bias_g, 1);
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        const bool make = kb + 1 < NHID;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const u32x4* xk = kb < NHID ? X[kb][0] : post[kb - NHID][0];
                const int g = b * 2 + s2;
                auto hook = [&](int m) {
                    const int n = g * 6 + m;
                    if (make) op.template sub<MODE>(Y[kb + 1], X[kb + 1], kb + 1, n / 3, n % 3);
                    if (kb == NKB - 1 && g == 3 && m == 0 && NOB > 2) Y[2] = bias16(bias_g, 2);
                };
                Blk16& y = b == 0 ? Y[0] : Z;
                x16_step(st, lane, kb * 4 + g, y, s2, xk, hook);
            }
    }
    Y[1] = Z;
#pragma unroll
    for (int ob = 2; ob < NOB; ++ob) {
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                auto hook = [&](int m) {
                    if (kb == NKB - 1 && s2 == 1 && m == 0 && ob + 1 < NOB) Y[ob + 1] = bias16(bias_g, ob + 1);
                };
                x16_step(st, lane, (ob * NKB + kb) * 2 + s2, Y[ob], s2, kb < NHID ? X[kb][0] : post[kb - NHID][0], hook);
            }
    }
}

// The kernel above in the 16x16x32 form.  `stream` / `small16`: the second half of the pack.  SAVE: the saved rows and
// sign words land in memory exactly as the 32x32 form writes them (rows are [row][32 points] whoever stores them; the
// sign words are re-ordered before their store, m16.h), so every backward kernel reads them unchanged.
template <bool SAVE>
__global__ __launch_bounds__(256, 1) void mlp_bf16x3_m16_kernel(const char* __restrict__ stream, const float* __restrict__ small16,
                                                                const float* __restrict__ a0, const float* __restrict__ a1,
                                                                const float* __restrict__ zbuf, int64_t z_stride, int tiles_per_ray,
                                                                int rays_mode, int64_t n_points, int64_t n_groups,
                                                                float4* __restrict__ raw, float* __restrict__ acts,
                                                                unsigned* __restrict__ masks) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5, g = lane >> 4;
    {
        const float4* src = reinterpret_cast<const float4*>(small16);
        float4* dst = reinterpret_cast<float4*>(smem + kXLdsConstOff);
        for (int i = threadIdx.x; i < kS16Floats / 4; i += 256) dst[i] = src[i];
    }
    __syncthreads();
    const float* cst = reinterpret_cast<const float*>(smem + kXLdsConstOff);
    const float* bias_g = cst + g * 16;

    XRingS st;
    xr_start(st, stream, smem, wave, lane);
    const SplitSel sel = SplitSel::make();

    for (int64_t tg = blockIdx.x; tg < n_groups; tg += gridDim.x) {
        const int64_t tile = tg * 4 + wave;
        int64_t pidx = tile * 32 + j;   // the point whose encoding this lane makes (32x32 form) and, in lanes 0..31, stores
        const bool valid = pidx < n_points;
        if (!valid) pidx = n_points - 1;
        float px, py, pz, dx, dy, dz;
        if (rays_mode) {
            const int64_t tile_c = pidx / 32;
            const int64_t ray = tile_c / tiles_per_ray;
            const int s0 = int(tile_c - ray * tiles_per_ray) * 32;
            const float ox = a0[ray * 3 + 0], oy = a0[ray * 3 + 1], oz = a0[ray * 3 + 2];
            dx = a1[ray * 3 + 0], dy = a1[ray * 3 + 1], dz = a1[ray * 3 + 2];
            const float nrm = sqrtf(fmaf(dz, dz, fmaf(dy, dy, dx * dx)));  // volume_renderer.py:230 (torch's 3-element norm)
            dx = dx / nrm, dy = dy / nrm, dz = dz / nrm;
            const float zz = zbuf[ray * z_stride + s0 + (valid ? j : int(pidx & 31))];
            px = __fadd_rn(ox, __fmul_rn(zz, dx));  // :231
            py = __fadd_rn(oy, __fmul_rn(zz, dy));
            pz = __fadd_rn(oz, __fmul_rn(zz, dz));
        } else {
            px = a0[pidx * 3 + 0], py = a0[pidx * 3 + 1], pz = a0[pidx * 3 + 2];
            dx = a1[pidx * 3 + 0], dy = a1[pidx * 3 + 1], dz = a1[pidx * 3 + 2];
        }
        u32x4 pe[2][2][3], dpe[1][2][3];
        float* arow = nullptr;     // this lane's base inside the tile's saved rows: 4 g * 32 + c (m16.h: saved_row_off)
        unsigned* mrow = nullptr;
        {
            const f32x16 e0 = encode_block<10, 0>(px, py, pz, h), e1 = encode_block<10, 1>(px, py, pz, h);
            const f32x16 ed = encode_block<4, 0>(dx, dy, dz, h);
            if (SAVE) {   // the encodings are made in the 32x32 form and saved from it
                float* arow32 = acts + tile * (int64_t(kActRows) * 32) + 4 * h * 32 + j;
                store_block(arow32 + kActPe * 32, 0, e0);
                store_block(arow32 + kActPe * 32, 1, e1);
                store_block(arow32 + kActDpe * 32, 0, ed);
                arow = acts + tile * (int64_t(kActRows) * 32) + 4 * g * 32 + (lane & 15);
                mrow = masks + tile * (kMaskLayers * 4 * 64) + lane;
            }
            split_block_x3(sel, to_m16(e0), pe[0]);
            split_block_x3(sel, to_m16(e1), pe[1]);
            split_block_x3(sel, to_m16(ed), dpe[0]);
        }

        u32x4 X[8][2][3];
        Blk16 Y[8], Z;
        constexpr int MODE = SAVE ? kOpFwdSave : kOpFwd;
        OpIn op{op_in(kActRelu, nullptr), 0.f, 0.f, 0.f, 0.f, 0u, 0u, sel};
        float sig0 = 0.f, sig1 = 0.f;
        for (int li = 1; li <= 8; ++li) {
            if (li == 1 || li == 6) x16_pe_layer(st, lane, bias_g + (li == 1 ? 0 : 5 * 512), li == 6, pe, Y);
            if (li == 8) {   // alpha_linear on relu(h7): one lump in front of feature_linear (OpInBase::alpha_head)
                const float* aw = cst + kS16AlphaW + g * 8;
#pragma unroll
                for (int kb = 0; kb < 8; ++kb) {
                    const f32x4 wl = *reinterpret_cast<const f32x4*>(aw + kb * 32), wh = *reinterpret_cast<const f32x4*>(aw + kb * 32 + 4);
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        const float w = t < 4 ? wl[t] : wh[t - 4];
                        const int b0 = __float_as_int(Y[kb][t]), b1 = __float_as_int(Y[kb][8 + t]);
                        sig0 = fmaf(w, __int_as_float(b0 > 0 ? b0 : 0), sig0);
                        sig1 = fmaf(w, __int_as_float(b1 > 0 ? b1 : 0), sig1);
                    }
                }
            }
            if (SAVE) {   // the layer's input h_{li-1} = relu(Y) and its signs (mask li - 1)
                op.save = arow + (kActH + (li - 1) * 256) * 32;
                op.msave = mrow + (li - 1) * 4 * 64;
            }
            x16_hidden_layer<8, 0, MODE>(st, lane, bias_g + li * 512, X, nullptr, Y, Z, op);
        }
        asm volatile("" : "+v"(sig0), "+v"(sig1));   // keep the head where it is (see mlp_bf16.hip)
        // feature -> views: no activation (saved as it is, no mask)
        OpIn op_views{op_in(kActNone, nullptr, SAVE ? arow + kActFeat * 32 : nullptr), 0.f, 0.f, 0.f, 0.f, 0u, 0u, sel};
        x16_hidden_layer<4, 1, MODE>(st, lane, bias_g + 9 * 512, X, dpe, Y, Z, op_views);   // views_linears.0
        float c0[3] = {0.f, 0.f, 0.f}, c1[3] = {0.f, 0.f, 0.f};
        {
            const float* rw = cst + kS16RgbW + g * 8;
            if (SAVE) {   // hv = relu(views output): rows and signs (mask 8), as OpFwdSave<true> does in the 32x32 form
                unsigned prev = 0u;
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
                    unsigned bits = 0u;
#pragma unroll
                    for (int t = 0; t < 16; ++t) {
                        const int bv = __float_as_int(Y[kb][t]);
                        const float hv = __int_as_float(bv > 0 ? bv : 0);
                        save_row(arow + (kActHv + 32 * kb) * 32 + saved_row_off<true>(t), hv);
                        bits |= (hv > 0.f ? 1u : 0u) << t;
                    }
                    const unsigned w = m16_mask_to_old(bits);
                    if (kb & 1) mrow[(8 * 4 + (kb >> 1)) * 64] = prev | (w << 16);
                    else prev = w;
                }
            }
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const f32x4 wl = *reinterpret_cast<const f32x4*>(rw + c * 128 + kb * 32);
                    const f32x4 wh = *reinterpret_cast<const f32x4*>(rw + c * 128 + kb * 32 + 4);
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        const float w = t < 4 ? wl[t] : wh[t - 4];
                        const int b0 = __float_as_int(Y[kb][t]), b1 = __float_as_int(Y[kb][8 + t]);
                        c0[c] = fmaf(w, __int_as_float(b0 > 0 ? b0 : 0), c0[c]);
                        c1[c] = fmaf(w, __int_as_float(b1 > 0 ? b1 : 0), c1[c]);
                    }
                }
        }
        // lanes 0..15 hold point c of the tile, lanes 16..31 point 16 + c: lane l < 32 stores point l
        const bool odd = (g & 1) != 0;
        const float r = add_groups(c0[0], c1[0], odd), gg = add_groups(c0[1], c1[1], odd);
        const float b = add_groups(c0[2], c1[2], odd), sg = add_groups(sig0, sig1, odd);
        if (lane < 32 && valid) {
            const float* rb = cst + kS16RgbB;
            raw[pidx] = make_float4(r + rb[0], gg + rb[1], b + rb[2], sg + cst[kS16AlphaB]);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA may outlive the workgroup
}

// ---- backward data-gradient chain on the same machinery (mlp_bwd.hip: mlp_dgrad_kernel is the fp32 MFMA form) -------
// d(input of layer l) = W_l^T . dY_l with the transposed weights as the three-term stream (views^T, feature^T,
// pts7^T .. pts1^T; nrf_pack_weights_transposed_bf16x3), dY_l = relu'(h_l) * d h_l from the sign words the training
// forward saved, every dY_l written once to HBM for the (fp32) wgrad GEMMs.  The operand split of a layer's input is
// the place where the mask, the alpha_linear term of h7 and the dY store happen (OpIn, MODE kOpBwd).
__global__ __launch_bounds__(256, 1) void mlp_dgrad_x3_kernel(const float* __restrict__ packed,   // fp32 pack: head rows
                                                              const char* __restrict__ packed_t,  // three-term W^T stream
                                                              const float4* __restrict__ graw, const unsigned* __restrict__ masks,
                                                              int64_t n_points, int64_t n_groups, float* __restrict__ dys) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    {
        const float4* src = reinterpret_cast<const float4*>(packed + kOffBias);
        float4* dst = reinterpret_cast<float4*>(smem + kXLdsConstOff);
        for (int i = threadIdx.x; i < kSmallFloats / 4; i += 256) dst[i] = src[i];
    }
    __syncthreads();
    const float* cst = reinterpret_cast<const float*>(smem + kXLdsConstOff);
    const float* aw = cst + (kOffAlphaW - kOffBias) + h * 16;
    const float* rw = cst + (kOffRgbW - kOffBias) + h * 16;

    XRing st;
    xr_start(st, packed_t, smem, wave, lane, kXBwdChunks);
    const SplitSel sel = SplitSel::make();

    for (int64_t tg = blockIdx.x; tg < n_groups; tg += gridDim.x) {
        const int64_t tile = tg * 4 + wave;
        int64_t pidx = tile * 32 + j;
        const bool valid = pidx < n_points;
        if (!valid) pidx = n_points - 1;
        float4 g = graw[pidx];
        if (!valid) g = make_float4(0.f, 0.f, 0.f, 0.f);
        const unsigned* mrow = masks + tile * (kMaskLayers * 4 * 64) + lane;
#ifdef NRF_ABL_L2WRAP
        float* drow = dys + (tile & NRF_ABL_L2WRAP) * (int64_t(kDyRows) * 32) + 4 * h * 32 + j;
#else
        float* drow = dys + tile * (int64_t(kDyRows) * 32) + 4 * h * 32 + j;
#endif
        if (h == 0) {  // d(raw) as 4 rows for the head wgrad jobs
            float* r = dys + tile * (int64_t(kDyRows) * 32) + kDyRaw * 32 + j;
            r[0] = g.x, r[32] = g.y, r[64] = g.z, r[96] = g.w;
        }
        u32x4 X[8][2][3];
        f32x16 Y[8], Z;
        // rgb_linear^T on the VALU (fp32): dYv = relu'(hv) * (W_rgb^T . d rgb), stored, and the input of views^T
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
                Y[kb] = mask16(v, ((kb < 2 ? m0 : m1) >> (16 * (kb & 1))) & 0xffffu);
            }
        }
        // d feature = W_views[:, :256]^T . dYv: four k-blocks in (masked above: all-ones mask here; the split stores them
        // as the dY rows of views_linears.0), eight blocks out
        OpIn op{op_in(kActNone, aw, drow + kDyV * 32), 0.f, 0.f, 0.f, 0.f, 0u, 0u, sel};
#pragma unroll
        for (int w = 0; w < 4; ++w) op.m[w] = ~0u;
        x3_hidden_layer<8, 0, kOpBwdOpen, 4>(st, lane, nullptr, X, nullptr, Y, Z, op);
        // li = 0: d h7 = W_feature^T . d feature (no activation between feature_linear and views_linears: all-ones mask,
        // stored as the dY rows of feature_linear); li = 1 .. 7: layer l = 8 - li, dY_l = relu'(h_l) * d h_l (+ the
        // alpha_linear term for h7), d h_{l-1} = W_l^T . dY_l.  One body, run-time loop.
#ifdef X3_BWD_ALPHA_ALWAYS
        for (int li = 0; li <= 7; ++li) {
            const int l = 8 - li;
            op.save = drow + (li == 0 ? kDyFeat : l * 256) * 32;
#pragma unroll
            for (int w = 0; w < 4; ++w) op.m[w] = li == 0 ? 0xffffffffu : mrow[(l * 4 + w) * 64];
            op.dsig = li == 1 ? g.w : 0.f;                       // h7 also feeds alpha_linear
            x3_hidden_layer<8, 0, kOpBwd>(st, lane, nullptr, X, nullptr, Y, Z, op);
        }
#else
#ifndef X3_BWD_MASKS_PER_LAYER   // (A/B switch)
        // every layer's sign words in one burst of loads before the tile's first MFMA (32 registers held through the tile)
        // instead of four loads at each layer's top, whose wait drains the ring's LDS-DMA queue nine times per tile: same bits,
        // chain 5.45 -> 5.40 ms per step, step 17.82 -> 17.76 ms on one box (three alternating runs)
        unsigned mq[8][4];   // mq[i]: layer 7 - i
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int w = 0; w < 4; ++w) mq[i][w] = mrow[((7 - i) * 4 + w) * 64];
#endif
        op.save = drow + kDyFeat * 32;                           // li = 0: feature^T
        x3_hidden_layer<8, 0, kOpBwdOpen>(st, lane, nullptr, X, nullptr, Y, Z, op);
        op.save = drow + 7 * 256 * 32;                           // li = 1: h7 also feeds alpha_linear
#pragma unroll
#ifndef X3_BWD_MASKS_PER_LAYER
        for (int w = 0; w < 4; ++w) op.m[w] = mq[0][w];
#else
        for (int w = 0; w < 4; ++w) op.m[w] = mrow[(7 * 4 + w) * 64];
#endif
        op.dsig = g.w;
        x3_hidden_layer<8, 0, kOpBwd>(st, lane, nullptr, X, nullptr, Y, Z, op);
        op.dsig = 0.f;
        for (int l = 6; l >= 1; --l) {
            op.save = drow + l * 256 * 32;
#ifndef X3_BWD_MASKS_PER_LAYER
#pragma unroll
            for (int w = 0; w < 4; ++w) op.m[w] = mq[1][w];
#pragma unroll
            for (int i = 1; i < 7; ++i)   // the queue moves up one layer
#pragma unroll
                for (int w = 0; w < 4; ++w) mq[i][w] = mq[i + 1][w];
#else
#pragma unroll
            for (int w = 0; w < 4; ++w) op.m[w] = mrow[(l * 4 + w) * 64];
#endif
            x3_hidden_layer<8, 0, kOpBwdPlain>(st, lane, nullptr, X, nullptr, Y, Z, op);
        }
#endif
        {   // dY_0 = relu'(h_0) * d h_0 (the encoding needs no gradient: the chain ends here)
            OpBwdMask last;
            last.dst = drow;
#pragma unroll
#if !defined(X3_BWD_MASKS_PER_LAYER) && !defined(X3_BWD_ALPHA_ALWAYS)
            for (int w = 0; w < 4; ++w) last.m[w] = mq[1][w];   // (after six moves: layer 0's)
#else
            for (int w = 0; w < 4; ++w) last.m[w] = mrow[w * 64];
#endif
            last.aw = aw;
            last.dsig = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) (void)last(i, Y[i]);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA may outlive the workgroup
}

int launch_x3(const void* packed, const float* a0, const float* a1, const float* z, int64_t z_stride, int tiles_per_ray,
              int rays_mode, int64_t n_points, float* raw, hipStream_t stream, float* acts = nullptr, unsigned* masks = nullptr) {
    const int64_t tiles = (n_points + 31) / 32;
    const int64_t groups = (tiles + 3) / 4;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (cus <= 0) cus = 256;
    static bool attr_set[64] = {};
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_bf16x3_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                kXLdsBytes) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_bf16x3_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                kXLdsBytes) != hipSuccess)
            return NRF_ERR_LAUNCH;
        attr_set[dev] = true;
    }
    const unsigned grid = unsigned(groups < cus ? groups : cus);
    nrf_launch_begin();
#ifndef X3_FWD_32X32   // (A/B switch: the 32x32x16 kernels on the first stream of the pack)
    {   // the 16x16x32 form, on the second stream of the pack
        static bool attr16_set[64] = {};
        if (dev >= 0 && dev < 64 && !attr16_set[dev]) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_bf16x3_m16_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    kX16LdsBytes) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_bf16x3_m16_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    kX16LdsBytes) != hipSuccess)
                return NRF_ERR_LAUNCH;
            attr16_set[dev] = true;
        }
        const char* base = static_cast<const char*>(packed);
        const float* small16 = reinterpret_cast<const float*>(base + kX16SmallOff);
        if (acts)
            mlp_bf16x3_m16_kernel<true><<<dim3(grid), 256, kX16LdsBytes, stream>>>(base + kX16StreamOff, small16, a0, a1, z, z_stride,
                                                                                  tiles_per_ray, rays_mode, n_points, groups,
                                                                                  reinterpret_cast<float4*>(raw), acts, masks);
        else
            mlp_bf16x3_m16_kernel<false><<<dim3(grid), 256, kX16LdsBytes, stream>>>(base + kX16StreamOff, small16, a0, a1, z, z_stride,
                                                                                   tiles_per_ray, rays_mode, n_points, groups,
                                                                                   reinterpret_cast<float4*>(raw), nullptr, nullptr);
        return nrf_launch_status();
    }
#endif
    if (acts)
        mlp_bf16x3_kernel<true><<<dim3(grid), 256, kXLdsBytes, stream>>>(static_cast<const char*>(packed), a0, a1, z, z_stride,
                                                                         tiles_per_ray, rays_mode, n_points, groups,
                                                                         reinterpret_cast<float4*>(raw), acts, masks);
    else
        mlp_bf16x3_kernel<false><<<dim3(grid), 256, kXLdsBytes, stream>>>(static_cast<const char*>(packed), a0, a1, z, z_stride,
                                                                          tiles_per_ray, rays_mode, n_points, groups,
                                                                          reinterpret_cast<float4*>(raw), nullptr, nullptr);
    return nrf_launch_status();
}

}  // namespace

// nrf_mlp_backward_bf16x3 (mlp_bwd.hip) calls this in place of mlp_dgrad_kernel
int launch_dgrad_x3(const float* packed, const void* packed_t, const float* grad_raw, const unsigned* masks, int64_t n_points,
                    float* dys, hipStream_t stream) {
    const int64_t groups = (n_points / 32 + 3) / 4;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (cus <= 0) cus = 256;
    static bool attr_set[64] = {};
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_dgrad_x3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                kXLdsBytes) != hipSuccess)
            return NRF_ERR_LAUNCH;
        attr_set[dev] = true;
    }
    mlp_dgrad_x3_kernel<<<dim3(unsigned(groups < cus ? groups : cus)), 256, kXLdsBytes, stream>>>(
        packed, static_cast<const char*>(packed_t), reinterpret_cast<const float4*>(grad_raw), masks, n_points, groups, dys);
    return NRF_OK;
}

NRF_EXPORT int64_t nrf_packed_bwd_bf16x3_bytes(void) { return int64_t(kXBwdSteps) * kXStepBytes; }

// The backward chain's stream in the order it is consumed (mlp_fp32.hip: nrf_pack_weights_transposed), three bf16 terms
// per weight: views^T (hidden columns only), feature^T, pts7^T .. pts1^T (pts5: hidden columns 63..318 only).
NRF_EXPORT int nrf_pack_weights_transposed_bf16x3(const float* const* w, void* packed_t, nrf_stream_t stream_) {
    if (!w || !packed_t) return NRF_ERR_NULL;
    for (int i = 0; i < NRF_NUM_LAYERS; ++i)
        if (!w[i]) return NRF_ERR_NULL;
    if (!nrf_aligned16(packed_t)) return NRF_ERR_ALIGN;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    unsigned short* base = static_cast<unsigned short*>(packed_t);
    nrf_launch_begin();
    int step = 0;
    XPackBatch batch;
    int n_batch = 0, max_total = 0;
    auto launch = [&](const float* W, int ldw, int in_blocks, int col0) {   // in_blocks = k-blocks = rows of W / 32
        batch.l[n_batch++] = XPackLayer{W, base + int64_t(step) * (kXStepBytes / 2), ldw, 8, 0, in_blocks, 0, 0, 0, 0, col0, 0, 2, 1};
        const int total = 8 * in_blocks * 2 * 512;
        max_total = total > max_total ? total : max_total;
        step += 8 * in_blocks * 2;
    };
    launch(w[10], 283, 4, 0);   // views^T: 128 outputs of views_linears.0 contracted, 256 feature columns out
    launch(w[8], 256, 8, 0);    // feature^T
    for (int l = 7; l >= 1; --l) launch(w[l], l == 5 ? 319 : 256, 8, l == 5 ? 63 : 0);
    xpack_layers_kernel<<<dim3((max_total + 255) / 256, n_batch), 256, 0, stream>>>(batch);
    return nrf_launch_status();
}


NRF_EXPORT int64_t nrf_packed_bf16x3_bytes(void) { return kXPackedBytes; }

NRF_EXPORT int nrf_pack_weights_bf16x3(const float* const* w, const float* const* b, void* packed, nrf_stream_t stream_) {
    if (!w || !b || !packed) return NRF_ERR_NULL;
    for (int i = 0; i < NRF_NUM_LAYERS; ++i)
        if (!w[i] || !b[i]) return NRF_ERR_NULL;
    if (!nrf_aligned16(packed)) return NRF_ERR_ALIGN;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    unsigned short* base = static_cast<unsigned short*>(packed);
    nrf_launch_begin();
    XPackBatch batch;
    int n_batch = 0, max_total = 0;
    auto launch = [&](XPackLayer p, int step0) {
        p.dst = base + int64_t(step0) * (kXStepBytes / 2);
        const int total = p.nob * (p.pre_kb + p.hid_kb + p.post_kb) * 2 * 512;
        batch.l[n_batch++] = p;
        max_total = total > max_total ? total : max_total;
    };
#ifdef X3_FWD_32X32
    xpack_forward_layers(w, launch);
    xpack_layers_kernel<<<dim3((max_total + 255) / 256, n_batch), 256, 0, stream>>>(batch);
    const PackSmall s = xpack_small(w, b, packed, kXSmallFloatOff);
    pack_small_kernel<<<(kSmallFloats + 255) / 256, 256, 0, stream>>>(s, 0, 0);
    n_batch = 0;
#endif
    // the layers as A fragments of the 16x16x32 form (split_stream.h: xpack_fetch, m16)
    base = reinterpret_cast<unsigned short*>(static_cast<char*>(packed) + kX16StreamOff);
    xpack_forward_layers(w, launch, 2, 1);
    xpack_layers_kernel<<<dim3((max_total + 255) / 256, n_batch), 256, 0, stream>>>(batch);
    PackSmall16 s16;
    for (int l = 0; l < 9; ++l) s16.bias[l] = b[l];
    s16.bias[9] = b[10];
    s16.alpha_w = w[9], s16.alpha_b = b[9], s16.rgb_w = w[11], s16.rgb_b = b[11];
    s16.dst = reinterpret_cast<float*>(static_cast<char*>(packed) + kX16SmallOff);
    pack_small16_kernel<<<(kS16Floats + 255) / 256, 256, 0, stream>>>(s16);
    return nrf_launch_status();
}

NRF_EXPORT int nrf_mlp_forward_rays_bf16x3(const void* packed, const float* rays_o, const float* rays_d, const float* z,
                                           int64_t z_stride, int64_t n_rays, int n_samples, float* raw, nrf_stream_t stream) {
    if (n_rays < 0 || n_samples <= 0 || n_samples % 32 != 0 || z_stride < 0) return NRF_ERR_SHAPE;
    if (z_stride != 0 && z_stride < n_samples) return NRF_ERR_SHAPE;
    if (n_rays == 0) return NRF_OK;
    if (!packed || !rays_o || !rays_d || !z || !raw) return NRF_ERR_NULL;
    if (!nrf_aligned16(packed) || !nrf_aligned16(raw)) return NRF_ERR_ALIGN;
    if (n_rays > (int64_t(1) << 40) / n_samples) return NRF_ERR_SHAPE;
    return launch_x3(packed, rays_o, rays_d, z, z_stride, n_samples / 32, 1, n_rays * n_samples, raw,
                     static_cast<hipStream_t>(stream));
}

NRF_EXPORT int nrf_mlp_forward_rays_train_bf16x3(const void* packed, const float* rays_o, const float* rays_d, const float* z,
                                                 int64_t z_stride, int64_t n_rays, int n_samples, float* raw, float* acts,
                                                 uint32_t* masks, nrf_stream_t stream) {
    if (n_rays < 0 || n_samples <= 0 || n_samples % 32 != 0 || z_stride < 0) return NRF_ERR_SHAPE;
    if (z_stride != 0 && z_stride < n_samples) return NRF_ERR_SHAPE;
    if (n_rays == 0) return NRF_OK;
    if (!packed || !rays_o || !rays_d || !z || !raw || !acts || !masks) return NRF_ERR_NULL;
    if (!nrf_aligned16(packed) || !nrf_aligned16(raw)) return NRF_ERR_ALIGN;
    if (n_rays > (int64_t(1) << 40) / n_samples) return NRF_ERR_SHAPE;
    return launch_x3(packed, rays_o, rays_d, z, z_stride, n_samples / 32, 1, n_rays * n_samples, raw,
                     static_cast<hipStream_t>(stream), acts, masks);
}

NRF_EXPORT int nrf_mlp_forward_points_bf16x3(const void* packed, const float* pts, const float* dirs, int64_t n_points,
                                             float* raw, nrf_stream_t stream) {
    if (n_points < 0 || n_points > (int64_t(1) << 40)) return NRF_ERR_SHAPE;
    if (n_points == 0) return NRF_OK;
    if (!packed || !pts || !dirs || !raw) return NRF_ERR_NULL;
    if (!nrf_aligned16(packed) || !nrf_aligned16(raw)) return NRF_ERR_ALIGN;
    return launch_x3(packed, pts, dirs, nullptr, 0, 1, 0, n_points, raw, static_cast<hipStream_t>(stream));
}
