// Device-side building blocks shared by the forward (mlp_fp32.hip) and backward (mlp_bwd.hip)
// register-chained MFMA kernels: the LDS weight ring, the chained GEMM layer, the encoding.
#pragma once
#include "nrf_common.h"

using namespace nrf;

#define NRF_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// The weight stream reaches the matrix cores through an LDS ring shared by the workgroup's 4
// waves: 16-step (16 KiB) chunks, filled by global_load_lds_dwordx4 (LDS-DMA, no VGPRs) kRing-1
// chunks ahead of use, one raw s_barrier per chunk and a COUNTED vmcnt so the prefetches stay in
// flight across it.  The kernel is persistent (one workgroup per CU) and the stream wraps, so
// the ring never drains between tiles.  Every layer is a whole number of chunks.
#ifndef NRF_CHUNK_STEPS
#define NRF_CHUNK_STEPS 16
#endif
#ifndef NRF_RING
#define NRF_RING 5
#endif
constexpr int kChunkSteps = NRF_CHUNK_STEPS;
constexpr int kChunkBytes = kChunkSteps * kStepFloats * 4;  // 16 KiB
constexpr int kRing = NRF_RING;
constexpr int kChunks = kStreamSteps / kChunkSteps;  // 145
static_assert(kStreamSteps % kChunkSteps == 0, "layers must be chunk aligned");
static_assert(kStepsL0 % kChunkSteps == 0 && kStepsHidden % kChunkSteps == 0 && kStepsSkip % kChunkSteps == 0 &&
              kStepsViews % kChunkSteps == 0, "layers must be chunk aligned");
constexpr int kLdsConstOff = kRing * kChunkBytes;              // biases + head weights live after the ring
constexpr int kLdsBytes = kLdsConstOff + kSmallFloats * 4;     // 80 KiB + 12.1 KiB

typedef __attribute__((address_space(3))) void* lds_vptr;
typedef const __attribute__((address_space(1))) void* glb_vptr;

struct WeightStream {
    const char* gnext;  // per lane: address of this lane's 16 B of piece 4w of the next chunk to prefetch
    const char* gbegin; // same for chunk 0
    const char* gend;   // same for chunk kChunks (one past the end: wrap point)
    unsigned lds_base;  // LDS byte address of smem + wave * 4 KiB (this wave fills pieces 4w..4w+3)
    char* smem;
    int pf_slot, rd_slot;
    const float4* rd;   // this lane's float4 of step 0 of the chunk being read
    float4 w;           // weights of the next step to be multiplied (already read from LDS)
};

// One LDS-DMA piece: 64 lanes x 16 B from per-lane global addresses to LDS[m0 .. m0 + 1 KiB).
// Inline asm on purpose: hipcc's waitcnt pass orders every ds_read behind ALL outstanding
// LDS-DMA it knows of (s_waitcnt vmcnt(0)), which would serialise the ring; issued from asm the
// loads are invisible to it and are ordered by ws_acquire's counted vmcnt + barrier instead.
__device__ __forceinline__ void glds16(const char* gsrc, unsigned lds_dst) {
#ifdef NRF_GLDS_SAVE_M0
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_dst)
        : "memory");
#else
    // M0 is not saved/restored: nothing else in this kernel uses it (checked in the .s: the only
    // m0 writes are these), and the restore would have to wait for the DMA to consume M0.
    asm volatile(
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, off"
        :
        : "v"(gsrc), "s"(lds_dst)
        : "memory");
#endif
}

// piece j (0..3) of this wave's share of the chunk being prefetched
__device__ __forceinline__ void ws_prefetch_piece(const WeightStream& st, int j) {
#if defined(NRF_ABLATE_NOSYNC) || defined(NRF_ABLATE_NOGLDS)  // diagnostic builds only (tools/ubench/mlp_ablate.hip): results are wrong
    return;
#endif
    glds16(st.gnext + j * 1024, st.lds_base + unsigned(st.pf_slot) * kChunkBytes + j * 1024);
}

__device__ __forceinline__ void ws_prefetch_advance(WeightStream& st) {
    const char* nx = st.gnext + kChunkBytes;
    st.gnext = nx == st.gend ? st.gbegin : nx;
    // keep the address a run-time value: the schedule repeats every tile, and LICM would otherwise
    // hoist all 145 chunk addresses out of the tile loop and spill them
    asm volatile("" : "+v"(st.gnext));
    st.pf_slot = st.pf_slot + 1 == kRing ? 0 : st.pf_slot + 1;
}

// Make the next chunk readable: my own 4 pieces of it have landed once at most 4*(kRing-2)
// younger VMEM ops are outstanding; the barrier then covers the other waves' pieces and also
// says every wave is done reading the previous chunk, whose slot the caller refills right away
// (ws_prefetch_piece x4 + ws_prefetch_advance, spread between MFMAs).
__device__ __forceinline__ void ws_acquire(WeightStream& st, int lane) {
#if !defined(NRF_ABLATE_NOSYNC) && !defined(NRF_ABLATE_NOBARRIER)
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)" ::"n"(4 * (kRing - 2)) : "memory");
    __builtin_amdgcn_s_barrier();
#endif
    st.rd_slot = st.rd_slot + 1 == kRing ? 0 : st.rd_slot + 1;
    st.rd = reinterpret_cast<const float4*>(st.smem + st.rd_slot * kChunkBytes) + lane;
}

__device__ __forceinline__ f32x16 relu16(f32x16 v) {
    // One v_max_i32 on the bit pattern: negative floats are negative ints (-> +0), non-negative ones order
    // like ints.  fmaxf() costs a second, canonicalising v_max; an inline-asm v_max_f32 is invisible to the
    // compiler's MFMA->VALU hazard pass and read stale registers right behind an MFMA chain.
    f32x16 r;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int b = __float_as_int(v[i]);
        r[i] = __int_as_float(b > 0 ? b : 0);
    }
    return r;
}

// One GEMM layer of the chain for this wave's 32 points: Y[ob] = init + sum_k W . [pre, X, post]
// where X = op(Y of the previous layer).  Y holds raw (pre-activation) outputs; the transformed
// copy X[i] = op(i, Y[i]) is made just in time while output block 0 is being accumulated (block i
// of Y is not overwritten before output block i starts, X[0] is made before Y[0] is
// re-initialised), so the VALU work runs under the matrix pipe and every layer has the same
// X -> Y orientation: identical layers share one unrolled body inside a run-time loop and the
// forward kernel (2112 unrolled MFMAs, ~48 KB of code) stays inside the 64 KB instruction cache.
// `op` is a functor: relu / copy in the plain forward, relu + save + sign mask in the training
// forward, mask + save in the backward (dgrad) chain.
enum { INIT_BIAS = 0, INIT_ACCUM = 1, INIT_ZERO = 2 };

struct OpCopy {
    __device__ __forceinline__ f32x16 operator()(int, f32x16 y) { return y; }
};
struct OpRelu {
    __device__ __forceinline__ f32x16 operator()(int, f32x16 y) {
#ifdef NRF_ABLATE_NOACT
        return y;
#else
        return relu16(y);
#endif
    }
};

template <int NOB, int NPRE, int NHID, int NPOST, int INIT, class InOp>
__device__ __forceinline__ void mfma_layer(WeightStream& st, int lane, const float* bias_h, const f32x16* pre,
                                           f32x16* X, const f32x16* post, f32x16* Y, InOp& op) {
    constexpr int NKB = NPRE + NHID + NPOST;
    constexpr int NS = NOB * NKB * 4;
    static_assert(NS % kChunkSteps == 0, "layer must be a whole number of chunks");
    // The weights of step s+1 are read from LDS before the MFMAs of step s are issued (the read
    // latency hides under 256 matrix-core cycles).  The ring hand-over to the next chunk -- also
    // the next layer's or next tile's first chunk -- happens ahead of a chunk's LAST step, and
    // the 4 LDS-DMA issues that refill the freed slot go one into each MFMA gap of that step.
    if (NHID > 0) X[0] = op(0, Y[0]);
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) {
        if (INIT == INIT_BIAS) Y[ob] = *reinterpret_cast<const f32x16*>(bias_h + ob * 32);
        if (INIT == INIT_ZERO) {
#pragma unroll
            for (int t = 0; t < 16; ++t) Y[ob][t] = 0.f;
        }
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            if (ob == 0 && kb > NPRE && kb < NPRE + NHID) X[kb - NPRE] = op(kb - NPRE, Y[kb - NPRE]);
            const f32x16 x = kb < NPRE ? pre[kb] : (kb < NPRE + NHID ? X[kb - NPRE] : post[kb - NPRE - NHID]);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int s = (ob * NKB + kb) * 4 + q;
                const bool hand = (s + 1) % kChunkSteps == 0;
                const float4 w = st.w;
                if (hand) ws_acquire(st, lane);
#ifndef NRF_ABLATE_NOLDS
                st.w = st.rd[((s + 1) % kChunkSteps) * 64];
#endif
                Y[ob] = NRF_MFMA(w.x, x[4 * q + 0], Y[ob]);
                if (hand) { __builtin_amdgcn_sched_barrier(0); ws_prefetch_piece(st, 0); __builtin_amdgcn_sched_barrier(0); }
                Y[ob] = NRF_MFMA(w.y, x[4 * q + 1], Y[ob]);
                if (hand) { __builtin_amdgcn_sched_barrier(0); ws_prefetch_piece(st, 1); __builtin_amdgcn_sched_barrier(0); }
                Y[ob] = NRF_MFMA(w.z, x[4 * q + 2], Y[ob]);
                if (hand) { __builtin_amdgcn_sched_barrier(0); ws_prefetch_piece(st, 2); __builtin_amdgcn_sched_barrier(0); }
                Y[ob] = NRF_MFMA(w.w, x[4 * q + 3], Y[ob]);
                if (hand) { __builtin_amdgcn_sched_barrier(0); ws_prefetch_piece(st, 3); ws_prefetch_advance(st); __builtin_amdgcn_sched_barrier(0); }
            }
        }
    }
}

// biases + head weights block ([layer][ob][h][16] accumulator order), shared by the fp32 and bf16 packers
struct PackSmall {
    const float* bias[10];  // pts0..7, feature, views
    const float* alpha_w;
    const float* alpha_b;
    const float* rgb_w;
    const float* rgb_b;
    float* packed;
    int64_t small_shift;  // floats; 0 for the fp32 layout
};

__attribute__((unused)) static __global__ void pack_small_kernel(PackSmall p, int64_t stream_pad_at, int pad_floats) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    float* dst = p.packed;
    if (idx < pad_floats) dst[stream_pad_at + idx] = 0.f;  // stream tail pad (fp32 layout only)
    if (idx >= kSmallFloats) return;
    const int64_t o = kOffBias + idx;
    dst += p.small_shift;  // where this layout keeps the small block relative to the fp32 layout's kOffBias
    if (o < kOffAlphaW) {  // [layer][ob][h][r]
        const int layer = idx >> 8, r = idx & 15, h = (idx >> 4) & 1, ob = (idx >> 5) & 7;
        dst[o] = p.bias[layer][ob * 32 + acc_row(r, h)];
    } else if (o < kOffAlphaB) {  // alpha weights [kb][h][t]
        const int i = int(o - kOffAlphaW), t = i & 15, h = (i >> 4) & 1, kb = i >> 5;
        dst[o] = p.alpha_w[32 * kb + acc_row(t, h)];
    } else if (o < kOffRgbW) {
        dst[o] = o == kOffAlphaB ? p.alpha_b[0] : 0.f;
    } else if (o < kOffRgbB) {  // rgb weights [c][kb][h][t]
        const int i = int(o - kOffRgbW), t = i & 15, h = (i >> 4) & 1, kb = (i >> 5) & 3, c = i >> 7;
        dst[o] = p.rgb_w[c * 128 + 32 * kb + acc_row(t, h)];
    } else {
        const int i = int(o - kOffRgbB);
        dst[o] = i < 3 ? p.rgb_b[i] : 0.f;
    }
}

// Start the ring: fill all kRing slots from the stream [gbegin, gend), then take chunk 0.
__device__ __forceinline__ void ws_start(WeightStream& st, const char* stream_base, int n_chunks, char* smem,
                                         int wave, int lane) {
    st.gbegin = stream_base + wave * 4096 + lane * 16;
    st.gend = st.gbegin + size_t(n_chunks) * kChunkBytes;
    st.gnext = st.gbegin;
    st.smem = smem;
    st.lds_base = __builtin_amdgcn_readfirstlane(
        unsigned(reinterpret_cast<uintptr_t>((lds_vptr)smem)) + unsigned(wave) * 4096u);
    st.pf_slot = 0;
#pragma unroll
    for (int i = 0; i < kRing; ++i) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) ws_prefetch_piece(st, jj);
        ws_prefetch_advance(st);
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (kRing - 1)) : "memory");
    __builtin_amdgcn_s_barrier();
    st.rd_slot = 0;
    st.rd = reinterpret_cast<const float4*>(smem) + lane;
    st.w = st.rd[0];
}

// ---- saved tensors of the training forward / backward (all [tile][row][32 points] fp32) ----
// acts rows: encoding (64, kernel slot order), h0..h7 (relu'd, 256 each), feature (256), dir
// encoding (32), hv (relu'd, 128).  dys rows: dY0..dY7, dY8 (= d feature), dYv (128), d raw (4).
constexpr int kActPe = 0, kActH = 64, kActFeat = 64 + 8 * 256, kActDpe = kActFeat + 256, kActHv = kActDpe + 32;
constexpr int kActRows = kActHv + 128;  // 2528
constexpr int kDyFeat = 8 * 256, kDyV = 9 * 256, kDyRaw = kDyV + 128;  // + d(raw) as 4 rows [r, g, b, sigma]
constexpr int kDyRows = kDyRaw + 32;    // 2464
constexpr int kMaskLayers = 9;          // h0..h7, hv: 128 sign bits per lane = 4 words

// A saved row (activation / dY) is written once and read once by another kernel, ~10 GB per pass: the stores carry the
// non-temporal hint (`nt`), so that the rows stream past the caches instead of pushing the weight stream's lines out.
// Measured (tools/debug/lib_ab_run.sh against NRF_ABL_TEMPORAL_ROWS builds, one box): fp16x2 step -4.3 %, bf16x3 -1.1 %.
__device__ __forceinline__ void save_row(float* p, float v) {
#ifdef NRF_ABL_TEMPORAL_ROWS
    *p = v;
#else
    __builtin_nontemporal_store(v, p);
#endif
}
// store a 32-row block held as (reg t, lane-half h) -> row 32*blk + acc_row(t, h); `base` already
// points at [tile][first row of the tensor][0] + (4*h*32 + point)
__device__ __forceinline__ void store_block(float* base, int blk, f32x16 v) {
#pragma unroll
    for (int t = 0; t < 16; ++t) save_row(base + (32 * blk + (t & 3) + 8 * (t >> 2)) * 32, v[t]);
}

__device__ __forceinline__ unsigned sign_bits16(f32x16 v) {
    unsigned m = 0;
#pragma unroll
    for (int t = 0; t < 16; ++t) m |= (v[t] > 0.f ? 1u : 0u) << t;
    return m;
}

// relu / copy + store the activated block + collect its sign bits (training forward)
template <bool RELU>
struct OpFwdSave {
    float* dst;      // acts + (tile * kActRows + first row) * 32 + 4 * h * 32 + point
    unsigned* mdst;  // masks + ((tile * kMaskLayers + layer) * 4) * 64 + lane, or nullptr
    unsigned cur;
    __device__ __forceinline__ f32x16 operator()(int i, f32x16 y) {
        const f32x16 x = RELU ? relu16(y) : y;
        store_block(dst, i, x);
        if (RELU) {
            const unsigned bits = sign_bits16(x);
            if (i & 1) mdst[(i >> 1) * 64] = cur | (bits << 16);
            else cur = bits;
        }
        return x;
    }
};

__device__ __forceinline__ f32x16 mask16(f32x16 v, unsigned bits) {
    f32x16 r;
#pragma unroll
    for (int t = 0; t < 16; ++t) r[t] = ((bits >> t) & 1u) ? v[t] : 0.f;
    return r;
}

// sin(a) for lane-half 0, cos(a) for lane-half 1, branch-free.  Cody-Waite reduction with a
// 3-term pi/2 (valid for |a| < 2^17, i.e. |x| < 256 at the 2^9 band; the scene lives in
// |x| <= far = 6) and degree-7/8 minimax polynomials: max abs error 9.2e-8 over |a| <= 2e4
// (tests/test_pe_emulation.py emulates it; torch's own sin is within 7e-8 of exact).
// Split in two so a caller can place the halves in different gaps of an MFMA chain (kilo.hip).
struct TrigReduced {
    float k, r, z, sp, cp;  // quadrant, reduced argument, its square, inner parts of the sin / cos polynomials
};
__device__ __forceinline__ TrigReduced trig_reduce(float a) {
    TrigReduced t;
    t.k = rintf(a * 0.63661977236758134308f);
    float r = fmaf(t.k, __uint_as_float(0xbfc90fdau), a);  // -1.570796251296997
    r = fmaf(t.k, __uint_as_float(0xb3a22168u), r);        // -7.549789415861596e-08
    r = fmaf(t.k, __uint_as_float(0xa7c234c4u), r);        // -5.390302529957765e-15
    t.r = r;
    const float z = r * r;
    t.z = z;
    t.sp = fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
    t.cp = fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f);
    return t;
}
__device__ __forceinline__ float trig_finish(TrigReduced t, int h) {
    const float r = t.r, z = t.z;
    const float sp = fmaf(t.sp * z, r, r);
    const float cp = fmaf(t.cp * z, z, fmaf(-0.5f, z, 1.0f));
    const int q = int(t.k) + h;  // cos(a) = sin(a + pi/2)
    const float v = (q & 1) ? cp : sp;
    return (q & 2) ? -v : v;
}
__device__ __forceinline__ float sin_or_cos(float a, int h) { return trig_finish(trig_reduce(a), h); }

// 16 encoding slots of one k-block: trig part for slots n < 3L, then the raw inputs.
// Slot n of lane-half h is sin (h = 0) or cos (h = 1) of argument n, so evaluated slot by slot both halves reduce
// the same argument and each throws one of the two polynomials away.  Instead the halves split the arguments: of a
// slot pair (2u, 2u+1) half 0 evaluates argument 2u and half 1 argument 2u+1, BOTH sin and cos, and one
// v_permlane32_swap hands each half what it needs ({S_lo, C_lo} -> slot 2u, {S_hi, C_hi} -> slot 2u+1).  Same
// polynomials, same bits, 40 % fewer VALU instructions.  Needs all 64 lanes active.
template <int L, int KB>
__device__ __forceinline__ f32x16 encode_block(float x, float y, float z, int h) {
    f32x16 r;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int n0 = KB * 16 + 2 * u, n1 = n0 + 1;
        if (n1 < 3 * L) {
#ifdef NRF_ABLATE_NOPE
            r[2 * u] = x + h, r[2 * u + 1] = y + h;
#else
            const float c0 = (n0 % 3) == 0 ? x : ((n0 % 3) == 1 ? y : z);
            const float c1 = (n1 % 3) == 0 ? x : ((n1 % 3) == 1 ? y : z);
            const float a = h ? c1 * float(1 << (n1 / 3)) : c0 * float(1 << (n0 / 3));
            const TrigReduced t = trig_reduce(a);
            const float sn = trig_finish(t, 0), cs = trig_finish(t, 1);
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(sn), __float_as_uint(cs), false, false);
            r[2 * u] = __uint_as_float(sw[0]);
            r[2 * u + 1] = __uint_as_float(sw[1]);
#endif
        } else {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int n = n0 + k;
                float v;
                if (n == 3 * L) v = h ? y : x;
                else if (n == 3 * L + 1) v = h ? 0.f : z;
                else v = 0.f;
                r[2 * u + k] = v;
            }
        }
    }
    return r;
}

// All k-blocks of an encoding at once from HALF the polynomial evaluations: only the even octaves go through
// trig_reduce / trig_finish, the odd ones follow by one double-angle step, sin 2a = 2 sin a cos a and
// cos 2a = 1 - 2 sin^2 a (3 VALU instead of ~27).  One step keeps the error at 1.8e-7 (sin) / 3.0e-7 (cos) max abs,
// 3.5e-8 / 6e-8 rms over |x| <= 6 at every octave (tests/test_pe_emulation.py restates both forms in numpy) against
// 9.3e-8 for the direct evaluation: below the fp32 accumulation noise of the networks that use it (the KiloNeRF
// cells, K <= 64) and far below a bf16 ulp.  The 8x256 fp32 chain keeps encode_block: the encoding is 2 % of its time.
// Exact argument e = 3g + c is coordinate c at octave 2g; the lane halves split the arguments as in encode_block.
// (V0, V1: the argument pairs [V0, V1) only -- a caller that spreads the encoding over several MFMA chains takes it in
// parts, the part with V0 == 0 also fills the raw-input slots; the default is all of it.)
template <int L, int V0 = 0, int V1 = (3 * (L / 2) + 1) / 2>
__device__ __forceinline__ void encode_pe_doubling(float x, float y, float z, int h, f32x16* out) {
    static_assert(L % 2 == 0, "octaves are taken in (even, odd) pairs");
    constexpr int NE = 3 * (L / 2);
    constexpr int NB = (3 * L + 2 + 15) / 16;
    static_assert(0 <= V0 && V0 < V1 && V1 <= (NE + 1) / 2, "argument pairs");
    if (V0 == 0) {
#pragma unroll
        for (int n = 3 * L; n < 16 * NB; ++n)
            out[n / 16][n % 16] = n == 3 * L ? (h ? y : x) : (n == 3 * L + 1 ? (h ? 0.f : z) : 0.f);
    }
#ifdef KILO_ABL_NOPE   // timing-only ablation: no trigonometry
#pragma unroll
    for (int n = 0; n < 3 * L; ++n) out[n / 16][n % 16] = x + float(n);
    return;
#endif
#pragma unroll
    for (int v = V0; v < V1; ++v) {
        const int e0 = 2 * v, e1 = 2 * v + 1;
        const int c0 = e0 % 3, g0 = e0 / 3, c1 = e1 % 3, g1 = e1 / 3;
        const float a0 = (c0 == 0 ? x : (c0 == 1 ? y : z)) * float(1 << (2 * g0));
        const float a1 = e1 < NE ? (c1 == 0 ? x : (c1 == 1 ? y : z)) * float(1 << (2 * g1)) : 0.f;
        const TrigReduced t = trig_reduce(h ? a1 : a0);
        const float sn = trig_finish(t, 0), cs = trig_finish(t, 1);
        const float d = sn + sn;
        const float s2 = d * cs, c2 = fmaf(-d, sn, 1.0f);
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(sn), __float_as_uint(cs), false, false);
        const auto sw2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(s2), __float_as_uint(c2), false, false);
        const int n0 = 6 * g0 + c0, n1 = 6 * g1 + c1;  // slot of (coordinate, octave 2g); octave 2g + 1 is 3 further
        out[n0 / 16][n0 % 16] = __uint_as_float(sw[0]);
        out[(n0 + 3) / 16][(n0 + 3) % 16] = __uint_as_float(sw2[0]);
        if (e1 < NE) {
            out[n1 / 16][n1 % 16] = __uint_as_float(sw[1]);
            out[(n1 + 3) / 16][(n1 + 3) % 16] = __uint_as_float(sw2[1]);
        }
    }
}

// X[i] = (Y[i] + dsig * w_alpha[i]) masked by the saved relu signs; stored as dY rows of the layer
struct OpBwdMask {
    float* dst;          // dys + (tile * kDyRows + first row) * 32 + 4 * h * 32 + point
    unsigned m[4];       // sign bits of the layer's forward output, 16 per 32-row block
    const float* aw;     // alpha weights [kb][h][t] in LDS (this lane-half's 16 at aw + kb*32)
    float dsig;          // d(raw sigma) when the layer output also fed alpha_linear, else 0
    __device__ __forceinline__ f32x16 operator()(int i, f32x16 y) {
        const f32x16 w = *reinterpret_cast<const f32x16*>(aw + i * 32);
        f32x16 v;
#pragma unroll
        for (int t = 0; t < 16; ++t) v[t] = fmaf(w[t], dsig, y[t]);
        const f32x16 x = mask16(v, (m[i >> 1] >> (16 * (i & 1))) & 0xffffu);
        store_block(dst, i, x);
        return x;
    }
};
