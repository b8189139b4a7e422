// Fused point construction + positional encoding + 8x256 NeRF MLP, fp32 on the CDNA4 matrix
// cores (v_mfma_f32_32x32x2_f32), plus the weight packer that feeds it.
//
// Replaces (reference paths): volume_renderer.py:227-231 (normalise, pts = o + z*d),
// :393-423 (_run_network chunk loop), network.py:171-195 (Network.forward, batchify),
// encoding/freq.py:31-32 (sin/cos encoding), network.py:49-74 (NeRF.forward).
//
// Design (MI355X-first, not a translation):
//   * Transposed GEMM chain  Y^T = W . X^T : the 32x32 MFMA's B operand is the activation
//     (k = feature, column = point), the A operand is the weight matrix.  The accumulator
//     layout of one layer (column = point on the lane, 16 rows in registers) is then exactly
//     the B-operand layout of the next layer, so activations never leave the register file:
//     no LDS, no barriers, no HBM round trips between the 10 GEMM layers.
//   * One wavefront = 32 points x all 256 features: 128 accumulator registers in, 128 out
//     (1 wave per SIMD, 512-register budget).
//   * Weights are pre-permuted (nrf_pack_weights) into the order the wave consumes them, so
//     the whole network is ONE linear 2.3 MiB stream of 1 KiB pieces per wavefront, read with
//     global_load_dwordx4 through an 8-deep register ring (L2-resident: every wave streams
//     the same bytes).  One float4 per lane feeds 4 MFMAs (256 matrix-core cycles).
//   * PE (63 ch) and dir PE (27 ch) are computed in registers in the k-order the packed
//     weights expect; alpha (256->1) and rgb (128->3) heads are VALU dot products.
#include "nrf_common.h"

using namespace nrf;

// Kernel PE slot n of lane-half h -> column of the reference encoding
// cat([x, sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(2^(L-1) x)]) (3 wide each), or -1.
// Half 0 holds the sin terms, half 1 the cos terms; slots 3L, 3L+1 hold the raw input.
__host__ __device__ static inline int pe_col(int n, int h, int L) {
    if (n < 3 * L) return 3 + 6 * (n / 3) + 3 * h + (n % 3);
    if (n == 3 * L) return h;               // x | y
    if (n == 3 * L + 1) return h ? -1 : 2;  // z | pad
    return -1;
}

// ------------------------------------------------------------------------------------
// weight packing
// ------------------------------------------------------------------------------------
struct PackLayer {
    const float* W;
    float* dst;
    int ldw, nob;
    int pre_kb, hid_kb, post_kb;  // K blocks of 32: PE before / hidden / PE after
    int pre_L, post_L;            // PE frequencies of the pre / post part
    int pre_col0, hid_col0, post_col0;  // first source column of each part
};

__global__ void pack_layer_kernel(PackLayer p) {
    const int nkb = p.pre_kb + p.hid_kb + p.post_kb;
    const int total = p.nob * nkb * 4 * kStepFloats;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int e = idx & 3, lane = (idx >> 2) & 63, step = idx >> 8;
    const int q = step & 3, kbpos = (step >> 2) % nkb, ob = (step >> 2) / nkb;
    const int h = lane >> 5, t = 4 * q + e;
    const int64_t row = int64_t(ob * 32 + (lane & 31)) * p.ldw;
    float v = 0.f;
    if (kbpos < p.pre_kb) {
        const int c = pe_col(kbpos * 16 + t, h, p.pre_L);
        if (c >= 0) v = p.W[row + p.pre_col0 + c];
    } else if (kbpos < p.pre_kb + p.hid_kb) {
        v = p.W[row + p.hid_col0 + 32 * (kbpos - p.pre_kb) + acc_row(t, h)];
    } else {
        const int c = pe_col((kbpos - p.pre_kb - p.hid_kb) * 16 + t, h, p.post_L);
        if (c >= 0) v = p.W[row + p.post_col0 + c];
    }
    p.dst[idx] = v;
}

struct PackSmall {
    const float* bias[10];  // pts0..7, feature, views
    const float* alpha_w;
    const float* alpha_b;
    const float* rgb_w;
    const float* rgb_b;
    float* packed;
};

__global__ void pack_small_kernel(PackSmall p) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    float* dst = p.packed;
    if (idx < kPrefetch * kStepFloats) dst[int64_t(kStreamSteps) * kStepFloats + idx] = 0.f;  // stream tail pad
    if (idx >= kSmallFloats) return;
    const int64_t o = kOffBias + idx;
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

NRF_EXPORT int64_t nrf_packed_floats(void) { return kPackedFloats; }

NRF_EXPORT int nrf_pack_weights(const float* const* w, const float* const* b, float* packed, nrf_stream_t stream_) {
    if (!w || !b || !packed) return NRF_ERR_NULL;
    for (int i = 0; i < NRF_NUM_LAYERS; ++i)
        if (!w[i] || !b[i]) return NRF_ERR_NULL;
    if (!nrf_aligned16(packed)) return NRF_ERR_ALIGN;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    nrf_launch_begin();
    auto launch = [&](PackLayer p, int layer_slot) {
        p.dst = packed + int64_t(kLayerStep[layer_slot]) * kStepFloats;
        const int total = p.nob * (p.pre_kb + p.hid_kb + p.post_kb) * 4 * kStepFloats;
        pack_layer_kernel<<<(total + 255) / 256, 256, 0, stream>>>(p);
    };
    // pts_linears.0: 63 PE -> 256
    launch(PackLayer{w[0], nullptr, 63, 8, 2, 0, 0, 10, 0, 0, 0, 0}, 0);
    for (int l = 1; l <= 7; ++l) {
        if (l == 5) {
            // network.py:58 cat([input_pts, h]): source columns 0..62 are the PE, 63..318 the hidden.  Streamed as
            // a plain 256->256 layer (so it shares the hidden-layer code) followed by a PE pass that accumulates.
            launch(PackLayer{w[5], nullptr, 319, 8, 0, 8, 0, 0, 0, 0, 63, 0}, 5);
            PackLayer pe_part{w[5], packed + int64_t(kLayerStep[5] + kStepsHidden) * kStepFloats, 319, 8, 2, 0, 0, 10, 0, 0, 0, 0};
            pack_layer_kernel<<<(kStepsL0 * kStepFloats + 255) / 256, 256, 0, stream>>>(pe_part);
        } else
            launch(PackLayer{w[l], nullptr, 256, 8, 0, 8, 0, 0, 0, 0, 0, 0}, l);
    }
    launch(PackLayer{w[8], nullptr, 256, 8, 0, 8, 0, 0, 0, 0, 0, 0}, 8);  // feature_linear
    // views_linears.0: network.py:63 cat([feature, input_views]): 256 hidden then 27 dir PE -> 128
    launch(PackLayer{w[10], nullptr, 283, 4, 0, 8, 1, 0, 4, 0, 0, 256}, 9);
    PackSmall s;
    for (int l = 0; l < 8; ++l) s.bias[l] = b[l];
    s.bias[8] = b[8];
    s.bias[9] = b[10];
    s.alpha_w = w[9];
    s.alpha_b = b[9];
    s.rgb_w = w[11];
    s.rgb_b = b[11];
    s.packed = packed;
    pack_small_kernel<<<(kSmallFloats + 255) / 256, 256, 0, stream>>>(s);
    return nrf_launch_status();
}

// ------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------
#define NRF_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

namespace {

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
    f32x16 r;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        float t;  // one v_max_f32 (fmaxf() costs a second, canonicalising v_max for signalling NaNs)
        asm("v_max_f32 %0, 0, %1" : "=v"(t) : "v"(v[i]));
        r[i] = t;
    }
    return r;
}

// One GEMM layer of the chain for this wave's 32 points: Y[ob] = bias + sum_k W . [pre, X, post]
// where X = act(Y of the previous layer).  Y holds raw (pre-activation) outputs; the activated
// copy X[i] = act(Y[i]) is made just in time while output block 0 is being accumulated (block i
// of Y is not overwritten before output block i starts, X[0] is made before Y[0] is
// re-initialised), so the VALU work runs under the matrix pipe and every layer has the same
// X -> Y orientation: identical layers share one unrolled body inside a run-time loop and the
// whole kernel (3136 unrolled MFMAs, ~45 KB of code) stays inside the 64 KB instruction cache.
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_COPY = 2 };
template <int NOB, int NPRE, int NHID, int NPOST, int ACT, bool ACCUM = false>
__device__ __forceinline__ void mfma_layer(WeightStream& st, int lane, const float* bias_h, const f32x16* pre,
                                           f32x16* X, const f32x16* post, f32x16* Y) {
    constexpr int NKB = NPRE + NHID + NPOST;
    constexpr int NS = NOB * NKB * 4;
    static_assert(NS % kChunkSteps == 0, "layer must be a whole number of chunks");
    // The weights of step s+1 are read from LDS before the MFMAs of step s are issued (the read
    // latency hides under 256 matrix-core cycles).  The ring hand-over to the next chunk -- also
    // the next layer's or next tile's first chunk -- happens ahead of a chunk's LAST step, and
    // the 4 LDS-DMA issues that refill the freed slot go one into each MFMA gap of that step.
#ifdef NRF_ABLATE_NOACT
#define NRF_ACT(v) (v)
#else
#define NRF_ACT(v) (ACT == ACT_RELU ? relu16(v) : (v))
#endif
    if (NHID > 0) X[0] = NRF_ACT(Y[0]);
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) {
        if (!ACCUM) Y[ob] = *reinterpret_cast<const f32x16*>(bias_h + ob * 32);
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            if (ob == 0 && kb > NPRE && kb < NPRE + NHID) X[kb - NPRE] = NRF_ACT(Y[kb - NPRE]);
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

// sin(a) for lane-half 0, cos(a) for lane-half 1, branch-free.  Cody-Waite reduction with a
// 3-term pi/2 (valid for |a| < 2^17, i.e. |x| < 256 at the 2^9 band; the scene lives in
// |x| <= far = 6) and degree-7/8 minimax polynomials: max abs error 9.2e-8 over |a| <= 2e4
// (tests/test_host_logic.py emulates it; torch's own sin is within 7e-8 of exact).
__device__ __forceinline__ float sin_or_cos(float a, int h) {
    const float k = rintf(a * 0.63661977236758134308f);
    float r = fmaf(k, __uint_as_float(0xbfc90fdau), a);  // -1.570796251296997
    r = fmaf(k, __uint_as_float(0xb3a22168u), r);        // -7.549789415861596e-08
    r = fmaf(k, __uint_as_float(0xa7c234c4u), r);        // -5.390302529957765e-15
    const float z = r * r;
    float sp = fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
    sp = fmaf(sp * z, r, r);
    float cp = fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f);
    cp = fmaf(cp * z, z, fmaf(-0.5f, z, 1.0f));
    const int q = int(k) + h;  // cos(a) = sin(a + pi/2)
    const float v = (q & 1) ? cp : sp;
    return (q & 2) ? -v : v;
}

// 16 encoding slots of one k-block: trig part for slots n < 3L, then the raw inputs.
template <int L, int KB>
__device__ __forceinline__ f32x16 encode_block(float x, float y, float z, int h) {
    f32x16 r;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const int n = KB * 16 + t;
        float v;
        if (n < 3 * L) {
            const float c = (n % 3) == 0 ? x : ((n % 3) == 1 ? y : z);
#ifdef NRF_ABLATE_NOPE
            v = c * float(1 << (n / 3)) + h;
#else
            v = sin_or_cos(c * float(1 << (n / 3)), h);
#endif
        } else if (n == 3 * L) {
            v = h ? y : x;
        } else if (n == 3 * L + 1) {
            v = h ? 0.f : z;
        } else {
            v = 0.f;
        }
        r[t] = v;
    }
    return r;
}

// Persistent: gridDim.x <= number of CUs, one 4-wave workgroup per CU, one wave per SIMD.
// Tile group tg = 4 consecutive 32-point tiles, one per wave; all waves run every group (the
// barriers of the weight ring are workgroup-wide), waves past the end just do not store.
__global__ __launch_bounds__(256, 1) void mlp_fwd_kernel(const float* __restrict__ packed,
                                                         const float* __restrict__ a0,  // rays_o | pts
                                                         const float* __restrict__ a1,  // rays_d | dirs
                                                         const float* __restrict__ zbuf, int64_t z_stride,
                                                         int tiles_per_ray, int rays_mode, int64_t n_points,
                                                         int64_t n_groups, float4* __restrict__ raw) {
    __shared__ __attribute__((aligned(16))) char smem[kLdsBytes];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;

    // biases and head weights: global -> LDS once per workgroup (plain loads, before any LDS-DMA)
    {
        const float4* src = reinterpret_cast<const float4*>(packed + kOffBias);
        float4* dst = reinterpret_cast<float4*>(smem + kLdsConstOff);
        for (int i = threadIdx.x; i < kSmallFloats / 4; i += 256) dst[i] = src[i];
    }
    __syncthreads();
    const float* cst = reinterpret_cast<const float*>(smem + kLdsConstOff);  // same offsets as packed + kOffBias
    const float* bias_h = cst + h * 16;                                      // [layer][ob][h][16]

    WeightStream st;
    st.gbegin = reinterpret_cast<const char*>(packed) + wave * 4096 + lane * 16;
    st.gend = st.gbegin + size_t(kChunks) * kChunkBytes;
    st.gnext = st.gbegin;
    st.smem = smem;
    st.lds_base = __builtin_amdgcn_readfirstlane(
        unsigned(reinterpret_cast<uintptr_t>((lds_vptr)smem)) + unsigned(wave) * 4096u);
    st.pf_slot = 0;
    st.rd_slot = kRing - 1;
#pragma unroll
    for (int i = 0; i < kRing; ++i) {  // fill the whole ring, then take chunk 0
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) ws_prefetch_piece(st, jj);
        ws_prefetch_advance(st);
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (kRing - 1)) : "memory");
    __builtin_amdgcn_s_barrier();
    st.rd_slot = 0;
    st.rd = reinterpret_cast<const float4*>(smem) + lane;
    st.w = st.rd[0];

    for (int64_t tg = blockIdx.x; tg < n_groups; tg += gridDim.x) {
        const int64_t tile = tg * 4 + wave;
        int64_t pidx = tile * 32 + j;
        const bool valid = pidx < n_points;
        if (!valid) pidx = n_points - 1;
        float px, py, pz, dx, dy, dz;
        if (rays_mode) {
            const int64_t tile_c = pidx / 32;  // clamped tile for waves past the end
            const int64_t ray = tile_c / tiles_per_ray;
            const int s0 = int(tile_c - ray * tiles_per_ray) * 32;
            const float ox = a0[ray * 3 + 0], oy = a0[ray * 3 + 1], oz = a0[ray * 3 + 2];
            dx = a1[ray * 3 + 0], dy = a1[ray * 3 + 1], dz = a1[ray * 3 + 2];
            // volume_renderer.py:230; torch-CPU's norm over 3 elements is this fma chain (bit-exact dhat
            // matters: a 1-ulp change of the point moves the 2^9-band encoding by ~2e-4)
            const float nrm = sqrtf(fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
            dx = dx / nrm, dy = dy / nrm, dz = dz / nrm;
            const float zz = zbuf[ray * z_stride + s0 + (valid ? j : int(pidx & 31))];
            px = __fadd_rn(ox, __fmul_rn(zz, dx));  // :231, separate mul and add like torch
            py = __fadd_rn(oy, __fmul_rn(zz, dy));
            pz = __fadd_rn(oz, __fmul_rn(zz, dz));
        } else {
            px = a0[pidx * 3 + 0], py = a0[pidx * 3 + 1], pz = a0[pidx * 3 + 2];
            dx = a1[pidx * 3 + 0], dy = a1[pidx * 3 + 1], dz = a1[pidx * 3 + 2];
        }

        f32x16 pe[2], dpe[1];
        pe[0] = encode_block<10, 0>(px, py, pz, h);
        pe[1] = encode_block<10, 1>(px, py, pz, h);
        dpe[0] = encode_block<4, 0>(dx, dy, dz, h);

        f32x16 X[8], Y[8];
        mfma_layer<8, 2, 0, 0, ACT_NONE>(st, lane, bias_h, pe, X, nullptr, Y);               // pts_linears.0
        for (int li = 1; li <= 8; ++li) {                                                     // pts 1..7, feature
            mfma_layer<8, 0, 8, 0, ACT_RELU>(st, lane, bias_h + li * 256, nullptr, X, nullptr, Y);
            if (li == 5)  // skip: cat([pe, h]) (network.py:58) = the hidden part above + this PE pass on top
                mfma_layer<8, 2, 0, 0, ACT_NONE, true>(st, lane, nullptr, pe, X, nullptr, Y);
        }
        // X = relu(h7) (made while feature_linear ran): alpha_linear as a VALU dot over this half's 128 features
        float sigma = 0.f;
        {
            const float* aw = cst + (kOffAlphaW - kOffBias) + h * 16;
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) {
                const f32x16 w = *reinterpret_cast<const f32x16*>(aw + kb * 32);
#pragma unroll
                for (int t = 0; t < 16; ++t) sigma = fmaf(w[t], X[kb][t], sigma);
            }
        }
        // views_linears.0 on cat([feature (no activation), dir PE]) -> 128
        mfma_layer<4, 0, 8, 1, ACT_COPY>(st, lane, bias_h + 9 * 256, nullptr, X, dpe, Y);
        float r = 0.f, g = 0.f, b = 0.f;
        {
            const float* rw = cst + (kOffRgbW - kOffBias) + h * 16;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                const f32x16 hv = relu16(Y[kb]);
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
        // combine the two lane halves (each holds half of every dot product)
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

int launch_mlp(const float* packed, const float* a0, const float* a1, const float* z, int64_t z_stride,
               int tiles_per_ray, int rays_mode, int64_t n_points, float* raw, hipStream_t stream) {
    const int64_t tiles = (n_points + 31) / 32;
    const int64_t groups = (tiles + 3) / 4;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (cus <= 0) cus = 256;
    const unsigned grid = unsigned(groups < cus ? groups : cus);
    nrf_launch_begin();
    mlp_fwd_kernel<<<dim3(grid), 256, 0, stream>>>(packed, a0, a1, z, z_stride, tiles_per_ray, rays_mode, n_points,
                                                   groups, reinterpret_cast<float4*>(raw));
    return nrf_launch_status();
}

int check_common(const float* packed, const void* a, const void* b, const void* raw) {
    if (!packed || !a || !b || !raw) return NRF_ERR_NULL;
    if (!nrf_aligned16(packed) || !nrf_aligned16(raw)) return NRF_ERR_ALIGN;
    return NRF_OK;
}

}  // namespace

NRF_EXPORT int nrf_mlp_forward_rays(const float* packed, const float* rays_o, const float* rays_d, const float* z,
                                    int64_t z_stride, int64_t n_rays, int n_samples, float* raw,
                                    nrf_stream_t stream) {
    if (n_rays < 0 || n_samples <= 0 || n_samples % 32 != 0 || z_stride < 0) return NRF_ERR_SHAPE;
    if (z_stride != 0 && z_stride < n_samples) return NRF_ERR_SHAPE;
    if (n_rays == 0) return NRF_OK;  // empty batch: nothing to read or write, pointers may be NULL
    if (int st = check_common(packed, rays_o, rays_d, raw)) return st;
    if (!z) return NRF_ERR_NULL;
    if (n_rays > (int64_t(1) << 40) / n_samples) return NRF_ERR_SHAPE;
    return launch_mlp(packed, rays_o, rays_d, z, z_stride, n_samples / 32, 1, n_rays * n_samples, raw,
                      static_cast<hipStream_t>(stream));
}

NRF_EXPORT int nrf_mlp_forward_points(const float* packed, const float* pts, const float* dirs, int64_t n_points,
                                      float* raw, nrf_stream_t stream) {
    if (n_points < 0 || n_points > (int64_t(1) << 40)) return NRF_ERR_SHAPE;
    if (n_points == 0) return NRF_OK;
    if (int st = check_common(packed, pts, dirs, raw)) return st;
    return launch_mlp(packed, pts, dirs, nullptr, 0, 1, 0, n_points, raw, static_cast<hipStream_t>(stream));
}
