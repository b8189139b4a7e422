// bf16 variant of the fused point construction + positional encoding + 8x256 NeRF MLP
// (BASELINE configs[4]: "bf16 MLP"; same reference path as mlp_fp32.hip: volume_renderer.py:227-231, 393-423,
// network.py:49-74, 171-195, encoding/freq.py:31-32).
//
// Numerics (the stated bf16 rule, restated in oracle/nerf_oracle.py:network_forward(bf16=True)):
//   * matrix weights and every layer INPUT (encodings, relu'd hidden activations, the feature vector) are rounded
//     to bf16 (round-to-nearest-even); products accumulate in fp32 on v_mfma_f32_32x32x16_bf16; biases are fp32
//     accumulator initialisers; point construction, sin/cos and the rgb head (fp32 weights, VALU) are fp32;
//   * the alpha head (256 -> 1) runs on the matrix pipe too, but with its weight NOT rounded: it travels as two bf16
//     terms, w_hi = bf16(w) and w_lo = bf16(w - w_hi) (16 significand bits, 2^-17 relative), in rows 0 and 1 of one
//     otherwise empty 32-row block behind feature_linear, its bias as that block's accumulator initialiser, and
//     sigma = row 0 + row 1.  As a VALU dot product over the packed bf16 activations it was ~770 instructions per
//     64-point tile that nothing overlapped (5.9 k of a tile's ~130 k cycles); as 16 k16-steps it is 1.35 % more MFMAs.
//
// Shape (what differs from the fp32 kernel): one MFMA now covers 16 k for 32 cycles (16x the fp32 rate), so the
// weight stream and everything that is not an MFMA, not the matrix pipe, are the things to manage.  Activations live
// in registers as bf16 B fragments (v_cvt_pk_bf16_f32 of accumulator registers 8s..8s+7 = the fragment of k-step s,
// cdna_hip_programming.md section 3): a finished 32-row output block is activated and packed while the next block
// accumulates, so only two fp32 accumulator blocks per point set are ever live and the layer's input and output
// ping-pong between two register arrays.  The LDS ring moves 32-KiB chunks (one barrier per 2048 matrix cycles per
// SIMD) and its LDS-DMA refills are spread over the chunk's steps.
//
// Matrix shape (round 3): v_mfma_f32_16x16x32_bf16 (m16.h) -- a step's 1-KiB weight fragment is 16 rows x 32 k and meets
// four 16-point accumulators (2 sets x 2 halves); everything else (stream positions, epilogue slices, ring) as it was on
// 32x32x16.  The chip holds a higher clock on this shape.
//
// Tiling (template NSETS = 2): four waves per workgroup, one per SIMD, each owning two 32-point sets that share every
// weight operand (one ds_read_b128 per 2 MFMAs); a workgroup covers 256 points per pass of the weight stream.
// NSETS = 1 -- eight waves, two per SIMD, one set each, everything in the 256 architectural VGPRs, no VGPR<->AGPR
// copies -- was built and measured in round 2: 8 % fewer cycles, but twice the LDS read bytes per MFMA, and the chip
// answered with a 9 % lower clock (1.72 vs 1.89 GHz, GRBM_GUI_ACTIVE): 10.99 vs 10.95 ms per 65,536-ray fine pass.  The
// template parameter stays (the code is generic in it); only NSETS = 2 is instantiated.
#include "m16.h"

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int kBChunkSteps = 32;                   // 1-KiB steps per ring chunk
constexpr int kBChunkBytes = kBChunkSteps * 1024;  // 32 KiB
constexpr int kBRing = 4;
#ifndef BF_LOOK
#define BF_LOOK 3
#endif
constexpr int kLook = BF_LOOK;                     // LDS -> register look-ahead of the weight operands, in steps
constexpr int kWQ = kLook < 4 ? 4 : 8;             // register ring of those operands (power of two > kLook)
constexpr int kBStepsL0 = 8 * 2 * 2;               // 8 out blocks x 2 PE k-blocks x 2 k16-steps
constexpr int kBStepsHidden = 8 * 8 * 2;           // 128
constexpr int kBStepsSkip = 8 * 10 * 2;            // 160: per output block 2 PE k-blocks then 8 hidden
constexpr int kBStepsViewsPad = 96;                // views: 4 x 9 x 2 = 72 used steps of 3 chunks
constexpr int kBStepsAlphaPad = 32;                // alpha_linear as one 32-row block: 8 x 2 = 16 used steps of 1 chunk
constexpr int kBStreamSteps = kBStepsL0 + 7 * kBStepsHidden + kBStepsSkip + kBStepsAlphaPad + kBStepsViewsPad;  // 1216
constexpr int kBChunks = kBStreamSteps / kBChunkSteps;                                         // 38
static_assert(kBStreamSteps % kBChunkSteps == 0, "stream must be whole chunks");
// stream offsets (steps): pts0, pts1..4, pts5 (skip), pts6, pts7, feature, views; the alpha block sits between the last two
constexpr int kBLayerStep[10] = {0, 32, 160, 288, 416, 544, 704, 832, 960, 1120};
constexpr int kBAlphaStep = 1088;
constexpr int64_t kBSmallFloatOff = int64_t(kBStreamSteps) * 256;  // small block (fp32) behind the stream
constexpr int kBLdsConstOff = kBRing * kBChunkBytes;               // 128 KiB
constexpr int kBLdsBytes = kBLdsConstOff + kS16Floats * 4;         // + 21.6 KiB (m16.h: the small block of the 16x16x32 form)
static_assert(kBLdsBytes <= 160 * 1024, "LDS budget");

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    const bf16x2 v = {(__bf16)lo, (__bf16)hi};  // v_cvt_pk_bf16_f32, round to nearest even
    return __builtin_bit_cast(unsigned, v);
}
// registers 8s..8s+7 of a 32x32 accumulator (or encoding) tile -> the bf16 B fragment of k16-step s
__device__ __forceinline__ void pack_tile(const f32x16& v, u32x4 out[2]) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i) out[s][i] = pack_bf16(v[8 * s + 2 * i], v[8 * s + 2 * i + 1]);
}

// ---- weight packing ------------------------------------------------------------------------
struct BPackLayer {
    const float* W;
    unsigned short* dst;
    int ldw, nob;
    int pre_kb, hid_kb, post_kb;
    int pre_L, post_L;
    int pre_col0, hid_col0, post_col0;
    int split_row;   // != 0: W is ONE row (alpha_linear); block row 0 = bf16(w), row 1 = bf16(w - row 0), rows 2..31 = 0
};

__device__ __forceinline__ unsigned short bf16_bits(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }

__global__ void bpack_layer_kernel(BPackLayer p) {
    const int nkb = p.pre_kb + p.hid_kb + p.post_kb;
    const int total = p.nob * nkb * 2 * 512;  // bf16 elements
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int jj = idx & 7, lane = (idx >> 3) & 63, step = idx >> 9;
    const int s2 = step & 1, kbpos = (step >> 1) % nkb, ob = (step >> 1) / nkb;
    // A fragment of the 16x16x32 form (split_stream.h: xpack_fetch, m16): rows 32 ob + 16 s2 + (lane & 15), slot phi of
    // the k-block = register t, lane half h of the 32x32 form the encoding blocks are made in
    const int phi = m16_slot(lane >> 4, jj);
    const int h = (phi >> 2) & 1, t = 4 * (phi >> 3) + (phi & 3);
    const int brow = 16 * s2 + (lane & 15);
    const int64_t row = int64_t(ob * 32 + brow) * p.ldw;
    float v = 0.f;
    if (kbpos < p.pre_kb) {
        const int c = pe_col(kbpos * 16 + t, h, p.pre_L);
        if (c >= 0) v = p.W[row + p.pre_col0 + c];
    } else if (kbpos < p.pre_kb + p.hid_kb) {
        if (p.split_row) {
            const int r = brow;
            const float wv = p.W[p.hid_col0 + 32 * (kbpos - p.pre_kb) + acc_row(t, h)];
            v = r == 0 ? wv : (r == 1 ? __fsub_rn(wv, float((__bf16)wv)) : 0.f);
        } else {
            v = p.W[row + p.hid_col0 + 32 * (kbpos - p.pre_kb) + acc_row(t, h)];
        }
    } else {
        const int c = pe_col((kbpos - p.pre_kb - p.hid_kb) * 16 + t, h, p.post_L);
        if (c >= 0) v = p.W[row + p.post_col0 + c];
    }
    p.dst[idx] = bf16_bits(v);
}

// ---- LDS ring of the bf16 stream --------------------------------------------------------------
struct BRing {
    // the stream is addressed as a UNIFORM base that walks the chunks (scalar registers) + one 32-bit lane offset
    // (global_load_lds_dwordx4 voff, s[base]; mlp_bf16x3.hip: XRingS): no per-lane 64-bit pointers, no vector address arithmetic
    const char* snext;   // this wave's piece 0 of the next chunk to prefetch
    const char* sbegin;
    const char* send;
    unsigned voff;       // lane * 16
    unsigned lds_base;   // LDS byte address of smem + wave * PW KiB (this wave fills pieces PW*w .. PW*w+PW-1)
    char* smem;
    int pf_slot, rd_slot;
    const u32x4* rd;     // this lane's 16 B of step 0 of the chunk being read
    u32x4 wq[kWQ];       // weights of steps s .. s+kLook-1 (slot = step % kWQ), read kLook steps ahead of their MFMAs
#ifdef BF_STAMP
    int stamp_base;      // >= 0: bf_layer stamps the start of every output block at g_bf_stamps[.. + stamp_base + ob]
    int stamp_wave;
#endif
};

__device__ __forceinline__ void br_prefetch_piece(const BRing& st, int j) {
#ifdef BF_ABL_NODMA  // timing-only ablation builds (tools/debug/bf16_ablate.py): results are wrong
    return;
#endif
    // readfirstlane: the slot index is uniform, but under SGPR pressure the compiler may keep it in a VGPR, and the
    // asm's M0 operand must be a scalar register (folds away when it already is one)
    asm volatile(
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, %1"
        :
        : "v"(st.voff), "s"(st.snext + j * 1024),
          "s"(__builtin_amdgcn_readfirstlane(st.lds_base + unsigned(st.pf_slot) * kBChunkBytes + j * 1024))
        : "memory");
}
__device__ __forceinline__ void br_prefetch_advance(BRing& st) {
    const char* nx = st.snext + kBChunkBytes;
    st.snext = nx == st.send ? st.sbegin : nx;
    asm volatile("" : "+s"(st.snext));  // keep it a run-time value (see ws_prefetch_advance)
    st.pf_slot = st.pf_slot + 1 == kBRing ? 0 : st.pf_slot + 1;
}
// Next chunk readable: it was issued two chunk-times ago; the two batches of PW pieces (PW = 32 / waves: this wave's
// share of a chunk) issued since may stay in flight.  The barrier also says every wave is done reading the current
// chunk (its slot is refilled next).
template <int PW>
__device__ __forceinline__ void br_acquire(BRing& st, int lane) {
#ifndef BF_ABL_NOBARRIER
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)" ::"n"(2 * PW) : "memory");   // (dropping the lgkmcnt(0): no change, A/B in one process)
    __builtin_amdgcn_s_barrier();
#endif
    st.rd_slot = st.rd_slot + 1 == kBRing ? 0 : st.rd_slot + 1;
    st.rd = reinterpret_cast<const u32x4*>(st.smem + st.rd_slot * kBChunkBytes) + lane;
}
// Chunks 0..2 are fetched up front; while chunk c is consumed the wave issues its PW pieces of chunk c+3 into the
// slot chunk c-1 just left.
template <int PW>
__device__ __forceinline__ void br_start(BRing& st, const char* stream_base, char* smem, int wave, int lane) {
    st.sbegin = stream_base + wave * (PW * 1024);
    st.send = st.sbegin + size_t(kBChunks) * kBChunkBytes;
    st.snext = st.sbegin;
    st.voff = unsigned(lane) * 16u;
    st.smem = smem;
    st.lds_base = __builtin_amdgcn_readfirstlane(unsigned(reinterpret_cast<uintptr_t>((lds_vptr)smem)) + unsigned(wave) * unsigned(PW * 1024));
    st.pf_slot = 0;
#pragma unroll
    for (int i = 0; i < kBRing - 1; ++i) {
#pragma unroll
        for (int jj = 0; jj < PW; ++jj) br_prefetch_piece(st, jj);
        br_prefetch_advance(st);
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PW) : "memory");
    __builtin_amdgcn_s_barrier();
    st.rd_slot = 0;
    st.rd = reinterpret_cast<const u32x4*>(smem) + lane;
#pragma unroll
    for (int i = 0; i < kLook; ++i) st.wq[i] = st.rd[i * 64];
}

// Diagnostic builds only (tools/ubench/bf16_phase.hip, -DBF_STAMP): s_memtime at phase boundaries of one tile per wave,
// written to a buffer of their own that nothing else reads.  In the shipped kernel the macro is empty.
#ifdef BF_STAMP
__device__ unsigned long long g_bf_stamps[256 * 8 * 128];
#define BF_STAMP_AT(i)                                                                                       \
    do {                                                                                                     \
        if (stamp_tile) {                                                                                    \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                      \
            if (lane == 0) g_bf_stamps[st.stamp_wave * 128 + (i)] = t_;                             \
        }                                                                                                    \
    } while (0)
#else
#define BF_STAMP_AT(i) do {} while (0)
#endif

#define NRF_MFMA_BF16(a, b, c) \
    __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, (a)), __builtin_bit_cast(bf16x8, (b)), (c), 0, 0, 0)

// Activation of a finished accumulator block, on the PACKED bf16 pair: v_pk_max_i16 against 0 is relu (negative bf16 are
// negative int16) and against 0x8000 the identity, so "relu or not" is a run-time scalar, not another code body.  (On the
// 32x32x16 kernel this form measured 2 % SLOWER than two v_max_i32 on the fp32 bits ahead of the conversion -- the
// activation waits for the conversion -- and was not used; on the 16x16x32 kernel, where the clock follows the energy a tile
// costs, the 768 instructions it saves per 64-point tile are -2.7 %: 130.6 -> 127.0 ms per frame, old and new alternately
// on one box.)
constexpr unsigned kPkRelu = 0u, kPkNone = 0x80008000u;

// Block epilogues come in SLICES.  A finished 32-row block needs 3 VALU instructions per packed output register
// (two v_max_i32 on the fp32 bits, one v_cvt_pk_bf16_f32), 48 or 96 per block; issued as one lump they stop the wave's
// in-order MFMA issue for hundreds of cycles (phase stamps, tools/ubench/bf16_phase.hip: 1340-1550 cycles per block of
// 1024 MFMA cycles).  Two or three VALU instructions per MFMA gap hide completely (tools/ubench/mfma_valu_overlap.hip:
// up to 6 per gap are free), so bf_layer hands every step of block ob its share of block ob-1's epilogue, and the LAST
// block's epilogue of a layer travels into the first block of the next layer (`pending`), where its output -- input
// block 7 -- is not read before step 14.
template <int NSETS>
struct EpiPack {  // hidden layers: activate + pack block ob of every set into the next layer's input
    static constexpr int Q = 8 * NSETS;   // slices per block: one packed output register each
    u32x4 (*Xout)[8][2];
    unsigned lim16;                       // kPkRelu / kPkNone: the activation's limit for a packed bf16 pair
    __device__ __forceinline__ void begin(int) {}
    __device__ __forceinline__ void slice(int ob, const f32x16 (&y)[NSETS], int q) {
        const int set = q / 8, s = (q % 8) / 4, i = q % 4;
        // relu AFTER the conversion, on the packed pair: one v_pk_max_i16 against 0 (or 0x8000: no activation) per half --
        // a negative bf16 is a negative int16, and rounding keeps the sign, so the bits are those of relu-then-convert with
        // two v_max_i32 less per pair.  (asm: written as a vector max the compiler un-packs the conversion; its operand is
        // a VALU result, so the MFMA hazard pass has nothing to see here.)
        const unsigned pk = pack_bf16(y[set][8 * s + 2 * i], y[set][8 * s + 2 * i + 1]);
        unsigned r;
        asm("v_pk_max_i16 %0, %1, %2" : "=v"(r) : "v"(pk), "s"(lim16));
        Xout[set][ob][s][i] = r;
    }
};
template <int NSETS>
struct EpiRgb {  // views layer: relu + the rgb head's partial dot products (fp32 weights on the VALU)
    static constexpr int Q = 16 * NSETS;  // one accumulator register (a row pair of the block) per slice
    const float* rw;  // cst + rgb weights + g * 8 (m16.h: [c][block][g][8])
    float acc[NSETS][2][3];   // [set][point half nb][c]
    float w0[8], w1[8], w2[8];
    __device__ __forceinline__ void begin(int ob) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(rw + ob * 32 + 4 * i);
            const f32x4 b = *reinterpret_cast<const f32x4*>(rw + 128 + ob * 32 + 4 * i);
            const f32x4 c = *reinterpret_cast<const f32x4*>(rw + 256 + ob * 32 + 4 * i);
#pragma unroll
            for (int t = 0; t < 4; ++t) w0[4 * i + t] = a[t], w1[4 * i + t] = b[t], w2[4 * i + t] = c[t];
        }
    }
    __device__ __forceinline__ void slice(int, const f32x16 (&y)[NSETS], int q) {
        const int set = q / 16, t = q % 16, nb = t / 8, k = t % 8;
        const int b = __float_as_int(y[set][t]);
        const float hv = __int_as_float(b > 0 ? b : 0);
        acc[set][nb][0] = fmaf(w0[k], hv, acc[set][nb][0]);
        acc[set][nb][1] = fmaf(w1[k], hv, acc[set][nb][1]);
        acc[set][nb][2] = fmaf(w2[k], hv, acc[set][nb][2]);
    }
};

// One GEMM layer for this wave's NSETS 32-point sets: block ob of the output = bias + sum_k W . [pre, Xin, post].
// `acc` ([ob & 1][set]) is owned by the caller so that the last block's accumulators survive the call: its epilogue is
// the NEXT layer's `pend` (PEND: the previous layer left one; the views layer, last of the tile, is followed by an
// explicit bf_finish).  NS steps are consumed; the layer occupies NSPAD steps of the stream.
template <int NSETS, int NOB, int NPRE, int NHID, int NPOST, int NSPAD, bool PEND, class Epi>
__device__ __forceinline__ void bf_layer(BRing& st, int lane, const float* bias_h, const u32x4 (*pre)[2][2],
                                         const u32x4 (*Xin)[8][2], const u32x4 (*post)[2], Epi& epi,
                                         f32x16 (&acc)[2][NSETS], EpiPack<NSETS>& pend) {
    constexpr int NKB = NPRE + NHID + NPOST;
    constexpr int SB = NKB * 2;                    // steps per output block
    constexpr int NS = NOB * SB;
    constexpr int PW = 4 * NSETS;                  // LDS-DMA pieces per wave per chunk: 32 / (8 / NSETS waves)
    constexpr int GAP = kBChunkSteps / PW;         // one piece every GAP steps
    constexpr int S0 = SB >= 8 ? 2 : 1;            // first step of a block that carries epilogue slices (the producing
                                                   // MFMAs have drained by then)
    constexpr int PE = (NPRE + 7) * 2 - 1;         // the pending epilogue writes input block 7, first read at step
                                                   // (NPRE + 7) * 2: its last slice comes two steps earlier
    static_assert(NOB % 2 == 0 || NOB == 1, "the last block must leave its accumulators in acc[1], the next layer starts in acc[0] "
                                            "(NOB = 1, the alpha block: it stays in acc[0], the caller reads it, the next layer has no pending epilogue)");
    static_assert(!PEND || (NHID == 8 && PE > S0), "a pending epilogue needs a layer that reads 8 input blocks");
    static_assert(NSPAD % kBChunkSteps == 0 && NSPAD >= NS && NSPAD - NS < kBChunkSteps, "layer padding");
    static_assert(NS % kWQ == 0 && (NS % kBChunkSteps == 0 || (NS % kBChunkSteps) % 4 == 0) && (NS % kBChunkSteps == 0 || NS % kBChunkSteps > kLook),
                  "register ring phase / tail length");
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) {
#ifdef BF_STAMP
        if (st.stamp_base >= 0) {
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();
            if (lane == 0) g_bf_stamps[st.stamp_wave * 128 + st.stamp_base + ob] = t_;
        }
#endif
        const f32x16 bias = *reinterpret_cast<const f32x16*>(bias_h + ob * 64);  // srcC of the block's first MFMAs ([g][16])
        if (ob > 0) epi.begin(ob - 1);
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                // Step s uses the weights in register slot s & 3 and issues the LDS read of step s + kLook (one
                // ds_read_b128 in flight across ~3 steps, so its latency never surfaces).  The hand-over to
                // the next ring chunk comes kLook steps before the chunk's end, right after the read of its last step
                // was issued: by then every wave has ISSUED all its reads of the chunk (and br_acquire drains them
                // before the barrier), so the chunk's slot may be refilled from the next chunk's step 0 on.
                const int sb = kb * 2 + s2;                                             // step inside the block
                const int s = ob * SB + sb;
                const int cs = s % kBChunkSteps;                                        // position inside the ring chunk
                const int cend = (s - cs + kBChunkSteps - 1) < NS - 1 ? (s - cs + kBChunkSteps - 1) : NS - 1;  // chunk's last step
                const bool tail = cend - (s - cs) + 1 < kBChunkSteps;                   // padded (short) last chunk of the layer
                const u32x4 w = st.wq[s % kWQ];
                const int r = s + kLook;
                if (r <= cend) st.wq[r % kWQ] = st.rd[(cs + kLook) * 64];
                const bool slot = cs % GAP == 0;                                        // this chunk-time's refill pieces
                const bool flush = tail && s == cend - kLook;                           // short chunk: issue what its steps never reach
                if (slot || flush) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (slot) br_prefetch_piece(st, cs / GAP);
                    if (flush) {
#pragma unroll
                        for (int jj = cs / GAP + 1; jj < PW; ++jj) br_prefetch_piece(st, jj);
                    }
                    if ((slot && cs / GAP == PW - 1) || flush) br_prefetch_advance(st);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (s == cend - kLook) br_acquire<PW>(st, lane);
                if (r > cend) st.wq[r % kWQ] = st.rd[(r - cend - 1) * 64];
                // this step's share of the previous block's epilogue (own layer: block ob-1; block 0: the pending one)
                if (sb >= S0) {
                    if (ob > 0) {
                        constexpr int Q = Epi::Q;
#pragma unroll
                        for (int q = Q * (sb - S0) / (SB - S0); q < Q * (sb - S0 + 1) / (SB - S0); ++q)
                            epi.slice(ob - 1, acc[(ob - 1) & 1], q);
                    } else if (PEND && sb < PE) {
                        constexpr int Q = EpiPack<NSETS>::Q;
#pragma unroll
                        for (int q = Q * (sb - S0) / (PE - S0); q < Q * (sb - S0 + 1) / (PE - S0); ++q)
                            pend.slice(7, acc[1], q);
                    }
                }
                // rows 16 s2 .. + 15 of the block against all 32 k of k-block kb: the accumulators of row half s2, two
                // point halves per set (x[0], x[1]); the k-block's first step of each row half starts from the bias
#pragma unroll
                for (int set = 0; set < NSETS; ++set) {
                    const u32x4* x = kb < NPRE ? pre[set][kb] : (kb < NPRE + NHID ? Xin[set][kb - NPRE] : post[set]);
                    const f32x16& src = kb == 0 ? bias : acc[ob & 1][set];
                    f32x4 ya, yb;
                    if (s2) blk16_get<1>(src, ya, yb);
                    else blk16_get<0>(src, ya, yb);
                    ya = NRF_MFMA_BF16(w, x[0], ya);
                    yb = NRF_MFMA_BF16(w, x[1], yb);
                    if (s2) blk16_put<1>(acc[ob & 1][set], ya, yb);
                    else blk16_put<0>(acc[ob & 1][set], ya, yb);
                }
                __builtin_amdgcn_sched_barrier(0);   // keep the slices where they are: one step = one scheduling region
            }
        }
    }
#ifdef BF_STAMP
    if (st.stamp_base >= 0) {
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();
        if (lane == 0) g_bf_stamps[st.stamp_wave * 128 + st.stamp_base + 8] = t_;
        if (lane == 0) g_bf_stamps[st.stamp_wave * 128 + st.stamp_base + 9] = t_;
    }
#endif
}
// the epilogue of a layer's last block when no further layer carries it (end of the tile)
template <int NSETS, int NOB, class Epi>
__device__ __forceinline__ void bf_finish(Epi& epi, f32x16 (&acc)[2][NSETS]) {
    epi.begin(NOB - 1);
#pragma unroll
    for (int q = 0; q < Epi::Q; ++q) epi.slice(NOB - 1, acc[(NOB - 1) & 1], q);
}

// Persistent, one workgroup of 8 / NSETS waves per CU (NSETS = 1: two waves per SIMD; NSETS = 2: one); a wave's tile =
// 32 * NSETS consecutive samples of one ray, or (POINTS: Network.forward on a flat batch, network.py:184-186) that many
// consecutive points with their own directions, used as given; then rays_o = pts (n, 3), rays_d = dirs (n, 3),
// n_points bounds the last tile.
template <bool POINTS, int NSETS>
__global__ __launch_bounds__(2048 / (4 * NSETS), 1) void mlp_bf16_kernel(const char* __restrict__ packed, const float* __restrict__ rays_o,
                                                                         const float* __restrict__ rays_d, const float* __restrict__ zbuf,
                                                                         int64_t z_stride, int tiles_per_ray, int64_t n_tiles,
                                                                         int64_t n_groups, int64_t n_points, float4* __restrict__ raw) {
    constexpr int WAVES = 8 / NSETS, PW = 4 * NSETS, TP = 32 * NSETS;   // waves per workgroup, DMA pieces per wave, points per tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5, g = lane >> 4;
    {
        const float4* src = reinterpret_cast<const float4*>(packed + kBSmallFloatOff * 4);
        float4* dst = reinterpret_cast<float4*>(smem + kBLdsConstOff);
        for (int i = threadIdx.x; i < kS16Floats / 4; i += 64 * WAVES) dst[i] = src[i];
    }
    __syncthreads();
    {   // the alpha block's accumulator initialiser, [g][16] like every bias block: alpha_b in row 0 (group 0, registers 0
        // and 8: the two point halves), zeros elsewhere.  It takes the place of the head's fp32 weights in LDS, which this
        // kernel does not read.
        float* ab = reinterpret_cast<float*>(smem + kBLdsConstOff) + kS16AlphaW;
        if (threadIdx.x < 64) ab[threadIdx.x] = (threadIdx.x == 0 || threadIdx.x == 8) ? ab[kS16AlphaB - kS16AlphaW] : 0.f;
    }
    __syncthreads();
    const float* cst = reinterpret_cast<const float*>(smem + kBLdsConstOff);
    const float* bias_h = cst + g * 16;   // this lane group's 16 of a block's [g][16] initialisers

    BRing st;
    br_start<PW>(st, packed, smem, wave, lane);

    for (int64_t tg = blockIdx.x; tg < n_groups; tg += gridDim.x) {
        int64_t tile = tg * WAVES + wave;
        const bool valid = tile < n_tiles;
        if (!valid) tile = n_tiles - 1;
#ifdef BF_STAMP
        const bool stamp_tile = tg == int64_t(blockIdx.x) + 4 * int64_t(gridDim.x);
        st.stamp_wave = blockIdx.x * WAVES + wave;
        st.stamp_base = -1;
#endif
        BF_STAMP_AT(0);
        u32x4 pe[NSETS][2][2], dpe[NSETS][2];
        float pdir[NSETS][3];   // POINTS: the sets' own directions, encoded in front of the views layer (below)
        if constexpr (POINTS) {
#pragma unroll
            for (int set = 0; set < NSETS; ++set) {
                int64_t pt = tile * TP + 32 * set + j;
                pt = pt < n_points ? pt : n_points - 1;   // tail lanes repeat the last point (stores are masked)
                const float px = rays_o[pt * 3], py = rays_o[pt * 3 + 1], pz = rays_o[pt * 3 + 2];
#pragma unroll
                for (int a = 0; a < 3; ++a) pdir[set][a] = rays_d[pt * 3 + a];
                f32x16 enc[2];
                encode_pe_doubling<10>(px, py, pz, h, enc);
                pack_tile(to_m16(enc[0]), pe[set][0]);   // made in the 32x32 form, handed over (m16.h)
                pack_tile(to_m16(enc[1]), pe[set][1]);
            }
        } else {
            const int64_t ray = tile / tiles_per_ray;
            const int s0 = int(tile - ray * tiles_per_ray) * TP;
            const float ox = rays_o[ray * 3 + 0], oy = rays_o[ray * 3 + 1], oz = rays_o[ray * 3 + 2];
            float dx = rays_d[ray * 3 + 0], dy = rays_d[ray * 3 + 1], dz = rays_d[ray * 3 + 2];
            const float nrm = sqrtf(fmaf(dz, dz, fmaf(dy, dy, dx * dx)));  // volume_renderer.py:230 (torch's 3-element norm)
            dx = dx / nrm, dy = dy / nrm, dz = dz / nrm;
#pragma unroll
            for (int set = 0; set < NSETS; ++set) {
                const float zz = zbuf[ray * z_stride + s0 + 32 * set + j];
                const float px = __fadd_rn(ox, __fmul_rn(zz, dx));  // :231
                const float py = __fadd_rn(oy, __fmul_rn(zz, dy));
                const float pz = __fadd_rn(oz, __fmul_rn(zz, dz));
                f32x16 enc[2];
                encode_pe_doubling<10>(px, py, pz, h, enc);
                pack_tile(to_m16(enc[0]), pe[set][0]);
                pack_tile(to_m16(enc[1]), pe[set][1]);
            }
            f32x16 denc[1];
            encode_pe_doubling<4>(dx, dy, dz, h, denc);
            pack_tile(to_m16(denc[0]), dpe[0]);  // all sets are samples of the same ray
#pragma unroll
            for (int set = 1; set < NSETS; ++set) dpe[set][0] = dpe[0][0], dpe[set][1] = dpe[0][1];
        }

        BF_STAMP_AT(1);
        u32x4 XA[NSETS][8][2], XB[NSETS][8][2];
        f32x16 acc[2][NSETS];                 // accumulator blocks [ob & 1][set], shared by all layers (see bf_layer)
        EpiPack<NSETS> toA{XA, kPkRelu};      // the epilogue a layer writing XA leaves pending for its successor
        EpiPack<NSETS> toB{XB, kPkRelu};
        bf_layer<NSETS, 8, 2, 0, 0, kBStepsL0, false>(st, lane, bias_h, pe, nullptr, nullptr, toA, acc, toA);     // pts_linears.0
        BF_STAMP_AT(2);
        // pts1..7 + feature as 4 pairs (XA -> XB -> XA); pair 2 starts with the skip layer, the last layer of pair 3
        // is feature_linear (no activation)
        for (int p = 0; p < 4; ++p) {
            const float* b0 = bias_h + (2 * p + 1) * 512;
#ifdef BF_STAMP
            st.stamp_base = stamp_tile ? 16 + 20 * p : -1;
#endif
            // first layer of the pair: XA -> XB; the layer before it (pts0 or the pair before) left XA's block 7 pending
            if (p == 2)
                bf_layer<NSETS, 8, 2, 8, 0, kBStepsSkip, true>(st, lane, b0, pe, XA, nullptr, toB, acc, toA);     // cat([pe, h]) (network.py:58)
            else
                bf_layer<NSETS, 8, 0, 8, 0, kBStepsHidden, true>(st, lane, b0, nullptr, XA, nullptr, toB, acc, toA);
            BF_STAMP_AT(3 + 2 * p);
#ifdef BF_STAMP
            st.stamp_base = stamp_tile ? 26 + 20 * p : -1;
#endif
            toA.lim16 = p == 3 ? kPkNone : kPkRelu;                                                               // feature_linear: no activation
            bf_layer<NSETS, 8, 0, 8, 0, kBStepsHidden, true>(st, lane, b0 + 512, nullptr, XB, nullptr, toA, acc, toB);
            BF_STAMP_AT(4 + 2 * p);
#ifdef BF_STAMP
            st.stamp_base = -1;
#endif
        }
        // XB = bf16(relu(h7)): alpha_linear as one more output block (rows 0 / 1 = the weight's two bf16 terms, bias in
        // srcC); the layer also completes feature_linear's last block (pending in acc[1]) into XA.  Block 0 stays in
        // acc[0]: sigma = row 0 + row 1 = registers 0 and 1 of lane half 0 (acc_row), read before views overwrites them.
        // Register allocation: the VALU head that stood here read all of XB and thereby settled where XA / XB live through
        // the pair loop; without such a use the allocator's choice spilled 112 registers per lane (and every scratch
        // reload's vmcnt(0) drains the LDS-DMA ring).  Naming XB as accumulation registers here -- where the alpha block's
        // MFMAs read it anyway -- restores a spill-free assignment (474 registers).
#pragma unroll
        for (int set = 0; set < NSETS; ++set)
#pragma unroll
            for (int kb = 0; kb < 8; ++kb)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) asm volatile("" : "+a"(XB[set][kb][s2]));
        bf_layer<NSETS, 1, 0, 8, 0, kBStepsAlphaPad, true>(st, lane, cst + kS16AlphaW + g * 16, nullptr, XB, nullptr, toB, acc, toA);
        // rows 0 and 1 of the block = registers 0, 1 (points c) and 8, 9 (points 16 + c) of lane group 0
        float sigma[NSETS][2];
#pragma unroll
        for (int set = 0; set < NSETS; ++set) {
            sigma[set][0] = acc[0][set][0] + acc[0][set][1];
            sigma[set][1] = acc[0][set][8] + acc[0][set][9];
            asm volatile("" : "+v"(sigma[set][0]), "+v"(sigma[set][1]));   // pinned here: sunk behind the views layer it would keep the block alive across it
        }
        BF_STAMP_AT(11);
        if constexpr (POINTS) {
            // every set has its own direction fragments here: carried through the eight hidden layers they are 8 registers
            // more than the kernel has (scratch reloads in the layer loop, each draining the LDS-DMA ring); three floats
            // per set are carried instead and encoded now
#pragma unroll
            for (int set = 0; set < NSETS; ++set) {
                f32x16 denc[1];
                encode_pe_doubling<4>(pdir[set][0], pdir[set][1], pdir[set][2], h, denc);
                pack_tile(to_m16(denc[0]), dpe[set]);
            }
        }
        // views_linears.0 on cat([feature (XA, no activation), dir PE]) -> 128, relu, rgb_linear
        EpiRgb<NSETS> er;
        er.rw = cst + kS16RgbW + g * 8;
#pragma unroll
        for (int set = 0; set < NSETS; ++set)
#pragma unroll
            for (int c = 0; c < 3; ++c) er.acc[set][0][c] = er.acc[set][1][c] = 0.f;
        bf_layer<NSETS, 4, 0, 8, 1, kBStepsViewsPad, false>(st, lane, bias_h + 9 * 512, nullptr, XA, dpe, er, acc, toA);
        bf_finish<NSETS, 4>(er, acc);
        BF_STAMP_AT(12);
        const float* rb = cst + kS16RgbB;
        const bool odd = (g & 1) != 0;
#pragma unroll
        for (int set = 0; set < NSETS; ++set) {
            // lane l < 32 stores point l of the set: group 0 holds points c, group 1 takes points 16 + c (m16.h: add_groups);
            // sigma is complete, bias included, in group 0: its second half goes to group 1
            const float r = add_groups(er.acc[set][0][0], er.acc[set][1][0], odd), gr = add_groups(er.acc[set][0][1], er.acc[set][1][1], odd);
            const float b = add_groups(er.acc[set][0][2], er.acc[set][1][2], odd);
            const float s1 = __shfl_xor(sigma[set][1], 16);
            const int64_t dst = tile * TP + 32 * set + j;
            if (lane < 32 && valid && (!POINTS || dst < n_points))
                raw[dst] = make_float4(r + rb[0], gr + rb[1], b + rb[2], odd ? s1 : sigma[set][0]);
        }
        BF_STAMP_AT(13);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA may outlive the workgroup
}

}  // namespace

NRF_EXPORT int64_t nrf_packed_bf16_bytes(void) { return kBSmallFloatOff * 4 + int64_t(kS16Floats) * 4; }

NRF_EXPORT int nrf_pack_weights_bf16(const float* const* w, const float* const* b, void* packed, nrf_stream_t stream_) {
    if (!w || !b || !packed) return NRF_ERR_NULL;
    for (int i = 0; i < NRF_NUM_LAYERS; ++i)
        if (!w[i] || !b[i]) return NRF_ERR_NULL;
    if (!nrf_aligned16(packed)) return NRF_ERR_ALIGN;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    unsigned short* base = static_cast<unsigned short*>(packed);
    nrf_launch_begin();
    (void)hipMemsetAsync(packed, 0, size_t(kBStreamSteps) * 1024, stream);  // the alpha / views layers' padding steps stay zero
    auto launch = [&](BPackLayer p, int step0) {
        p.dst = base + int64_t(step0) * 512;
        const int total = p.nob * (p.pre_kb + p.hid_kb + p.post_kb) * 2 * 512;
        bpack_layer_kernel<<<(total + 255) / 256, 256, 0, stream>>>(p);
    };
    launch(BPackLayer{w[0], nullptr, 63, 8, 2, 0, 0, 10, 0, 0, 0, 0, 0}, kBLayerStep[0]);
    for (int l = 1; l <= 7; ++l) {
        if (l == 5)  // network.py:58 cat([input_pts, h]): PE columns 0..62 then hidden 63..318, per output block
            launch(BPackLayer{w[5], nullptr, 319, 8, 2, 8, 0, 10, 0, 0, 63, 0, 0}, kBLayerStep[5]);
        else
            launch(BPackLayer{w[l], nullptr, 256, 8, 0, 8, 0, 0, 0, 0, 0, 0, 0}, kBLayerStep[l]);
    }
    launch(BPackLayer{w[8], nullptr, 256, 8, 0, 8, 0, 0, 0, 0, 0, 0, 0}, kBLayerStep[8]);     // feature_linear
    launch(BPackLayer{w[9], nullptr, 256, 1, 0, 8, 0, 0, 0, 0, 0, 0, 1}, kBAlphaStep);        // alpha_linear: one block, two-term weight
    launch(BPackLayer{w[10], nullptr, 283, 4, 0, 8, 1, 0, 4, 0, 0, 256, 0}, kBLayerStep[9]);  // views_linears.0
    PackSmall16 s;
    for (int l = 0; l < 9; ++l) s.bias[l] = b[l];
    s.bias[9] = b[10];
    s.alpha_w = w[9], s.alpha_b = b[9], s.rgb_w = w[11], s.rgb_b = b[11];
    s.dst = static_cast<float*>(packed) + kBSmallFloatOff;
    pack_small16_kernel<<<(kS16Floats + 255) / 256, 256, 0, stream>>>(s);
    return nrf_launch_status();
}

template <bool POINTS, int NSETS>
static int launch_bf16(const void* packed_bf16, const float* a, const float* b, const float* z, int64_t z_stride,
                       int tiles_per_ray, int64_t n_tiles, int64_t n_points, float* raw, nrf_stream_t stream) {
    constexpr int WAVES = 8 / NSETS;
    const int64_t groups = (n_tiles + WAVES - 1) / WAVES;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (cus <= 0) cus = 256;
    static bool attr_set[64] = {};
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_bf16_kernel<POINTS, NSETS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                kBLdsBytes) != hipSuccess)
            return NRF_ERR_LAUNCH;
        attr_set[dev] = true;
    }
    const unsigned grid = unsigned(groups < cus ? groups : cus);
    nrf_launch_begin();
    mlp_bf16_kernel<POINTS, NSETS><<<dim3(grid), 64 * WAVES, kBLdsBytes, static_cast<hipStream_t>(stream)>>>(
        static_cast<const char*>(packed_bf16), a, b, z, z_stride, tiles_per_ray, n_tiles, groups, n_points,
        reinterpret_cast<float4*>(raw));
    return nrf_launch_status();
}

NRF_EXPORT int nrf_mlp_forward_rays_bf16(const void* packed_bf16, const float* rays_o, const float* rays_d, const float* z,
                                         int64_t z_stride, int64_t n_rays, int n_samples, float* raw, nrf_stream_t stream) {
    if (n_rays < 0 || n_samples <= 0 || n_samples % 64 != 0 || z_stride < 0) return NRF_ERR_SHAPE;
    if (z_stride != 0 && z_stride < n_samples) return NRF_ERR_SHAPE;
    if (n_rays == 0) return NRF_OK;
    if (!packed_bf16 || !rays_o || !rays_d || !z || !raw) return NRF_ERR_NULL;
    if (!nrf_aligned16(packed_bf16) || !nrf_aligned16(raw)) return NRF_ERR_ALIGN;
    if (n_rays > (int64_t(1) << 40) / n_samples) return NRF_ERR_SHAPE;
    const int tiles_per_ray = n_samples / 64;
    return launch_bf16<false, 2>(packed_bf16, rays_o, rays_d, z, z_stride, tiles_per_ray, n_rays * tiles_per_ray, 0, raw, stream);
}

// Network.forward on a flat batch with the bf16 MLP: pts (n, 3), dirs (n, 3) used as given.  raw: (n, 4).
NRF_EXPORT int nrf_mlp_forward_points_bf16(const void* packed_bf16, const float* pts, const float* dirs, int64_t n_points,
                                           float* raw, nrf_stream_t stream) {
    if (n_points < 0 || n_points > (int64_t(1) << 40)) return NRF_ERR_SHAPE;
    if (n_points == 0) return NRF_OK;
    if (!packed_bf16 || !pts || !dirs || !raw) return NRF_ERR_NULL;
    if (!nrf_aligned16(packed_bf16) || !nrf_aligned16(raw)) return NRF_ERR_ALIGN;
    return launch_bf16<true, 2>(packed_bf16, pts, dirs, nullptr, 0, 0, (n_points + 63) / 64, n_points, raw, stream);
}
