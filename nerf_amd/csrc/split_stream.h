// What the operand-split chain kernels share (mlp_bf16x3.hip: three bf16 terms per operand; mlp_fp16x2.hip: two fp16
// terms): the order of the weight stream in k16-steps, the packer's source lookup, and the input op of a hidden-type
// layer up to the point where the operand is split.
#pragma once
#include "m16.h"

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kXChunkSteps = 8;                       // k16-steps per LDS ring chunk (both kernels)
constexpr int kXStepsL0 = 8 * 2 * 2;                  // 8 out blocks x 2 PE k-blocks x 2 k16-steps
constexpr int kXStepsHidden = 8 * 8 * 2;              // 128
constexpr int kXStepsViews = 4 * 9 * 2;               // 72
// stream order (steps): pts0, pts1..4, pts5 hidden part, pts5 PE pass, pts6, pts7, feature, views
constexpr int kXLayerStep[11] = {0, 32, 160, 288, 416, 544, 672, 704, 832, 960, 1088};
constexpr int kXStreamSteps = 1088 + kXStepsViews;    // 1160
constexpr int kXChunks = kXStreamSteps / kXChunkSteps;  // 145
static_assert(kXStreamSteps % kXChunkSteps == 0 && kXStepsL0 % kXChunkSteps == 0 && kXStepsHidden % kXChunkSteps == 0 &&
              kXStepsViews % kXChunkSteps == 0, "layers must be whole chunks");
// backward data-gradient chain: views^T (128 -> 256), then eight 256 x 256 layers
constexpr int kXBwdSteps = 8 * 4 * 2 + 8 * kXStepsHidden;   // 1088
constexpr int kXBwdChunks = kXBwdSteps / kXChunkSteps;      // 136
static_assert(kXBwdSteps % kXChunkSteps == 0, "backward layers must be chunk aligned");

// ---- weight packing: which weight sits at (step, lane, element jj of the lane's 8) of a layer's stream -------------
struct XPackLayer {
    const float* W;
    unsigned short* dst;
    int ldw, nob;
    int pre_kb, hid_kb, post_kb;
    int pre_L, post_L;
    int pre_col0, hid_col0, post_col0;
    int lead;     // hidden-type layers: the stream opens with blocks 0 .. lead-1 interleaved, [kb][ob < lead][s2] (0: plain)
    int transposed;   // backward chain: A = W^T (hidden part only): row = a column of W, k = a row of W
    int m16;          // A fragments of v_mfma_*_16x16x32 (mlp_bf16x3.hip, M16): the step at the same stream position holds rows
                      // 32 ob + 16 s2 + (lane & 15) and, in lane group g = lane >> 4, the 8 k of feature slots m16_slot(g, jj)
};
__device__ __forceinline__ int xpack_total(const XPackLayer& p) { return p.nob * (p.pre_kb + p.hid_kb + p.post_kb) * 2 * 512; }
__device__ __forceinline__ float xpack_fetch(const XPackLayer& p, int idx, int& step, int& lane, int& jj) {
    const int nkb = p.pre_kb + p.hid_kb + p.post_kb;
    jj = idx & 7, lane = (idx >> 3) & 63, step = idx >> 9;
    int s2 = step & 1, kbpos = (step >> 1) % nkb, ob = (step >> 1) / nkb;
    if (p.lead > 1 && step < 2 * p.lead * nkb) kbpos = step / (2 * p.lead), ob = (step >> 1) % p.lead;
    int h = lane >> 5, t = 8 * s2 + jj;
    int64_t row = int64_t(ob * 32 + (lane & 31)) * p.ldw;
    if (p.m16) {
        // slot phi of the block = register t, lane half h of the 32x32 form with acc_row(t, h) == phi: the encoding
        // blocks are made in that form and handed over register by register (mlp_bf16x3.hip: to_m16)
        const int phi = m16_slot(lane >> 4, jj);
        t = 4 * (phi >> 3) + (phi & 3), h = (phi >> 2) & 1;
        row = int64_t(ob * 32 + 16 * s2 + (lane & 15)) * p.ldw;
    }
    float v = 0.f;
    if (p.transposed) {
        v = p.W[int64_t(32 * kbpos + acc_row(t, h)) * p.ldw + p.hid_col0 + ob * 32 + (lane & 31)];
    } else if (kbpos < p.pre_kb) {
        const int c = pe_col(kbpos * 16 + t, h, p.pre_L);
        if (c >= 0) v = p.W[row + p.pre_col0 + c];
    } else if (kbpos < p.pre_kb + p.hid_kb) {
        v = p.W[row + p.hid_col0 + 32 * (kbpos - p.pre_kb) + acc_row(t, h)];
    } else {
        const int c = pe_col((kbpos - p.pre_kb - p.hid_kb) * 16 + t, h, p.post_L);
        if (c >= 0) v = p.W[row + p.post_col0 + c];
    }
    return v;
}
// the forward stream's layers in stream order (kXLayerStep): launch(layer description, first step)
template <class Launch>
inline void xpack_forward_layers(const float* const* w, Launch&& launch, int lead = 2, int m16 = 0) {
    launch(XPackLayer{w[0], nullptr, 63, 8, 2, 0, 0, 10, 0, 0, 0, 0, 0, 0, m16}, kXLayerStep[0]);
    for (int l = 1; l <= 4; ++l) launch(XPackLayer{w[l], nullptr, 256, 8, 0, 8, 0, 0, 0, 0, 0, 0,lead, 0, m16}, kXLayerStep[l]);
    // network.py:58 cat([input_pts, h]): hidden columns 63..318 as a plain layer, then the PE columns 0..62 accumulate
    launch(XPackLayer{w[5], nullptr, 319, 8, 0, 8, 0, 0, 0, 0, 63, 0,lead, 0, m16}, kXLayerStep[5]);
    launch(XPackLayer{w[5], nullptr, 319, 8, 2, 0, 0, 10, 0, 0, 0, 0, 0, 0, m16}, kXLayerStep[6]);
    launch(XPackLayer{w[6], nullptr, 256, 8, 0, 8, 0, 0, 0, 0, 0, 0,lead, 0, m16}, kXLayerStep[7]);
    launch(XPackLayer{w[7], nullptr, 256, 8, 0, 8, 0, 0, 0, 0, 0, 0,lead, 0, m16}, kXLayerStep[8]);
    launch(XPackLayer{w[8], nullptr, 256, 8, 0, 8, 0, 0, 0, 0, 0, 0,lead, 0, m16}, kXLayerStep[9]);      // feature_linear
    launch(XPackLayer{w[10], nullptr, 283, 4, 0, 8, 1, 0, 4, 0, 0, 256,lead, 0, m16}, kXLayerStep[10]);   // views_linears.0
}
// the small fp32 block behind a stream (biases, alpha / rgb head rows: pack_small_kernel's layout)
inline PackSmall xpack_small(const float* const* w, const float* const* b, void* packed, int64_t small_float_off) {
    PackSmall s;
    for (int l = 0; l < 8; ++l) s.bias[l] = b[l];
    s.bias[8] = b[8];
    s.bias[9] = b[10];
    s.alpha_w = w[9];
    s.alpha_b = b[9];
    s.rgb_w = w[11];
    s.rgb_b = b[11];
    s.packed = static_cast<float*>(packed);
    s.small_shift = small_float_off - kOffBias;
    return s;
}

constexpr int kActRelu = 0, kActNone = int(0x80000000u);
constexpr int kOpFwd = 0, kOpFwdSave = 1, kOpBwd = 2;   // input-op modes

// Input op of a hidden-type layer up to the operand split: the activated fp32 value of a register pair of the previous
// layer's block (relu as max(bits, lim): lim = 0, or INT_MIN for "no activation").  The layer whose input is relu(h7)
// (feature_linear) also feeds alpha_linear (256 -> 1, fp32 weights, VALU).
struct OpInBase {
    int lim;
    const float* aw;   // cst + alpha weights + h * 16
    float sigma;
    // training forward (kOpFwdSave): the activated input of the layer goes to `save` (acts rows of this layer's input,
    // this lane's column: mlp_fp32.hip's layout) and its relu signs to `msave` (nullptr: no activation, no mask)
    float* save;
    unsigned* msave;
    unsigned mcur, mprev;
    // backward chain (kOpBwd): the layer's input is dY = relu'(h) * (y + w_alpha * dsig), masked by the sign words
    // the training forward saved (mlp_bwd.hip: OpBwdMask) and written to `save` as the dY rows of the wgrad GEMMs
    unsigned m[4];
    float dsig;
    float kscale;      // SCALED stores (the fp16x2 backward chain runs at a power-of-two loss scale): saved = value * kscale
    template <int MODE, bool SCALED = false, bool M16 = false>   // M16: the lane holds the block in the 16x16x32 form (m16.h)
    __device__ __forceinline__ void keep(int blk, int q, float a, float b) {
        if (MODE == kOpFwd) return;
#ifdef NRF_ABL_NOKEEP   // timing only (tools/debug/lib_ab.sh)
        return;
#endif
        const int t0 = 2 * q, t1 = 2 * q + 1;
#ifdef NRF_ABL_FAKESTORE   // timing only (tools/debug/lib_ab.sh): the values and sign words are made, the row stores are not issued
        asm volatile("" ::"v"(SCALED ? a * kscale : a), "v"(SCALED ? b * kscale : b));
#elif defined(NRF_ABL_STORE2)   // timing only: the same bytes and lines with one 8-byte store per pair (wrong layout)
        *reinterpret_cast<float2*>(save + (32 * blk + (t0 & 3) + 8 * (t0 >> 2)) * 32 + (threadIdx.x & 31)) =
            make_float2(SCALED ? a * kscale : a, SCALED ? b * kscale : b);
#else
        save_row(save + 32 * blk * 32 + saved_row_off<M16>(t0), SCALED ? a * kscale : a);
        save_row(save + 32 * blk * 32 + saved_row_off<M16>(t1), SCALED ? b * kscale : b);
#endif
        if (MODE == kOpFwdSave && msave) {
            if (q == 0) mcur = 0u;
            mcur |= ((a > 0.f ? 1u : 0u) << t0) | ((b > 0.f ? 1u : 0u) << t1);   // (as min(bits, 1) << t | mcur: no faster)
            if (q == 7) {
                const unsigned w = M16 ? m16_mask_to_old(mcur) : mcur;   // the words in memory keep the 32x32 form's bit order
                if (blk & 1) msave[(blk >> 1) * 64] = mprev | (w << 16);
                else mprev = w;
            }
        }
    }
    // alpha_linear (network.py:61) on relu(h7): Y holds layer 7's raw outputs when the loop reaches feature_linear.  ONE
    // lump per tile, in front of that layer -- as a run-time `if (alpha)` inside the shared layer body (once per k-block)
    // the compiler gave EVERY layer's k-block a second relu of the block and sixteen moves around the branch: 32 wasted
    // VALU instructions per k-block, 256 per layer, and the block's real relu pulled out of its MFMA gaps into the same
    // lump.  Same terms in the same order (k-block, then register): sigma keeps its bits.
    __device__ __forceinline__ void alpha_head(const f32x16* Y) {
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
            const f32x16 w = *reinterpret_cast<const f32x16*>(aw + kb * 32);
            const f32x16 hx = relu16(Y[kb]);
#pragma unroll
            for (int t = 0; t < 16; ++t) sigma = fmaf(w[t], hx[t], sigma);
        }
    }
    template <int MODE, bool SCALED = false, class YT>   // YT: f32x16, or the 16x16x32 form's block (mlp_bf16x3.hip: Blk16)
    __device__ __forceinline__ void input(const YT& y, int blk, int q, float& a, float& b) {
        if (MODE == kOpBwd) {
            const float2 w = *reinterpret_cast<const float2*>(aw + blk * 32 + 2 * q);
            // the sign bit as an all-ones / all-zeros word (one v_bfe_i32) and an AND: two instructions per element
            // where test + compare + select were three (chain 4.43 -> 4.37 ms); a cleared element is +0 either way
            const int word = int(m[blk >> 1]), at = 16 * (blk & 1) + 2 * q;
            a = __int_as_float(__float_as_int(fmaf(w.x, dsig, y[2 * q])) & __builtin_amdgcn_sbfe(word, at, 1));
            b = __int_as_float(__float_as_int(fmaf(w.y, dsig, y[2 * q + 1])) & __builtin_amdgcn_sbfe(word, at + 1, 1));
        } else {
            const int b0 = __float_as_int(y[2 * q]), b1 = __float_as_int(y[2 * q + 1]);
            a = __int_as_float(b0 > lim ? b0 : lim), b = __int_as_float(b1 > lim ? b1 : lim);
        }
        keep<MODE, SCALED, is_blk16<YT>::value>(blk, q, a, b);
    }
};
__device__ __forceinline__ OpInBase op_in(int lim, const float* aw, float* save = nullptr) {
    return OpInBase{lim, aw, 0.f, save, nullptr, 0u, 0u, {0u, 0u, 0u, 0u}, 0.f, 1.f};
}

}  // namespace
