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
//     no LDS or HBM round trips of activations between the 10 GEMM layers.
//   * One wavefront = 32 points x all 256 features: 128 accumulator registers in, 128 out
//     (1 wave per SIMD, 512-register budget).
//   * Weights are pre-permuted (nrf_pack_weights) into the order the wave consumes them, so
//     the whole network is ONE linear 2.3 MiB stream of 1 KiB steps.  The stream is fetched once
//     per 4-wave workgroup (not per wave) by LDS-DMA (global_load_lds_dwordx4) into a 5 x 16 KiB
//     LDS ring, 4 chunks ahead of use, one raw s_barrier per 16-step chunk (mlp_chain.h); each
//     wave reads its A operands from the ring with ds_read_b128 one step ahead of the MFMAs.
//     One float4 per lane feeds 4 MFMAs (256 matrix-core cycles).  L2-resident: every
//     workgroup streams the same bytes.
//   * PE (63 ch) and dir PE (27 ch) are computed in registers in the k-order the packed
//     weights expect; alpha (256->1) and rgb (128->3) heads are VALU dot products.
#include "mlp_chain.h"

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
    int transpose;                      // hidden part only: stream W^T (rows = input features) for the dgrad chain
};

// all layers of a stream in one launch (blockIdx.y = layer; a training step re-packs four streams after every update,
// and eleven 4-us launches per stream were a third of a millisecond per step)
constexpr int kPackBatch = 11;
struct PackBatch { PackLayer l[kPackBatch]; };
__global__ void pack_layers_kernel(PackBatch b) {
    const PackLayer& p = b.l[blockIdx.y];
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
        if (p.transpose)  // A[i = input feature][k = output feature] = W[k][i]
            v = p.W[int64_t(32 * (kbpos - p.pre_kb) + acc_row(t, h)) * p.ldw + p.hid_col0 + ob * 32 + (lane & 31)];
        else
            v = p.W[row + p.hid_col0 + 32 * (kbpos - p.pre_kb) + acc_row(t, h)];
    } else {
        const int c = pe_col((kbpos - p.pre_kb - p.hid_kb) * 16 + t, h, p.post_L);
        if (c >= 0) v = p.W[row + p.post_col0 + c];
    }
    p.dst[idx] = v;
}

NRF_EXPORT int64_t nrf_packed_floats(void) { return kPackedFloats; }

NRF_EXPORT int nrf_pack_weights(const float* const* w, const float* const* b, float* packed, nrf_stream_t stream_) {
    if (!w || !b || !packed) return NRF_ERR_NULL;
    for (int i = 0; i < NRF_NUM_LAYERS; ++i)
        if (!w[i] || !b[i]) return NRF_ERR_NULL;
    if (!nrf_aligned16(packed)) return NRF_ERR_ALIGN;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    nrf_launch_begin();
    PackBatch batch;
    int n_batch = 0, max_total = 0;
    auto add = [&](const PackLayer& p) {
        batch.l[n_batch++] = p;
        const int total = p.nob * (p.pre_kb + p.hid_kb + p.post_kb) * 4 * kStepFloats;
        max_total = total > max_total ? total : max_total;
    };
    auto launch = [&](PackLayer p, int layer_slot) {
        p.dst = packed + int64_t(kLayerStep[layer_slot]) * kStepFloats;
        add(p);
    };
    // pts_linears.0: 63 PE -> 256
    launch(PackLayer{w[0], nullptr, 63, 8, 2, 0, 0, 10, 0, 0, 0, 0}, 0);
    for (int l = 1; l <= 7; ++l) {
        if (l == 5) {
            // network.py:58 cat([input_pts, h]): source columns 0..62 are the PE, 63..318 the hidden.  Streamed as
            // a plain 256->256 layer (so it shares the hidden-layer code) followed by a PE pass that accumulates.
            launch(PackLayer{w[5], nullptr, 319, 8, 0, 8, 0, 0, 0, 0, 63, 0}, 5);
            PackLayer pe_part{w[5], packed + int64_t(kLayerStep[5] + kStepsHidden) * kStepFloats, 319, 8, 2, 0, 0, 10, 0, 0, 0, 0};
            add(pe_part);
        } else
            launch(PackLayer{w[l], nullptr, 256, 8, 0, 8, 0, 0, 0, 0, 0, 0}, l);
    }
    launch(PackLayer{w[8], nullptr, 256, 8, 0, 8, 0, 0, 0, 0, 0, 0}, 8);  // feature_linear
    // views_linears.0: network.py:63 cat([feature, input_views]): 256 hidden then 27 dir PE -> 128
    launch(PackLayer{w[10], nullptr, 283, 4, 0, 8, 1, 0, 4, 0, 0, 256}, 9);
    pack_layers_kernel<<<dim3((max_total + 255) / 256, n_batch), 256, 0, stream>>>(batch);
    PackSmall s;
    for (int l = 0; l < 8; ++l) s.bias[l] = b[l];
    s.bias[8] = b[8];
    s.bias[9] = b[10];
    s.alpha_w = w[9];
    s.alpha_b = b[9];
    s.rgb_w = w[11];
    s.rgb_b = b[11];
    s.packed = packed;
    s.small_shift = 0;
    pack_small_kernel<<<(kSmallFloats + 255) / 256, 256, 0, stream>>>(s, int64_t(kStreamSteps) * kStepFloats, kPrefetch * kStepFloats);
    return nrf_launch_status();
}

NRF_EXPORT int64_t nrf_packed_bwd_floats(void) { return kBwdPackedFloats; }

// Stream for the backward (dgrad) chain, in the order it is consumed: views^T (hidden columns
// only: the direction encoding needs no gradient), feature^T, pts7^T ... pts1^T (pts5: hidden
// columns 63..318 only).  d(input of layer l) = W_l^T . d(output of layer l).
NRF_EXPORT int nrf_pack_weights_transposed(const float* const* w, float* packed_t, nrf_stream_t stream_) {
    if (!w || !packed_t) return NRF_ERR_NULL;
    for (int i = 0; i < NRF_NUM_LAYERS; ++i)
        if (!w[i]) return NRF_ERR_NULL;
    if (!nrf_aligned16(packed_t)) return NRF_ERR_ALIGN;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    nrf_launch_begin();
    PackBatch batch;
    int n_batch = 0, max_total = 0;
    int step = 0;
    auto launch = [&](const float* W, int ldw, int in_blocks, int out_blocks, int col0) {
        const PackLayer p{W, packed_t + int64_t(step) * kStepFloats, ldw, in_blocks, 0, out_blocks, 0, 0, 0, 0, col0, 0, 1};
        const int total = in_blocks * out_blocks * 4 * kStepFloats;
        batch.l[n_batch++] = p;
        max_total = total > max_total ? total : max_total;
        step += in_blocks * out_blocks * 4;
    };
    launch(w[10], 283, 8, 4, 0);  // views^T: 256 inputs x 128 outputs
    launch(w[8], 256, 8, 8, 0);   // feature^T
    for (int l = 7; l >= 1; --l) launch(w[l], l == 5 ? 319 : 256, 8, 8, l == 5 ? 63 : 0);
    pack_layers_kernel<<<dim3((max_total + 255) / 256, n_batch), 256, 0, stream>>>(batch);
    (void)hipMemsetAsync(packed_t + int64_t(kBwdSteps) * kStepFloats, 0, size_t(kPrefetch) * kStepFloats * 4, stream);
    return nrf_launch_status();
}

// ------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------
namespace {

// Persistent: gridDim.x <= number of CUs, one 4-wave workgroup per CU, one wave per SIMD.
// Tile group tg = 4 consecutive 32-point tiles, one per wave; all waves run every group (the
// barriers of the weight ring are workgroup-wide), waves past the end just do not store.
// SAVE (training): also writes every layer's activated input (`acts`) and relu sign bits (`masks`)
// for the backward kernels; both buffers must cover n_groups * 4 tiles.
template <bool SAVE>
__global__ __launch_bounds__(256, 1) void mlp_fwd_kernel(const float* __restrict__ packed,
                                                         const float* __restrict__ a0,  // rays_o | pts
                                                         const float* __restrict__ a1,  // rays_d | dirs
                                                         const float* __restrict__ zbuf, int64_t z_stride,
                                                         int tiles_per_ray, int rays_mode, int64_t n_points,
                                                         int64_t n_groups, float4* __restrict__ raw,
                                                         float* __restrict__ acts, unsigned* __restrict__ masks) {
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
    ws_start(st, reinterpret_cast<const char*>(packed), kChunks, smem, wave, lane);

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
        float* arow = nullptr;     // this lane's column of the tile's saved rows
        unsigned* mrow = nullptr;
        if (SAVE) {
            arow = acts + tile * (int64_t(kActRows) * 32) + 4 * h * 32 + j;
            mrow = masks + tile * (kMaskLayers * 4 * 64) + lane;
            store_block(arow + kActPe * 32, 0, pe[0]);
            store_block(arow + kActPe * 32, 1, pe[1]);
            store_block(arow + kActDpe * 32, 0, dpe[0]);
        }

        f32x16 X[8], Y[8];
        OpCopy op_none;
        OpRelu op_relu;
        mfma_layer<8, 2, 0, 0, INIT_BIAS>(st, lane, bias_h, pe, X, nullptr, Y, op_none);      // pts_linears.0
        for (int li = 1; li <= 8; ++li) {                                                      // pts 1..7, feature
            if (SAVE) {  // X = h_{li-1} = relu(Y): saved as the input of layer li, its signs as mask li-1
                OpFwdSave<true> op{arow + (kActH + (li - 1) * 256) * 32, mrow + (li - 1) * 4 * 64, 0u};
                mfma_layer<8, 0, 8, 0, INIT_BIAS>(st, lane, bias_h + li * 256, nullptr, X, nullptr, Y, op);
            } else {
                mfma_layer<8, 0, 8, 0, INIT_BIAS>(st, lane, bias_h + li * 256, nullptr, X, nullptr, Y, op_relu);
            }
            if (li == 5)  // skip: cat([pe, h]) (network.py:58) = the hidden part above + this PE pass on top
                mfma_layer<8, 2, 0, 0, INIT_ACCUM>(st, lane, nullptr, pe, X, nullptr, Y, op_none);
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
        if (SAVE) {
            OpFwdSave<false> op{arow + kActFeat * 32, nullptr, 0u};
            mfma_layer<4, 0, 8, 1, INIT_BIAS>(st, lane, bias_h + 9 * 256, nullptr, X, dpe, Y, op);
        } else {
            mfma_layer<4, 0, 8, 1, INIT_BIAS>(st, lane, bias_h + 9 * 256, nullptr, X, dpe, Y, op_none);
        }
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
               int tiles_per_ray, int rays_mode, int64_t n_points, float* raw, hipStream_t stream,
               float* acts = nullptr, unsigned* masks = nullptr) {
    const int64_t tiles = (n_points + 31) / 32;
    const int64_t groups = (tiles + 3) / 4;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (cus <= 0) cus = 256;
    const unsigned grid = unsigned(groups < cus ? groups : cus);
    nrf_launch_begin();
    if (acts)
        mlp_fwd_kernel<true><<<dim3(grid), 256, 0, stream>>>(packed, a0, a1, z, z_stride, tiles_per_ray, rays_mode,
                                                             n_points, groups, reinterpret_cast<float4*>(raw), acts, masks);
    else
        mlp_fwd_kernel<false><<<dim3(grid), 256, 0, stream>>>(packed, a0, a1, z, z_stride, tiles_per_ray, rays_mode,
                                                              n_points, groups, reinterpret_cast<float4*>(raw), nullptr,
                                                              nullptr);
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

// Training forward: nrf_mlp_forward_rays + the saved tensors the backward needs.
NRF_EXPORT int64_t nrf_mlp_train_tiles(int64_t n_rays, int n_samples) {
    if (n_rays < 0 || n_samples <= 0 || n_samples % 32 != 0) return NRF_ERR_SHAPE;
    const int64_t tiles = n_rays * (n_samples / 32);
    return (tiles + 3) / 4 * 4;  // buffers cover whole 4-tile groups
}
NRF_EXPORT int64_t nrf_mlp_acts_bytes(int64_t n_rays, int n_samples) {
    const int64_t t = nrf_mlp_train_tiles(n_rays, n_samples);
    return t < 0 ? t : t * int64_t(kActRows) * 32 * 4;
}
NRF_EXPORT int64_t nrf_mlp_masks_bytes(int64_t n_rays, int n_samples) {
    const int64_t t = nrf_mlp_train_tiles(n_rays, n_samples);
    return t < 0 ? t : t * int64_t(kMaskLayers) * 4 * 64 * 4;
}

NRF_EXPORT int nrf_mlp_forward_rays_train(const float* packed, const float* rays_o, const float* rays_d, const float* z,
                                          int64_t z_stride, int64_t n_rays, int n_samples, float* raw, float* acts,
                                          uint32_t* masks, nrf_stream_t stream) {
    if (n_rays < 0 || n_samples <= 0 || n_samples % 32 != 0 || z_stride < 0) return NRF_ERR_SHAPE;
    if (z_stride != 0 && z_stride < n_samples) return NRF_ERR_SHAPE;
    if (n_rays == 0) return NRF_OK;
    if (int st = check_common(packed, rays_o, rays_d, raw)) return st;
    if (!z || !acts || !masks) return NRF_ERR_NULL;
    if (n_rays > (int64_t(1) << 40) / n_samples) return NRF_ERR_SHAPE;
    return launch_mlp(packed, rays_o, rays_d, z, z_stride, n_samples / 32, 1, n_rays * n_samples, raw,
                      static_cast<hipStream_t>(stream), acts, masks);
}
