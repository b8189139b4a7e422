// Occupancy-grid bake from the teacher NeRF: src/models/kilonerf/renderer/volumn_renderer.py:142-232
// (`_evaluate_grid`).  Per grid cell 27 sub-points (centre +- {0, 1/4 cell} per axis, :169-176) are pushed through
// the teacher's fine 8x256 MLP (the fused MFMA kernel of mlp_fp32.hip, zero view directions as the reference's
// dummy_viewdirs :204) and the cell is occupied when any raw sigma exceeds the threshold (:212-220).
// 256^3 cells x 27 = 453 M MLP evaluations = 538 TFLOP: MFMA-bound, a few seconds per bake; the point generator
// and the 27-way OR are HBM-trivial (40 B per point of workspace traffic against 1.19 MFLOP).
#include "nrf_common.h"

namespace {

struct BakeGrid {
    float mn[3], mx[3];
    int res[3];
};

// torch.linspace(start, end, steps)[i] in fp32: step = (end - start) / (steps - 1); the lower half counts up from
// start, the upper half down from end (:159-161).
__device__ __forceinline__ float linspace_at(float start, float end, int steps, int i) {
    const float step = __fdiv_rn(__fsub_rn(end, start), float(steps - 1));
    return i < steps / 2 ? __fadd_rn(start, __fmul_rn(step, float(i))) : __fsub_rn(end, __fmul_rn(step, float(steps - i - 1)));
}

__global__ void occ_points_kernel(BakeGrid g, int64_t first_cell, int64_t n_cells, float* __restrict__ pts) {
    const int64_t p = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (p >= n_cells * 27) return;
    const int64_t cell = first_cell + p / 27;
    const int sub = int(p % 27);
    const int idx[3] = {int(cell / (int64_t(g.res[1]) * g.res[2])), int((cell / g.res[2]) % g.res[1]), int(cell % g.res[2])};
    const int o[3] = {sub / 9, (sub / 3) % 3, sub % 3};  // meshgrid 'ij' of {-0.25, 0, 0.25}^3, flattened (:170-174)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float cs = __fdiv_rn(__fsub_rn(g.mx[a], g.mn[a]), float(g.res[a]));                          // :156
        const float centre = __fadd_rn(linspace_at(g.mn[a], g.mx[a], g.res[a] + 1, idx[a]), __fmul_rn(cs, 0.5f));  // :159-164
        const float off = __fmul_rn(float(o[a] - 1) * 0.25f, cs);                                          // :176
        pts[p * 3 + a] = __fadd_rn(centre, off);                                                            // :194
    }
}

// The NeRF renderer's own grid (src/models/nerf/renderer/volume_renderer.py:156-197, `_build_occ_grid`): ONE point per
// voxel at torch.linspace(min, max, res)[i] per axis (:170-175, grid VERTICES spanning the whole box, not centres).
__global__ void occ_vertex_points_kernel(BakeGrid g, int64_t first_cell, int64_t n_cells, float* __restrict__ pts) {
    const int64_t p = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (p >= n_cells) return;
    const int64_t cell = first_cell + p;
    const int idx[3] = {int(cell / (int64_t(g.res[1]) * g.res[2])), int((cell / g.res[2]) % g.res[1]), int(cell % g.res[2])};
#pragma unroll
    for (int a = 0; a < 3; ++a) pts[p * 3 + a] = g.res[a] > 1 ? linspace_at(g.mn[a], g.mx[a], g.res[a], idx[a]) : g.mn[a];
}

template <int SUB>
__global__ void occ_reduce_kernel(const float4* __restrict__ raw, int64_t n_cells, float threshold,
                                  unsigned char* __restrict__ occ) {
    const int64_t c = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (c >= n_cells) return;
    bool any = false;
#pragma unroll
    for (int s = 0; s < SUB; ++s) any |= raw[c * SUB + s].w > threshold;  // :217-220 (kilonerf), :195 (nerf)
    occ[c] = any ? 1 : 0;
}

inline int64_t a256(int64_t n) { return (n + 255) & ~int64_t(255); }

}  // namespace

// workspace: pts (sub n x 3) | zero dirs (sub n x 3) | raw (sub n x 4)
static int64_t bake_workspace_bytes(int64_t n_cells, int sub) {
    if (n_cells < 0 || n_cells > (int64_t(1) << 32)) return NRF_ERR_SHAPE;
    return 2 * a256(n_cells * sub * 12) + a256(n_cells * sub * 16);
}

static int bake(const float* packed, const float* aabb_host, const int* res_host, float threshold, int64_t first_cell,
                int64_t n_cells, uint8_t* occ, void* workspace, int64_t workspace_bytes, bool vertices, nrf_stream_t stream_,
                int kind = 0 /* teacher kernel: 0 fp32 MFMA, 1 fp16x2, 2 bf16x3 */) {
    if (n_cells < 0 || first_cell < 0) return NRF_ERR_SHAPE;
    if (n_cells == 0) return NRF_OK;
    if (!packed || !aabb_host || !res_host || !occ || !workspace) return NRF_ERR_NULL;
    BakeGrid g;
    for (int a = 0; a < 3; ++a) g.mn[a] = aabb_host[a], g.mx[a] = aabb_host[3 + a], g.res[a] = res_host[a];
    if (g.res[0] <= 0 || g.res[1] <= 0 || g.res[2] <= 0) return NRF_ERR_SHAPE;
    if (first_cell + n_cells > int64_t(g.res[0]) * g.res[1] * g.res[2]) return NRF_ERR_SHAPE;
    const int sub = vertices ? 1 : 27;
    const int64_t need = bake_workspace_bytes(n_cells, sub);
    if (need < 0) return int(need);
    if (workspace_bytes < need) return NRF_ERR_WORKSPACE;
    if (!nrf_aligned16(workspace)) return NRF_ERR_ALIGN;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int64_t np = n_cells * sub;
    char* wsb = static_cast<char*>(workspace);
    float* pts = reinterpret_cast<float*>(wsb);
    float* dirs = reinterpret_cast<float*>(wsb + a256(np * 12));
    float* raw = reinterpret_cast<float*>(wsb + 2 * a256(np * 12));
    nrf_launch_begin();
    if (hipMemsetAsync(dirs, 0, size_t(np) * 12, stream) != hipSuccess) return NRF_ERR_LAUNCH;
    const dim3 pgrid(unsigned((np + 255) / 256)), cgrid(unsigned((n_cells + 255) / 256));
    if (vertices) occ_vertex_points_kernel<<<pgrid, 256, 0, stream>>>(g, first_cell, n_cells, pts);
    else occ_points_kernel<<<pgrid, 256, 0, stream>>>(g, first_cell, n_cells, pts);
    if (int st = nrf_launch_status()) return st;
    if (int st = kind == 1   ? nrf_mlp_forward_points_fp16x2(packed, pts, dirs, np, raw, stream_)
                 : kind == 2 ? nrf_mlp_forward_points_bf16x3(packed, pts, dirs, np, raw, stream_)
                             : nrf_mlp_forward_points(packed, pts, dirs, np, raw, stream_))
        return st;
    nrf_launch_begin();
    if (vertices)
        occ_reduce_kernel<1><<<cgrid, 256, 0, stream>>>(reinterpret_cast<const float4*>(raw), n_cells, threshold, occ + first_cell);
    else
        occ_reduce_kernel<27><<<cgrid, 256, 0, stream>>>(reinterpret_cast<const float4*>(raw), n_cells, threshold, occ + first_cell);
    return nrf_launch_status();
}

NRF_EXPORT int64_t nrf_occ_bake_workspace_bytes(int64_t n_cells) { return bake_workspace_bytes(n_cells, 27); }

NRF_EXPORT int nrf_occ_bake(const float* packed_teacher, const float* aabb_host, const int* res_host, float threshold,
                            int64_t first_cell, int64_t n_cells, uint8_t* occ, void* workspace, int64_t workspace_bytes,
                            nrf_stream_t stream) {
    return bake(packed_teacher, aabb_host, res_host, threshold, first_cell, n_cells, occ, workspace, workspace_bytes, false,
                stream);
}

// nrf_occ_bake with the teacher evaluated by nrf_mlp_forward_points_fp16x2 (packed_teacher_h2: nrf_pack_weights_fp16x2)
NRF_EXPORT int nrf_occ_bake_fp16x2(const void* packed_teacher_h2, const float* aabb_host, const int* res_host, float threshold,
                                   int64_t first_cell, int64_t n_cells, uint8_t* occ, void* workspace, int64_t workspace_bytes,
                                   nrf_stream_t stream) {
    return bake(static_cast<const float*>(packed_teacher_h2), aabb_host, res_host, threshold, first_cell, n_cells, occ, workspace,
                workspace_bytes, false, stream, 1);
}

// nrf_occ_bake with the teacher evaluated by nrf_mlp_forward_points_bf16x3 (packed_teacher_x3: nrf_pack_weights_bf16x3):
// the fp32 kernel's 24-bit operands and range on the bf16 matrix cores
NRF_EXPORT int nrf_occ_bake_bf16x3(const void* packed_teacher_x3, const float* aabb_host, const int* res_host, float threshold,
                                   int64_t first_cell, int64_t n_cells, uint8_t* occ, void* workspace, int64_t workspace_bytes,
                                   nrf_stream_t stream) {
    return bake(static_cast<const float*>(packed_teacher_x3), aabb_host, res_host, threshold, first_cell, n_cells, occ, workspace,
                workspace_bytes, false, stream, 2);
}

NRF_EXPORT int64_t nrf_occ_bake_vertices_workspace_bytes(int64_t n_cells) { return bake_workspace_bytes(n_cells, 1); }

NRF_EXPORT int nrf_occ_bake_vertices(const float* packed, const float* aabb_host, const int* res_host, float threshold,
                                     int64_t first_cell, int64_t n_cells, uint8_t* occ, void* workspace,
                                     int64_t workspace_bytes, nrf_stream_t stream) {
    return bake(packed, aabb_host, res_host, threshold, first_cell, n_cells, occ, workspace, workspace_bytes, true, stream);
}
