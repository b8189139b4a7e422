"""ctypes binding of libnerf_amd.so (the C ABI declared in include/nerf_amd.h).

There is no fallback: if the HIP library is missing or fails to load, importing the
renderer raises.  Build it with ``make -C nerf_amd/csrc`` (or ``__graft_entry__.build()``).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libnerf_amd.so")

# every symbol include/nerf_amd.h declares (tests/test_cabi_symbols.py checks the two agree)
SYMBOLS = (
    "nrf_version", "nrf_strerror", "nrf_packed_floats", "nrf_pack_weights", "nrf_raygen",
    "nrf_perturb_samples", "nrf_mlp_forward_rays", "nrf_mlp_forward_points", "nrf_composite",
    "nrf_sample_fine", "nrf_render_workspace_bytes", "nrf_render", "nrf_render_bf16",
    "nrf_packed_bwd_floats", "nrf_pack_weights_transposed", "nrf_mlp_train_tiles", "nrf_mlp_acts_bytes",
    "nrf_mlp_masks_bytes", "nrf_mlp_forward_rays_train", "nrf_composite_backward", "nrf_model_param_floats",
    "nrf_mlp_backward_workspace_bytes", "nrf_mlp_backward", "nrf_mse_grad", "nrf_clip_adam", "nrf_clip_adam_l2", "nrf_kilo_forward_packing", "nrf_kilo_pack_weights_cells", "nrf_kilo_train_workspace_bytes", "nrf_kilo_train_forward_backward", "nrf_kilo_clip_adam", "nrf_mlp_forward_rays_train_bf16x3", "nrf_packed_bwd_bf16x3_bytes", "nrf_pack_weights_transposed_bf16x3",
    "nrf_mlp_backward_bf16x3",
    "nrf_kilo_packed_floats", "nrf_kilo_pack_weights", "nrf_kilo_route", "nrf_kilo_forward_workspace_bytes",
    "nrf_kilo_forward", "nrf_kilo_march_workspace_bytes", "nrf_kilo_march_count", "nrf_kilo_march_write",
    "nrf_kilo_composite", "nrf_occ_bake_workspace_bytes", "nrf_occ_bake",
    "nrf_occ_bake_vertices_workspace_bytes", "nrf_occ_bake_vertices",
    "nrf_packed_bf16_bytes", "nrf_pack_weights_bf16", "nrf_mlp_forward_rays_bf16", "nrf_mlp_forward_points_bf16",
    "nrf_kilo_packed_bwd_floats", "nrf_kilo_pack_weights_transposed", "nrf_kilo_composite_backward",
    "nrf_kilo_backward_workspace_bytes", "nrf_kilo_backward", "nrf_l2_reg",
    "nrf_kilo_record_floats", "nrf_kilo_cells_gather", "nrf_kilo_cells_scatter",
    "nrf_render_static",
    "nrf_packed_bf16x3_bytes", "nrf_pack_weights_bf16x3", "nrf_mlp_forward_rays_bf16x3", "nrf_mlp_forward_points_bf16x3",
    "nrf_render_bf16x3",
    "nrf_packed_fp16x2_bytes", "nrf_pack_weights_fp16x2", "nrf_packed_fp16x2_status", "nrf_mlp_forward_rays_fp16x2", "nrf_mlp_forward_points_fp16x2",
    "nrf_render_fp16x2", "nrf_mlp_forward_rays_train_fp16x2", "nrf_kilo_pack_weights_fp16x2", "nrf_kilo_forward_fp16x2", "nrf_occ_bake_fp16x2", "nrf_occ_bake_bf16x3", "nrf_packed_bwd_fp16x2_bytes", "nrf_pack_weights_transposed_fp16x2", "nrf_mlp_backward_fp16x2",
    "nrf_kilo_dir_table_bytes", "nrf_kilo_encode_dirs", "nrf_kilo_forward_encoded",
    "nrf_kilo_packed_bf16x3_floats", "nrf_kilo_pack_weights_bf16x3", "nrf_kilo_forward_bf16x3",
    "nrf_occ_bricks_bytes", "nrf_occ_pack_bricks", "nrf_kilo_march_count_bricks",
)

P = C.c_void_p
I64 = C.c_int64
INT = C.c_int


class RenderArgs(C.Structure):
    """struct nrf_render_args (include/nerf_amd.h)."""
    _fields_ = [
        ("packed_coarse", P), ("packed_fine", P), ("rays_o", P), ("rays_d", P), ("bins", P),
        ("u", P), ("u2", P), ("u_coarse", P), ("u_fine", P),
        ("n_rays", I64), ("n_coarse", INT), ("n_importance", INT), ("white_bkgd", INT),
        ("coarse_rgb", P), ("coarse_depth", P), ("fine_rgb", P), ("fine_depth", P), ("fine_acc", P),
        ("workspace", P), ("workspace_bytes", I64),
    ]


class KiloTrainArgs(C.Structure):
    """struct nrf_kilo_train_args (include/nerf_amd.h)."""
    _fields_ = [
        ("rays_o", P), ("rays_d", P), ("n_rays", I64), ("occ", P), ("occ_res", P), ("aabb", P), ("res", P),
        ("num_cells", INT), ("t_near", C.c_float), ("t_far", C.c_float), ("step", C.c_float), ("ert_eps", C.c_float),
        ("white_bkgd", INT), ("target_rgb", P), ("target_alpha", P), ("tensors", P), ("grads", P), ("reach", P),
        ("packed", P), ("packed_t", P), ("rgb_map", P), ("depth_map", P), ("acc_map", P), ("loss_accum", P),
        ("touched", P), ("live", P), ("n_samples", P), ("max_samples", I64), ("workspace", P), ("workspace_bytes", I64),
    ]


class NrfError(RuntimeError):
    pass


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NrfError(f"{LIB_PATH} not built: run `make -C nerf_amd/csrc` (hipcc --offload-arch=gfx950); "
                       "there is no CPU/PyTorch fallback for the render path")
    # torch bundles its own libamdhip64.so.7; it must be the one already mapped when our library's
    # NEEDED entry is resolved, or the process ends up with two HIP runtimes (launches then fail).
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    L.nrf_version.restype = INT
    L.nrf_strerror.restype = C.c_char_p
    L.nrf_strerror.argtypes = [INT]
    L.nrf_packed_floats.restype = I64
    L.nrf_pack_weights.argtypes = [P, P, P, P]
    L.nrf_raygen.argtypes = [P, INT, INT, C.c_double, I64, I64, P, P, P, P]
    L.nrf_perturb_samples.argtypes = [P, I64, P, I64, INT, P, P]
    L.nrf_mlp_forward_rays.argtypes = [P, P, P, P, I64, I64, INT, P, P]
    L.nrf_mlp_forward_points.argtypes = [P, P, P, I64, P, P]
    L.nrf_composite.argtypes = [P, P, I64, I64, INT, INT, P, P, P, P, P]
    L.nrf_sample_fine.argtypes = [P, P, I64, P, I64, P, P, P, I64, INT, INT, P, P, P, P]
    L.nrf_render_workspace_bytes.restype = I64
    L.nrf_render_workspace_bytes.argtypes = [I64, INT, INT]
    L.nrf_render.argtypes = [C.POINTER(RenderArgs), P]
    L.nrf_render_bf16.argtypes = [C.POINTER(RenderArgs), P]
    L.nrf_render_bf16.restype = INT
    L.nrf_render_bf16x3.argtypes = [C.POINTER(RenderArgs), P]
    L.nrf_render_bf16x3.restype = INT
    L.nrf_render_fp16x2.argtypes = [C.POINTER(RenderArgs), P]
    L.nrf_render_fp16x2.restype = INT
    L.nrf_packed_fp16x2_bytes.restype = I64
    L.nrf_packed_fp16x2_bytes.argtypes = []
    L.nrf_pack_weights_fp16x2.restype = INT
    L.nrf_packed_fp16x2_status.restype = INT
    L.nrf_mlp_forward_rays_train_fp16x2.argtypes = [P, P, P, P, I64, I64, INT, P, P, P, P]
    L.nrf_mlp_forward_rays_train_fp16x2.restype = INT
    L.nrf_packed_fp16x2_status.argtypes = [P, P]
    L.nrf_pack_weights_fp16x2.argtypes = [P, P, P, P]
    L.nrf_mlp_forward_rays_fp16x2.restype = INT
    L.nrf_mlp_forward_rays_fp16x2.argtypes = [P, P, P, P, I64, I64, INT, P, P]
    L.nrf_mlp_forward_points_fp16x2.restype = INT
    L.nrf_mlp_forward_points_fp16x2.argtypes = [P, P, P, I64, P, P]
    L.nrf_packed_bf16x3_bytes.restype = I64
    L.nrf_packed_bf16x3_bytes.argtypes = []
    L.nrf_pack_weights_bf16x3.restype = INT
    L.nrf_pack_weights_bf16x3.argtypes = [P, P, P, P]
    L.nrf_mlp_forward_rays_bf16x3.restype = INT
    L.nrf_mlp_forward_rays_bf16x3.argtypes = [P, P, P, P, I64, I64, INT, P, P]
    L.nrf_mlp_forward_points_bf16x3.restype = INT
    L.nrf_mlp_forward_points_bf16x3.argtypes = [P, P, P, I64, P, P]
    L.nrf_render_static.argtypes = [C.POINTER(RenderArgs), P]
    L.nrf_render_static.restype = INT
    L.nrf_packed_bwd_floats.restype = I64
    L.nrf_pack_weights_transposed.argtypes = [P, P, P]
    for name in ("nrf_mlp_train_tiles", "nrf_mlp_acts_bytes", "nrf_mlp_masks_bytes", "nrf_mlp_backward_workspace_bytes"):
        getattr(L, name).restype = I64
        getattr(L, name).argtypes = [I64, INT]
    L.nrf_model_param_floats.restype = I64
    L.nrf_mlp_forward_rays_train.argtypes = [P, P, P, P, I64, I64, INT, P, P, P, P]
    L.nrf_composite_backward.argtypes = [P, P, I64, I64, INT, INT, P, P, P, P, P]
    L.nrf_mlp_backward.argtypes = [P, P, P, P, P, I64, INT, P, I64, P, P]
    F = C.c_float
    L.nrf_mse_grad.argtypes = [P, P, I64, P, P, P]
    L.nrf_clip_adam.argtypes = [P, P, P, P, I64, F, F, F, F, F, F, INT, P]
    L.nrf_mse_grad.restype = INT
    L.nrf_clip_adam.restype = INT
    L.nrf_clip_adam_l2.argtypes = [P, P, P, P, I64, F, F, F, F, F, F, INT, I64, F, F, P, P]
    L.nrf_clip_adam_l2.restype = INT
    L.nrf_kilo_forward_packing.argtypes = [P, P, P, P, INT, P, P, P, P, P, I64, P, P, I64, P]
    L.nrf_kilo_forward_packing.restype = INT
    L.nrf_kilo_pack_weights_cells.argtypes = [P, INT, P, P, P, P]
    L.nrf_kilo_pack_weights_cells.restype = INT
    L.nrf_kilo_train_workspace_bytes.argtypes = [I64, I64, INT, C.c_float, C.c_float, C.c_float]
    L.nrf_kilo_train_workspace_bytes.restype = I64
    L.nrf_kilo_train_forward_backward.argtypes = [C.POINTER(KiloTrainArgs), P]
    L.nrf_kilo_train_forward_backward.restype = INT
    L.nrf_kilo_clip_adam.argtypes = [P, P, P, P, INT, P, F, F, F, F, F, F, INT, F, F, P, P]
    L.nrf_kilo_clip_adam.restype = INT
    L.nrf_mlp_forward_rays_train_bf16x3.argtypes = [P, P, P, P, I64, I64, INT, P, P, P, P]
    L.nrf_mlp_forward_rays_train_bf16x3.restype = INT
    L.nrf_packed_bwd_bf16x3_bytes.restype = I64
    L.nrf_packed_bwd_bf16x3_bytes.argtypes = []
    L.nrf_pack_weights_transposed_bf16x3.argtypes = [P, P, P]
    L.nrf_pack_weights_transposed_bf16x3.restype = INT
    L.nrf_mlp_backward_bf16x3.argtypes = [P, P, P, P, P, I64, INT, P, I64, P, P]
    L.nrf_mlp_backward_bf16x3.restype = INT
    L.nrf_packed_bwd_fp16x2_bytes.restype = I64
    L.nrf_packed_bwd_fp16x2_bytes.argtypes = []
    L.nrf_pack_weights_transposed_fp16x2.argtypes = [P, P, P]
    L.nrf_pack_weights_transposed_fp16x2.restype = INT
    L.nrf_mlp_backward_fp16x2.argtypes = [P, P, P, P, P, I64, INT, P, I64, P, P]
    L.nrf_mlp_backward_fp16x2.restype = INT
    L.nrf_packed_bf16_bytes.restype = I64
    L.nrf_packed_bf16_bytes.argtypes = []
    L.nrf_pack_weights_bf16.restype = INT
    L.nrf_pack_weights_bf16.argtypes = [P, P, P, P]
    L.nrf_mlp_forward_rays_bf16.restype = INT
    L.nrf_mlp_forward_rays_bf16.argtypes = [P, P, P, P, I64, I64, INT, P, P]
    L.nrf_mlp_forward_points_bf16.restype = INT
    L.nrf_mlp_forward_points_bf16.argtypes = [P, P, P, I64, P, P]
    L.nrf_occ_bake_workspace_bytes.restype = I64
    L.nrf_occ_bake_workspace_bytes.argtypes = [I64]
    L.nrf_occ_bake.restype = INT
    L.nrf_occ_bake.argtypes = [P, P, P, F, I64, I64, P, P, I64, P]
    L.nrf_occ_bake_fp16x2.restype = INT
    L.nrf_occ_bake_fp16x2.argtypes = [P, P, P, F, I64, I64, P, P, I64, P]
    L.nrf_occ_bake_bf16x3.restype = INT
    L.nrf_occ_bake_bf16x3.argtypes = [P, P, P, F, I64, I64, P, P, I64, P]
    L.nrf_occ_bake_vertices_workspace_bytes.restype = I64
    L.nrf_occ_bake_vertices_workspace_bytes.argtypes = [I64]
    L.nrf_occ_bake_vertices.restype = INT
    L.nrf_occ_bake_vertices.argtypes = [P, P, P, F, I64, I64, P, P, I64, P]
    L.nrf_kilo_packed_floats.restype = I64
    L.nrf_kilo_packed_floats.argtypes = [INT]
    L.nrf_kilo_pack_weights.argtypes = [P, INT, P, P]
    L.nrf_kilo_pack_weights_fp16x2.argtypes = [P, INT, P, P]
    L.nrf_kilo_pack_weights_fp16x2.restype = INT
    L.nrf_kilo_forward_fp16x2.argtypes = [P, INT, P, P, P, P, P, I64, P, P, I64, P]
    L.nrf_kilo_forward_fp16x2.restype = INT
    L.nrf_kilo_packed_bf16x3_floats.restype = I64
    L.nrf_kilo_packed_bf16x3_floats.argtypes = [INT]
    L.nrf_kilo_pack_weights_bf16x3.argtypes = [P, INT, P, P]
    L.nrf_kilo_pack_weights_bf16x3.restype = INT
    L.nrf_kilo_forward_bf16x3.argtypes = [P, INT, P, P, P, P, P, I64, P, P, I64, P]
    L.nrf_kilo_forward_bf16x3.restype = INT
    L.nrf_kilo_dir_table_bytes.restype = I64
    L.nrf_kilo_dir_table_bytes.argtypes = [I64]
    L.nrf_kilo_encode_dirs.argtypes = [P, I64, INT, P, P]
    L.nrf_kilo_encode_dirs.restype = INT
    L.nrf_kilo_forward_encoded.argtypes = [P, INT, INT, P, P, P, P, I64, P, I64, P, P, I64, P]
    L.nrf_kilo_forward_encoded.restype = INT
    L.nrf_kilo_route.argtypes = [P, I64, P, P, P, P]
    L.nrf_kilo_forward_workspace_bytes.restype = I64
    L.nrf_kilo_forward_workspace_bytes.argtypes = [I64, INT]
    L.nrf_kilo_forward.argtypes = [P, INT, P, P, P, P, P, I64, P, P, I64, P]
    L.nrf_kilo_march_workspace_bytes.restype = I64
    L.nrf_kilo_march_workspace_bytes.argtypes = [I64, F, F, F]
    L.nrf_kilo_march_count.argtypes = [P, P, I64, P, P, P, F, F, F, P, P, P, P, I64, P]
    L.nrf_kilo_march_write.argtypes = [P, P, I64, P, P, P, F, F, F, P, P, P, P, P, P, P, I64, P]
    L.nrf_occ_bricks_bytes.restype = I64
    L.nrf_occ_bricks_bytes.argtypes = [P]
    L.nrf_occ_pack_bricks.argtypes = [P, P, P, P]
    L.nrf_occ_pack_bricks.restype = INT
    L.nrf_kilo_march_count_bricks.argtypes = [P, P, I64, P, P, P, F, F, F, P, P, P, P, I64, P]
    L.nrf_kilo_march_count_bricks.restype = INT
    L.nrf_kilo_composite.argtypes = [P, P, P, P, P, I64, F, INT, P, P, P, P]
    L.nrf_kilo_packed_bwd_floats.restype = I64
    L.nrf_kilo_packed_bwd_floats.argtypes = [INT]
    L.nrf_kilo_pack_weights_transposed.argtypes = [P, INT, P, P]
    L.nrf_kilo_composite_backward.argtypes = [P, P, P, P, P, I64, F, INT, P, P, P, P, P]
    L.nrf_kilo_backward_workspace_bytes.restype = I64
    L.nrf_kilo_backward_workspace_bytes.argtypes = [I64, INT]
    L.nrf_kilo_backward.argtypes = [P, P, INT, P, P, P, I64, P, P, I64, P, P, I64, P]
    L.nrf_l2_reg.argtypes = [P, I64, F, P, P, P]
    L.nrf_kilo_record_floats.restype = I64
    L.nrf_kilo_record_floats.argtypes = []
    L.nrf_kilo_cells_gather.argtypes = [P, P, INT, INT, P, P]
    L.nrf_kilo_cells_scatter.argtypes = [P, P, INT, INT, P, P]
    L.nrf_kilo_cells_gather.restype = INT
    L.nrf_kilo_cells_scatter.restype = INT
    for name in ("nrf_kilo_pack_weights_transposed", "nrf_kilo_composite_backward", "nrf_kilo_backward", "nrf_l2_reg"):
        getattr(L, name).restype = INT
    for name in ("nrf_kilo_pack_weights", "nrf_kilo_route", "nrf_kilo_forward", "nrf_kilo_march_count",
                 "nrf_kilo_march_write", "nrf_kilo_composite"):
        getattr(L, name).restype = INT
    for name in ("nrf_pack_weights_transposed", "nrf_mlp_forward_rays_train", "nrf_composite_backward", "nrf_mlp_backward"):
        getattr(L, name).restype = INT
    for name in ("nrf_pack_weights", "nrf_raygen", "nrf_perturb_samples", "nrf_mlp_forward_rays",
                 "nrf_mlp_forward_points", "nrf_composite", "nrf_sample_fine", "nrf_render"):
        getattr(L, name).restype = INT
    _lib = L
    return L


def check(status: int, what: str) -> None:
    if status != 0:
        raise NrfError(f"{what}: {lib().nrf_strerror(status).decode()} (status {status})")
