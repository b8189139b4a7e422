"""Tensor-level wrappers over the C ABI: marshal ``tensor.data_ptr()`` + sizes + the current
HIP stream into libnerf_amd.so.  No arithmetic happens here; torch only owns the memory."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import check, lib

# order of include/nerf_amd.h `w[12]` / `b[12]` (reference state-dict names, network.py:22-47)
LAYER_NAMES = tuple([f"pts_linears.{i}" for i in range(8)] +
                    ["feature_linear", "alpha_linear", "views_linears.0", "rgb_linear"])
LAYER_SHAPES = ((256, 63), (256, 256), (256, 256), (256, 256), (256, 256), (256, 319), (256, 256), (256, 256),
                (256, 256), (1, 256), (128, 283), (3, 128))


def _stream(t: torch.Tensor):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _f32(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise _lib.NrfError(f"{name}: expected a cuda tensor (the render path has no CPU fallback)")
    if t.dtype != torch.float32:
        raise _lib.NrfError(f"{name}: expected float32, got {t.dtype}")
    return t.contiguous()


def _ptr(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


def packed_floats() -> int:
    return int(lib().nrf_packed_floats())


def pack_weights(weights, biases, out: torch.Tensor | None = None) -> torch.Tensor:
    """12 (out,in) weight tensors + 12 biases (LAYER_NAMES order) -> packed streaming buffer."""
    assert len(weights) == 12 and len(biases) == 12
    ws = [_f32(w.detach(), "weight") for w in weights]
    bs = [_f32(b.detach(), "bias") for b in biases]
    for w, b, shp in zip(ws, bs, LAYER_SHAPES):
        if tuple(w.shape) != shp or tuple(b.shape) != (shp[0],):
            raise _lib.NrfError(f"unexpected layer shape {tuple(w.shape)} (want {shp}): only the 8x256 NeRF MLP "
                                "with 10/4-frequency encodings is supported")
    dev = ws[0].device
    if out is None:
        out = torch.empty(packed_floats(), dtype=torch.float32, device=dev)
    wp = (C.c_void_p * 12)(*[w.data_ptr() for w in ws])
    bp = (C.c_void_p * 12)(*[b.data_ptr() for b in bs])
    with torch.cuda.device(dev):
        check(lib().nrf_pack_weights(wp, bp, _ptr(out), _stream(out)), "nrf_pack_weights")
    return out


def pack_state_dict(sd: dict, prefix: str, device) -> torch.Tensor:
    ws = [torch.as_tensor(sd[f"{prefix}.{n}.weight"]).to(device=device, dtype=torch.float32) for n in LAYER_NAMES]
    bs = [torch.as_tensor(sd[f"{prefix}.{n}.bias"]).to(device=device, dtype=torch.float32) for n in LAYER_NAMES]
    return pack_weights(ws, bs)


def pack_weights_bf16(weights, biases, out: torch.Tensor | None = None) -> torch.Tensor:
    """Same inputs as pack_weights -> the bf16 streaming layout of nrf_mlp_forward_rays_bf16 (a uint8 buffer)."""
    assert len(weights) == 12 and len(biases) == 12
    ws = [_f32(w.detach(), "weight") for w in weights]
    bs = [_f32(b.detach(), "bias") for b in biases]
    for w, b, shp in zip(ws, bs, LAYER_SHAPES):
        if tuple(w.shape) != shp or tuple(b.shape) != (shp[0],):
            raise _lib.NrfError(f"unexpected layer shape {tuple(w.shape)} (want {shp})")
    dev = ws[0].device
    if out is None:
        out = torch.empty(int(lib().nrf_packed_bf16_bytes()), dtype=torch.uint8, device=dev)
    wp = (C.c_void_p * 12)(*[w.data_ptr() for w in ws])
    bp = (C.c_void_p * 12)(*[b.data_ptr() for b in bs])
    with torch.cuda.device(dev):
        check(lib().nrf_pack_weights_bf16(wp, bp, _ptr(out), _stream(out)), "nrf_pack_weights_bf16")
    return out


def pack_state_dict_bf16(sd: dict, prefix: str, device) -> torch.Tensor:
    ws = [torch.as_tensor(sd[f"{prefix}.{n}.weight"]).to(device=device, dtype=torch.float32) for n in LAYER_NAMES]
    bs = [torch.as_tensor(sd[f"{prefix}.{n}.bias"]).to(device=device, dtype=torch.float32) for n in LAYER_NAMES]
    return pack_weights_bf16(ws, bs)


def mlp_forward_rays_bf16(packed_bf16, rays_o, rays_d, z) -> torch.Tensor:
    rays_o, rays_d, z = _f32(rays_o, "rays_o"), _f32(rays_d, "rays_d"), _f32(z, "z")
    R = rays_o.shape[0]
    S = z.shape[-1]
    stride = 0 if z.ndim == 1 else S
    raw = torch.empty((R, S, 4), dtype=torch.float32, device=rays_o.device)
    with torch.cuda.device(raw.device):
        check(lib().nrf_mlp_forward_rays_bf16(_ptr(packed_bf16), _ptr(rays_o), _ptr(rays_d), _ptr(z), stride, R, S,
                                              _ptr(raw), _stream(raw)), "nrf_mlp_forward_rays_bf16")
    return raw


def pack_weights_bf16x3(weights, biases, out: torch.Tensor | None = None) -> torch.Tensor:
    """Same inputs as pack_weights -> the three-term bf16 streaming layout of nrf_mlp_forward_rays_bf16x3 (uint8 buffer)."""
    assert len(weights) == 12 and len(biases) == 12
    ws = [_f32(w.detach(), "weight") for w in weights]
    bs = [_f32(b.detach(), "bias") for b in biases]
    for w, b, shp in zip(ws, bs, LAYER_SHAPES):
        if tuple(w.shape) != shp or tuple(b.shape) != (shp[0],):
            raise _lib.NrfError(f"unexpected layer shape {tuple(w.shape)} (want {shp})")
    dev = ws[0].device
    if out is None:
        out = torch.empty(int(lib().nrf_packed_bf16x3_bytes()), dtype=torch.uint8, device=dev)
    wp = (C.c_void_p * 12)(*[w.data_ptr() for w in ws])
    bp = (C.c_void_p * 12)(*[b.data_ptr() for b in bs])
    with torch.cuda.device(dev):
        check(lib().nrf_pack_weights_bf16x3(wp, bp, _ptr(out), _stream(out)), "nrf_pack_weights_bf16x3")
    return out


def pack_state_dict_bf16x3(sd: dict, prefix: str, device) -> torch.Tensor:
    ws = [torch.as_tensor(sd[f"{prefix}.{n}.weight"]).to(device=device, dtype=torch.float32) for n in LAYER_NAMES]
    bs = [torch.as_tensor(sd[f"{prefix}.{n}.bias"]).to(device=device, dtype=torch.float32) for n in LAYER_NAMES]
    return pack_weights_bf16x3(ws, bs)


def mlp_forward_rays_bf16x3(packed_x3, rays_o, rays_d, z) -> torch.Tensor:
    rays_o, rays_d, z = _f32(rays_o, "rays_o"), _f32(rays_d, "rays_d"), _f32(z, "z")
    R = rays_o.shape[0]
    S = z.shape[-1]
    stride = 0 if z.ndim == 1 else S
    raw = torch.empty((R, S, 4), dtype=torch.float32, device=rays_o.device)
    with torch.cuda.device(raw.device):
        check(lib().nrf_mlp_forward_rays_bf16x3(_ptr(packed_x3), _ptr(rays_o), _ptr(rays_d), _ptr(z), stride, R, S,
                                                _ptr(raw), _stream(raw)), "nrf_mlp_forward_rays_bf16x3")
    return raw


def mlp_forward_points_bf16x3(packed_x3, pts, dirs) -> torch.Tensor:
    pts, dirs = _f32(pts, "pts"), _f32(dirs, "dirs")
    n = pts.shape[0]
    if dirs.shape[0] != n:
        raise _lib.NrfError(f"dirs has {dirs.shape[0]} rows for {n} points")
    raw = torch.empty((n, 4), dtype=torch.float32, device=pts.device)
    with torch.cuda.device(raw.device):
        check(lib().nrf_mlp_forward_points_bf16x3(_ptr(packed_x3), _ptr(pts), _ptr(dirs), n, _ptr(raw), _stream(raw)),
              "nrf_mlp_forward_points_bf16x3")
    return raw


def pack_weights_fp16x2(weights, biases, out: torch.Tensor | None = None, check_range: bool = True) -> torch.Tensor:
    """Same inputs as pack_weights -> the two-term fp16 streaming layout of nrf_mlp_forward_rays_fp16x2 (uint8 buffer).
    ``check_range`` waits for the packer and raises NrfError when a weight lies outside fp16's range (|w| >= 65520 or
    not finite); without it such a buffer makes the forward kernels return NaN."""
    assert len(weights) == 12 and len(biases) == 12
    ws = [_f32(w.detach(), "weight") for w in weights]
    bs = [_f32(b.detach(), "bias") for b in biases]
    for w, b, shp in zip(ws, bs, LAYER_SHAPES):
        if tuple(w.shape) != shp or tuple(b.shape) != (shp[0],):
            raise _lib.NrfError(f"unexpected layer shape {tuple(w.shape)} (want {shp})")
    dev = ws[0].device
    if out is None:
        out = torch.empty(int(lib().nrf_packed_fp16x2_bytes()), dtype=torch.uint8, device=dev)
    wp = (C.c_void_p * 12)(*[w.data_ptr() for w in ws])
    bp = (C.c_void_p * 12)(*[b.data_ptr() for b in bs])
    with torch.cuda.device(dev):
        check(lib().nrf_pack_weights_fp16x2(wp, bp, _ptr(out), _stream(out)), "nrf_pack_weights_fp16x2")
    if check_range:
        with torch.cuda.device(dev):
            check(lib().nrf_packed_fp16x2_status(_ptr(out), _stream(out)), "nrf_pack_weights_fp16x2")
    return out


def pack_state_dict_fp16x2(sd: dict, prefix: str, device, check_range: bool = True) -> torch.Tensor:
    ws = [torch.as_tensor(sd[f"{prefix}.{n}.weight"]).to(device=device, dtype=torch.float32) for n in LAYER_NAMES]
    bs = [torch.as_tensor(sd[f"{prefix}.{n}.bias"]).to(device=device, dtype=torch.float32) for n in LAYER_NAMES]
    return pack_weights_fp16x2(ws, bs, check_range=check_range)


def mlp_forward_rays_fp16x2(packed_h2, rays_o, rays_d, z) -> torch.Tensor:
    rays_o, rays_d, z = _f32(rays_o, "rays_o"), _f32(rays_d, "rays_d"), _f32(z, "z")
    R = rays_o.shape[0]
    S = z.shape[-1]
    stride = 0 if z.ndim == 1 else S
    raw = torch.empty((R, S, 4), dtype=torch.float32, device=rays_o.device)
    with torch.cuda.device(raw.device):
        check(lib().nrf_mlp_forward_rays_fp16x2(_ptr(packed_h2), _ptr(rays_o), _ptr(rays_d), _ptr(z), stride, R, S,
                                                _ptr(raw), _stream(raw)), "nrf_mlp_forward_rays_fp16x2")
    return raw


def mlp_forward_points_fp16x2(packed_h2, pts, dirs) -> torch.Tensor:
    pts, dirs = _f32(pts, "pts"), _f32(dirs, "dirs")
    n = pts.shape[0]
    if dirs.shape[0] != n:
        raise _lib.NrfError(f"dirs has {dirs.shape[0]} rows for {n} points")
    raw = torch.empty((n, 4), dtype=torch.float32, device=pts.device)
    with torch.cuda.device(raw.device):
        check(lib().nrf_mlp_forward_points_fp16x2(_ptr(packed_h2), _ptr(pts), _ptr(dirs), n, _ptr(raw), _stream(raw)),
              "nrf_mlp_forward_points_fp16x2")
    return raw


def raygen(c2w, H: int, W: int, focal: float, device, first_ray: int = 0, n_rays: int | None = None,
           pixel_index: torch.Tensor | None = None):
    c2w = np.ascontiguousarray(np.asarray(c2w, dtype=np.float64)[:3, :4])
    flat = np.ascontiguousarray(np.concatenate([c2w, np.zeros((0, 4))]).reshape(-1))
    # the C side reads rows of 4: [r00 r01 r02 t0, r10 ...]
    if pixel_index is not None:
        n = int(pixel_index.numel())
        pixel_index = pixel_index.to(device=device, dtype=torch.int64).contiguous()
    else:
        n = H * W - first_ray if n_rays is None else int(n_rays)
    xyz = torch.empty((n, 3), dtype=torch.float32, device=device)
    dirs = torch.empty((n, 3), dtype=torch.float32, device=device)
    with torch.cuda.device(device):
        check(lib().nrf_raygen(flat.ctypes.data_as(C.c_void_p), H, W, float(focal), first_ray, n,
                               _ptr(pixel_index), _ptr(xyz), _ptr(dirs), _stream(xyz)), "nrf_raygen")
    return xyz, dirs


def perturb_samples(z: torch.Tensor, u: torch.Tensor) -> torch.Tensor:
    u = _f32(u, "u")
    R, n = u.shape
    z = _f32(z, "z")
    stride = 0 if z.ndim == 1 else n
    out = torch.empty_like(u)
    with torch.cuda.device(u.device):
        check(lib().nrf_perturb_samples(_ptr(z), stride, _ptr(u), R, n, _ptr(out), _stream(u)), "nrf_perturb_samples")
    return out


def mlp_forward_rays(packed, rays_o, rays_d, z) -> torch.Tensor:
    rays_o, rays_d, z = _f32(rays_o, "rays_o"), _f32(rays_d, "rays_d"), _f32(z, "z")
    R = rays_o.shape[0]
    S = z.shape[-1]
    stride = 0 if z.ndim == 1 else S
    raw = torch.empty((R, S, 4), dtype=torch.float32, device=rays_o.device)
    with torch.cuda.device(raw.device):
        check(lib().nrf_mlp_forward_rays(_ptr(packed), _ptr(rays_o), _ptr(rays_d), _ptr(z), stride, R, S, _ptr(raw),
                                         _stream(raw)), "nrf_mlp_forward_rays")
    return raw


def mlp_forward_points(packed, pts, dirs) -> torch.Tensor:
    pts, dirs = _f32(pts, "pts"), _f32(dirs, "dirs")
    n = pts.shape[0]
    raw = torch.empty((n, 4), dtype=torch.float32, device=pts.device)
    with torch.cuda.device(raw.device):
        check(lib().nrf_mlp_forward_points(_ptr(packed), _ptr(pts), _ptr(dirs), n, _ptr(raw), _stream(raw)),
              "nrf_mlp_forward_points")
    return raw


def mlp_forward_points_bf16(packed_bf16, pts, dirs) -> torch.Tensor:
    pts, dirs = _f32(pts, "pts"), _f32(dirs, "dirs")
    n = pts.shape[0]
    if dirs.shape[0] != n:
        raise _lib.NrfError(f"dirs has {dirs.shape[0]} rows for {n} points")
    raw = torch.empty((n, 4), dtype=torch.float32, device=pts.device)
    with torch.cuda.device(raw.device):
        check(lib().nrf_mlp_forward_points_bf16(_ptr(packed_bf16), _ptr(pts), _ptr(dirs), n, _ptr(raw), _stream(raw)),
              "nrf_mlp_forward_points_bf16")
    return raw


def composite(raw, z, white_bkgd: bool = True, want_weights: bool = True):
    raw, z = _f32(raw, "raw"), _f32(z, "z")
    R, S = raw.shape[0], raw.shape[1]
    stride = 0 if z.ndim == 1 else S
    dev = raw.device
    rgb = torch.empty((R, 3), dtype=torch.float32, device=dev)
    depth = torch.empty((R,), dtype=torch.float32, device=dev)
    acc = torch.empty((R, 1), dtype=torch.float32, device=dev)
    w = torch.empty((R, S), dtype=torch.float32, device=dev) if want_weights else None
    with torch.cuda.device(dev):
        check(lib().nrf_composite(_ptr(raw), _ptr(z), stride, R, S, int(bool(white_bkgd)), _ptr(rgb), _ptr(depth),
                                  _ptr(acc), _ptr(w), _stream(raw)), "nrf_composite")
    return rgb, depth, w, acc


def sample_fine(weights, bins, z_coarse, u, u2, u_fine=None, want_debug: bool = False):
    weights, u, u2 = _f32(weights, "weights"), _f32(u, "u"), _f32(u2, "u2")
    bins, z_coarse = _f32(bins, "bins"), _f32(z_coarse, "z_coarse")
    R, nc = weights.shape
    ni = u.shape[1]
    dev = weights.device
    z_out = torch.empty((R, nc + ni), dtype=torch.float32, device=dev)
    fine = torch.empty((R, ni), dtype=torch.float32, device=dev) if want_debug else None
    idx = torch.empty((R, ni), dtype=torch.int32, device=dev) if want_debug else None
    if u_fine is not None:
        u_fine = _f32(u_fine, "u_fine")
    with torch.cuda.device(dev):
        check(lib().nrf_sample_fine(_ptr(weights), _ptr(bins), 0 if bins.ndim == 1 else nc, _ptr(z_coarse),
                                    0 if z_coarse.ndim == 1 else nc, _ptr(u), _ptr(u2), _ptr(u_fine), R, nc, ni,
                                    _ptr(z_out), _ptr(fine), _ptr(idx), _stream(weights)), "nrf_sample_fine")
    return (z_out, fine, idx) if want_debug else z_out


# ---- training step ------------------------------------------------------------------------
def model_param_floats() -> int:
    return int(lib().nrf_model_param_floats())


def pack_weights_transposed(weights, out: torch.Tensor | None = None) -> torch.Tensor:
    ws = [_f32(w.detach(), "weight") for w in weights]
    dev = ws[0].device
    if out is None:
        out = torch.empty(int(lib().nrf_packed_bwd_floats()), dtype=torch.float32, device=dev)
    wp = (C.c_void_p * 12)(*[w.data_ptr() for w in ws])
    with torch.cuda.device(dev):
        check(lib().nrf_pack_weights_transposed(wp, _ptr(out), _stream(out)), "nrf_pack_weights_transposed")
    return out


def pack_weights_transposed_bf16x3(weights, out: torch.Tensor | None = None) -> torch.Tensor:
    """The backward chain's weights as the three-term bf16 stream (nrf_pack_weights_transposed_bf16x3, uint8 buffer)."""
    ws = [_f32(w.detach(), "weight") for w in weights]
    dev = ws[0].device
    if out is None:
        out = torch.empty(int(lib().nrf_packed_bwd_bf16x3_bytes()), dtype=torch.uint8, device=dev)
    wp = (C.c_void_p * 12)(*[w.data_ptr() for w in ws])
    with torch.cuda.device(dev):
        check(lib().nrf_pack_weights_transposed_bf16x3(wp, _ptr(out), _stream(out)), "nrf_pack_weights_transposed_bf16x3")
    return out


def pack_weights_transposed_fp16x2(weights, out: torch.Tensor | None = None) -> torch.Tensor:
    """The backward chain's weights as the two-term fp16 stream (nrf_pack_weights_transposed_fp16x2, uint8 buffer)."""
    ws = [_f32(w.detach(), "weight") for w in weights]
    dev = ws[0].device
    if out is None:
        out = torch.empty(int(lib().nrf_packed_bwd_fp16x2_bytes()), dtype=torch.uint8, device=dev)
    wp = (C.c_void_p * 12)(*[w.data_ptr() for w in ws])
    with torch.cuda.device(dev):
        check(lib().nrf_pack_weights_transposed_fp16x2(wp, _ptr(out), _stream(out)), "nrf_pack_weights_transposed_fp16x2")
    return out


def mlp_forward_rays_train(packed, rays_o, rays_d, z, kind: str = "fp32"):
    """-> (raw (R,S,4), acts, masks): the saved tensors are opaque buffers for mlp_backward.  ``kind``: "fp32"
    (``packed`` from pack_weights) or "bf16x3" (``packed`` from pack_weights_bf16x3: the forward on the three-term
    bf16 kernel, same saved layout) or "fp16x2" (pack_weights_fp16x2: the two-term fp16 kernel, same saved layout)."""
    rays_o, rays_d, z = _f32(rays_o, "rays_o"), _f32(rays_d, "rays_d"), _f32(z, "z")
    R, S = rays_o.shape[0], z.shape[-1]
    stride = 0 if z.ndim == 1 else S
    dev = rays_o.device
    L = lib()
    raw = torch.empty((R, S, 4), dtype=torch.float32, device=dev)
    acts = torch.empty(max(int(L.nrf_mlp_acts_bytes(R, S)), 4) // 4, dtype=torch.float32, device=dev)
    masks = torch.empty(max(int(L.nrf_mlp_masks_bytes(R, S)), 4) // 4, dtype=torch.int32, device=dev)
    if kind not in ("fp32", "bf16x3", "fp16x2"):
        raise ValueError(f"kind {kind!r}: expected 'fp32', 'bf16x3' or 'fp16x2'")
    if S % 32 != 0:
        raise _lib.NrfError(f"n_samples {S}: the MLP kernels take whole 32-point tiles per ray")
    fn = {"fp32": L.nrf_mlp_forward_rays_train, "bf16x3": L.nrf_mlp_forward_rays_train_bf16x3,
          "fp16x2": L.nrf_mlp_forward_rays_train_fp16x2}[kind]
    with torch.cuda.device(dev):
        check(fn(_ptr(packed), _ptr(rays_o), _ptr(rays_d), _ptr(z), stride, R, S, _ptr(raw), _ptr(acts), _ptr(masks),
                 _stream(raw)), "nrf_mlp_forward_rays_train" + ("" if kind == "fp32" else "_" + kind))
    return raw, acts, masks


def composite_backward(raw, z, white_bkgd, g_rgb=None, g_depth=None, g_acc=None) -> torch.Tensor:
    raw, z = _f32(raw, "raw"), _f32(z, "z")
    R, S = raw.shape[0], raw.shape[1]
    stride = 0 if z.ndim == 1 else S
    gs = [None if g is None else _f32(g, "grad").reshape(R, -1) for g in (g_rgb, g_depth, g_acc)]
    g_raw = torch.empty_like(raw)
    with torch.cuda.device(raw.device):
        check(lib().nrf_composite_backward(_ptr(raw), _ptr(z), stride, R, S, int(bool(white_bkgd)), _ptr(gs[0]), _ptr(gs[1]),
                                           _ptr(gs[2]), _ptr(g_raw), _stream(raw)), "nrf_composite_backward")
    return g_raw


def mlp_backward(packed, packed_t, g_raw, acts, masks, workspace: torch.Tensor | None = None,
                 out: torch.Tensor | None = None, kind: str = "fp32") -> torch.Tensor:
    """-> flat gradient (595,844,) of one model in state-dict order (weight then bias per layer).  ``kind`` "bf16x3":
    ``packed_t`` from pack_weights_transposed_bf16x3, the data-gradient chain on the three-term bf16 kernel; "fp16x2":
    pack_weights_transposed_fp16x2, the chain on the two-term fp16 kernel at a device-chosen loss scale."""
    g_raw = _f32(g_raw, "g_raw")
    R, S = g_raw.shape[0], g_raw.shape[1]
    dev = g_raw.device
    L = lib()
    need = int(L.nrf_mlp_backward_workspace_bytes(R, S))
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(max(need, 16), dtype=torch.uint8, device=dev)
    flat = torch.zeros(model_param_floats(), dtype=torch.float32, device=dev) if out is None else out
    if kind not in ("fp32", "bf16x3", "fp16x2"):
        raise ValueError(f"kind {kind!r}: expected 'fp32', 'bf16x3' or 'fp16x2'")
    fn = {"fp32": L.nrf_mlp_backward, "bf16x3": L.nrf_mlp_backward_bf16x3, "fp16x2": L.nrf_mlp_backward_fp16x2}[kind]
    with torch.cuda.device(dev):
        check(fn(_ptr(packed), _ptr(packed_t), _ptr(g_raw), _ptr(acts), _ptr(masks), R, S, _ptr(workspace),
                 workspace.numel(), _ptr(flat), _stream(flat)), "nrf_mlp_backward" + ("" if kind == "fp32" else "_" + kind))
    return flat


def backward_scale_block(workspace: torch.Tensor, n_rays: int, n_samples: int) -> torch.Tensor:
    """The fp16x2 backward's scale block (64 uint32 words as an int32 tensor, a copy) as the last nrf_mlp_backward_fp16x2
    call with these sizes left it in ``workspace``: word 0 holds the bits of max |d raw| of the batch, from which the
    chain picked its power-of-two loss scale (``loss_scale_exponent``), words 1.. the maxima of the scaled dY row sets
    (csrc/split2h.h: kGmax*).  No synchronisation: a device-side copy on the current stream."""
    need = int(lib().nrf_mlp_backward_workspace_bytes(n_rays, n_samples))
    return workspace[need - 256:need].view(torch.int32).clone()


def loss_scale_exponent(max_bits: int, target: int = 4) -> int:
    """S of the chain's loss scale 2^S (csrc/split2h.h: scale_exponent with kChainScaleTarget = 4: max |d raw| lands in
    [16, 32)); 0 for a zero, subnormal or non-finite maximum."""
    e = (int(max_bits) >> 23) & 0xFF
    if e in (0, 255):
        return 0
    return max(-100, min(100, target - (e - 127)))


def split_flat_grad(flat: torch.Tensor):
    """flat (595,844,) -> 12 (weight, bias) pairs shaped like LAYER_SHAPES (views, no copies)."""
    out, o = [], 0
    for (fo, fi) in LAYER_SHAPES:
        w = flat[o:o + fo * fi].view(fo, fi)
        o += fo * fi
        b = flat[o:o + fo]
        o += fo
        out.append((w, b))
    return out


def mse_grad(x: torch.Tensor, target: torch.Tensor, loss_accum: torch.Tensor | None = None) -> torch.Tensor:
    x, target = _f32(x, "x"), _f32(target, "target")
    assert x.numel() == target.numel()
    g = torch.empty_like(x)
    with torch.cuda.device(x.device):
        check(lib().nrf_mse_grad(_ptr(x), _ptr(target), x.numel(), _ptr(g), _ptr(loss_accum), _stream(x)), "nrf_mse_grad")
    return g


def clip_adam(params, grads, exp_avg, exp_avg_sq, step: int, lr=5e-4, beta1=0.9, beta2=0.999, eps=1e-8,
              clip_value=40.0, grad_scale=1.0, reg_begin=None, reg_grad_coeff=0.0, reg_loss_coeff=0.0, loss_accum=None) -> None:
    """In-place on flat fp32 buffers (trainer.py:74-75 + optimizer.py:9-29 defaults).  ``reg_begin``: fold an L2
    regulariser on params[reg_begin:] into the pass (nrf_clip_adam_l2: grads[reg_begin:] += 2 reg_grad_coeff p, written
    back; loss_accum += reg_loss_coeff sum p^2)."""
    for t in (params, grads, exp_avg, exp_avg_sq):
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() == params.numel()
    if reg_begin is not None:
        with torch.cuda.device(params.device):
            check(lib().nrf_clip_adam_l2(_ptr(params), _ptr(grads), _ptr(exp_avg), _ptr(exp_avg_sq), params.numel(),
                                         float(grad_scale), float(clip_value), float(lr), float(beta1), float(beta2),
                                         float(eps), int(step), int(reg_begin), float(reg_grad_coeff), float(reg_loss_coeff),
                                         _ptr(loss_accum), _stream(params)), "nrf_clip_adam_l2")
        return
    with torch.cuda.device(params.device):
        check(lib().nrf_clip_adam(_ptr(params), _ptr(grads), _ptr(exp_avg), _ptr(exp_avg_sq), params.numel(),
                                  float(grad_scale), float(clip_value), float(lr), float(beta1), float(beta2), float(eps),
                                  int(step), _stream(params)), "nrf_clip_adam")
