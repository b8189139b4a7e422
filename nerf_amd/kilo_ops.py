"""Tensor-level wrappers over the KiloNeRF entry points of the C ABI (include/nerf_amd.h, nrf_kilo_*).
torch only owns the memory; all arithmetic happens in libnerf_amd.so."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from ._lib import KiloTrainArgs, NrfError, check, lib
from .ops import _f32, _ptr, _stream
from .weights import KILO_TENSORS


MAX_POINTS = 1 << 28        # n_points limit of nrf_kilo_forward (int32 offsets, 2^28 x 16 B of raw)


class TooManySamples(NrfError):
    """The march of one call kept more samples than the packed path supports; render fewer rays per call."""

    def __init__(self, total: int, n_rays: int):
        super().__init__(f"kilo_march: {total} samples for {n_rays} rays exceed the {MAX_POINTS} a single call supports "
                         "(dense occupancy grid / fine step): split the rays over several calls")
        self.total = total


def _grid_args(aabb, res):
    aabb = np.ascontiguousarray(np.asarray(aabb, dtype=np.float32).reshape(6))
    res = np.ascontiguousarray(np.asarray(res, dtype=np.int32).reshape(3))
    return aabb, res, aabb.ctypes.data_as(C.c_void_p), res.ctypes.data_as(C.c_void_p)


KILO_KINDS = {"fp32": 0, "fp16x2": 1, "bf16x3": 2}   # the C ABI's `kind` argument (include/nerf_amd.h)


def kilo_pack_weights(tensors, out: torch.Tensor | None = None, kind: str = "fp32") -> torch.Tensor:
    """12 batched (cells, ...) tensors in KILO_TENSORS order -> per-cell MFMA streaming layout.  ``kind`` "fp16x2": the
    records of nrf_kilo_forward_fp16x2 (same size; every matrix weight as two fp16 terms); "bf16x3": the records of
    nrf_kilo_forward_bf16x3 (every matrix weight as three bf16 terms = exactly its fp32 value; 1.5x the size)."""
    assert len(tensors) == 12
    if kind not in KILO_KINDS:
        raise ValueError(f"kind {kind!r}: expected one of {sorted(KILO_KINDS)}")
    ts = [_f32(t.detach(), "kilo tensor") for t in tensors]
    num_cells = ts[0].shape[0]
    for t, (name, shape, _) in zip(ts, KILO_TENSORS):
        if tuple(t.shape) != (num_cells,) + shape:
            raise NrfError(f"{name}: shape {tuple(t.shape)}, want {(num_cells,) + shape} "
                           "(only D=2, W=32 cells with 10/4-frequency encodings are supported)")
    dev = ts[0].device
    L = lib()
    n = int((L.nrf_kilo_packed_bf16x3_floats if kind == "bf16x3" else L.nrf_kilo_packed_floats)(num_cells))
    if out is None or out.numel() != n:
        out = torch.empty(n, dtype=torch.float32, device=dev)
    ptrs = (C.c_void_p * 12)(*[t.data_ptr() for t in ts])
    with torch.cuda.device(dev):
        fn = {"fp32": L.nrf_kilo_pack_weights, "fp16x2": L.nrf_kilo_pack_weights_fp16x2, "bf16x3": L.nrf_kilo_pack_weights_bf16x3}[kind]
        check(fn(ptrs, num_cells, _ptr(out), _stream(out)), "nrf_kilo_pack_weights" + ("" if kind == "fp32" else "_" + kind))
    return out


def kilo_pack_state_dict(sd: dict, device, prefix: str = "model.", kind: str = "fp32") -> torch.Tensor:
    return kilo_pack_weights([torch.as_tensor(sd[prefix + n]).to(device=device, dtype=torch.float32)
                              for n, _, _ in KILO_TENSORS], kind=kind)


def kilo_route(pts: torch.Tensor, aabb, res) -> torch.Tensor:
    pts = _f32(pts, "pts").reshape(-1, 3)
    _a, _r, pa, pr = _grid_args(aabb, res)
    cell = torch.empty(pts.shape[0], dtype=torch.int32, device=pts.device)
    with torch.cuda.device(pts.device):
        check(lib().nrf_kilo_route(_ptr(pts), pts.shape[0], pa, pr, _ptr(cell), _stream(pts)), "nrf_kilo_route")
    return cell


class ParamPointers:
    """The 12 batched tensors as the host array of device pointers the C ABI takes, built once for tensors whose
    storage does not move (KiloTrainStep's views of its flat buffer): a training step hands it to kilo_forward /
    kilo_pack_weights_cells without re-checking twelve tensors per call."""

    def __init__(self, tensors):
        ts = [_f32(t.detach(), "kilo tensor") for t in tensors]
        if len(ts) != 12:
            raise NrfError("expected the 12 batched KiloNeRF tensors")
        self.num_cells = int(ts[0].shape[0])
        for t, (name, shape, _) in zip(ts, KILO_TENSORS):
            if tuple(t.shape) != (self.num_cells,) + shape:
                raise NrfError(f"{name}: shape {tuple(t.shape)}, want {(self.num_cells,) + shape}")
        self.tensors = ts                                     # keeps the storage alive
        self.array = (C.c_void_p * 12)(*[t.data_ptr() for t in ts])
        self.device = ts[0].device


def _param_pointers(p) -> "ParamPointers":
    return p if isinstance(p, ParamPointers) else ParamPointers(p)


def kilo_pack_weights_cells(params, cell_mask: torch.Tensor, packed: torch.Tensor, packed_t: torch.Tensor | None = None):
    """Records of the cells with cell_mask != 0 (int32, one per cell) from the 12 tensors into ``packed`` (and the
    transposed layout into ``packed_t``); other records are left alone (nrf_kilo_pack_weights_cells)."""
    pp = _param_pointers(params)
    if cell_mask.dtype != torch.int32 or cell_mask.numel() != pp.num_cells or not cell_mask.is_cuda or not cell_mask.is_contiguous():
        raise NrfError("cell_mask: expected a contiguous int32 cuda tensor with one entry per cell")
    if packed.numel() != kilo_packed_floats(pp.num_cells) or (packed_t is not None and
                                                              packed_t.numel() != kilo_packed_bwd_floats(pp.num_cells)):
        raise NrfError("packed / packed_t: wrong size for this grid")
    with torch.cuda.device(pp.device):
        check(lib().nrf_kilo_pack_weights_cells(pp.array, pp.num_cells, _ptr(cell_mask), _ptr(packed), _ptr(packed_t),
                                                _stream(packed)), "nrf_kilo_pack_weights_cells")


def kilo_packed_floats(num_cells: int) -> int:
    return int(lib().nrf_kilo_packed_floats(num_cells))


def kilo_packed_bwd_floats(num_cells: int) -> int:
    return int(lib().nrf_kilo_packed_bwd_floats(num_cells))


def kilo_encode_dirs(dirs, kind: str = "fp32", out: torch.Tensor | None = None) -> torch.Tensor:
    """The direction encodings of (m, 3) directions once per row (nrf_kilo_encode_dirs): a uint8 table for
    kilo_forward(..., dir_table=...).  ``kind`` must be the kind of the forward that reads it."""
    dirs = _f32(dirs, "dirs").reshape(-1, 3)
    m = dirs.shape[0]
    need = int(lib().nrf_kilo_dir_table_bytes(m)) if m else 0
    if out is None or out.numel() < need:
        out = torch.empty(max(need, 256), dtype=torch.uint8, device=dirs.device)
    with torch.cuda.device(dirs.device):
        check(lib().nrf_kilo_encode_dirs(_ptr(dirs), m, KILO_KINDS[kind], _ptr(out), _stream(dirs)), "nrf_kilo_encode_dirs")
    return out


def kilo_forward(packed, num_cells: int, aabb, res, pts, dirs, dir_index=None, workspace=None,
                 return_workspace: bool = False, pack_from=None, packed_t=None, packed_cells=None, kind: str = "fp32",
                 encode_dirs_per_ray: bool = True):
    """raw (n, 4) = [r, g, b, sigma] per point; dirs (n, 3) or (m, 3) with int32 dir_index (n,).
    With ``dir_index`` (one direction per ray) the direction encodings are evaluated once per row of ``dirs`` and the
    tiny-MLP kernel reads them from a table (nrf_kilo_encode_dirs + nrf_kilo_forward_encoded: bit-identical raw);
    ``encode_dirs_per_ray=False`` keeps the per-point evaluation (nrf_kilo_forward / _fp16x2).
    ``kind`` "fp16x2": ``packed`` from kilo_pack_weights(kind="fp16x2"), the tiny MLPs on the fp16 matrix cores
    (fp32-class; operands beyond fp16's range give NaN); "bf16x3": kilo_pack_weights(kind="bf16x3"), the tiny MLPs on the
    bf16 matrix cores with every operand as three bf16 terms (fp32's 24 bits and range, six MFMAs per product).
    ``return_workspace``: also return the workspace (the points' order by cell), which kilo_backward needs.
    ``pack_from``: the 12 parameter tensors (or a ParamPointers); the cells this batch routes to are packed from them
    into ``packed`` (and ``packed_t``) on the way (nrf_kilo_forward_packing), except those ``packed_cells`` (int32 per
    cell) marks as packed already -- both buffers are then good for this batch only."""
    pts, dirs = _f32(pts, "pts").reshape(-1, 3), _f32(dirs, "dirs").reshape(-1, 3)
    n = pts.shape[0]
    if dir_index is None:
        if dirs.shape[0] != n:
            raise NrfError(f"dirs has {dirs.shape[0]} rows for {n} points and no dir_index")
    else:
        if dir_index.dtype != torch.int32 or dir_index.numel() != n or not dir_index.is_cuda:
            raise NrfError("dir_index: expected int32 cuda tensor with one entry per point")
        dir_index = dir_index.contiguous()
    _a, _r, pa, pr = _grid_args(aabb, res)
    raw = torch.empty((n, 4), dtype=torch.float32, device=pts.device)
    need = int(lib().nrf_kilo_forward_workspace_bytes(n, num_cells))
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(max(need, 256), dtype=torch.uint8, device=pts.device)
    with torch.cuda.device(pts.device):
        if kind not in KILO_KINDS or (kind != "fp32" and pack_from is not None):
            raise NrfError(f"kilo_forward: kind {kind!r} (packing on the way is fp32 only)")
        if pack_from is not None:
            pp = _param_pointers(pack_from)
            if pp.num_cells != num_cells:
                raise NrfError("pack_from: expected the 12 batched KiloNeRF tensors of this grid")
            if packed.numel() != kilo_packed_floats(num_cells) or (
                    packed_t is not None and packed_t.numel() != kilo_packed_bwd_floats(num_cells)):
                raise NrfError("packed / packed_t: wrong size for this grid")
            if packed_cells is not None and (packed_cells.dtype != torch.int32 or packed_cells.numel() != num_cells or
                                             not packed_cells.is_cuda or not packed_cells.is_contiguous()):
                raise NrfError("packed_cells: expected a contiguous int32 cuda tensor with one entry per cell")
            check(lib().nrf_kilo_forward_packing(pp.array, _ptr(packed_cells), _ptr(packed), _ptr(packed_t), num_cells, pa, pr,
                                                 _ptr(pts), _ptr(dirs),
                                                 _ptr(dir_index), n, _ptr(raw), _ptr(workspace), workspace.numel(),
                                                 _stream(pts)), "nrf_kilo_forward_packing")
        elif dir_index is not None and encode_dirs_per_ray and n > 0 and dirs.shape[0] <= (1 << 24):
            table = kilo_encode_dirs(dirs, kind)
            check(lib().nrf_kilo_forward_encoded(_ptr(packed), KILO_KINDS[kind], num_cells, pa, pr, _ptr(pts), _ptr(table),
                                                 dirs.shape[0], _ptr(dir_index), n, _ptr(raw), _ptr(workspace),
                                                 workspace.numel(), _stream(pts)), "nrf_kilo_forward_encoded")
        else:
            L = lib()
            fn = {"fp32": L.nrf_kilo_forward, "fp16x2": L.nrf_kilo_forward_fp16x2, "bf16x3": L.nrf_kilo_forward_bf16x3}[kind]
            check(fn(_ptr(packed), num_cells, pa, pr, _ptr(pts), _ptr(dirs), _ptr(dir_index), n,
                     _ptr(raw), _ptr(workspace), workspace.numel(), _stream(pts)), "nrf_kilo_forward" + ("" if kind == "fp32" else "_" + kind))
    return (raw, workspace) if return_workspace else raw


def kilo_pack_weights_transposed(tensors, out: torch.Tensor | None = None) -> torch.Tensor:
    """A operands of the backward data-gradient chain (nrf_kilo_pack_weights_transposed)."""
    ts = [_f32(t.detach(), "kilo tensor") for t in tensors]
    num_cells = ts[0].shape[0]
    dev = ts[0].device
    n = int(lib().nrf_kilo_packed_bwd_floats(num_cells))
    if out is None:
        out = torch.empty(n, dtype=torch.float32, device=dev)
    ptrs = (C.c_void_p * 12)(*[t.data_ptr() for t in ts])
    with torch.cuda.device(dev):
        check(lib().nrf_kilo_pack_weights_transposed(ptrs, num_cells, _ptr(out), _stream(out)),
              "nrf_kilo_pack_weights_transposed")
    return out


def kilo_backward(packed, packed_t, num_cells: int, pts, dirs, dir_index, grad_raw, forward_workspace, grads=None,
                  workspace=None):
    """Gradients of the 12 parameter tensors (KILO_TENSORS order, reference shapes) for the batch the last
    kilo_forward(..., return_workspace=True) evaluated.  Every element of every gradient is written."""
    pts, dirs = _f32(pts, "pts").reshape(-1, 3), _f32(dirs, "dirs").reshape(-1, 3)
    n = pts.shape[0]
    grad_raw = _f32(grad_raw, "grad_raw").reshape(-1, 4)
    if grad_raw.shape[0] != n:
        raise NrfError(f"grad_raw has {grad_raw.shape[0]} rows for {n} points")
    if dir_index is None:
        if dirs.shape[0] != n:
            raise NrfError(f"dirs has {dirs.shape[0]} rows for {n} points and no dir_index")
    elif dir_index.dtype != torch.int32 or dir_index.numel() != n or not dir_index.is_cuda or not dir_index.is_contiguous():
        raise NrfError("dir_index: expected a contiguous int32 cuda tensor with one entry per point")
    dev = packed.device
    if grads is None:
        grads = [torch.empty((num_cells,) + shape, dtype=torch.float32, device=dev) for _, shape, _ in KILO_TENSORS]
    for gten, (name, shape, _) in zip(grads, KILO_TENSORS):
        if tuple(gten.shape) != (num_cells,) + shape or gten.dtype != torch.float32 or not gten.is_contiguous():
            raise NrfError(f"grad of {name}: expected a contiguous float32 tensor of shape {(num_cells,) + shape}")
    ptrs = (C.c_void_p * 12)(*[t.data_ptr() for t in grads])
    need = int(lib().nrf_kilo_backward_workspace_bytes(n, num_cells))
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(max(need, 256), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        check(lib().nrf_kilo_backward(_ptr(packed), _ptr(packed_t), num_cells, _ptr(pts), _ptr(dirs), _ptr(dir_index), n,
                                      _ptr(grad_raw), _ptr(forward_workspace), forward_workspace.numel(), ptrs,
                                      _ptr(workspace), workspace.numel(), _stream(packed)), "nrf_kilo_backward")
    return grads


def kilo_composite_backward(raw, t_start, t_end, offsets, counts, ert_eps: float, white_bkgd: bool, grad_rgb,
                            grad_depth=None, grad_acc=None) -> torch.Tensor:
    """d(loss)/d(raw) (n, 4) from the gradients of kilo_composite's outputs."""
    R = offsets.shape[0]
    dev = offsets.device
    grad_raw = torch.empty((raw.shape[0], 4), dtype=torch.float32, device=dev)
    if raw.shape[0] == 0:
        return grad_raw
    g_rgb = _f32(grad_rgb, "grad_rgb").reshape(R, 3)
    g_depth = None if grad_depth is None else _f32(grad_depth, "grad_depth").reshape(R)
    g_acc = None if grad_acc is None else _f32(grad_acc, "grad_acc").reshape(R)
    with torch.cuda.device(dev):
        check(lib().nrf_kilo_composite_backward(_ptr(_f32(raw, "raw")), _ptr(t_start), _ptr(t_end), _ptr(offsets),
                                                _ptr(counts), R, float(ert_eps), int(bool(white_bkgd)), _ptr(g_rgb),
                                                _ptr(g_depth), _ptr(g_acc), _ptr(grad_raw), _stream(offsets)),
              "nrf_kilo_composite_backward")
    return grad_raw


def l2_reg(param: torch.Tensor, coeff: float, grad_accum: torch.Tensor | None, loss_accum: torch.Tensor | None):
    """loss_accum += coeff * sum(param^2); grad_accum += 2 coeff param (nrf_l2_reg)."""
    p = _f32(param.detach(), "param")
    with torch.cuda.device(p.device):
        check(lib().nrf_l2_reg(_ptr(p), p.numel(), float(coeff), _ptr(grad_accum), _ptr(loss_accum), _stream(p)),
              "nrf_l2_reg")


def _cell_tensors(tensors):
    ts = list(tensors)
    if len(ts) != 12:
        raise NrfError("expected the 12 batched KiloNeRF tensors (KILO_TENSORS order)")
    for t in ts:
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.shape[0] == ts[0].shape[0]):
            raise NrfError("kilo tensors: contiguous fp32 cuda tensors with the cell count as the first dimension")
    return ts


def kilo_cells_gather(tensors, cells: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    """(len(cells), 6212) records: the 12 per-cell slices of every listed cell, back to back (nrf_kilo_cells_gather).
    ``cells``: distinct int32 cell indices on the device."""
    ts = _cell_tensors(tensors)
    num_cells, dev = ts[0].shape[0], ts[0].device
    if cells.dtype != torch.int32 or not cells.is_cuda or cells.ndim != 1:
        raise NrfError("cells: expected a 1-D int32 cuda tensor")
    n, rec = cells.shape[0], int(lib().nrf_kilo_record_floats())
    if out is None:
        out = torch.empty((n, rec), dtype=torch.float32, device=dev)
    elif out.shape != (n, rec) or out.dtype != torch.float32 or not out.is_contiguous() or out.device != dev:
        raise NrfError("out: expected a contiguous (len(cells), %d) fp32 tensor" % rec)
    ptrs = (C.c_void_p * 12)(*[t.data_ptr() for t in ts])
    with torch.cuda.device(dev):
        check(lib().nrf_kilo_cells_gather(ptrs, _ptr(cells.contiguous()), n, num_cells, _ptr(out), _stream(out)),
              "nrf_kilo_cells_gather")
    return out


def kilo_cells_scatter(records: torch.Tensor, cells: torch.Tensor, tensors) -> None:
    """Inverse of kilo_cells_gather: writes the records back into the listed cells' slices; other cells are untouched."""
    ts = _cell_tensors(tensors)
    num_cells, dev = ts[0].shape[0], ts[0].device
    n, rec = cells.shape[0], int(lib().nrf_kilo_record_floats())
    if cells.dtype != torch.int32 or not cells.is_cuda or cells.ndim != 1:
        raise NrfError("cells: expected a 1-D int32 cuda tensor")
    if records.shape != (n, rec) or records.dtype != torch.float32 or not records.is_contiguous() or records.device != dev:
        raise NrfError("records: expected a contiguous (len(cells), %d) fp32 tensor" % rec)
    ptrs = (C.c_void_p * 12)(*[t.data_ptr() for t in ts])
    with torch.cuda.device(dev):
        check(lib().nrf_kilo_cells_scatter(_ptr(records), _ptr(cells.contiguous()), n, num_cells, ptrs, _stream(records)),
              "nrf_kilo_cells_scatter")


_pinned_total = {}


def occ_pack_bricks(occ: torch.Tensor) -> torch.Tensor:
    """The (rx,ry,rz) uint8 grid as bit bricks (one int64 word per 4 x 4 x 4 voxels) for ``kilo_march(..., bricks=)``:
    built once per grid, 1/8 of a byte per voxel, so the count pass's lookups stay in L2."""
    if occ.dtype != torch.uint8 or occ.ndim != 3 or not occ.is_cuda:
        raise NrfError("occ: expected a 3-D uint8 cuda tensor")
    occ = occ.contiguous()
    res = np.ascontiguousarray(np.asarray(occ.shape, dtype=np.int32))
    pr = res.ctypes.data_as(C.c_void_p)
    need = int(lib().nrf_occ_bricks_bytes(pr))
    if need < 0:
        check(need, "nrf_occ_bricks_bytes")
    bricks = torch.empty(need // 8, dtype=torch.int64, device=occ.device)
    with torch.cuda.device(occ.device):
        check(lib().nrf_occ_pack_bricks(_ptr(occ), pr, _ptr(bricks), _stream(occ)), "nrf_occ_pack_bricks")
    return bricks


def kilo_march(rays_o, rays_d, occ: torch.Tensor, aabb, t_near: float, t_far: float, step: float,
               reuse_masks: bool = True, after_count=None, bricks: torch.Tensor | None = None):
    """Occupancy-grid marching.  Returns (pts (N,3), ray_idx (N,), t_start (N,), t_end (N,),
    offsets (R,), counts (R,)); one host sync to learn N between the count and the write pass.  ``reuse_masks``:
    the write pass takes the kept steps from the count pass's ballots instead of walking the grid again (same bits).
    ``after_count``: called once the count pass and the read-back of N are enqueued and before the host waits for
    them -- GPU work that does not depend on N goes here and runs while the host wakes up and sizes the buffers.
    ``bricks``: ``occ_pack_bricks(occ)``; the count pass then looks the bit bricks up instead of the bytes (same result)."""
    rays_o, rays_d = _f32(rays_o, "rays_o").reshape(-1, 3), _f32(rays_d, "rays_d").reshape(-1, 3)
    if occ.dtype != torch.uint8 or occ.ndim != 3 or not occ.is_cuda:
        raise NrfError("occ: expected a 3-D uint8 cuda tensor")
    occ = occ.contiguous()
    if bricks is not None and (bricks.dtype != torch.int64 or not bricks.is_cuda or not bricks.is_contiguous()
                               or bricks.numel() * 8 != int(lib().nrf_occ_bricks_bytes(
                                   np.ascontiguousarray(np.asarray(occ.shape, dtype=np.int32)).ctypes.data_as(C.c_void_p)))):
        raise NrfError("bricks: expected occ_pack_bricks(occ) of this grid")
    R = rays_o.shape[0]
    dev = rays_o.device
    _a, _r, pa, pr = _grid_args(aabb, occ.shape)
    if R == 0:
        e = torch.empty(0, dtype=torch.float32, device=dev)
        z = torch.empty(0, dtype=torch.int32, device=dev)
        return e.reshape(0, 3), z, e, e.clone(), z.clone(), z.clone()
    counts = torch.empty(R, dtype=torch.int32, device=dev)      # the count pass writes every ray's entry,
    offsets = torch.empty(R, dtype=torch.int32, device=dev)     # the scan every offset and the total
    total = torch.empty(1, dtype=torch.int64, device=dev)
    need = int(lib().nrf_kilo_march_workspace_bytes(R, float(t_near), float(t_far), float(step)))
    if need < 0:
        check(need, "nrf_kilo_march_workspace_bytes")
    ws = torch.empty(max(need, 256), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        if bricks is not None:
            check(lib().nrf_kilo_march_count_bricks(_ptr(rays_o), _ptr(rays_d), R, _ptr(bricks), pr, pa, float(t_near),
                                                    float(t_far), float(step), _ptr(counts), _ptr(offsets), _ptr(total),
                                                    _ptr(ws), ws.numel(), _stream(rays_o)), "nrf_kilo_march_count_bricks")
        else:
            check(lib().nrf_kilo_march_count(_ptr(rays_o), _ptr(rays_d), R, _ptr(occ), pr, pa, float(t_near), float(t_far),
                                             float(step), _ptr(counts), _ptr(offsets), _ptr(total), _ptr(ws), ws.numel(),
                                             _stream(rays_o)), "nrf_kilo_march_count")
        if after_count is None:
            n = int(total.item())
        else:
            host = _pinned_total.get(dev)
            if host is None:
                host = _pinned_total[dev] = torch.empty(1, dtype=torch.int64, pin_memory=True)
            host.copy_(total, non_blocking=True)
            done = torch.cuda.Event()
            done.record()
            after_count()
            done.synchronize()
            n = int(host[0])
        if n > MAX_POINTS:
            raise TooManySamples(n, R)
        pts = torch.empty((n, 3), dtype=torch.float32, device=dev)
        ray_idx = torch.empty(n, dtype=torch.int32, device=dev)
        t_start = torch.empty(n, dtype=torch.float32, device=dev)
        t_end = torch.empty(n, dtype=torch.float32, device=dev)
        if n:
            check(lib().nrf_kilo_march_write(_ptr(rays_o), _ptr(rays_d), R, _ptr(occ), pr, pa, float(t_near),
                                             float(t_far), float(step), _ptr(offsets), _ptr(pts), _ptr(ray_idx),
                                             _ptr(t_start), _ptr(t_end), _ptr(counts),
                                             _ptr(ws) if reuse_masks else None, ws.numel() if reuse_masks else 0,
                                             _stream(rays_o)), "nrf_kilo_march_write")
    return pts, ray_idx, t_start, t_end, offsets, counts


def kilo_composite(raw, t_start, t_end, offsets, counts, ert_eps: float, white_bkgd: bool):
    R = offsets.shape[0]
    dev = offsets.device
    rgb = torch.empty((R, 3), dtype=torch.float32, device=dev)
    depth = torch.empty((R,), dtype=torch.float32, device=dev)
    acc = torch.empty((R,), dtype=torch.float32, device=dev)
    if raw.shape[0]:
        raw, t_start, t_end = _f32(raw, "raw"), _f32(t_start, "t_start"), _f32(t_end, "t_end")
    with torch.cuda.device(dev):
        check(lib().nrf_kilo_composite(_ptr(raw), _ptr(t_start), _ptr(t_end), _ptr(offsets), _ptr(counts), R,
                                       float(ert_eps), int(bool(white_bkgd)), _ptr(rgb), _ptr(depth), _ptr(acc),
                                       _stream(offsets)), "nrf_kilo_composite")
    return rgb, depth, acc


def occ_bake(packed_teacher: torch.Tensor, aabb, res, threshold: float, chunk_cells: int = 1 << 20,
             vertices: bool = False, kind: str = "fp32") -> torch.Tensor:
    """Occupancy grid (rx,ry,rz) bool from a packed 8x256 MLP, baked `chunk_cells` cells per launch group.
    ``kind`` "fp16x2" / "bf16x3": ``packed_teacher`` from ops.pack_weights_fp16x2 / _bf16x3, the teacher on that kernel
    (27-point rule only; "bf16x3" carries the fp32 kernel's 24-bit operands at about twice its rate).
    Default: KiloNeRF's `_evaluate_grid` (volumn_renderer.py:142-232: 27 sub-points per cell, teacher fine model,
    1080 B of workspace per cell).  ``vertices``: the NeRF renderer's `_build_occ_grid`
    (nerf/renderer/volume_renderer.py:156-197: one linspace vertex per voxel, 40 B per cell)."""
    dev = packed_teacher.device
    _a, r, pa, pr = _grid_args(aabb, res)
    n = int(r[0]) * int(r[1]) * int(r[2])
    occ = torch.zeros(n, dtype=torch.uint8, device=dev)
    chunk = min(chunk_cells, n)
    L = lib()
    if kind not in ("fp32", "fp16x2", "bf16x3") or (kind != "fp32" and vertices):
        raise NrfError(f"occ_bake: kind {kind!r}" + (" with vertices" if vertices else ""))
    size_fn, bake_fn = ((L.nrf_occ_bake_vertices_workspace_bytes, L.nrf_occ_bake_vertices) if vertices else
                        (L.nrf_occ_bake_workspace_bytes, {"fp32": L.nrf_occ_bake, "fp16x2": L.nrf_occ_bake_fp16x2,
                                                          "bf16x3": L.nrf_occ_bake_bf16x3}[kind]))
    ws = torch.empty(max(int(size_fn(chunk)), 256), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        for first in range(0, n, chunk):
            m = min(chunk, n - first)
            check(bake_fn(_ptr(packed_teacher), pa, pr, float(threshold), first, m, _ptr(occ), _ptr(ws), ws.numel(),
                          _stream(occ)), "nrf_occ_bake")
    return occ.view(int(r[0]), int(r[1]), int(r[2])).bool()


class KiloTrainCall:
    """nrf_kilo_train_forward_backward with its buffers kept between steps: the workspace (grown when a batch has
    more samples than it was sized for), the per-ray outputs, and the argument block.  ``params`` / ``grads``: the 12
    parameter tensors and their gradient tensors (storage must not move, e.g. views of flat buffers)."""

    NRF_ERR_WORKSPACE = -4

    def __init__(self, params, grads, packed, packed_t, loss):
        self.pp, self.gp = ParamPointers(params), ParamPointers(grads)
        self.packed, self.packed_t, self.loss = packed, packed_t, loss
        self.ws = None
        self.max_samples = 0
        self._rays = None
        self._n = (C.c_int64 * 1)()
        self.args = KiloTrainArgs()

    def _size(self, n_rays, need, near, far, step, dev):
        self.max_samples = max(int(need * 1.25), 1 << 16)
        b = int(lib().nrf_kilo_train_workspace_bytes(n_rays, self.max_samples, self.pp.num_cells, near, far, step))
        if b < 0:
            check(b, "nrf_kilo_train_workspace_bytes")
        self.ws = torch.empty(b, dtype=torch.uint8, device=dev)

    def __call__(self, rays_o, rays_d, occ, aabb, res, near, far, step, ert, white_bkgd, target_rgb, target_alpha, reach,
                 touched=None, live=None):
        """Returns (rgb, depth, acc, n_samples); grads and loss are filled; ``touched`` (int32 per cell) is set to 1
        where the batch has samples, ``live`` gets those cells OR-ed in."""
        o, d = _f32(rays_o, "rays_o").reshape(-1, 3), _f32(rays_d, "rays_d").reshape(-1, 3)
        R, dev = o.shape[0], o.device
        if occ.dtype != torch.uint8 or occ.ndim != 3 or not occ.is_cuda or not occ.is_contiguous():
            raise NrfError("occ: expected a contiguous 3-D uint8 cuda tensor")
        t_rgb = _f32(target_rgb, "target_rgb").reshape(-1)
        t_alpha = None if target_alpha is None else _f32(target_alpha, "target_alpha").reshape(-1)
        if R == 0 or t_rgb.numel() != 3 * R or (t_alpha is not None and t_alpha.numel() != R):
            raise NrfError("KiloTrainCall: need at least one ray and one target per ray")
        for m in (reach, touched, live):
            if m is not None and (m.dtype != torch.int32 or m.numel() != self.pp.num_cells or not m.is_cuda):
                raise NrfError("reach / touched: expected int32 cuda tensors with one entry per cell")
        _a, _r, pa, pr = _grid_args(aabb, res)
        _o, _s, _po, pocc = _grid_args(aabb, occ.shape)
        near, far, step = float(near), float(far), float(step)
        if self.ws is None or self._rays != R:
            self._rays = R
            self._size(R, self.max_samples, near, far, step, dev)
        rgb = torch.empty((R, 3), dtype=torch.float32, device=dev)
        depth = torch.empty((R,), dtype=torch.float32, device=dev)
        acc = torch.empty((R,), dtype=torch.float32, device=dev)
        a = self.args
        a.rays_o, a.rays_d, a.n_rays = o.data_ptr(), d.data_ptr(), R
        a.occ, a.occ_res, a.aabb, a.res = occ.data_ptr(), pocc, pa, pr
        a.num_cells = self.pp.num_cells
        a.t_near, a.t_far, a.step, a.ert_eps, a.white_bkgd = near, far, step, float(ert), int(bool(white_bkgd))
        a.target_rgb, a.target_alpha = t_rgb.data_ptr(), _ptr(t_alpha)
        a.tensors, a.grads = C.cast(self.pp.array, C.c_void_p), C.cast(self.gp.array, C.c_void_p)
        a.reach, a.packed, a.packed_t = _ptr(reach), self.packed.data_ptr(), self.packed_t.data_ptr()
        a.rgb_map, a.depth_map, a.acc_map, a.loss_accum = rgb.data_ptr(), depth.data_ptr(), acc.data_ptr(), self.loss.data_ptr()
        a.touched, a.live, a.n_samples = _ptr(touched), _ptr(live), C.cast(self._n, C.c_void_p)
        with torch.cuda.device(dev):
            stream = _stream(o)
            for _attempt in range(2):
                a.max_samples, a.workspace, a.workspace_bytes = self.max_samples, self.ws.data_ptr(), self.ws.numel()
                rc = lib().nrf_kilo_train_forward_backward(C.byref(a), stream)
                if rc != self.NRF_ERR_WORKSPACE:
                    break
                n = int(self._n[0])
                if n > MAX_POINTS:
                    raise TooManySamples(n, R)
                self._size(R, n, near, far, step, dev)        # this batch has more samples than any before: grow, again
            check(rc, "nrf_kilo_train_forward_backward")
        return rgb, depth, acc, int(self._n[0])


def kilo_clip_adam(params, grads, exp_avg, exp_avg_sq, step: int, live=None, lr=5e-4, beta1=0.9, beta2=0.999, eps=1e-8,
                   clip_value=40.0, grad_scale=1.0, reg_grad_coeff=0.0, reg_loss_coeff=0.0, loss_accum=None) -> None:
    """clip_grad_value_ + Adam over the 12 batched tensors, one block per cell (nrf_kilo_clip_adam).  Each of the four
    arguments: the 12 tensors or a ParamPointers.  ``live`` (int32 per cell): cells that have had a gradient since the
    moments were zeroed; the others are skipped in the unregularised tensors (their update is the identity)."""
    pp, gp, mp, vp = (_param_pointers(x) for x in (params, grads, exp_avg, exp_avg_sq))
    if live is not None and (live.dtype != torch.int32 or live.numel() != pp.num_cells or not live.is_cuda or
                             not live.is_contiguous()):
        raise NrfError("live: expected a contiguous int32 cuda tensor with one entry per cell")
    with torch.cuda.device(pp.device):
        check(lib().nrf_kilo_clip_adam(pp.array, gp.array, mp.array, vp.array, pp.num_cells, _ptr(live), float(grad_scale),
                                       float(clip_value), float(lr), float(beta1), float(beta2), float(eps), int(step),
                                       float(reg_grad_coeff), float(reg_loss_coeff), _ptr(loss_accum), _stream(pp.tensors[0])),
              "nrf_kilo_clip_adam")
