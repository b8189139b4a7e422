"""CPU restatement (oracle) of the reference's KiloNeRF path.

TEST INFRASTRUCTURE ONLY (same rule as nerf_oracle.py: used by ``tests/``, ``smoke()`` and the
``cpu_baseline`` leg of ``bench.py``; nothing under ``nerf_amd/`` imports it).

Parity status, per function:
  * ``route``, ``network_forward``, ``color_reg_loss``: **pinned** -- checked against fixtures made
    by running the real reference (``tools/gen_golden_kilo.py`` -> ``tests/golden/kilo_*.npz``).
    They restate src/models/kilonerf/network.py (paths relative to /root/reference).
  * gradients of ``network_forward`` w.r.t. the parameter tensors (torch autograd on this restatement): **pinned**
    by ``tests/golden/kilo_network_grad.npz`` (autograd through the real reference network).
  * ``march``, ``composite_packed``, ``composite_packed_torch`` (its differentiable form), ``render``,
    ``train_loss``'s rendering part: **parity unpinned**.  The reference delegates
    occupancy-grid sampling, early ray termination and packed rendering to the third-party
    ``nerfacc`` package (src/models/kilonerf/renderer/volumn_renderer.py:47-51, 78-86, 107-113), which is
    neither vendored in the reference nor installed here, and whose version the reference does
    not pin (no requirements entry).  These functions state THIS repository's rule (DESIGN.md
    section 7), following nerfacc's published algorithm (fixed-step marching through a binary
    occupancy grid, transmittance-threshold termination, w = T * alpha accumulation) and the
    reference's own call sites for everything around it (mid-point positions :90-92, sigmoid /
    relu / alpha :97-100, background :116, output shapes :118-122).
"""
from __future__ import annotations

import numpy as np
import torch

from .nerf_oracle import pos_enc

F32 = np.float32


# --------------------------------------------------------------------------------------
# routing -- network.py:314-334 and :246-253
# --------------------------------------------------------------------------------------
def route(pts: np.ndarray, aabb_min, aabb_max, res):
    """``Network._get_index`` + the flat index.  fp32, op for op: (p - min) / (max - min) * res
    (:322-325), clamp(min=0) (:329), min(res - 1) (:330-331), truncate (:334);
    flat = i * ry * rz + j * rz + k (:251-253)."""
    p = np.asarray(pts, dtype=F32).reshape(-1, 3)
    mn, mx = np.asarray(aabb_min, dtype=F32), np.asarray(aabb_max, dtype=F32)
    res = np.asarray(res, dtype=np.int64)
    with np.errstate(invalid="ignore"):
        s = ((p - mn) / (mx - mn)).astype(F32) * res.astype(F32)
        s = np.where(s < 0, F32(0), s)                       # torch.clamp(min=0) keeps NaN
        s = np.where(s > (res.astype(F32) - F32(1)), res.astype(F32) - F32(1), s)
        s = np.where(np.isnan(s), F32(0), s)
    idx3 = s.astype(np.int64)
    flat = idx3[:, 0] * (res[1] * res[2]) + idx3[:, 1] * res[2] + idx3[:, 2]
    return idx3, flat


# --------------------------------------------------------------------------------------
# the tiny MLPs -- network.py:123-167 (BatchedKiloNeRF.forward), :223-311 (Network.forward)
# --------------------------------------------------------------------------------------
def _bl(h, w, b, idx):
    """One gathered layer: bmm(h[:,None], w[idx]^T) + b[idx] (:134-137)."""
    return torch.einsum("ni,noi->no", h, w[idx]) + b[idx]


def network_forward(sd: dict, pts: torch.Tensor, dirs: torch.Tensor, aabb_min, aabb_max, res,
                    dtype=None, chunk: int = 32768, prefix: str = "model.") -> torch.Tensor:
    """raw (n, 4) = [r, g, b, sigma] (:163).  ``pts`` (n,3) with ``dirs`` (n,3), or (R,S,3) with (R,3)
    (dirs expanded per sample, :243).  ``dtype=torch.float64`` is the rounding yardstick (routing
    stays fp32 so the same cell is used)."""
    if pts.ndim == 3:
        dirs = dirs[:, None].expand(pts.shape)
    p, d = pts.reshape(-1, 3), dirs.reshape(-1, 3)
    _, flat = route(p.numpy(), aabb_min, aabb_max, res)
    flat = torch.from_numpy(flat)
    g = {k[len(prefix):]: (torch.as_tensor(v) if dtype is None else torch.as_tensor(v).to(dtype))
         for k, v in sd.items() if k.startswith(prefix)}
    if dtype is not None:
        p, d = p.to(dtype), d.to(dtype)
    outs = []
    for i in range(0, p.shape[0], chunk):
        idx = flat[i:i + chunk]
        h = pos_enc(p[i:i + chunk], 10)                                        # :292
        ed = pos_enc(d[i:i + chunk], 4)                                        # :293
        for layer in range(2):                                                 # :132-138
            h = torch.relu(_bl(h, g[f"pts_linears_weights.{layer}"], g[f"pts_linears_biases.{layer}"], idx))
        sigma = _bl(h, g["sigma_linear_weights"], g["sigma_linear_biases"], idx)        # :142-144
        feat = _bl(h, g["feature_linear_weights"], g["feature_linear_biases"], idx)     # :147-149
        h = torch.relu(_bl(torch.cat([feat, ed], -1), g["view_linears_weights"], g["view_linears_biases"], idx))
        rgb = _bl(h, g["rgb_linear_weights"], g["rgb_linear_biases"], idx)              # :159-161
        outs.append(torch.cat([rgb, sigma], -1))
    return torch.cat(outs, 0)


def color_reg_loss(sd: dict, prefix: str = "model.") -> float:
    """``Network.get_color_reg_loss`` (:378-382): sum of squares of the view and rgb layers."""
    tot = torch.zeros((), dtype=torch.float32)
    for n in ("view_linears_weights", "view_linears_biases", "rgb_linear_weights", "rgb_linear_biases"):
        tot = tot + torch.sum(torch.as_tensor(sd[prefix + n]) ** 2)
    return float(tot)


# --------------------------------------------------------------------------------------
# occupancy-grid marching -- own rule (parity unpinned; see module docstring)
# --------------------------------------------------------------------------------------
def march(rays_o: np.ndarray, rays_d: np.ndarray, occ: np.ndarray, aabb_min, aabb_max,
          t_near: float, t_far: float, step: float):
    """Per ray: [t0, t1] = slab overlap of the ray with the scene box, clipped to [near, far];
    interval k = [t0 + k*step, min(t0 + (k+1)*step, t1)), k < ceil((t1 - t0) / step); kept when the
    occupancy voxel containing the interval MID-point is set.  All arithmetic fp32 in the order
    written here (the HIP kernel uses the same order, so the sample sets are bit-identical).
    Returns pts (N,3), ray_idx (N,), t_start (N,), t_end (N,), offsets (R,), counts (R,)."""
    o, d = np.asarray(rays_o, dtype=F32).reshape(-1, 3), np.asarray(rays_d, dtype=F32).reshape(-1, 3)
    mn, mx = np.asarray(aabb_min, dtype=F32), np.asarray(aabb_max, dtype=F32)
    res = np.asarray(occ.shape, dtype=np.int64)
    resf = res.astype(F32)
    step = F32(step)
    P, RI, TS, TE, counts = [], [], [], [], np.zeros(o.shape[0], dtype=np.int32)
    for r in range(o.shape[0]):
        t0, t1 = F32(t_near), F32(t_far)
        hit = True
        for a in range(3):
            if d[r, a] == 0:
                if o[r, a] < mn[a] or o[r, a] > mx[a]:
                    hit = False
                continue
            inv = F32(1) / d[r, a]
            ta, tb = (mn[a] - o[r, a]) * inv, (mx[a] - o[r, a]) * inv
            if ta > tb:
                ta, tb = tb, ta
            t0, t1 = max(t0, ta), min(t1, tb)
        if not hit or not (t0 < t1):
            continue
        n = int(np.ceil((t1 - t0) / step))
        k = np.arange(n, dtype=np.int64)
        ts = (t0 + k.astype(F32) * step).astype(F32)
        te = np.minimum((ts + step).astype(F32), t1)
        tm = (F32(0.5) * (ts + te).astype(F32)).astype(F32)
        p = (o[r][None] + (d[r][None] * tm[:, None]).astype(F32)).astype(F32)
        s = (((p - mn).astype(F32) / (mx - mn).astype(F32)).astype(F32) * resf).astype(F32)
        v = np.minimum(np.maximum(s, F32(0)), resf - F32(1)).astype(np.int64)
        keep = (ts < t1) & (occ[v[:, 0], v[:, 1], v[:, 2]] != 0)
        counts[r] = int(keep.sum())
        P.append(p[keep]), TS.append(ts[keep]), TE.append(te[keep])
        RI.append(np.full(counts[r], r, dtype=np.int32))
    offsets = (np.cumsum(counts) - counts).astype(np.int32)
    cat = lambda xs, shp, dt: np.concatenate(xs) if xs else np.zeros(shp, dtype=dt)  # noqa: E731
    return (cat(P, (0, 3), F32), cat(RI, (0,), np.int32), cat(TS, (0,), F32), cat(TE, (0,), F32), offsets, counts)


# --------------------------------------------------------------------------------------
# packed compositing with early ray termination -- own rule around volumn_renderer.py:89-122
# --------------------------------------------------------------------------------------
def composite_packed(raw: np.ndarray, t_start, t_end, offsets, counts, ert_eps: float, white_bkgd: bool,
                     want_margin: bool = False):
    """alpha = 1 - exp(-relu(sigma) * (t_end - t_start)) (:99-100); rgb = sigmoid(raw rgb) (:97);
    T_i = prod_{j<i}(1 - alpha_j); w_i = T_i * alpha_i while T_i >= ert_eps, else 0 (the sample and
    everything behind it are terminated); colour = sum w rgb + (1 - acc) * white (:116);
    depth = sum w t_mid / max(acc, fp32 eps); acc = sum w.  float64 accumulation (yardstick).
    ``want_margin`` also returns min_i |T_i - ert_eps| per ray: a ray whose margin is at fp32 rounding level
    may legitimately terminate one sample earlier or later in an fp32 implementation."""
    raw = np.asarray(raw, dtype=np.float64).reshape(-1, 4)
    ts, te = np.asarray(t_start, dtype=np.float64), np.asarray(t_end, dtype=np.float64)
    R = len(offsets)
    rgb, depth, acc, margin = np.zeros((R, 3)), np.zeros(R), np.zeros(R), np.full(R, np.inf)
    for r in range(R):
        b, n = int(offsets[r]), int(counts[r])
        sl = slice(b, b + n)
        alpha = 1.0 - np.exp(-np.maximum(raw[sl, 3], 0.0) * (te[sl] - ts[sl]))
        T = np.concatenate([[1.0], np.cumprod(1.0 - alpha)[:-1]]) if n else np.zeros(0)
        w = np.where(T >= ert_eps, T * alpha, 0.0)
        if n:
            margin[r] = np.abs(T - ert_eps).min()
        c = 1.0 / (1.0 + np.exp(-raw[sl, :3]))
        acc[r] = w.sum()
        rgb[r] = (w[:, None] * c).sum(0) + ((1.0 - acc[r]) if white_bkgd else 0.0)
        depth[r] = (w * 0.5 * (ts[sl] + te[sl])).sum() / max(acc[r], float(np.finfo(np.float32).eps))
    return (rgb, depth, acc, margin) if want_margin else (rgb, depth, acc)


def composite_packed_torch(raw: torch.Tensor, t_start, t_end, offsets, counts, ert_eps: float, white_bkgd: bool):
    """``composite_packed`` in differentiable torch float64 (autograd gives the yardstick gradients of the packed
    compositing rule; the termination mask is a constant, as in the HIP backward)."""
    raw = raw.double().reshape(-1, 4)
    ts = torch.as_tensor(np.asarray(t_start, dtype=np.float64))
    te = torch.as_tensor(np.asarray(t_end, dtype=np.float64))
    rgbs, depths, accs = [], [], []
    for r in range(len(offsets)):
        b, n = int(offsets[r]), int(counts[r])
        sl = slice(b, b + n)
        alpha = 1.0 - torch.exp(-torch.relu(raw[sl, 3]) * (te[sl] - ts[sl]))
        T = torch.cat([torch.ones(1, dtype=torch.float64), torch.cumprod(1.0 - alpha, 0)[:-1]]) if n else alpha
        w = torch.where(T.detach() >= ert_eps, T * alpha, torch.zeros_like(alpha))
        c = torch.sigmoid(raw[sl, :3])
        acc = w.sum()
        rgbs.append((w[:, None] * c).sum(0) + ((1.0 - acc) if white_bkgd else 0.0))
        depths.append((w * 0.5 * (ts[sl] + te[sl])).sum() / torch.clamp(acc, min=float(np.finfo(np.float32).eps)))
        accs.append(acc)
    return torch.stack(rgbs), torch.stack(depths), torch.stack(accs)


def train_loss(rgb_map: torch.Tensor, alpha_map: torch.Tensor, target_rgb, target_alpha, stage: str, l2_reg: float,
               sd: dict | None = None, prefix: str = "model."):
    """``NetworkWrapper.forward`` (src/train/trainers/kilonerf.py:43-61): distillation = mse(teacher_rgb, rgb) +
    mse(teacher_alpha, alpha) [+ l2_reg * colour regulariser (:52-55)]; fine-tuning = mse(gt_rgb, rgb)."""
    mse = torch.nn.functional.mse_loss
    if stage == "distillation":
        loss = mse(target_rgb.reshape(-1, 3).to(rgb_map.dtype), rgb_map.reshape(-1, 3)) + \
            mse(target_alpha.reshape(-1, 1).to(alpha_map.dtype), alpha_map.reshape(-1, 1))
        if l2_reg > 0:
            reg = 0.0
            for n in ("view_linears_weights", "view_linears_biases", "rgb_linear_weights", "rgb_linear_biases"):
                reg = reg + torch.sum(sd[prefix + n].to(rgb_map.dtype) ** 2)
            loss = loss + l2_reg * reg
        return loss
    return mse(target_rgb.reshape(-1, 3).to(rgb_map.dtype), rgb_map.reshape(-1, 3))


def render(sd: dict, rays_o, rays_d, occ, aabb_min, aabb_max, res, t_near, t_far, step, ert_eps, white_bkgd):
    """``Renderer.forward`` (volumn_renderer.py:57-123) under the rules above, for (R, 3) rays.
    Returns the reference's dict shapes for batch size 1: rgb_map (1,R,3), depth_map (1,R,1), alpha_map (1,R,1)."""
    pts, ri, ts, te, off, cnt = march(rays_o, rays_d, occ, aabb_min, aabb_max, t_near, t_far, step)
    if len(pts):
        raw = network_forward(sd, torch.from_numpy(pts), torch.from_numpy(np.asarray(rays_d, dtype=F32)[ri]),
                              aabb_min, aabb_max, res).numpy()
    else:
        raw = np.zeros((0, 4), dtype=F32)
    rgb, depth, acc = composite_packed(raw, ts, te, off, cnt, ert_eps, white_bkgd)
    R = len(off)
    return {"rgb_map": rgb.reshape(1, R, 3).astype(F32), "depth_map": depth.reshape(1, R, 1).astype(F32),
            "alpha_map": acc.reshape(1, R, 1).astype(F32), "n_samples": int(cnt.sum())}


# --------------------------------------------------------------------------------------
# occupancy-grid bake -- volumn_renderer.py:142-232 (`_evaluate_grid`), restated with the same torch ops
# --------------------------------------------------------------------------------------
def bake_points(aabb_min, aabb_max, res) -> torch.Tensor:
    """The (cells, 27, 3) evaluation points: linspace cell corners + cell_size / 2 (:156-164), offsets
    {-0.25, 0, 0.25}^3 * cell_size in meshgrid 'ij' order (:170-176), centre + offset (:194)."""
    mn, mx = torch.tensor(aabb_min, dtype=torch.float32), torch.tensor(aabb_max, dtype=torch.float32)
    resf = torch.tensor(res, dtype=torch.float32)
    cell = (mx - mn) / resf
    axes = [torch.linspace(float(mn[a]), float(mx[a]), res[a] + 1)[:-1] for a in range(3)]
    gx, gy, gz = torch.meshgrid(*axes, indexing="ij")
    centres = torch.stack([gx, gy, gz], -1) + cell / 2.0
    o = torch.tensor([-0.25, 0.0, 0.25])
    offs = torch.stack(torch.meshgrid([o, o, o], indexing="ij"), -1).view(-1, 3) * cell
    return centres.view(-1, 1, 3) + offs[None]


def bake(teacher_sd: dict, aabb_min, aabb_max, res, threshold: float, want_sigma: bool = False):
    """Teacher fine-model raw sigma at the 27 points, zero view directions (:204-207); occupied = any > threshold
    (:212-220).  Returns the (rx,ry,rz) bool grid (and the (cells, 27) sigmas)."""
    from .nerf_oracle import network_forward as nerf_forward
    pts = bake_points(aabb_min, aabb_max, res).reshape(-1, 3)
    raw = nerf_forward(teacher_sd, pts, torch.zeros_like(pts), "fine")
    sig = raw[:, 3].reshape(-1, 27)
    occ = (sig > threshold).any(-1).reshape(*res)
    return (occ.numpy(), sig.numpy()) if want_sigma else occ.numpy()
