"""CPU restatement (oracle) of the reference's classic NeRF volume-rendering path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``nerf_amd/`` imports this file; it is used by
``tests/``, by ``__graft_entry__.smoke()`` as the checker and by ``bench.py`` for the timed
``cpu_baseline`` leg.  The product path is the HIP library behind ``include/nerf_amd.h``.

Every function cites the reference lines it restates (paths relative to /root/reference).
Parity status: **pinned** -- ``tests/test_oracle_golden.py`` checks each stage against
fixtures produced by running the real reference in the build container
(``tools/gen_golden.py`` -> ``tests/golden/*.npz``).

Numerics that matter for bit-exact sample indices are written as explicit models of what
torch-CPU does (verified against torch in ``tools/gen_golden.py``):
  * ``torch.sum`` over a 64-wide inner dim: 8 lanes x 4 partial sums (``sum64_model``);
  * ``torch.cumsum`` / ``torch.cumprod`` on fp32: running *double* accumulator, each prefix
    rounded to fp32 (``cumsum_model`` / ``cumprod_model``).
Everything else is plain fp32 torch-CPU / numpy arithmetic where the reference uses the
same op, so tolerances (1e-4 abs on rgb/depth/acc) are the bar there, not bit equality.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

F32 = np.float32


# --------------------------------------------------------------------------------------
# ray generation -- src/datasets/nerf/blender.py:76-101
# --------------------------------------------------------------------------------------
def raygen(c2w, H: int, W: int, camera_angle_x: float, u=None, v=None):
    """Pixel grid -> (origins, un-normalised world directions), float64 math, cast to fp32.

    blender.py:76 focal; :88-90 row-major pixel grid (v slow, u fast); :94 camera dir
    ``[(u-W/2)/focal, -(v-H/2)/focal, -1]`` (no +0.5, not normalised); :95 ``R @ dir``;
    :96 origin broadcast; :99-100 cast to float32.
    """
    c2w = np.asarray(c2w, dtype=np.float64)
    focal = 0.5 * W / np.tan(0.5 * camera_angle_x)
    if u is None:
        v, u = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
        u = u.flatten()
        v = v.flatten()
    u = np.asarray(u)
    v = np.asarray(v)
    d = np.stack([(u - W / 2) / focal, -(v - H / 2) / focal, -np.ones_like(u)], axis=-1)
    R = c2w[:3, :3]
    # explicit left-to-right products/sums (what a 3x3 @ 3xN reduces to per element)
    dw = np.stack([R[i, 0] * d[:, 0] + R[i, 1] * d[:, 1] + R[i, 2] * d[:, 2] for i in range(3)], -1)
    o = np.broadcast_to(c2w[:3, 3], dw.shape)
    return o.astype(np.float32), dw.astype(np.float32)


# --------------------------------------------------------------------------------------
# stratified coarse sampling -- volume_renderer.py:209-220, 336-349
# --------------------------------------------------------------------------------------
def linspace01(n: int) -> np.ndarray:
    """fp32 ``linspace(0, 1, n)`` with the two-sided formula torch uses (scalar form).

    torch computes ``start + step*i`` for the first half and ``end - step*(n-1-i)`` for the
    second half; its CPU kernel is vectorised and differs from this scalar form by <= 1 ulp
    on a few entries (machine dependent), so kernels take the table as an *input* and the
    golden fixture carries the reference's own table.
    """
    step = F32(F32(1.0) / F32(n - 1))
    out = np.zeros(n, dtype=np.float32)
    for i in range(n):
        if i < n // 2:
            out[i] = F32(step * F32(i))
        else:
            out[i] = F32(F32(1.0) - F32(step * F32(n - 1 - i)))
    return out


def t_bins(near: float, far: float, n: int, s=None) -> np.ndarray:
    """``near*(1-s) + far*s`` in fp32 (volume_renderer.py:214-216)."""
    s = linspace01(n) if s is None else np.asarray(s, dtype=np.float32)
    return (F32(near) * (F32(1.0) - s) + F32(far) * s).astype(np.float32)


def perturb(z: np.ndarray, u: np.ndarray) -> np.ndarray:
    """``add_perturbation`` (volume_renderer.py:336-349) with the uniforms supplied."""
    z = np.asarray(z, dtype=np.float32)
    mid = F32(0.5) * (z[:, 1:] + z[:, :-1])
    upper = np.concatenate([mid, z[:, -1:]], -1)
    lower = np.concatenate([z[:, :1], mid], -1)
    return (lower + (upper - lower) * np.asarray(u, dtype=np.float32)).astype(np.float32)


# --------------------------------------------------------------------------------------
# positional encoding -- src/models/encoding/freq.py:7-32, encoding/__init__.py:7-18
# --------------------------------------------------------------------------------------
def pos_enc(x: torch.Tensor, L: int) -> torch.Tensor:
    """``cat([x, sin(1x), cos(1x), sin(2x), cos(2x), ... sin(2^(L-1)x), cos(2^(L-1)x)])``."""
    outs = [x]
    for i in range(L):
        f = float(2.0 ** i)  # freq_bands = 2**linspace(0, L-1, L): exact powers of two
        outs.append(torch.sin(x * f))
        outs.append(torch.cos(x * f))
    return torch.cat(outs, -1)


# --------------------------------------------------------------------------------------
# the 8x256 MLP -- src/models/nerf/network.py:49-74, 171-195
# --------------------------------------------------------------------------------------
def _q_bf16(t: torch.Tensor) -> torch.Tensor:
    """Round to bf16 (nearest even) and back: what an MFMA operand of the bf16 kernel holds."""
    return t.to(torch.bfloat16).to(t.dtype)


def _split3(t: torch.Tensor):
    """t = t0 + t1 + t2 with three bf16-representable terms (the residuals are exact in fp32): what the bf16x3 kernel
    (mlp_bf16x3.hip) feeds the bf16 matrix cores."""
    t0 = _q_bf16(t)
    r = t - t0
    t1 = _q_bf16(r)
    return t0, t1, _q_bf16(r - t1)


def _linear_split3(h: torch.Tensor, w: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """One layer of nrf_mlp_forward_rays_bf16x3: the six partial products with i + j <= 2 (each product of two bf16
    values is exact in fp32), accumulated in fp32, small terms first, plus the fp32 bias."""
    hs, ws = _split3(h), _split3(w)
    acc = None
    for i, j in ((0, 2), (2, 0), (1, 1), (0, 1), (1, 0), (0, 0)):       # (x term, w term)
        p = F.linear(hs[i], ws[j])
        acc = p if acc is None else acc + p
    return acc + b


def _split2_f16(t: torch.Tensor):
    """t ~ hi + lo / 2048 with hi = fp16(t), lo = fp16((t - hi) * 2048), both round-to-nearest-even (22 significand bits):
    what the fp16x2 kernel (mlp_fp16x2.hip, split2h.h) feeds the fp16 matrix cores."""
    hi = t.to(torch.float16).to(t.dtype)
    return hi, ((t - hi) * 2048.0).to(torch.float16).to(t.dtype)


def _linear_split2_f16(h: torch.Tensor, w: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """One layer of nrf_mlp_forward_rays_fp16x2: hi.hi (+ the fp32 bias) in the main accumulator, hi.lo + lo.hi in a
    second one at scale 2^11, result = main + 2^-11 * second (each product of two fp16 values is exact in fp32)."""
    hh, hl = _split2_f16(h)
    wh, wl = _split2_f16(w)
    return (F.linear(hh, wh) + b) + (F.linear(hl, wh) + F.linear(hh, wl)) * (1.0 / 2048.0)


def mlp_forward(sd: dict, prefix: str, emb_pts: torch.Tensor, emb_dirs: torch.Tensor, bf16: bool = False,
                split3: bool = False, split2_f16: bool = False) -> torch.Tensor:
    """``NeRF.forward`` (network.py:49-74): raw ``[r, g, b, sigma]`` per point, no activation.

    ``bf16=True`` restates the numerics of nrf_mlp_forward_rays_bf16 (mlp_bf16.hip): the weights and the input of
    every MFMA layer (pts_linears, feature_linear, views_linears) are rounded to bf16, accumulation and biases are
    fp32; alpha_linear reads the bf16-rounded relu(h7) with fp32 weights, rgb_linear is all fp32."""
    q = _q_bf16 if bf16 else (lambda t: t)

    def lin(name, h, mfma=True):
        w = sd[f"{prefix}.{name}.weight"]
        if split3 and mfma:       # ``split3=True``: the bf16x3 kernel's rule; the alpha / rgb heads stay fp32 (VALU)
            return _linear_split3(h, w, sd[f"{prefix}.{name}.bias"])
        if split2_f16 and mfma:   # the fp16x2 kernel's rule, same fp32 heads
            return _linear_split2_f16(h, w, sd[f"{prefix}.{name}.bias"])
        return F.linear(q(h), q(w), sd[f"{prefix}.{name}.bias"]) if mfma else F.linear(h, w, sd[f"{prefix}.{name}.bias"])

    h = emb_pts
    for i in range(8):
        h = torch.relu(lin(f"pts_linears.{i}", h))
        if i == 4:                                   # network.py:57-58: [input_pts, h]
            h = torch.cat([emb_pts, h], -1)
    alpha = lin("alpha_linear", q(h), mfma=False)    # :61, raw
    feature = lin("feature_linear", h)               # :62, no activation
    h = torch.cat([feature, emb_dirs], -1)           # :63
    h = torch.relu(lin("views_linears.0", h))        # :65-67
    rgb = lin("rgb_linear", h, mfma=False)           # :69
    return torch.cat([rgb, alpha], -1)               # :70


def network_forward(sd: dict, pts: torch.Tensor, dirs: torch.Tensor, model: str = "",
                    chunk: int = 65536, dtype=None, bf16: bool = False, split3: bool = False,
                    split2_f16: bool = False) -> torch.Tensor:
    """``Network.forward`` (network.py:171-195). ``pts`` (R,S,3) with ``dirs`` (R,3), or
    ``pts`` (P,3) with ``dirs`` (P,3).  Returns (..., 4).

    ``dtype=torch.float64`` evaluates encoding + MLP in double from the same fp32 inputs: the
    yardstick tests use to tell fp32 rounding noise (ours vs the reference's) from a bug."""
    prefix = "model_fine" if model == "fine" else "model"
    if dtype is not None:
        sd = {k: v.to(dtype) for k, v in sd.items() if k.startswith(prefix + ".")}
        pts, dirs = pts.to(dtype), dirs.to(dtype)
    flat = pts.reshape(-1, 3)
    d = dirs[:, None].expand(pts.shape) if pts.ndim == 3 else dirs
    dflat = d.reshape(-1, 3)
    outs = []
    for i in range(0, flat.shape[0], chunk):
        e = pos_enc(flat[i:i + chunk], 10)
        ed = pos_enc(dflat[i:i + chunk], 4)
        outs.append(mlp_forward(sd, prefix, e, ed, bf16, split3, split2_f16))
    out = torch.cat(outs, 0)
    return out.reshape(list(pts.shape[:-1]) + [4])


# --------------------------------------------------------------------------------------
# explicit torch-CPU reduction models (see module docstring)
# --------------------------------------------------------------------------------------
def sum64_model(x: np.ndarray) -> np.ndarray:
    """torch-CPU ``x.sum(-1)`` for fp32 ``x`` of shape (R, 64): 8 lanes x 4 accumulators."""
    x = np.asarray(x, dtype=np.float32)
    assert x.shape[1] == 64
    tot = np.zeros(x.shape[0], dtype=np.float32)
    for l in range(8):
        p = [(x[:, 8 * a + l] + x[:, 32 + 8 * a + l]).astype(np.float32) for a in range(4)]
        lane = (((p[0] + p[1]).astype(np.float32) + p[2]).astype(np.float32) + p[3]).astype(np.float32)
        tot = (tot + lane).astype(np.float32)
    return tot


def cumsum_model(x: np.ndarray) -> np.ndarray:
    x = np.asarray(x, dtype=np.float32)
    acc = np.zeros(x.shape[0], dtype=np.float64)
    out = np.empty_like(x)
    for k in range(x.shape[1]):
        acc = acc + x[:, k].astype(np.float64)
        out[:, k] = acc.astype(np.float32)
    return out


def cumprod_model(x: np.ndarray) -> np.ndarray:
    x = np.asarray(x, dtype=np.float32)
    acc = np.ones(x.shape[0], dtype=np.float64)
    out = np.empty_like(x)
    for k in range(x.shape[1]):
        acc = acc * x[:, k].astype(np.float64)
        out[:, k] = acc.astype(np.float32)
    return out


# --------------------------------------------------------------------------------------
# activation + alpha compositing -- volume_renderer.py:222-224, 236-265 (coarse), 291-334
# --------------------------------------------------------------------------------------
def composite(raw: torch.Tensor, z: torch.Tensor, white_bkgd: bool = True):
    """raw (R,S,4), z (R,S) -> rgb_map (R,3), depth (R,), weights (R,S), acc (R,1)."""
    R, S = z.shape
    delta = torch.cat([z[:, 1:] - z[:, :-1], torch.full((R, 1), 1e10, dtype=z.dtype)], -1)  # :222-224
    sigma = torch.relu(raw[..., 3])                                 # :244
    rgb = torch.sigmoid(raw[..., :3])                               # :246
    alpha = 1 - torch.exp(-sigma * delta)                           # :248
    # :249 -- torch-CPU cumprod = running double accumulator rounded per prefix (== cumprod_model,
    # asserted in tests/test_oracle_golden.py); the torch op keeps the oracle differentiable.
    T = torch.cumprod((1 - alpha)[:, :-1], dim=-1)
    T = torch.cat([torch.ones((R, 1), dtype=z.dtype), T], -1)       # :250 exclusive
    w = T * alpha                                                   # :252
    rgb_map = torch.sum(w.unsqueeze(-1) * rgb, dim=-2)              # :254
    acc = torch.sum(w, dim=-1, keepdim=True)                        # :256
    if white_bkgd:
        rgb_map = rgb_map + (1.0 - torch.sum(w, -1).unsqueeze(-1))  # :259-261
    depth = torch.sum(w * z, dim=-1)                                # :263 (not divided by acc)
    return rgb_map, depth, w, acc


# --------------------------------------------------------------------------------------
# importance resampling -- volume_renderer.py:363-391
# --------------------------------------------------------------------------------------
def importance_sampling(w: np.ndarray, bins: np.ndarray, u: np.ndarray, u2: np.ndarray):
    """w (R,Nc) coarse weights, bins (R,Nc) or (Nc,) unperturbed coarse z, u/u2 (R,Ni).

    Returns (samples (R,Ni) fp32, idx (R,Ni) int64).  Bit-exact model of the reference on
    torch-CPU: pdf = w / (sum(w) + 1e-10) (:374); cdf = [0, cumsum(pdf)] (:375-376);
    idx = clamp(searchsorted(cdf, u, right=True) - 1, 0, Nc-1) (:381-384);
    lo = bins[idx], hi = bins[min(idx+1, Nc-1)] (:387-388); lo + (hi-lo)*u2 (:390).
    """
    w = np.asarray(w, dtype=np.float32)
    R, Nc = w.shape
    bins = np.broadcast_to(np.asarray(bins, dtype=np.float32), (R, Nc))
    tot = sum64_model(w) if Nc == 64 else w.sum(-1, dtype=np.float32)
    pdf = (w / (tot[:, None] + F32(1e-10))).astype(np.float32)
    cdf = np.concatenate([np.zeros((R, 1), np.float32), cumsum_model(pdf)], -1)   # (R, Nc+1)
    u = np.asarray(u, dtype=np.float32)
    # searchsorted(right=True): number of cdf entries <= u
    cnt = (cdf[:, None, :] <= u[:, :, None]).sum(-1)
    idx = np.clip(cnt - 1, 0, Nc - 1).astype(np.int64)
    lo = np.take_along_axis(bins, idx, 1)
    hi = np.take_along_axis(bins, np.minimum(idx + 1, Nc - 1), 1)
    samples = (lo + ((hi - lo).astype(np.float32) * np.asarray(u2, np.float32)).astype(np.float32)).astype(np.float32)
    return samples, idx


def merge_sort(zc: np.ndarray, zf: np.ndarray) -> np.ndarray:
    """``sort(cat([coarse, fine]))`` values (volume_renderer.py:288-289)."""
    zc = np.broadcast_to(np.asarray(zc, np.float32), (zf.shape[0], zc.shape[-1]))
    return np.sort(np.concatenate([zc, np.asarray(zf, np.float32)], -1), axis=-1)


# --------------------------------------------------------------------------------------
# whole path -- volume_renderer.py:143-153 (intended glue), 201-334
# --------------------------------------------------------------------------------------
def points(o: torch.Tensor, d: torch.Tensor, z: torch.Tensor):
    """:227-231 normalise dirs, pts = o + z * dhat. Returns (pts (R,S,3), dhat (R,3))."""
    dhat = d / torch.norm(d, dim=-1, keepdim=True)
    return o[:, None, :] + z[..., None] * dhat[:, None, :], dhat


def render(sd: dict, rays_o, rays_d, near=2.0, far=6.0, n_coarse=64, n_importance=128,
           white_bkgd=True, u=None, u2=None, bins_table=None, u_coarse=None, u_fine=None,
           return_intermediates=False, bf16=False):
    """Coarse + fine render of R rays.  ``sd``: state dict of fp32 torch CPU tensors.

    ``u``/``u2``: the two ``torch.rand(R, Ni)`` draws of ``importance_sampling``.
    ``u_coarse`` (R,Nc) / ``u_fine`` (R,Ni): perturbation uniforms (training, perturb>0).
    ``bf16``: both MLPs with the rounding points of the bf16 kernel (``mlp_forward(bf16=True)``); everything else fp32.
    """
    o = torch.as_tensor(rays_o, dtype=torch.float32)
    d = torch.as_tensor(rays_d, dtype=torch.float32)
    R = o.shape[0]
    tb = t_bins(near, far, n_coarse) if bins_table is None else np.asarray(bins_table, np.float32)
    bins = np.broadcast_to(tb, (R, n_coarse)).copy()
    zc = bins if u_coarse is None else perturb(bins, u_coarse)
    zc_t = torch.from_numpy(np.ascontiguousarray(zc))
    pts, dhat = points(o, d, zc_t)
    raw_c = network_forward(sd, pts, dhat, bf16=bf16)
    rgb_c, depth_c, w_c, acc_c = composite(raw_c, zc_t, white_bkgd)
    zf, idx = importance_sampling(w_c.detach().numpy(), bins, np.asarray(u), np.asarray(u2))  # fine z carries no grad
    if u_fine is not None:
        zf = perturb(zf, u_fine)
    z_all = torch.from_numpy(merge_sort(zc, zf))
    pts_f, _ = points(o, d, z_all)
    raw_f = network_forward(sd, pts_f, dhat, "fine", bf16=bf16)
    rgb_f, depth_f, w_f, acc_f = composite(raw_f, z_all, white_bkgd)
    out = {
        "coarse_rgb_map": rgb_c, "coarse_depth_map": depth_c,
        "fine_rgb_map": rgb_f, "fine_depth_map": depth_f, "alpha_map": acc_f,
    }
    if return_intermediates:
        out.update(raw_coarse=raw_c, w_coarse=w_c, acc_coarse=acc_c, z_coarse=zc_t,
                   fine_samples=torch.from_numpy(zf), bin_indices=torch.from_numpy(idx),
                   z_fine=z_all, raw_fine=raw_f, w_fine=w_f)
    return out


def loss(out: dict, gt: torch.Tensor) -> torch.Tensor:
    """``mse(coarse, gt) + mse(fine, gt)`` (src/train/trainers/nerf.py:25-27)."""
    return F.mse_loss(out["coarse_rgb_map"], gt) + F.mse_loss(out["fine_rgb_map"], gt)


def state_dict_from_arrays(arrs: dict) -> dict:
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in arrs.items()}


# --------------------------------------------------------------------------------------
# the renderer's occupancy-grid path -- volume_renderer.py:50-139 (live `forward`), 156-197 (`_build_occ_grid`)
# --------------------------------------------------------------------------------------
def build_occ_grid(sd: dict, aabb_min, aabb_max, res, occ_thre: float = 0.01, want_sigma: bool = False):
    """``_build_occ_grid`` with the same torch ops: ``linspace(min, max, res)`` per axis (:170-172), meshgrid 'ij'
    (:175), the network's default (coarse) model with zero view directions (:183-185), ``sigma > occ_thre`` (:195).
    The network part is pinned by the network fixtures; the grid construction is torch's own linspace."""
    axes = [torch.linspace(float(aabb_min[a]), float(aabb_max[a]), int(res[a])) for a in range(3)]
    gx, gy, gz = torch.meshgrid(*axes, indexing="ij")
    pts = torch.stack([gx, gy, gz], -1).view(-1, 3)
    with torch.no_grad():
        sig = network_forward(sd, pts, torch.zeros_like(pts))[:, 3]
    occ = (sig > occ_thre).view(*[int(r) for r in res]).numpy()
    return (occ, sig.view(*[int(r) for r in res]).numpy()) if want_sigma else occ


def render_occupancy(sd: dict, rays_o, rays_d, occ, aabb_min, aabb_max, near=2.0, far=6.0, step=1e-3, early_stop_eps=1e-4,
                     white_bkgd=True):
    """The live ``forward`` (:50-139) under this repository's sampling / compositing rule (**parity unpinned**: the
    reference delegates both to nerfacc, see oracle/kilo_oracle.py): intervals of ``step`` through the occupancy
    grid, ``self.net(pts, viewdirs[ray])`` = coarse model with the directions as given (:104-112), packed
    compositing + background (:123-132).  Shapes for batch size 1."""
    from . import kilo_oracle as K
    o, d = np.asarray(rays_o, np.float32).reshape(-1, 3), np.asarray(rays_d, np.float32).reshape(-1, 3)
    pts, ri, ts, te, off, cnt = K.march(o, d, occ, aabb_min, aabb_max, near, far, step)
    if len(pts):
        with torch.no_grad():
            raw = network_forward(sd, torch.from_numpy(pts), torch.from_numpy(d[ri])).numpy()
    else:
        raw = np.zeros((0, 4), np.float32)
    rgb, depth, acc = K.composite_packed(raw, ts, te, off, cnt, early_stop_eps, white_bkgd)
    R = len(off)
    return {"fine_rgb_map": rgb.reshape(1, R, 3).astype(np.float32), "fine_depth_map": depth.reshape(1, R, 1).astype(np.float32),
            "alpha_map": acc.reshape(1, R, 1).astype(np.float32), "n_samples": int(cnt.sum())}
