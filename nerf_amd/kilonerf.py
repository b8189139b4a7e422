"""KiloNeRF host side: ``KiloNetwork`` and ``KiloRenderer`` with the reference's interfaces.

``KiloNetwork`` has the module tree / state-dict keys of
/root/reference/src/models/kilonerf/network.py:172-220 (``model.pts_linears_weights.{0,1}``, ...,
buffers ``aabb_min``, ``aabb_max``, ``grid_resolution``) and the ``forward(pts, viewdirs)`` signature of
:223-311.  ``KiloRenderer`` is ``Renderer(net)`` of
/root/reference/src/models/kilonerf/renderer/volumn_renderer.py:14-123: ``forward(batch, is_training)``
returns ``rgb_map (B,R,3)``, ``depth_map (B,R,1)``, ``alpha_map (B,R,1)``.

With ``is_training`` under autograd the outputs are connected to the network's 12 parameter tensors through one
``torch.autograd.Function`` whose backward runs the HIP backward kernels (nrf_kilo_composite_backward,
nrf_kilo_backward), so the reference's trainer / optimizer work unchanged (src/train/trainers/kilonerf.py:34-63);
``nerf_amd/kilo_train.py`` is the fused form of that step.

All arithmetic runs in libnerf_amd.so (nrf_kilo_*); there is no torch fallback.  Sampling, early
termination and packed compositing follow this repository's own rule (the reference delegates them to
nerfacc, which is absent and unpinned): see DESIGN.md section 7 and oracle/kilo_oracle.py.
"""
from __future__ import annotations

import math
from types import SimpleNamespace

import torch
import torch.nn as nn

from . import kilo_ops
from .weights import KILO_TENSORS


def default_kilo_cfg(**overrides) -> SimpleNamespace:
    """The keys of /root/reference/configs/kilonerf/lego.yaml:26-67 this path reads."""
    task = dict(white_bkgd=1, aabb={"min": [-1.0, -1.0, -1.0], "max": [1.0, 1.0, 1.0]}, grid_resolution=[16, 16, 16],
                chunk_size=32768, l2_reg=1e-6)
    sampler = dict(occ_factor=16, occ_threshold=10, ert_threshold=0.01, near=2, far=6, render_step_size=0.005)
    for k, v in overrides.items():
        (task if k in task else sampler)[k] = v
    return SimpleNamespace(task_arg=SimpleNamespace(**task), sampler=SimpleNamespace(**sampler))


class BatchedKiloNeRF(nn.Module):
    """Parameter container with the reference's names and shapes (network.py:72-113, D=2, W=32)."""

    def __init__(self, num_models: int):
        super().__init__()
        self.num_models = num_models
        shapes = {n: s for n, s, _ in KILO_TENSORS}
        mk = lambda n: nn.Parameter(torch.empty((num_models,) + shapes[n]))  # noqa: E731
        self.pts_linears_weights = nn.ParameterList([mk("pts_linears_weights.0"), mk("pts_linears_weights.1")])
        self.pts_linears_biases = nn.ParameterList([mk("pts_linears_biases.0"), mk("pts_linears_biases.1")])
        for n in ("feature_linear", "sigma_linear", "view_linears", "rgb_linear"):
            setattr(self, n + "_weights", mk(n + "_weights"))
            setattr(self, n + "_biases", mk(n + "_biases"))
        for p in self.parameters():  # network.py:117-121: kaiming_uniform(a=sqrt(5)) on every >1-D tensor
            if p.dim() > 1:
                nn.init.kaiming_uniform_(p, a=math.sqrt(5))
            else:
                nn.init.zeros_(p)

    def tensors(self):
        """The 12 tensors in include/nerf_amd.h order."""
        out = []
        for n, _, _ in KILO_TENSORS:
            obj = self
            for part in n.split("."):
                obj = obj[int(part)] if part.isdigit() else getattr(obj, part)
            out.append(obj)
        return out


class KiloNetwork(nn.Module):
    def __init__(self, cfg=None):
        super().__init__()
        cfg = cfg or default_kilo_cfg()
        t = cfg.task_arg
        self.register_buffer("aabb_min", torch.tensor(t.aabb["min"], dtype=torch.float32))
        self.register_buffer("aabb_max", torch.tensor(t.aabb["max"], dtype=torch.float32))
        self.register_buffer("grid_resolution", torch.tensor(t.grid_resolution, dtype=torch.long))
        res = [int(v) for v in t.grid_resolution]
        self.num_models = res[0] * res[1] * res[2]
        self._grid, self._grid_key = None, None
        self.white_bkgd = t.white_bkgd
        self.model = BatchedKiloNeRF(self.num_models)
        self._packed = None
        self._packed_key = None
        # "fp32" (the reference's arithmetic on the fp32 MFMA), "bf16x3" (the same 24-bit operands as three bf16 terms each
        # on the bf16 MFMA) or "fp16x2" (fp32-class on the fp16 MFMA: every operand as two fp16 terms); the last two are
        # inference forms -- the training kernels read the fp32 records
        self.mlp_dtype = "fp32"

    def grid(self):
        """(aabb, res) from the registered buffers the reference routes with (network.py:189-191, :322), which
        load_state_dict / load_network overwrite; cached until a buffer changes (one host read per change)."""
        return _grid_from_buffers(self, self)

    def packed(self) -> torch.Tensor:
        """Per-cell streaming layout, repacked only when a parameter changed (tensor versions)."""
        ts = self.model.tensors()
        key = tuple((t.data_ptr(), t._version) for t in ts) + (self.mlp_dtype,)
        if self._packed is None or key != self._packed_key or self._packed.device != ts[0].device:
            self._packed = kilo_ops.kilo_pack_weights(ts, self._packed if self._packed is not None and
                                                      self._packed.device == ts[0].device else None, kind=self.mlp_dtype)
            self._packed_key = key
        return self._packed

    @torch.no_grad()
    def forward(self, pts, viewdirs):
        """(R,S,3)+(R,3) or (P,3)+(P,3) -> rgb (R,S,3), sigma (R,S,1) raw (network.py:223-311)."""
        if pts.ndim == 2:
            R, S = pts.shape[0], 1
            di = None
        else:
            R, S = pts.shape[:2]
            di = torch.arange(R, device=pts.device, dtype=torch.int32).repeat_interleave(S)
        aabb, res = self.grid()
        raw = kilo_ops.kilo_forward(self.packed(), self.num_models, aabb, res, pts.reshape(-1, 3),
                                    viewdirs, di, kind=self.mlp_dtype).view(R, S, 4)
        return raw[..., :3], raw[..., 3:4]

    def get_color_reg_loss(self):
        """network.py:357-384: plain torch on four parameter tensors (training-side, differentiable)."""
        m = self.model
        return (torch.sum(m.view_linears_weights ** 2) + torch.sum(m.view_linears_biases ** 2) +
                torch.sum(m.rgb_linear_weights ** 2) + torch.sum(m.rgb_linear_biases ** 2))


def _grid_from_buffers(net, cache):
    bufs = (net.aabb_min, net.aabb_max, net.grid_resolution)
    key = tuple((b.data_ptr(), b._version) for b in bufs)
    if cache._grid is None or cache._grid_key != key:
        aabb = [float(v) for v in bufs[0].tolist()] + [float(v) for v in bufs[1].tolist()]
        res = [int(v) for v in bufs[2].tolist()]
        if res[0] * res[1] * res[2] != int(net.model.pts_linears_biases[0].shape[0]):
            raise kilo_ops.NrfError(f"grid_resolution {res} does not match the {int(net.model.pts_linears_biases[0].shape[0])} cells of the parameter tensors")
        cache._grid, cache._grid_key = (aabb, res), key
    return cache._grid


class KiloNetView:
    """What the renderer needs from a KiloNeRF network, for ``KiloNetwork`` and equally for the reference's own
    ``src.models.kilonerf.network.Network`` (same parameter names ``model.pts_linears_weights.0`` ... and buffers
    ``aabb_min`` / ``aabb_max`` / ``grid_resolution``, network.py:172-220): the 12 tensors in C-ABI order, the grid,
    and a packed copy that is rebuilt when a parameter's version changes."""

    def __init__(self, net: nn.Module):
        self.net = net
        self._packed = None
        self._packed_key = None
        self._grid, self._grid_key = None, None
        self._mlp_dtype = "fp32"

    @property
    def mlp_dtype(self) -> str:
        return getattr(self.net, "mlp_dtype", self._mlp_dtype)

    @mlp_dtype.setter
    def mlp_dtype(self, kind: str):
        if kind not in kilo_ops.KILO_KINDS:
            raise kilo_ops.NrfError(f"mlp_dtype {kind!r}: expected one of {sorted(kilo_ops.KILO_KINDS)}")
        if hasattr(self.net, "mlp_dtype"):
            self.net.mlp_dtype = kind
        self._mlp_dtype = kind

    def tensors(self):
        model = getattr(self.net, "model", None)
        if hasattr(model, "tensors"):
            return model.tensors()
        named = dict(self.net.named_parameters())
        try:
            return [named["model." + n] for n, _, _ in KILO_TENSORS]
        except KeyError as e:
            raise kilo_ops.NrfError(f"network has no parameter {e}: expected the reference's BatchedKiloNeRF layout") from e

    @property
    def num_models(self) -> int:
        return int(self.tensors()[0].shape[0])

    def grid(self):
        if hasattr(self.net, "grid"):
            return self.net.grid()
        return _grid_from_buffers(self.net, self)

    def packed(self) -> torch.Tensor:
        if hasattr(self.net, "packed"):
            return self.net.packed()
        ts = self.tensors()
        key = tuple((t.data_ptr(), t._version) for t in ts) + (self.mlp_dtype,)
        if self._packed is None or key != self._packed_key or self._packed.device != ts[0].device:
            same_dev = self._packed is not None and self._packed.device == ts[0].device
            self._packed = kilo_ops.kilo_pack_weights(ts, self._packed if same_dev else None, kind=self.mlp_dtype)
            self._packed_key = key
        return self._packed

    def invalidate(self):
        self._packed_key = None
        if hasattr(self.net, "_packed_key"):
            self.net._packed_key = None


def _ref_layer_params(model):
    """Weights / biases of a reference-style NeRF module in include/nerf_amd.h order."""
    from .ops import LAYER_NAMES
    sd = dict(model.named_parameters())
    return [sd[n + ".weight"] for n in LAYER_NAMES], [sd[n + ".bias"] for n in LAYER_NAMES]


class _KiloRenderFn(torch.autograd.Function):
    """march -> tiny MLPs -> packed compositing, differentiable w.r.t. the 12 parameter tensors.  Sample positions
    and the termination mask are constants (the reference marches under no_grad too, volumn_renderer.py:78-86)."""

    @staticmethod
    def forward(ctx, renderer, o, d, *params):
        net = renderer.view
        aabb, res = net.grid()
        pts, ray_idx, ts, te, offsets, counts = kilo_ops.kilo_march(o, d, renderer.occ_grid, aabb, renderer.near,
                                                                    renderer.far, renderer.step, bricks=renderer.occ_bricks())
        renderer.last_num_samples = int(pts.shape[0])
        if net.mlp_dtype != "fp32":
            raise kilo_ops.NrfError("KiloRenderer: the backward kernels read the fp32 records; train with mlp_dtype='fp32'")
        packed = net.packed()
        raw, ws = kilo_ops.kilo_forward(packed, net.num_models, aabb, res, pts, d, ray_idx, return_workspace=True)
        rgb, depth, acc = kilo_ops.kilo_composite(raw, ts, te, offsets, counts, renderer.ert, renderer.white_bkgd)
        ctx.renderer = renderer
        ctx.params = params
        # `packed` is the network's cached buffer, rewritten in place by the next repack, and the backward packs the
        # transposed operands from the live parameters: both are only the forward's weights while no parameter changes
        ctx.versions = tuple(p._version for p in params)
        ctx.saved = (packed, pts, d, ray_idx, ts, te, offsets, counts, raw, ws)
        return rgb, depth, acc

    @staticmethod
    def backward(ctx, g_rgb, g_depth, g_acc):
        r = ctx.renderer
        if tuple(p._version for p in ctx.params) != ctx.versions:
            raise kilo_ops.NrfError("KiloRenderer: a parameter was modified between forward and backward (optimizer.step() "
                                    "or another in-place update); call backward before updating the parameters")
        packed, pts, d, ray_idx, ts, te, offsets, counts, raw, ws = ctx.saved
        g_raw = kilo_ops.kilo_composite_backward(raw, ts, te, offsets, counts, r.ert, r.white_bkgd, g_rgb.contiguous(),
                                                 g_depth.contiguous(), g_acc.contiguous())
        packed_t = kilo_ops.kilo_pack_weights_transposed(ctx.params)
        grads = kilo_ops.kilo_backward(packed, packed_t, r.view.num_models, pts, d, ray_idx, g_raw, ws)
        return (None, None, None) + tuple(grads)


class KiloRenderer(nn.Module):
    """``Renderer(net)`` for KiloNeRF.  ``occ_grid``: (rx,ry,rz) bool/uint8 tensor (the reference loads or bakes
    it in ``__init__``, volumn_renderer.py:36-44; here it is handed in or set later with ``set_occ_grid``)."""

    def __init__(self, net: KiloNetwork, cfg=None, occ_grid: torch.Tensor | None = None, mlp_dtype: str | None = None):
        """``mlp_dtype``: "fp32" (default), "bf16x3" -- inference with the tiny MLPs on the bf16 matrix cores, every operand
        as three bf16 terms (nrf_kilo_forward_bf16x3: fp32's 24 bits and range) -- or "fp16x2" -- on the fp16 matrix cores
        (nrf_kilo_forward_fp16x2: fp32-class, operands must lie in fp16's range); training always reads fp32 records."""
        super().__init__()
        self.net = net
        self.view = KiloNetView(net)
        if mlp_dtype is not None:
            self.view.mlp_dtype = mlp_dtype
        ref_cfg = None
        if cfg is None:
            try:  # inside the reference tree: its global config (volumn_renderer.py:5)
                from src.config import cfg as ref_cfg  # type: ignore
                cfg = ref_cfg
            except Exception:
                cfg = default_kilo_cfg()
        s = cfg.sampler
        self.near, self.far = float(s.near), float(s.far)
        self.step, self.ert = float(s.render_step_size), float(s.ert_threshold)
        self.white_bkgd = bool(cfg.task_arg.white_bkgd)
        self.occ_grid_resolution = [g * s.occ_factor for g in cfg.task_arg.grid_resolution]
        self.occ_grid = None
        self.last_num_samples = 0
        # trainer.py:58-66 drives `renderer.estimator.update_every_n_steps(step, occ_eval_fn)` every iteration; the
        # reference's estimator re-estimates the grid from the student's density while training (nerf_amd/estimator.py)
        from .estimator import GridEstimator
        self.estimator = GridEstimator(self, lambda: self.view.grid()[0], self.occ_grid_resolution)
        if occ_grid is not None:
            self.set_occ_grid(occ_grid)
        elif ref_cfg is not None:
            self._init_grid_from_reference_cfg(ref_cfg)

    def _init_grid_from_reference_cfg(self, cfg):
        """volumn_renderer.py:36-44: load ``<result_dir>/occ_grid_res{X}_{Y}_{Z}_thresh{t}.pth`` if it exists, else bake
        it from the teacher checkpoint (``cfg.teacher_model_dir``, :149-150) and save it."""
        import os
        from .checkpoint import load_network, resolve_checkpoint
        self.grid_path = os.path.join(cfg.result_dir, self.grid_filename(cfg.sampler.occ_threshold))
        dev = torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")
        if os.path.exists(self.grid_path):
            self.load_occ_grid(self.grid_path, dev)
        elif getattr(cfg, "teacher_model_dir", None) and resolve_checkpoint(cfg.teacher_model_dir):
            from .network import Network
            teacher = Network()
            load_network(teacher, cfg.teacher_model_dir)
            self.bake_occ_grid(teacher.to(dev), float(cfg.sampler.occ_threshold))
            os.makedirs(cfg.result_dir, exist_ok=True)
            self.save_occ_grid(self.grid_path)

    def set_occ_grid(self, occ_grid: torch.Tensor):
        if list(occ_grid.shape) != list(self.occ_grid_resolution):
            raise ValueError(f"occupancy grid {tuple(occ_grid.shape)} != {self.occ_grid_resolution}")
        self.occ_grid = occ_grid.to(torch.uint8).contiguous()

    def occ_bricks(self):
        """``kilo_ops.occ_pack_bricks`` of the current grid for the march's count pass, rebuilt when ``occ_grid`` is another
        tensor or has been written in place (storage address + torch's version counter are the key)."""
        g = self.occ_grid
        if g is None or not g.is_cuda:
            return None
        key = (g.data_ptr(), g._version, g.device, tuple(g.shape))
        if getattr(self, "_bricks_key", None) != key:
            self._bricks, self._bricks_key = kilo_ops.occ_pack_bricks(g), key
        return self._bricks

    def grid_filename(self, occ_threshold) -> str:
        """The reference's cache file name (volumn_renderer.py:37-38)."""
        r = self.occ_grid_resolution
        return f"occ_grid_res{r[0]}_{r[1]}_{r[2]}_thresh{occ_threshold}.pth"

    @torch.no_grad()
    def bake_occ_grid(self, teacher, occ_threshold: float = 10.0, chunk_cells: int = 1 << 20, mlp_dtype: str = "fp32"):
        """`_evaluate_grid` (volumn_renderer.py:142-232): `teacher` is the NeRF `Network` (its fine model is queried).
        ``mlp_dtype`` "bf16x3": the teacher on the three-term bf16 kernel (every operand exactly, about twice the rate);
        "fp16x2": on the fp16x2 kernel (3.7x the rate; fp32-class densities, operands in fp16's range)."""
        from . import ops
        ws, bs = teacher.model_fine.layer_params() if hasattr(teacher.model_fine, "layer_params") else _ref_layer_params(
            teacher.model_fine)
        packed = {"fp32": ops.pack_weights, "bf16x3": ops.pack_weights_bf16x3, "fp16x2": ops.pack_weights_fp16x2}[mlp_dtype](ws, bs)
        aabb, _ = self.view.grid()
        self.set_occ_grid(kilo_ops.occ_bake(packed, aabb, self.occ_grid_resolution, occ_threshold, chunk_cells,
                                            kind=mlp_dtype))
        return self.occ_grid

    def save_occ_grid(self, path: str):
        torch.save(self.occ_grid.bool().cpu(), path)      # the reference's format: a (rx,ry,rz) bool tensor (:126-132)

    def load_occ_grid(self, path: str, device=None):
        self.set_occ_grid(torch.load(path, map_location=device or "cpu", weights_only=True))

    def forward(self, batch, is_training: bool = True):
        if self.occ_grid is None:
            raise RuntimeError("KiloRenderer: no occupancy grid (set_occ_grid)")
        rays_o, viewdirs = batch["xyz"], batch["viewdirs"]
        B, R, _ = rays_o.shape
        o, d = rays_o.reshape(-1, 3).contiguous(), viewdirs.reshape(-1, 3).contiguous()
        if self.occ_grid.device != o.device:
            self.occ_grid = self.occ_grid.to(o.device)
        params = self.view.tensors()
        if is_training and torch.is_grad_enabled() and any(p.requires_grad for p in params):
            rgb, depth, acc = _KiloRenderFn.apply(self, o, d, *params)
        else:
            with torch.no_grad():
                rgb, depth, acc = self._render(o, d)
        return {"rgb_map": rgb.view(B, R, 3), "depth_map": depth.view(B, R, 1), "alpha_map": acc.view(B, R, 1)}

    def _render(self, o, d):
        """One march / forward / composite when the samples fit a call, else the rays in halves (dense grids)."""
        try:
            self.last_num_samples = 0
            return self._render_once(o, d)
        except kilo_ops.TooManySamples:
            h = o.shape[0] // 2
            if h == 0:
                raise
            a = self._render(o[:h].contiguous(), d[:h].contiguous())
            n = self.last_num_samples
            b = self._render(o[h:].contiguous(), d[h:].contiguous())
            self.last_num_samples += n
            return tuple(torch.cat([x, y]) for x, y in zip(a, b))

    def _render_once(self, o, d):
        aabb, res = self.view.grid()
        occ = self.occ_grid
        pts, ray_idx, ts, te, offsets, counts = kilo_ops.kilo_march(o, d, occ, aabb, self.near, self.far, self.step,
                                                                    bricks=self.occ_bricks())
        self.last_num_samples = int(pts.shape[0])
        if pts.shape[0]:
            raw = kilo_ops.kilo_forward(self.view.packed(), self.view.num_models, aabb, res, pts, d, ray_idx,
                                        kind=self.view.mlp_dtype)
        else:
            raw = torch.empty((0, 4), dtype=torch.float32, device=o.device)
        return kilo_ops.kilo_composite(raw, ts, te, offsets, counts, self.ert, self.white_bkgd)

    def render(self, batch, is_training: bool = True):
        return self.forward(batch, is_training)
