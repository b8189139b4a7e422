"""Drop-in ``Renderer`` for the reference's ``renderer_module`` plug-in point.

Mirrors /root/reference/src/models/nerf/renderer/volume_renderer.py: ``Renderer(net)``
(:11-21), ``forward(batch, is_training)`` (:49) and the ``render`` name the callers use
(src/train/trainers/nerf.py:21, run.py:39, gen_dist_data.py:72), with the classic
coarse + fine body the reference keeps at :143-153 / :201-334.  All arithmetic runs in
libnerf_amd.so (HIP, gfx950); this class only owns tensors, the packed-weight cache and
the random draws, and fails loudly when the library or a GPU is missing.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch
import torch.nn as nn

from . import ops
from ._lib import NrfError, RenderArgs, check, lib
from .config import default_cfg


def near_far_table(near: float, far: float, n: int) -> np.ndarray:
    """``near*(1-s) + far*s`` with ``s = linspace(0, 1, n)`` in fp32 (volume_renderer.py:214-215).

    Host-side, 64 entries, identical for every ray when ``perturb == 0``.  Uses torch's
    two-sided linspace formula in scalar form (start + i*step below the middle,
    end - (n-1-i)*step above), so the table does not depend on the host's SIMD width.
    """
    f = np.float32
    step = f(f(1.0) / f(n - 1))
    s = np.array([f(step * f(i)) if i < n // 2 else f(f(1.0) - f(step * f(n - 1 - i))) for i in range(n)], dtype=f)
    return (f(near) * (f(1.0) - s) + f(far) * s).astype(f)


_PACK = {"fp32": ops.pack_weights, "bf16": ops.pack_weights_bf16, "bf16x3": ops.pack_weights_bf16x3, "fp16x2": ops.pack_weights_fp16x2}
_MLP_RAYS = {"fp32": ops.mlp_forward_rays, "bf16": ops.mlp_forward_rays_bf16, "bf16x3": ops.mlp_forward_rays_bf16x3,
             "fp16x2": ops.mlp_forward_rays_fp16x2}
_MLP_POINTS = {"fp32": ops.mlp_forward_points, "bf16": ops.mlp_forward_points_bf16, "bf16x3": ops.mlp_forward_points_bf16x3,
               "fp16x2": ops.mlp_forward_points_fp16x2}


class _RenderTrainFn(torch.autograd.Function):
    """Coarse + fine render whose backward runs the HIP backward kernels.

    Inputs after the renderer / ray arguments are the 48 parameters (coarse then fine model, each
    in ops.LAYER_NAMES order, weight then bias) so autograd routes the flat gradients back to them.
    """

    @staticmethod
    def forward(ctx, renderer, rays_o, rays_d, u, u2, u_coarse, u_fine, *params):
        dev = rays_o.device
        bins = renderer._bins
        white = renderer.white_bkgd > 0
        # mlp_dtype "bf16x3": both forward passes, the data-gradient chain and the weight-gradient GEMMs on the three-term
        # bf16 kernels (fp32-class); the fp32 pack still supplies the head rows of the backward
        # "fp16x2": forward passes, the data-gradient chain (at a loss scale it picks itself) and the weight-gradient
        # GEMMs (wgrad_h2p_kernel, dY scaled per row set) all on the two-term fp16 kernels
        kind = renderer.mlp_dtype
        bkind = kind
        pc, pc_t = renderer._packed_model("model", dev), renderer._packed_model_t("model", dev, bkind)
        pf, pf_t = renderer._packed_model("model_fine", dev), renderer._packed_model_t("model_fine", dev, bkind)
        fc = pc if kind == "fp32" else renderer._packed_model("model", dev, kind)
        ff = pf if kind == "fp32" else renderer._packed_model("model_fine", dev, kind)
        zc = bins if u_coarse is None else ops.perturb_samples(bins, u_coarse)
        raw_c, acts_c, masks_c = ops.mlp_forward_rays_train(fc, rays_o, rays_d, zc, kind)
        c_rgb, c_depth, w_c, _ = ops.composite(raw_c, zc, white, True)
        z_f = ops.sample_fine(w_c, bins, zc, u, u2, u_fine)      # integer bins + constants: no gradient
        raw_f, acts_f, masks_f = ops.mlp_forward_rays_train(ff, rays_o, rays_d, z_f, kind)
        f_rgb, f_depth, _, f_acc = ops.composite(raw_f, z_f, white, False)
        ctx.renderer, ctx.white, ctx.kind = renderer, white, bkind
        # pc / pf / pc_t / pf_t are the renderer's cached packed buffers, rewritten in place by the next repack: they
        # are the forward's weights only while no parameter changes before backward
        ctx.params, ctx.versions = params, tuple(p._version for p in params)
        ctx.saved = (pc, pc_t, raw_c, zc, acts_c, masks_c, pf, pf_t, raw_f, z_f, acts_f, masks_f)
        renderer.coarse_samples, renderer.fine_samples = zc, z_f
        return c_rgb, c_depth, f_rgb, f_depth, f_acc

    @staticmethod
    def backward(ctx, g_c_rgb, g_c_depth, g_f_rgb, g_f_depth, g_f_acc):
        if tuple(p._version for p in ctx.params) != ctx.versions:
            raise NrfError("Renderer: a parameter was modified between forward and backward (optimizer.step() or another "
                           "in-place update); call backward before updating the parameters")
        pc, pc_t, raw_c, zc, acts_c, masks_c, pf, pf_t, raw_f, z_f, acts_f, masks_f = ctx.saved
        r = ctx.renderer
        g_raw_c = ops.composite_backward(raw_c, zc, ctx.white, g_c_rgb, g_c_depth, None)
        flat_c = ops.mlp_backward(pc, pc_t, g_raw_c, acts_c, masks_c, r._bwd_workspace(raw_c), kind=ctx.kind)
        g_raw_f = ops.composite_backward(raw_f, z_f, ctx.white, g_f_rgb, g_f_depth, g_f_acc)
        flat_f = ops.mlp_backward(pf, pf_t, g_raw_f, acts_f, masks_f, r._bwd_workspace(raw_f), kind=ctx.kind)
        ctx.saved = None
        grads = []
        for flat in (flat_c, flat_f):
            for w, b in ops.split_flat_grad(flat):
                grads += [w, b]
        return (None,) * 7 + tuple(grads)


class Renderer(nn.Module):
    """``mlp_dtype``: "fp32" (default, the reference's precision), "bf16x3" (fp32-class results from three-term bf16
    operand splits on the bf16 matrix cores), "fp16x2" (fp32-class results from two-term fp16 operand splits, 22-bit
    operands, for networks whose weights and activations lie in fp16's range; under autograd its backward runs on the
    fp16x2 kernels too, at power-of-two scales chosen on the device) or "bf16" (BASELINE configs[4]: the MLP on the bf16
    matrix cores, inference only -- training runs with "fp32", "bf16x3" or "fp16x2"; ``cfg.task_arg.mlp_dtype`` is read when the
    argument is None)."""

    def __init__(self, net, cfg=None, mlp_dtype: str | None = None):
        super().__init__()
        if cfg is None:
            try:  # inside the reference tree: its global config (volume_renderer.py:8)
                from src.config import cfg as ref_cfg  # type: ignore
                cfg = ref_cfg
            except Exception:
                cfg = default_cfg()
        ta = cfg.task_arg
        self.net = net
        self.N_samples = int(ta.N_samples)
        self.N_importance = int(ta.N_importance)
        self.white_bkgd = ta.white_bkgd
        self.perturb = ta.perturb
        self.raw_noise_std = ta.raw_noise_std
        self.near = ta.near
        self.far = ta.far
        self.chunk_size = ta.chunk_size if getattr(ta, "chunk_size", 0) > 0 else 1024
        self.mlp_dtype = mlp_dtype or getattr(ta, "mlp_dtype", "fp32")
        if self.mlp_dtype not in ("fp32", "bf16", "bf16x3", "fp16x2"):
            raise NrfError(f"mlp_dtype {self.mlp_dtype!r}: expected 'fp32', 'bf16', 'bf16x3' or 'fp16x2'")
        if self.mlp_dtype == "bf16" and (self.N_samples % 64 or (self.N_samples + self.N_importance) % 64):
            raise NrfError("the bf16 MLP kernel needs sample counts that are multiples of 64")
        if self.raw_noise_std and self.raw_noise_std > 0:
            # the reference calls an undefined bare add_noise() here (:241, :310) and would crash
            raise NrfError("raw_noise_std > 0 is not supported (the reference's noise branch is broken)")
        lib()  # fail now, not at the first frame, if the HIP library is absent
        # state the reference keeps between the two passes (:216, :220, :285)
        self.t_bins = None
        self.coarse_samples = None
        self.fine_samples = None
        self.max_rays_per_call = 1 << 20
        self._packed = {}      # model name -> (version key, packed tensor)
        self._packed_t = {}    # model name -> (version key, transposed stream for the backward)
        self._repacks = {}     # fp16x2 slots: repack count (range-check cadence)
        self._bwd_ws = None
        self._bins = None
        self._workspace = None
        # the reference's second path: occupancy-grid skipping around the (coarse) network, its live `forward`
        # (volume_renderer.py:37-47, 50-139); off unless asked for (render_occupancy / cfg.task_arg.occupancy_forward)
        aabb = getattr(ta, "aabb", None) or {"min": [-1.0, -1.0, -1.0], "max": [1.0, 1.0, 1.0]}
        self.scene_aabb = [float(v) for v in aabb["min"]] + [float(v) for v in aabb["max"]]
        grid = getattr(ta, "grid_resolution", [16, 16, 16])
        self.occ_grid_resolution = [int(g_) * int(getattr(ta, "occ_factor", 16)) for g_ in grid]
        self.occ_grid = None
        self.occupancy_forward = bool(getattr(ta, "occupancy_forward", False))
        self.render_step_size, self.early_stop_eps = 1e-3, 1e-4   # nerfacc's defaults, which the reference does not override (:90-96)
        self.last_num_samples = 0
        # what trainer.py:58-66 calls every iteration (the reference's nerfacc estimator, :38-47); it re-estimates
        # `occ_grid` only while the occupancy path is the one in use -- the classic path does not sample through it
        from .estimator import GridEstimator
        self.estimator = GridEstimator(self, self.scene_aabb, self.occ_grid_resolution, active=lambda: self.occupancy_forward)

    def set_occ_grid(self, occ_grid: torch.Tensor):
        if list(occ_grid.shape) != list(self.occ_grid_resolution):
            raise ValueError(f"occupancy grid {tuple(occ_grid.shape)} != {self.occ_grid_resolution}")
        self.occ_grid = occ_grid.to(torch.uint8).contiguous()

    # ---- occupancy-grid path (volume_renderer.py:50-139 + 156-197; sampling / compositing rule: DESIGN.md section 7) ----
    @torch.no_grad()
    def build_occ_grid(self, occ_thre: float = 0.01, device=None, chunk_cells: int = 1 << 22):
        """`_build_occ_grid` (:156-197): the network's default (coarse) model at one linspace vertex per voxel of the
        (grid_resolution * occ_factor)^3 grid, zero view directions, occupied = raw sigma > occ_thre."""
        from . import kilo_ops
        dev = torch.device(device) if device is not None else next(self.net.parameters()).device
        packed = self._packed_model("model", dev)
        self.set_occ_grid(kilo_ops.occ_bake(packed, self.scene_aabb, self.occ_grid_resolution, occ_thre, chunk_cells,
                                            vertices=True))
        return self.occ_grid

    @torch.no_grad()
    def render_occupancy(self, batch):
        """The reference's live `forward` (:50-139): occupancy-grid sampling between near and far, the network at the
        interval mid-points (`self.net(pts, viewdirs[ray])`: default = coarse model, directions as given, :104-112),
        packed compositing with the background (:123-132).  The sampling / termination / compositing rule is this
        repository's (nerfacc is absent and unpinned: kilo.hip), with nerfacc's default step
        1e-3 and early-stop threshold 1e-4.  Returns fine_rgb_map (B,R,3), fine_depth_map (B,R,1), alpha_map (B,R,1)."""
        from . import kilo_ops
        xyz, viewdirs = batch["xyz"], batch["viewdirs"]
        if not xyz.is_cuda:
            raise NrfError("batch tensors must be on a GPU: this renderer has no CPU path")
        B, R, _ = xyz.shape
        o = xyz.reshape(-1, 3).to(torch.float32).contiguous()
        d = viewdirs.reshape(-1, 3).to(torch.float32).contiguous()
        dev = o.device
        if self.occ_grid is None:
            self.build_occ_grid(device=dev)                       # the reference bakes it in __init__ (:46-47)
        if self.occ_grid.device != dev:
            self.occ_grid = self.occ_grid.to(dev)
        self.last_num_samples = 0

        def rays(o, d):
            """One march / MLP / composite when the samples fit a call, else the rays in halves (an 800x800 frame at step
            1e-3 through a mostly occupied grid is ~2e9 samples)."""
            try:
                pts, ray_idx, ts, te, offsets, counts = kilo_ops.kilo_march(o, d, self.occ_grid, self.scene_aabb, self.near,
                                                                            self.far, self.render_step_size)
            except kilo_ops.TooManySamples:
                h = o.shape[0] // 2
                if h == 0:
                    raise
                a, b = rays(o[:h].contiguous(), d[:h].contiguous()), rays(o[h:].contiguous(), d[h:].contiguous())
                return tuple(torch.cat([x, y]) for x, y in zip(a, b))
            n = int(pts.shape[0])
            self.last_num_samples += n
            if n:
                fwd = _MLP_POINTS[self.mlp_dtype]
                raw = fwd(self._packed_model("model", dev, self.mlp_dtype), pts, d.index_select(0, ray_idx.long()))
            else:
                raw = torch.empty((0, 4), dtype=torch.float32, device=dev)
            return kilo_ops.kilo_composite(raw, ts, te, offsets, counts, self.early_stop_eps, self.white_bkgd > 0)

        rgb, depth, acc = rays(o, d)
        rgb, depth, acc = rgb.view(B, R, 3), depth.view(B, R, 1), acc.view(B, R, 1)
        return {"fine_rgb_map": rgb, "fine_depth_map": depth, "alpha_map": acc,
                "rgb_map": rgb, "depth_map": depth, "acc_map": acc}

    # ---- weights ---------------------------------------------------------------------
    def _packed_model(self, name: str, device, kind="fp32") -> torch.Tensor:
        kind = {True: "bf16", False: "fp32"}.get(kind, kind)
        model = getattr(self.net, name)
        params = []
        for ln in ops.LAYER_NAMES:
            mod = model
            for part in ln.split("."):
                mod = mod[int(part)] if part.isdigit() else getattr(mod, part)
            params += [mod.weight, mod.bias]
        key = tuple((p.data_ptr(), p._version) for p in params) + (str(device),)
        slot = name + ("" if kind == "fp32" else ":" + kind)
        hit = self._packed.get(slot)
        if hit is not None and hit[0] == key:
            return hit[1]
        buf = hit[1] if hit is not None and hit[1].device == torch.device(device) else None
        ws = [p if p.device == torch.device(device) else p.to(device) for p in params[0::2]]
        bs = [p if p.device == torch.device(device) else p.to(device) for p in params[1::2]]
        if kind == "fp16x2":
            # the packer's range check waits for the stream: always for inference (weights are packed once), but under
            # autograd -- a repack per training step -- only on the first repack and every 64th after it; in between a
            # weight that has outgrown fp16 shows as NaN outputs (TrainStep does the same)
            n = self._repacks.get(slot, 0)
            self._repacks[slot] = n + 1
            buf = _PACK[kind](ws, bs, out=buf, check_range=(not torch.is_grad_enabled()) or n % 64 == 0)
        else:
            buf = _PACK[kind](ws, bs, out=buf)
        self._packed[slot] = (key, buf)
        return buf

    def _model_params(self, name: str):
        model = getattr(self.net, name)
        params = []
        for ln in ops.LAYER_NAMES:
            mod = model
            for part in ln.split("."):
                mod = mod[int(part)] if part.isdigit() else getattr(mod, part)
            params += [mod.weight, mod.bias]
        return params

    def _packed_model_t(self, name: str, device, kind: str = "fp32") -> torch.Tensor:
        params = self._model_params(name)
        key = tuple((p.data_ptr(), p._version) for p in params[0::2]) + (str(device),)
        slot = (name, kind)
        hit = self._packed_t.get(slot)
        if hit is not None and hit[0] == key:
            return hit[1]
        pack = {"fp32": ops.pack_weights_transposed, "bf16x3": ops.pack_weights_transposed_bf16x3,
                "fp16x2": ops.pack_weights_transposed_fp16x2}[kind]
        buf = pack(params[0::2], out=hit[1] if hit is not None else None)
        self._packed_t[slot] = (key, buf)
        return buf

    def _bwd_workspace(self, raw: torch.Tensor) -> torch.Tensor:
        need = int(lib().nrf_mlp_backward_workspace_bytes(raw.shape[0], raw.shape[1]))
        if self._bwd_ws is None or self._bwd_ws.device != raw.device or self._bwd_ws.numel() < need:
            self._bwd_ws = torch.empty(need, dtype=torch.uint8, device=raw.device)
        return self._bwd_ws

    def render_rays_train(self, rays_o, rays_d, u, u2, u_coarse=None, u_fine=None):
        """Differentiable render (w.r.t. the network parameters): the training-step path."""
        dev = rays_o.device
        if self._bins is None or self._bins.device != dev:
            self._bins = torch.from_numpy(near_far_table(self.near, self.far, self.N_samples)).to(dev)
        params = self._model_params("model") + self._model_params("model_fine")
        c_rgb, c_depth, f_rgb, f_depth, f_acc = _RenderTrainFn.apply(self, rays_o, rays_d, u, u2, u_coarse, u_fine, *params)
        return {
            "coarse_rgb_map": c_rgb, "coarse_depth_map": c_depth,
            "fine_rgb_map": f_rgb, "fine_depth_map": f_depth, "alpha_map": f_acc,
            "rgb_map": f_rgb, "depth_map": f_depth, "acc_map": f_acc,
        }

    # ---- the reference's public surface ------------------------------------------------
    def render(self, batch, is_training: bool = True):
        return self.forward(batch, is_training)

    def forward(self, batch, is_training: bool = True):
        if self.occupancy_forward and not (is_training and torch.is_grad_enabled()):
            return self.render_occupancy(batch)
        xyz, viewdirs = batch["xyz"], batch["viewdirs"]
        _ = batch["gt_rgb"]  # the classic path requires the key (:209, :276)
        if xyz.ndim != 3 or xyz.shape[0] != 1:
            raise NrfError(f"batch['xyz'] must be (1, R, 3) like the reference's classic path, got {tuple(xyz.shape)}")
        if not xyz.is_cuda:
            raise NrfError("batch tensors must be on a GPU: this renderer has no CPU path")
        needs_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.net.parameters())
        dev = xyz.device
        rays_o = xyz[0].to(torch.float32).contiguous()
        rays_d = viewdirs[0].to(torch.float32).contiguous()
        R = rays_o.shape[0]
        nc, ni = self.N_samples, self.N_importance
        if self._bins is None or self._bins.device != dev:
            self._bins = torch.from_numpy(near_far_table(self.near, self.far, nc)).to(dev)
        self.t_bins = self._bins
        perturbed = bool(self.perturb > 0 and is_training)
        # same draw order as the reference: coarse perturb (:348), u (:379), u2 (:389), fine perturb (:348)
        u_coarse = torch.rand((R, nc), dtype=torch.float32, device=dev) if perturbed else None
        u = torch.rand((R, ni), device=dev)
        u2 = torch.rand(R, ni, device=dev)
        u_fine = torch.rand((R, ni), dtype=torch.float32, device=dev) if perturbed else None
        if needs_grad:
            if self.mlp_dtype == "bf16":
                raise NrfError("training runs the fp32-class kernels: construct the Renderer with mlp_dtype='fp32', 'bf16x3' "
                               "or 'fp16x2'")
            return self.render_rays_train(rays_o, rays_d, u, u2, u_coarse, u_fine)
        return self.render_rays(rays_o, rays_d, u, u2, u_coarse, u_fine)

    def render_rays_staged(self, rays_o, rays_d, u, u2, u_coarse=None, u_fine=None, events=None):
        """Same kernels, same order as nrf_render, but one C-ABI call per stage so a profiler
        hook can bracket each stage.  ``events``: dict name -> list of (start, end)
        ``torch.cuda.Event`` pairs appended per stage (recorded on the launch stream)."""
        dev = rays_o.device
        nc = self.N_samples
        if self._bins is None or self._bins.device != dev:
            self._bins = torch.from_numpy(near_far_table(self.near, self.far, nc)).to(dev)
        mlp = _MLP_RAYS[self.mlp_dtype]
        pc = self._packed_model("model", dev, self.mlp_dtype)
        pf = self._packed_model("model_fine", dev, self.mlp_dtype)
        white = self.white_bkgd > 0

        def stage(name, fn, *a, **k):
            if events is None:
                return fn(*a, **k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*a, **k)
            e1.record()
            events.setdefault(name, []).append((e0, e1))
            return out

        zc = self._bins if u_coarse is None else stage("perturb", ops.perturb_samples, self._bins, u_coarse)
        raw = stage("mlp_coarse", mlp, pc, rays_o, rays_d, zc)
        c_rgb, c_depth, w_c, _ = stage("composite_coarse", ops.composite, raw, zc, white, True)
        z_f = stage("sample_fine", ops.sample_fine, w_c, self._bins, zc, u, u2, u_fine)
        del raw
        raw = stage("mlp_fine", mlp, pf, rays_o, rays_d, z_f)
        f_rgb, f_depth, _, f_acc = stage("composite_fine", ops.composite, raw, z_f, white, False)
        return {
            "coarse_rgb_map": c_rgb, "coarse_depth_map": c_depth,
            "fine_rgb_map": f_rgb, "fine_depth_map": f_depth, "alpha_map": f_acc,
            "rgb_map": f_rgb, "depth_map": f_depth, "acc_map": f_acc,
        }

    def render_rays(self, rays_o, rays_d, u, u2, u_coarse=None, u_fine=None):
        """(R,3) origins / un-normalised dirs + the uniforms -> the reference's output dict."""
        dev = rays_o.device
        R = rays_o.shape[0]
        nc, ni = self.N_samples, self.N_importance
        if self._bins is None or self._bins.device != dev:
            self._bins = torch.from_numpy(near_far_table(self.near, self.far, nc)).to(dev)
        pc = self._packed_model("model", dev, self.mlp_dtype)
        pf = self._packed_model("model_fine", dev, self.mlp_dtype)
        c_rgb = torch.empty((R, 3), dtype=torch.float32, device=dev)
        c_depth = torch.empty((R,), dtype=torch.float32, device=dev)
        f_rgb = torch.empty((R, 3), dtype=torch.float32, device=dev)
        f_depth = torch.empty((R,), dtype=torch.float32, device=dev)
        f_acc = torch.empty((R, 1), dtype=torch.float32, device=dev)
        L = lib()
        step = min(R, self.max_rays_per_call)
        need = int(L.nrf_render_workspace_bytes(step, nc, ni))
        if self._workspace is None or self._workspace.device != dev or self._workspace.numel() < need:
            self._workspace = torch.empty(need, dtype=torch.uint8, device=dev)
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        with torch.cuda.device(dev):
            for r0 in range(0, R, step):
                n = min(step, R - r0)
                a = RenderArgs()
                a.packed_coarse, a.packed_fine = pc.data_ptr(), pf.data_ptr()
                a.rays_o, a.rays_d = rays_o[r0:].data_ptr(), rays_d[r0:].data_ptr()
                a.bins = self._bins.data_ptr()
                a.u, a.u2 = u[r0:].data_ptr(), u2[r0:].data_ptr()
                a.u_coarse = u_coarse[r0:].data_ptr() if u_coarse is not None else None
                a.u_fine = u_fine[r0:].data_ptr() if u_fine is not None else None
                a.n_rays, a.n_coarse, a.n_importance = n, nc, ni
                a.white_bkgd = int(self.white_bkgd > 0)
                a.coarse_rgb, a.coarse_depth = c_rgb[r0:].data_ptr(), c_depth[r0:].data_ptr()
                a.fine_rgb, a.fine_depth, a.fine_acc = f_rgb[r0:].data_ptr(), f_depth[r0:].data_ptr(), f_acc[r0:].data_ptr()
                a.workspace, a.workspace_bytes = self._workspace.data_ptr(), self._workspace.numel()
                entry = {"fp32": L.nrf_render, "bf16": L.nrf_render_bf16, "bf16x3": L.nrf_render_bf16x3,
                         "fp16x2": L.nrf_render_fp16x2}[self.mlp_dtype]
                check(entry(C.byref(a), stream), "nrf_render")
        return {
            "coarse_rgb_map": c_rgb, "coarse_depth_map": c_depth,
            "fine_rgb_map": f_rgb, "fine_depth_map": f_depth, "alpha_map": f_acc,
            # north-star aliases
            "rgb_map": f_rgb, "depth_map": f_depth, "acc_map": f_acc,
        }

    def render_view_sharded(self, c2w, H: int, W: int, focal: float, rank: int, world: int, device, group=None,
                            gather: bool = True, generator=None):
        """ONE view over ``world`` GPUs (the north star's ray-tile sharding): this rank generates and
        renders rays ``shard_range(H * W, rank, world)`` of the row-major pixel grid with its replica of the
        weights, then (``gather``) one all_gather of 36 B per ray assembles the five per-ray outputs on every rank.
        Returns (dict, (first_ray, end_ray)).  The reference renders a view on one device (run.py:45-93)."""
        from .parallel import render_frame_sharded

        def rows(first, count):
            rays_o, rays_d = ops.raygen(c2w, H, W, focal, device, first_ray=first, n_rays=count)
            u = torch.rand((count, self.N_importance), device=device, generator=generator)
            u2 = torch.rand((count, self.N_importance), device=device, generator=generator)
            out = self.render_rays(rays_o, rays_d, u, u2)
            return {k: out[k] for k in ("coarse_rgb_map", "coarse_depth_map", "fine_rgb_map", "fine_depth_map", "alpha_map")}

        with torch.no_grad():
            out, span = render_frame_sharded(rows, H * W, rank, world, group=group, gather=gather)
        out.update(rgb_map=out["fine_rgb_map"], depth_map=out["fine_depth_map"], acc_map=out["alpha_map"])
        return out, span
