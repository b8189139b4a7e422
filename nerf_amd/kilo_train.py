"""One KiloNeRF training step entirely on HIP kernels.

Semantics of the reference's step for the KiloNeRF wrapper (src/train/trainers/kilonerf.py:34-63 inside
src/train/trainers/trainer.py:68-75, optimiser src/train/optimizer.py:9-29, schedule train_kilonerf.py:70-96):

  * stage ``'distillation'``: ``loss = mse(teacher_rgb, rgb_map) + mse(teacher_alpha, alpha_map)
    + l2_reg * net.get_color_reg_loss()`` (:43-55);
  * stage ``'fine-tuning'``: ``loss = mse(gt_rgb, rgb_map)`` (:57-61), with a fresh optimiser at the stage switch
    (train_kilonerf.py:79-86 -> ``reset_optimizer``);
  * backward, gradient mean over ranks, ``clip_grad_value_(40)``, Adam.

MI355X shape: the 12 batched parameter tensors live in ONE flat fp32 buffer (25.4 M floats; the module's
parameters are re-pointed at views of it), the backward kernel writes ONE flat gradient buffer (every element,
no memset), so data parallelism is a single RCCL all-reduce and the optimiser one fused kernel; with more than one
rank the all-reduce carries only the cells some rank sampled (``parallel.allreduce_cell_gradients``: a quarter of
the 102 MB for the sphere scene), the dense form stays available (``sparse_allreduce=False``).  The autograd
route (``KiloRenderer.forward`` with ``is_training``) stays available for callers that own their optimiser.
"""
from __future__ import annotations

import torch

from . import kilo_ops, ops
from .parallel import allreduce_cell_gradients, allreduce_gradients
from .weights import KILO_TENSORS

_REG_TENSORS = ("view_linears_weights", "view_linears_biases", "rgb_linear_weights", "rgb_linear_biases")


class KiloTrainStep:
    def __init__(self, renderer, device, lr=5e-4, clip_value=40.0, l2_reg=1e-6, process_group=None,
                 sparse_allreduce: bool = True):
        self.renderer = renderer
        self.net = renderer.net
        self.view = renderer.view
        self.device = torch.device(device)
        self.lr, self.clip, self.l2_reg = lr, clip_value, l2_reg
        self.group = process_group
        self.sparse_allreduce = bool(sparse_allreduce)   # ranks exchange only the cells some rank sampled
        self.world = 1
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            self.world = torch.distributed.get_world_size(process_group)
        params = self.view.tensors()
        sizes = [p.numel() for p in params]
        self.flat = torch.empty(sum(sizes), dtype=torch.float32, device=self.device)
        self.grad = torch.empty_like(self.flat)
        self.params, self.grads = [], []
        off = 0
        for p, n in zip(params, sizes):
            v = self.flat[off:off + n].view(p.shape)
            v.copy_(p.detach())
            p.data = v                                   # parameters now alias the flat buffer
            self.params.append(v)
            self.grads.append(self.grad[off:off + n].view(p.shape))
            off += n
        names = [n for n, _, _ in KILO_TENSORS]
        assert tuple(names[-4:]) == _REG_TENSORS, "colour-regularised tensors must be the last four of the flat buffer"
        self._reg_begin = sum(sizes[:-4])
        self.reset_optimizer(lr)
        self._packed = None
        self._packed_t = None
        self.loss = torch.zeros(1, dtype=torch.float32, device=self.device)
        self.num_samples = 0
        self.touched = None          # (cells,) int32 mask of the cells the last deferred-regulariser batch sampled
        self._reg_pending = False

    def reset_optimizer(self, lr=None):
        """Fresh Adam state (the stage switch re-creates the optimiser, train_kilonerf.py:79-86)."""
        if lr is not None:
            self.lr = lr
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self.t = 0

    def forward_backward(self, rays_o, rays_d, target_rgb, target_alpha=None, stage="distillation", defer_reg=False):
        """Fills self.grad (all 12 tensors, flat) and self.loss; returns the render outputs.  ``defer_reg``: leave
        the colour regulariser out (``step`` adds it after a compact gradient exchange) and record in
        ``self.touched`` which cells this batch sampled."""
        if stage not in ("distillation", "fine-tuning"):
            raise ValueError(f"stage {stage!r}: expected 'distillation' or 'fine-tuning' (train_kilonerf.py:74-78)")
        if stage == "distillation" and target_alpha is None:
            raise ValueError("the distillation stage needs teacher_alpha (trainers/kilonerf.py:44-47)")
        if self.renderer.occ_grid is None:
            raise RuntimeError("KiloRenderer: no occupancy grid (set_occ_grid)")
        r, net = self.renderer, self.view
        o, d = rays_o.reshape(-1, 3).contiguous(), rays_d.reshape(-1, 3).contiguous()
        aabb, res = net.grid()
        if r.occ_grid.device != o.device:
            r.occ_grid = r.occ_grid.to(o.device)
        # march first: its count pass ends in the step's one host sync (the sample count), and the repack of all
        # 4096 cells (0.09 ms of GPU time, no dependence on the samples) then runs while the host enqueues the rest
        pts, ray_idx, ts, te, offsets, counts = kilo_ops.kilo_march(o, d, r.occ_grid, aabb, r.near, r.far, r.step)
        self._packed = kilo_ops.kilo_pack_weights(self.params, out=self._packed)
        self._packed_t = kilo_ops.kilo_pack_weights_transposed(self.params, out=self._packed_t)
        self.num_samples = r.last_num_samples = int(pts.shape[0])
        raw, ws = kilo_ops.kilo_forward(self._packed, net.num_models, aabb, res, pts, d, ray_idx, return_workspace=True)
        rgb, depth, acc = kilo_ops.kilo_composite(raw, ts, te, offsets, counts, r.ert, r.white_bkgd)
        self.loss.zero_()
        g_rgb = ops.mse_grad(rgb, target_rgb.reshape(-1, 3), self.loss)
        g_acc = None
        distill = stage == "distillation"
        if distill:
            g_acc = ops.mse_grad(acc, target_alpha.reshape(-1), self.loss)
        g_raw = kilo_ops.kilo_composite_backward(raw, ts, te, offsets, counts, r.ert, r.white_bkgd, g_rgb, None, g_acc)
        kilo_ops.kilo_backward(self._packed, self._packed_t, net.num_models, pts, d, ray_idx, g_raw, ws, grads=self.grads)
        self._reg_pending = bool(distill and self.l2_reg > 0)
        if defer_reg:
            self.touched = torch.zeros(net.num_models, dtype=torch.int32, device=o.device)
            if pts.shape[0]:
                self.touched.index_fill_(0, kilo_ops.kilo_route(pts, aabb, res).long(), 1)
        elif self._reg_pending:   # the four regularised tensors are the tail of the flat buffer: one launch
            kilo_ops.l2_reg(self.flat[self._reg_begin:], self.l2_reg, self.grad[self._reg_begin:], self.loss)
            self._reg_pending = False
        return {"rgb_map": rgb, "depth_map": depth, "alpha_map": acc}

    def step(self, rays_o, rays_d, target_rgb, target_alpha=None, stage="distillation"):
        sparse = self.world > 1 and self.sparse_allreduce
        out = self.forward_backward(rays_o, rays_d, target_rgb, target_alpha, stage, defer_reg=sparse)
        if sparse:
            # DDP semantics (mean gradient) with a compact exchange: only cells some rank sampled travel.  The
            # regulariser is the same on every rank and non-zero everywhere, so it is added afterwards, times
            # world (the optimiser kernel divides the sum by world); its loss term is this rank's own.
            scale = allreduce_cell_gradients(self.grads, self.touched, self.group)
            if self._reg_pending:
                tail = self.flat[self._reg_begin:]
                kilo_ops.l2_reg(tail, self.l2_reg * self.world, self.grad[self._reg_begin:], None)
                kilo_ops.l2_reg(tail, self.l2_reg, None, self.loss)
                self._reg_pending = False
        else:
            scale = allreduce_gradients(self.grad, self.group)   # ONE flat sum all-reduce, mean in the kernel
        self.t += 1
        ops.clip_adam(self.flat, self.grad, self.exp_avg, self.exp_avg_sq, self.t, lr=self.lr, clip_value=self.clip,
                      grad_scale=scale)
        self.view.invalidate()        # the in-place update does not bump tensor versions: drop the inference pack
        return out
