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
        self._call = None                                     # kilo_ops.KiloTrainCall, built on the first batch
        self._reach, self._reach_key = None, None
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
        # nrf_kilo_clip_adam's two-bit mask per cell.  Bit 0: the cell has had a gradient since the moments were zeroed
        # (elsewhere the eight unregularised tensors have g = m = v = 0 and Adam's update is the identity, so the pass
        # skips them); bit 1: the same for the four colour-regularised tensors, whose moments are non-zero in EVERY
        # cell from the first regularised step on (`_reg_moments`).
        self.live = torch.zeros(self.params[0].shape[0], dtype=torch.int32, device=self.device)
        self._reg_moments = False
        views = lambda flat: [flat[g.storage_offset():g.storage_offset() + g.numel()].view(g.shape) for g in self.grads]  # noqa: E731
        self._adam_ptrs = tuple(kilo_ops.ParamPointers(x) for x in (self.params, self.grads, views(self.exp_avg),
                                                                    views(self.exp_avg_sq)))

    def load_optimizer_state(self, exp_avg, exp_avg_sq, t: int):
        """Adam moments from elsewhere (the reference's resume: net_utils.py:288-322 restores 'optim'): flat fp32
        tensors in this step's parameter order.  Nothing is known about which cells they are zero in, so the optimiser
        pass stops skipping (every cell live in both bits)."""
        self.exp_avg.copy_(exp_avg.reshape(-1))
        self.exp_avg_sq.copy_(exp_avg_sq.reshape(-1))
        self.t = int(t)
        self.live.fill_(3)
        self._reg_moments = True

    def optimizer_state_dict(self) -> dict:
        """Adam state in the reference's checkpoint layout (one torch.optim.Adam group over net.parameters(),
        src/train/optimizer.py:9-29): see nerf_amd.train.adam_state_dict."""
        from .train import adam_state_dict, flat_slices
        return adam_state_dict(flat_slices(self.net, self.flat), self.exp_avg, self.exp_avg_sq, self.t, self.lr)

    def load_optimizer_state_dict(self, sd: dict) -> None:
        """Resume from a checkpoint's 'optim' entry (net_utils.py:288-322).  Nothing is known about where foreign
        moments are zero, so the per-cell skipping is switched off (every cell live in both bits)."""
        from .train import flat_slices, load_adam_state_dict
        self.t, lr = load_adam_state_dict(sd, flat_slices(self.net, self.flat), self.exp_avg, self.exp_avg_sq)
        if lr is not None:
            self.lr = lr
        self.live.fill_(3)
        self._reg_moments = True

    def _reachable_cells(self, occ, res):
        """int32 mask over the cells: 1 where some occupancy voxel inside the cell is set.  A hint, not a contract (the
        forward packs any cell the batch routes to that this mask misses); all ones when the occupancy grid is not a
        whole multiple of the cell grid.  Recomputed when the grid tensor changes (estimator updates)."""
        key = (occ.data_ptr(), occ._version, tuple(occ.shape), tuple(res))
        if key != self._reach_key:
            shape, res = list(occ.shape), [int(v) for v in res]
            if all(s % c == 0 for s, c in zip(shape, res)):
                f = [s // c for s, c in zip(shape, res)]
                m = occ.reshape(res[0], f[0], res[1], f[1], res[2], f[2]).amax(dim=(1, 3, 5))
                self._reach = (m != 0).to(torch.int32).reshape(-1).contiguous()      # flat index (x * ry + y) * rz + z
            else:
                self._reach = torch.ones(res[0] * res[1] * res[2], dtype=torch.int32, device=occ.device)
            self._reach_key = key
        return self._reach

    def forward_backward(self, rays_o, rays_d, target_rgb, target_alpha=None, stage="distillation", defer_reg=False,
                         want_touched=None):
        """Fills self.grad (all 12 tensors, flat) and self.loss; returns the render outputs.  ``defer_reg``: leave
        the colour regulariser out (``step`` folds it into the optimiser pass); ``want_touched`` (default: same as
        ``defer_reg``): record in ``self.touched`` which cells this batch sampled (the compact gradient exchange)."""
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
        # The parameters changed in the last step, so the kernels' records are repacked from the live tensors -- only
        # for the cells this batch can reach.  Samples exist only in occupied voxels: the cells that hold one
        # (`_reachable_cells`, about a third of the grid for a scene inside its occupancy grid) are packed while the
        # host waits for the march's sample count; the forward then packs whatever else the batch routes to (normally
        # nothing) before its MLPs.  Nobody reads the other cells' records.  The whole chain -- march, forward,
        # compositing, losses, backward -- is ONE C call (nrf_kilo_train_forward_backward): thirty launches at 2-3 us
        # each instead of Python's 10-20 us, which the 5-15 us kernels of this step cannot hide.
        if self._call is None:
            self._packed = torch.empty(kilo_ops.kilo_packed_floats(net.num_models), dtype=torch.float32, device=o.device)
            self._packed_t = torch.empty(kilo_ops.kilo_packed_bwd_floats(net.num_models), dtype=torch.float32, device=o.device)
            self._call = kilo_ops.KiloTrainCall(self.params, self.grads, self._packed, self._packed_t, self.loss)
        distill = stage == "distillation"
        touched = None
        if defer_reg if want_touched is None else want_touched:
            if self.touched is None:
                self.touched = torch.empty(net.num_models, dtype=torch.int32, device=o.device)
            touched = self.touched
        rgb, depth, acc, self.num_samples = self._call(
            o, d, r.occ_grid, aabb, res, r.near, r.far, r.step, r.ert, r.white_bkgd, target_rgb,
            target_alpha if distill else None, self._reachable_cells(r.occ_grid, res), touched,
            self.live if self.world == 1 else None)
        r.last_num_samples = self.num_samples
        self._reg_pending = bool(distill and self.l2_reg > 0)
        if not defer_reg and self._reg_pending:   # the four regularised tensors are the tail of the flat buffer: one launch
            kilo_ops.l2_reg(self.flat[self._reg_begin:], self.l2_reg, self.grad[self._reg_begin:], self.loss)
            self._reg_pending = False
        return {"rgb_map": rgb, "depth_map": depth, "alpha_map": acc}

    def step(self, rays_o, rays_d, target_rgb, target_alpha=None, stage="distillation"):
        sparse = self.world > 1 and self.sparse_allreduce
        out = self.forward_backward(rays_o, rays_d, target_rgb, target_alpha, stage, defer_reg=True, want_touched=sparse)
        live = self.live
        if sparse:
            # DDP semantics (mean gradient) with a compact exchange: only cells some rank sampled travel
            scale = allreduce_cell_gradients(self.grads, self.touched, self.group)
            self.live.bitwise_or_(self.touched * 3)              # (now the union over the ranks; both mask bits)
        else:
            scale = allreduce_gradients(self.grad, self.group)   # ONE flat sum all-reduce, mean in the kernel
            if self.world > 1:
                live = None                                      # which cells the other ranks sampled is not known here
        self.t += 1
        # The regulariser is the same on every rank and non-zero everywhere: it rides in the optimiser pass, which reads
        # the parameters anyway -- times world on the summed gradient (the kernel divides the sum by world), its loss
        # term this rank's own.  self.grad ends as nrf_l2_reg would have left it.
        reg = self._reg_pending
        self._reg_pending = False
        if reg and not self._reg_moments:
            # from this step on every cell has moments in the four regularised tensors: a later step WITHOUT the
            # regulariser (fine-tuning without reset_optimizer) must keep updating them everywhere, as torch Adam does
            self.live.bitwise_or_(2)
            self._reg_moments = True
        kilo_ops.kilo_clip_adam(*self._adam_ptrs, self.t, live=live, lr=self.lr, clip_value=self.clip, grad_scale=scale,
                                reg_grad_coeff=self.l2_reg * self.world if reg else 0.0,
                                reg_loss_coeff=self.l2_reg if reg else 0.0, loss_accum=self.loss)
        self.view.invalidate()        # the in-place update does not bump tensor versions: drop the inference pack
        return out
