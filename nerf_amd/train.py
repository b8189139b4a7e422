"""One NeRF training step entirely on HIP kernels (BASELINE config 3).

Semantics of the reference's step (src/train/trainers/trainer.py:68-75 with
src/train/trainers/nerf.py:19-27 and src/train/optimizer.py:9-29): render coarse + fine with
perturbed samples, ``loss = mse(coarse, gt) + mse(fine, gt)``, backward, gradient mean over
ranks (DDP), ``clip_grad_value_(40)``, Adam(lr, 0.9, 0.999, 1e-8).

MI355X shape: both models' parameters live in ONE flat fp32 buffer (the module's parameters are
re-pointed at views of it, state-dict order), the backward writes ONE flat gradient buffer, so
data parallelism is a single RCCL all-reduce of 4.77 MB per step and the optimizer is one fused
kernel; the autograd route (``Renderer.render_rays_train``) stays available for callers that
own their optimizer, like the reference trainer.
"""
from __future__ import annotations

import torch

from . import ops
from .parallel import allreduce_gradients
from .renderer import near_far_table


def exponential_lr(base_lr: float, epoch: int, decay_epochs: int = 500, gamma: float = 0.1) -> float:
    """The reference's per-epoch schedule (src/utils/optimizer/lr_scheduler.py:75-79 with configs/nerf/lego.yaml:83-90:
    ``base_lr * gamma ** (epoch / decay_epochs)``; KiloNeRF: gamma 0.9 / 100 epochs while distilling, 0.1 / 500 while
    fine-tuning, configs/kilonerf/lego.yaml:124-133).  Assign the result to ``TrainStep.lr`` / ``KiloTrainStep.lr``
    once per epoch, as ``scheduler.step()`` does (train.py / train_kilonerf.py:96)."""
    return base_lr * gamma ** (epoch / decay_epochs)


def flat_slices(net, flat: torch.Tensor):
    """(parameter, offset, numel) in ``net.parameters()`` order for parameters that alias ``flat`` (fp32)."""
    base, out = flat.data_ptr(), []
    for p in net.parameters():
        off = (p.data_ptr() - base) // 4
        if not (0 <= off and off + p.numel() <= flat.numel()):
            raise RuntimeError("a parameter of the network no longer aliases the step's flat buffer")
        out.append((p, int(off), p.numel()))
    return out


def adam_state_dict(slices, exp_avg, exp_avg_sq, t: int, lr: float) -> dict:
    """``torch.optim.Adam(net.parameters(), lr, eps=1e-8).state_dict()`` carrying flat moments (the reference's 'optim'
    checkpoint entry: one group over net.parameters(), src/train/optimizer.py:9-29, net_utils.py:325-340); built by
    filling a real torch optimiser, so the layout is the installed torch's."""
    opt = torch.optim.Adam([p for p, _, _ in slices], lr=lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0)
    if t > 0:
        for p, off, n in slices:
            opt.state[p] = {"step": torch.tensor(float(t)), "exp_avg": exp_avg[off:off + n].view_as(p).clone(),
                            "exp_avg_sq": exp_avg_sq[off:off + n].view_as(p).clone()}
    return opt.state_dict()


def load_adam_state_dict(sd: dict, slices, exp_avg, exp_avg_sq):
    """The inverse -> (step count, lr or None).  Parameters without state (a fresh optimiser) get zero moments."""
    groups = sd.get("param_groups", [])
    ids = [i for g in groups for i in g["params"]]
    if len(ids) != len(slices):
        raise ValueError(f"optimizer state over {len(ids)} parameters, this network has {len(slices)}")
    exp_avg.zero_()
    exp_avg_sq.zero_()
    steps = set()
    for i, (p, off, n) in zip(ids, slices):
        st = sd["state"].get(i)
        if st is None:
            continue
        if st["exp_avg"].numel() != n:
            raise ValueError(f"optimizer state of parameter {i}: {tuple(st['exp_avg'].shape)} against {tuple(p.shape)}")
        exp_avg[off:off + n].copy_(st["exp_avg"].reshape(-1))
        exp_avg_sq[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
        steps.add(int(float(st["step"])))
    if len(steps) > 1:
        raise ValueError(f"parameters at different Adam steps {sorted(steps)}: the fused kernel keeps one count")
    return (steps.pop() if steps else 0), (float(groups[0]["lr"]) if groups else None)


class TrainStep:
    def __init__(self, net, device, near=2.0, far=6.0, n_coarse=64, n_importance=128, white_bkgd=True,
                 lr=5e-4, clip_value=40.0, process_group=None, forward_dtype="fp32", backward_dtype=None,
                 range_check_every: int = 64):
        """``forward_dtype``: "fp32" (the fp32 MFMA kernels), "bf16x3" -- both forward passes, the backward's
        data-gradient chain and the weight-gradient GEMMs on the three-term bf16 kernels (fp32-class: DESIGN 3.7; the
        forward saves the same activations) -- or "fp16x2": the two forward passes on the two-term fp16 kernel
        (same saved activations) and the backward's data-gradient chain on it too, at a power-of-two loss
        scale the kernel picks from the batch's largest |d raw|, and the weight-gradient GEMMs on the fp16 pipe as well
        (dY scaled per row set from the chain's maxima; the two head layers on a VALU streaming kernel).
        ``backward_dtype``: "bf16x3" or "fp16x2" to choose the chain's kernel separately (default: the forward's).
        ``range_check_every`` (fp16x2 only): every that many steps -- and on the first -- the packer's status word is
        read back (nrf_packed_fp16x2_status: one wait on the stream) and a weight that has outgrown fp16's range raises
        NrfError naming NRF_ERR_RANGE at that step, instead of surfacing as a NaN loss some steps later; 0 = never."""
        if forward_dtype not in ("fp32", "bf16x3", "fp16x2"):
            raise ValueError(f"forward_dtype {forward_dtype!r}: expected 'fp32', 'bf16x3' or 'fp16x2'")
        self.backward_dtype = backward_dtype or forward_dtype
        if self.backward_dtype not in ("fp32", "bf16x3", "fp16x2") or (self.backward_dtype == "fp32") != (forward_dtype == "fp32"):
            raise ValueError(f"backward_dtype {backward_dtype!r} with forward_dtype {forward_dtype!r}")
        self.forward_dtype = forward_dtype
        self.range_check_every = int(range_check_every)
        self.net = net
        self.device = torch.device(device)
        self.n_coarse, self.n_importance, self.white = n_coarse, n_importance, bool(white_bkgd)
        self.lr, self.clip = lr, clip_value
        self.group = process_group
        self.world = 1
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            self.world = torch.distributed.get_world_size(process_group)
        n = ops.model_param_floats()
        self.flat = torch.empty(2 * n, dtype=torch.float32, device=self.device)
        self.models = []
        for k, name in enumerate(("model", "model_fine")):
            model = getattr(net, name)
            if hasattr(model, "layer_params"):
                ws, bs = model.layer_params()
            else:   # the reference's own NeRF module: same layer names (network.py:10-47)
                named = dict(model.named_parameters())
                ws = [named[n + ".weight"] for n in ops.LAYER_NAMES]
                bs = [named[n + ".bias"] for n in ops.LAYER_NAMES]
            views = ops.split_flat_grad(self.flat[k * n:(k + 1) * n])
            for (w, b), (vw, vb) in zip(zip(ws, bs), views):
                vw.copy_(w.detach())
                vb.copy_(b.detach())
                w.data, b.data = vw, vb            # parameters now alias the flat buffer
            self.models.append((ws, bs))
        self.grad = torch.zeros_like(self.flat)
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self.t = 0
        self.bins = torch.from_numpy(near_far_table(near, far, n_coarse)).to(self.device)
        self._packed = [None, None]
        self._packed_t = [None, None]
        self._packed_x3 = [None, None]
        self._ws = None
        self.loss = torch.zeros(1, dtype=torch.float32, device=self.device)
        # fp16x2 backward: with `log_scales` set (a list), every step appends the (coarse, fine) scale blocks of its two
        # chains (ops.backward_scale_block: device tensors, no wait) -- the loss scale picked per step is word 0
        self.log_scales = None

    def _pack(self):
        for k, (ws, bs) in enumerate(self.models):
            self._packed[k] = ops.pack_weights(ws, bs, out=self._packed[k])
            if self.forward_dtype == "bf16x3":
                self._packed_x3[k] = ops.pack_weights_bf16x3(ws, bs, out=self._packed_x3[k])
            elif self.forward_dtype == "fp16x2":
                # the range check waits for the stream, so it runs every `range_check_every` steps only; between two
                # checks a weight beyond fp16 shows as a NaN loss (the kernels return NaN for such a stream)
                due = self.range_check_every > 0 and self.t % self.range_check_every == 0
                self._packed_x3[k] = ops.pack_weights_fp16x2(ws, bs, out=self._packed_x3[k], check_range=due)
            if self.backward_dtype == "bf16x3":
                self._packed_t[k] = ops.pack_weights_transposed_bf16x3(ws, out=self._packed_t[k])
            elif self.backward_dtype == "fp16x2":
                self._packed_t[k] = ops.pack_weights_transposed_fp16x2(ws, out=self._packed_t[k])
            else:
                self._packed_t[k] = ops.pack_weights_transposed(ws, out=self._packed_t[k])

    def forward_backward(self, rays_o, rays_d, gt, u, u2, u_coarse=None, u_fine=None):
        """Fills self.grad (both models, flat) and self.loss; returns the render outputs."""
        n = ops.model_param_floats()
        self._pack()
        pc, pf = self._packed
        pc_t, pf_t = self._packed_t
        zc = self.bins if u_coarse is None else ops.perturb_samples(self.bins, u_coarse)
        fc, ff = (pc, pf) if self.forward_dtype == "fp32" else self._packed_x3
        raw_c, acts_c, masks_c = ops.mlp_forward_rays_train(fc, rays_o, rays_d, zc, self.forward_dtype)
        c_rgb, c_depth, w_c, _ = ops.composite(raw_c, zc, self.white, True)
        z_f = ops.sample_fine(w_c, self.bins, zc, u, u2, u_fine)
        raw_f, acts_f, masks_f = ops.mlp_forward_rays_train(ff, rays_o, rays_d, z_f, self.forward_dtype)
        f_rgb, f_depth, _, f_acc = ops.composite(raw_f, z_f, self.white, False)
        self.loss.zero_()
        g_c = ops.mse_grad(c_rgb, gt, self.loss)                     # trainers/nerf.py:25-27
        g_f = ops.mse_grad(f_rgb, gt, self.loss)
        need = int(ops.lib().nrf_mlp_backward_workspace_bytes(raw_f.shape[0], raw_f.shape[1]))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        g_raw_c = ops.composite_backward(raw_c, zc, self.white, g_c)
        ops.mlp_backward(pc, pc_t, g_raw_c, acts_c, masks_c, self._ws, out=self.grad[:n], kind=self.backward_dtype)
        log = self.log_scales is not None and self.backward_dtype == "fp16x2"
        blk_c = ops.backward_scale_block(self._ws, raw_c.shape[0], raw_c.shape[1]) if log else None
        del g_raw_c, acts_c, masks_c, raw_c
        g_raw_f = ops.composite_backward(raw_f, z_f, self.white, g_f)
        ops.mlp_backward(pf, pf_t, g_raw_f, acts_f, masks_f, self._ws, out=self.grad[n:], kind=self.backward_dtype)
        if log:
            self.log_scales.append((blk_c, ops.backward_scale_block(self._ws, raw_f.shape[0], raw_f.shape[1])))
        return {"coarse_rgb_map": c_rgb, "fine_rgb_map": f_rgb, "fine_depth_map": f_depth, "alpha_map": f_acc}

    def step(self, rays_o, rays_d, gt, u, u2, u_coarse=None, u_fine=None):
        out = self.forward_backward(rays_o, rays_d, gt, u, u2, u_coarse, u_fine)
        scale = allreduce_gradients(self.grad, self.group)   # DDP semantics: ONE flat sum all-reduce, mean in the kernel
        self.t += 1
        ops.clip_adam(self.flat, self.grad, self.exp_avg, self.exp_avg_sq, self.t, lr=self.lr, clip_value=self.clip,
                      grad_scale=scale)
        self._bump_versions()
        return out

    # ---- checkpoint interop: the reference saves optim.state_dict() of ONE torch.optim.Adam group over
    # net.parameters() (src/train/optimizer.py:9-29, src/utils/net_utils.py:325-340) and resumes from it (:288-322) ----
    def _flat_slices(self):
        return flat_slices(self.net, self.flat)

    def optimizer_state_dict(self) -> dict:
        """The step's Adam state as ``torch.optim.Adam(net.parameters(), lr, eps=1e-8).state_dict()`` would hold it after
        the same steps: what ``checkpoint.save_model(..., optim_state=...)`` writes as ``'optim'`` for the reference's
        ``load_model``."""
        return adam_state_dict(self._flat_slices(), self.exp_avg, self.exp_avg_sq, self.t, self.lr)

    def load_optimizer_state_dict(self, sd: dict) -> None:
        """Resume from the ``'optim'`` entry of a checkpoint (this class's or the reference's own Adam over the same network)."""
        self.t, lr = load_adam_state_dict(sd, self._flat_slices(), self.exp_avg, self.exp_avg_sq)
        if lr is not None:
            self.lr = lr

    def _bump_versions(self):
        """The optimiser kernel writes the flat buffer behind torch's back, so the parameters (views of it) would keep
        their version counters -- and anything keyed on them would go on using the old weights: a ``Renderer`` over
        the same network (the reference's trainer validates with one every epoch, trainer.py:111-147) caches its packed
        weight streams per (data_ptr, _version).  Same contract as ``optimizer.step()``: every parameter's version
        moves."""
        bump = torch.autograd.graph.increment_version
        for ws, bs in self.models:
            for p in list(ws) + list(bs):
                bump(p)
