"""``renderer.estimator``: the occupancy-grid refresh the reference's trainer drives every iteration.

``Trainer.train`` (/root/reference/src/train/trainers/trainer.py:58-66) calls
``self.network.renderer.estimator.update_every_n_steps(step=..., occ_eval_fn=...)`` before every forward; in the
reference ``estimator`` is a ``nerfacc.OccGridEstimator`` (volumn_renderer.py:47-54, volume_renderer.py:38-47) whose
``binaries`` the renderer seeds with the baked grid and which, while training, re-estimates the grid from the network's
own density.  nerfacc is neither vendored nor pinned, so this is a restatement of its *published* update rule
(``OccGridEstimator.update_every_n_steps`` / ``_update``, defaults ``occ_thre=1e-2, ema_decay=0.95, warmup_steps=256,
n=16``) -- **parity unpinned**, like the sampling rule:

  every ``n`` steps: cells = all of them while ``step < warmup_steps``, afterwards a uniform quarter of the grid plus
  the currently occupied cells (at most as many again); one uniformly jittered point per cell -> ``occ_eval_fn`` ->
  ``occs[cell] = max(occs[cell] * ema_decay, occ)``; ``binaries = occs > min(mean(occs[occs >= 0]), occ_thre)``.

Host-side control logic on torch tensors (index bookkeeping only); the density evaluations inside ``occ_eval_fn`` run
on the HIP kernels of whatever network the trainer passes (``KiloNetwork.forward`` -> nrf_kilo_forward).
The grid lives on the owning renderer (``owner.occ_grid``, uint8 (rx, ry, rz)); ``binaries`` is a view of it in
nerfacc's (1, rx, ry, rz) bool shape.
"""
from __future__ import annotations

import torch


class GridEstimator:
    def __init__(self, owner, aabb, resolution, active=None):
        self.owner = owner
        # `aabb`: six floats, or a callable returning them -- the KiloNeRF box lives in the network's registered buffers,
        # which load_state_dict / load_network overwrite AFTER the renderer is constructed, so it is read per update
        self._aabb = aabb if callable(aabb) else [float(v) for v in aabb]
        self.resolution = [int(r) for r in resolution]
        self.cells = self.resolution[0] * self.resolution[1] * self.resolution[2]
        self.occs = None                     # (cells,) float32 EMA of the evaluated density, created on first update
        self._active = active or (lambda: True)
        self.updates = 0

    @property
    def aabb(self):
        return [float(v) for v in (self._aabb() if callable(self._aabb) else self._aabb)]

    # nerfacc's attribute: volumn_renderer.py:54 assigns it, callers read it
    @property
    def binaries(self):
        g = self.owner.occ_grid
        return None if g is None else g.bool()[None]

    @binaries.setter
    def binaries(self, value):
        v = value[0] if value.ndim == 4 else value
        self.owner.set_occ_grid(v)

    def _grid_coords(self, idx: torch.Tensor) -> torch.Tensor:
        ry, rz = self.resolution[1], self.resolution[2]
        return torch.stack([idx // (ry * rz), (idx // rz) % ry, idx % rz], -1).to(torch.float32)

    @torch.no_grad()
    def update_every_n_steps(self, step: int, occ_eval_fn, occ_thre: float = 1e-2, ema_decay: float = 0.95,
                             warmup_steps: int = 256, n: int = 16) -> bool:
        """Returns True when the grid was re-estimated on this step.  A renderer that does not sample through the grid
        (the classic coarse + fine path) is inactive: the call returns False without evaluating anything."""
        if not getattr(self.owner, "training", True):
            raise RuntimeError("update_every_n_steps is a training-time call (nerfacc raises here as well); "
                               "put the module in train() mode")
        if not self._active() or step % n != 0:
            return False
        self._update(step, occ_eval_fn, occ_thre, ema_decay, warmup_steps)
        return True

    @torch.no_grad()
    def _update(self, step, occ_eval_fn, occ_thre, ema_decay, warmup_steps):
        grid = self.owner.occ_grid
        if grid is not None:
            dev = grid.device
        else:
            params = getattr(self.owner, "parameters", None)
            dev = next(params()).device if params is not None else torch.device("cpu")
        if self.occs is None or self.occs.device != dev:
            self.occs = torch.zeros(self.cells, dtype=torch.float32, device=dev)
        if step < warmup_steps or grid is None:
            idx = torch.arange(self.cells, device=dev)
        else:
            quarter = self.cells // 4
            uniform = torch.randint(self.cells, (quarter,), device=dev)
            occupied = grid.reshape(-1).nonzero().flatten()
            if occupied.numel() > quarter:
                occupied = occupied[torch.randint(occupied.numel(), (quarter,), device=dev)]
            idx = torch.cat([uniform, occupied])
        res = torch.tensor(self.resolution, dtype=torch.float32, device=dev)
        aabb = self.aabb
        lo = torch.tensor(aabb[:3], dtype=torch.float32, device=dev)
        hi = torch.tensor(aabb[3:], dtype=torch.float32, device=dev)
        x = (self._grid_coords(idx) + torch.rand((idx.numel(), 3), device=dev)) / res
        x = lo + x * (hi - lo)
        occ = occ_eval_fn(x).reshape(-1).to(torch.float32)
        # duplicates in idx (uniform draws, overlap with the occupied list): the last write wins, as with torch indexing
        self.occs[idx] = torch.maximum(self.occs[idx] * ema_decay, occ)
        thre = torch.clamp(self.occs[self.occs >= 0].mean(), max=occ_thre)
        self.owner.set_occ_grid((self.occs > thre).view(*self.resolution))
        self.updates += 1
