"""Distillation data generation: what /root/reference/gen_dist_data.py:16-90 does
with the teacher NeRF renderer, sharded over the GPUs of a node.

For every training view the teacher renders the full image (``renderer.render(batch, is_training=False)``, :72)
and a dict ``{'xyz' (R,3), 'viewdirs' (R,3), 'teacher_rgb' (R,3), 'teacher_alpha' (R,1)}`` is written as
``dist_{i:04d}.pt`` (:76-85) -- the files ``DistDataset`` reloads (src/datasets/kilonerf/blender.py:168-220).
Views are dealt round-robin over ranks (no data-path collective); rays come from nrf_raygen, so no DataLoader,
no host-side ray tensors and no H2D copies are involved.
"""
from __future__ import annotations

import os

import torch

from . import ops
from .parallel import view_schedule


def distill_view(renderer, c2w, H: int, W: int, focal: float, device) -> dict:
    """One view's record, tensors on the CPU like the reference's (:76-81)."""
    rays_o, rays_d = ops.raygen(c2w, H, W, focal, device)
    with torch.no_grad():
        out = renderer.render({"xyz": rays_o[None], "viewdirs": rays_d[None], "gt_rgb": rays_o[None]}, is_training=False)
    return {"xyz": rays_o.cpu(), "viewdirs": rays_d.cpu(), "teacher_rgb": out["fine_rgb_map"].cpu(),
            "teacher_alpha": out["alpha_map"].cpu()}


def generate(renderer, poses, H: int, W: int, focal: float, out_dir: str, device, rank: int = 0, world: int = 1,
             save=torch.save) -> list:
    """Writes ``dist_{view:04d}.pt`` for this rank's share of ``poses``; returns the paths written."""
    os.makedirs(out_dir, exist_ok=True)
    paths = []
    for view, _tile in view_schedule(len(poses), rank, world):
        path = os.path.join(out_dir, f"dist_{view:04d}.pt")
        save(distill_view(renderer, poses[view], H, W, focal, device), path)
        paths.append(path)
    return paths


class DistRecords:
    """The reader side: ``DistDataset`` of /root/reference/src/datasets/kilonerf/blender.py:168-220.  All ``.pt``
    records of ``data_root`` (sorted by name, :183-187) are loaded once (``torch.load(weights_only=True)``) and
    kept on ``device`` -- 100 full 800x800 views are 2.6 GB, nothing next to 288 GB of HBM, so a training batch is
    an index_select on the GPU instead of a DataLoader worker + H2D copy.  ``self[i]`` is one item (``N_rays``
    rays of view i drawn without replacement, :207-217); ``batch(view_ids)`` stacks items the way the DataLoader's
    collate does: every key ``(B, N_rays, C)``."""

    KEYS = ("xyz", "viewdirs", "teacher_rgb", "teacher_alpha")

    def __init__(self, data_root: str, n_rays: int = 1024, device="cpu", generator: torch.Generator | None = None):
        self.files = sorted(os.path.join(data_root, f) for f in os.listdir(data_root) if f.endswith(".pt"))
        if not self.files:
            raise FileNotFoundError(f"No distilled data (.pt files) found in {data_root}")   # :189-190
        self.n_rays = int(n_rays)
        self.device = torch.device(device)
        self.generator = generator
        self.records = []
        for path in self.files:
            rec = torch.load(path, map_location="cpu", weights_only=True)
            missing = [k for k in self.KEYS if k not in rec]
            if missing:
                raise KeyError(f"{path}: missing {missing}")
            self.records.append({k: rec[k].to(self.device) for k in self.KEYS})

    def __len__(self):
        return len(self.files)                                # one item per original image (:199-201)

    def __getitem__(self, index: int) -> dict:
        rec = self.records[index]
        n = rec["xyz"].shape[0]
        if self.n_rays > n:
            raise ValueError(f"record {index} has {n} rays, fewer than N_rays = {self.n_rays}")
        sel = torch.randperm(n, device=self.device, generator=self.generator)[: self.n_rays]
        return {k: rec[k].index_select(0, sel) for k in self.KEYS}

    def batch(self, view_ids) -> dict:
        items = [self[int(i)] for i in view_ids]
        return {k: torch.stack([it[k] for it in items], 0) for k in self.KEYS}
