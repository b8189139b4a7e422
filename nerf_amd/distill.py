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
