#!/usr/bin/env python3
"""How many of a KiloNeRF frame's marched samples lie BEFORE their ray's early-termination point (T >= ert)?  The frame
evaluates the tiny MLP on every marched sample and drops the ones behind the termination point at compositing; this
prints what a depth-ordered evaluation that stops at termination could skip (bench.py --mode kilo's scene)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nerf_amd import kilo_ops, ops  # noqa: E402
from nerf_amd.camera import focal_from_angle, test_set_poses  # noqa: E402
from nerf_amd.kilonerf import KiloNetwork, KiloRenderer, default_kilo_cfg  # noqa: E402
from nerf_amd.weights import make_kilo_state_dict_arrays  # noqa: E402

dev = torch.device("cuda:0")
net = KiloNetwork()
net.load_state_dict({k: torch.from_numpy(v) for k, v in make_kilo_state_dict_arrays(0, 4096, 10.0, 0.5).items()}, strict=False)
net.to(dev)
c = (torch.arange(256, device=dev, dtype=torch.float32) + 0.5) / 256 * 2 - 1
occ = ((c[:, None, None] ** 2 + c[None, :, None] ** 2 + c[None, None, :] ** 2).sqrt() < 0.75).to(torch.uint8)
r = KiloRenderer(net, default_kilo_cfg(), occ)
aabb, res = net.grid() if hasattr(net, "grid") else r.view.grid()
o, d = ops.raygen(test_set_poses(200)[0], 800, 800, focal_from_angle(800), dev)
pts, ray_idx, ts, te, offsets, counts = kilo_ops.kilo_march(o, d, occ, aabb, r.near, r.far, r.step)
raw = kilo_ops.kilo_forward(r.view.packed(), r.view.num_models, aabb, res, pts, d, ray_idx)
sig = torch.relu(raw[:, 3]).double()
tau = sig * (te - ts).double()
csum = torch.cumsum(tau, 0)
start = torch.zeros(o.shape[0], dtype=torch.float64, device=dev)
nz = counts > 0
first = offsets[nz].long()
start[nz] = csum[first] - tau[first]
excl = csum - tau - start[ray_idx.long()]            # optical depth in front of each sample
T = torch.exp(-excl)
alive = T >= r.ert
n = pts.shape[0]
print(f"samples {n}, before termination {int(alive.sum())} = {float(alive.sum()) / n:.3f}")
pos = torch.arange(n, device=dev) - offsets[ray_idx.long()].long()
for k in (16, 32, 64, 96, 128, 192):
    print(f"  samples with index < {k} in their ray: {float((pos < k).sum()) / n:.3f}; of the live ones {float((alive & (pos < k)).sum()) / float(alive.sum()):.3f}")
last = torch.zeros(o.shape[0], dtype=torch.long, device=dev)
last.scatter_reduce_(0, ray_idx.long(), torch.where(alive, pos + 1, torch.zeros_like(pos)), reduce="amax")
print("rays with samples", int(nz.sum()), "mean live samples per such ray", float(last[nz].double().mean()), "mean marched", float(counts[nz].double().mean()))
