#!/usr/bin/env python3
"""Small fixed workload for rocprofv3 runs: each hot kernel a few times on 65,536 rays.

    rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o run -- python3 tools/profile_kernels.py
    rocprofv3 --pmc SQ_WAVE_CYCLES ... -- python3 tools/profile_kernels.py --only mlp
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_amd import ops  # noqa: E402
from nerf_amd.camera import pose_spherical, focal_from_angle  # noqa: E402
from nerf_amd.renderer import near_far_table  # noqa: E402
from nerf_amd.weights import make_state_dict_arrays  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rays", type=int, default=65536)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--only", default="all", choices=["all", "mlp", "composite", "sample"])
a = ap.parse_args()

dev = torch.device("cuda:0")
sd = make_state_dict_arrays(0, 30.0, 2.0)
pc, pf = ops.pack_state_dict(sd, "model", dev), ops.pack_state_dict(sd, "model_fine", dev)
o, d = ops.raygen(pose_spherical(0.0), 800, 800, focal_from_angle(800), dev, first_ray=300 * 800, n_rays=a.rays)
bins = torch.from_numpy(near_far_table(2, 6, 64)).to(dev)
g = torch.Generator(device=dev).manual_seed(0)
u = torch.rand((a.rays, 128), device=dev, generator=g)
u2 = torch.rand((a.rays, 128), device=dev, generator=g)
raw_c = ops.mlp_forward_rays(pc, o, d, bins)
_, _, w_c, _ = ops.composite(raw_c, bins, True, True)
z_f = ops.sample_fine(w_c, bins, bins, u, u2)
raw_f = ops.mlp_forward_rays(pf, o, d, z_f)
torch.cuda.synchronize()
for _ in range(a.reps):
    if a.only in ("all", "mlp"):
        ops.mlp_forward_rays(pc, o, d, bins)
        ops.mlp_forward_rays(pf, o, d, z_f)
    if a.only in ("all", "composite"):
        ops.composite(raw_c, bins, True, True)
        ops.composite(raw_f, z_f, True, False)
    if a.only in ("all", "sample"):
        ops.sample_fine(w_c, bins, bins, u, u2)
torch.cuda.synchronize()
print("done")
