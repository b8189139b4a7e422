#!/usr/bin/env python3
"""cProfile of KiloTrainStep.step on the host side (200 steps)."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nerf_amd import ops, kilo_train  # noqa: E402
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
ts = kilo_train.KiloTrainStep(r, dev, lr=5e-4, clip_value=40.0, l2_reg=1e-6)
poses, focal = test_set_poses(200), focal_from_angle(800)
gen = torch.Generator(device=dev).manual_seed(977)
R = 4096


def run(n, k0):
    for k in range(k0, k0 + n):
        pix = torch.randint(0, 640000, (R,), device=dev, generator=gen)
        o, d = ops.raygen(poses[k % 200], 800, 800, focal, dev, pixel_index=pix)
        ts.step(o, d, torch.rand((R, 3), device=dev, generator=gen), torch.rand((R, 1), device=dev, generator=gen),
                stage="distillation")
    torch.cuda.synchronize()


run(20, 0)
pr = cProfile.Profile()
pr.enable()
run(200, 20)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
