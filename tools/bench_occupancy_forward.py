#!/usr/bin/env python3
"""Time the renderer's occupancy-grid path on a full 800x800 view: 256^3 vertex bake + render_occupancy
(nerfacc's default step 1e-3, early stop 1e-4) next to the classic 64 + 128 path on the same weights."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_amd import ops  # noqa: E402
from nerf_amd.camera import focal_from_angle, test_set_poses  # noqa: E402
from nerf_amd.config import default_cfg  # noqa: E402
from nerf_amd.network import Network  # noqa: E402
from nerf_amd.renderer import Renderer  # noqa: E402
from nerf_amd.weights import make_state_dict  # noqa: E402

dev = torch.device("cuda:0")
net = Network()
net.load_state_dict(make_state_dict(5, 30.0, -1.0))
r = Renderer(net.to(dev), default_cfg(perturb=0))
o, d = ops.raygen(test_set_poses(200)[0], 800, 800, focal_from_angle(800), dev)
batch = {"xyz": o[None], "viewdirs": d[None], "gt_rgb": o[None]}


def timed(fn, n=2):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n, out


t_bake, occ = timed(lambda: r.build_occ_grid(0.01), 1)
print(f"bake 256^3 vertices: {t_bake * 1e3:.1f} ms, occupied {float(occ.float().mean()):.3f}")
t_occ, out = timed(lambda: r.render_occupancy(batch))
print(f"render_occupancy: {t_occ * 1e3:.1f} ms, {r.last_num_samples / 1e6:.1f} M samples, "
      f"{r.last_num_samples * 1186816 / t_occ / 1e12:.1f} TFLOP/s over the whole call")
with torch.no_grad():
    t_cls, _ = timed(lambda: r(batch, is_training=False), 1)
print(f"classic 64 + 128: {t_cls * 1e3:.1f} ms")
rb = Renderer(net, default_cfg(perturb=0), mlp_dtype="bf16")
rb.occ_grid = r.occ_grid
t_b, out_b = timed(lambda: rb.render_occupancy(batch))
err = float((out_b["fine_rgb_map"] - out["fine_rgb_map"]).abs().max())
print(f"render_occupancy, bf16 MLP: {t_b * 1e3:.1f} ms ({rb.last_num_samples * 1186816 / t_b / 1e12:.0f} TFLOP/s over the whole call), "
      f"max |rgb - fp32| {err:.2e}")
