#!/usr/bin/env python3
"""Time nrf_kilo_backward alone on the kilo_train bench batch (4096 random rays, sphere occupancy).
usage: kilo_bwd_time.py [path/to/libnerf_amd.so]   (ablation builds; default: the in-tree library)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import nerf_amd._lib as _lib  # noqa: E402

if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from nerf_amd import kilo_ops, ops  # noqa: E402
from nerf_amd.camera import focal_from_angle, test_set_poses  # noqa: E402
from nerf_amd.weights import KILO_TENSORS, make_kilo_state_dict_arrays  # noqa: E402

dev = torch.device("cuda:0")
AABB, RES = [-1.0, -1.0, -1.0, 1.0, 1.0, 1.0], [16, 16, 16]
sd = make_kilo_state_dict_arrays(0, 4096, 10.0, 0.5)
ts = [torch.from_numpy(sd["model." + n]).to(dev) for n, _, _ in KILO_TENSORS]
packed, packed_t = kilo_ops.kilo_pack_weights(ts), kilo_ops.kilo_pack_weights_transposed(ts)
c = (torch.arange(256, device=dev, dtype=torch.float32) + 0.5) / 256 * 2 - 1
occ = ((c[:, None, None] ** 2 + c[None, :, None] ** 2 + c[None, None, :] ** 2).sqrt() < 0.75).to(torch.uint8)
R = int(os.environ.get("RAYS", "4096"))
gen = torch.Generator(device=dev)
gen.manual_seed(977)
pix = torch.randint(0, 800 * 800, (R,), device=dev, generator=gen)
o, d = ops.raygen(test_set_poses(200)[0], 800, 800, focal_from_angle(800), dev, pixel_index=pix)
pts, ray_idx, t0, t1, offsets, counts = kilo_ops.kilo_march(o, d, occ, AABB, 2.0, 6.0, 0.005)
raw, ws = kilo_ops.kilo_forward(packed, 4096, AABB, RES, pts, d, ray_idx, return_workspace=True)
cell = kilo_ops.kilo_route(pts, AABB, RES)
cnt = torch.bincount(cell.long(), minlength=4096)
g_raw = torch.randn(raw.shape, device=dev, generator=gen)
grads = None
for it in range(3):
    grads = kilo_ops.kilo_backward(packed, packed_t, 4096, pts, d, ray_idx, g_raw, ws, grads=grads)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
N = 20
e0.record()
for it in range(N):
    kilo_ops.kilo_backward(packed, packed_t, 4096, pts, d, ray_idx, g_raw, ws, grads=grads)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / N
n = pts.shape[0]
print(f"{os.path.basename(_lib.LIB_PATH)}: samples {n}, cells {int((cnt > 0).sum())}, max/cell {int(cnt.max())}, "
      f"tiles {int(((cnt + 31) // 32).sum())}: backward {ms:.3f} ms = {ms * 1e3 / max(1, n / 32):.2f} us per 32-point tile "
      f"(whole chip), checksum {float(sum(g.double().abs().sum() for g in grads)):.6e}")
print("  per tensor:", " ".join(f"{float(g.double().abs().sum()):.7e}" for g in grads))
