#!/usr/bin/env python3
"""Where a wave of kilo_mlp_x3_kernel spends its time: a -DKF_STAMP build of kilo.hip sums s_memtime ticks (100 MHz) per
phase over all waves, on bench.py --mode kilo's frame.
    bash tools/debug/lib_ab.sh kilo.hip stamp=-DKF_STAMP     (here)
    python tools/debug/kilo_fwd_phases.py nerf_amd/lib/ab_stamp.so    (on the GPU box)"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import nerf_amd._lib as _lib  # noqa: E402

_lib.LIB_PATH = os.path.abspath(sys.argv[1])
import torch  # noqa: E402

from nerf_amd import kilo_ops, ops  # noqa: E402
from nerf_amd.camera import focal_from_angle, test_set_poses  # noqa: E402
from nerf_amd.kilonerf import KiloNetwork, KiloRenderer, default_kilo_cfg  # noqa: E402
from nerf_amd.weights import make_kilo_state_dict_arrays  # noqa: E402

L = _lib.lib()
L.nrf_debug_kf_stamps.argtypes = [C.c_void_p, C.c_int]
out = (C.c_ulonglong * 16)()
dev = torch.device("cuda:0")
net = KiloNetwork()
net.load_state_dict({k: torch.from_numpy(v) for k, v in make_kilo_state_dict_arrays(0, 4096, 10.0, 0.5).items()}, strict=False)
net.to(dev)
c = (torch.arange(256, device=dev, dtype=torch.float32) + 0.5) / 256 * 2 - 1
occ = ((c[:, None, None] ** 2 + c[None, :, None] ** 2 + c[None, None, :] ** 2).sqrt() < 0.75).to(torch.uint8)
r = KiloRenderer(net, default_kilo_cfg(), occ, mlp_dtype="bf16x3")
aabb, res = r.view.grid()
o, d = ops.raygen(test_set_poses(200)[0], 800, 800, focal_from_angle(800), dev)
pts, ray_idx, ts, te, offsets, counts = kilo_ops.kilo_march(o, d, occ, aabb, r.near, r.far, r.step)
packed = r.view.packed()
for _ in range(2):
    kilo_ops.kilo_forward(packed, r.view.num_models, aabb, res, pts, d, ray_idx, kind="bf16x3")
torch.cuda.synchronize()
L.nrf_debug_kf_stamps(out, 1)
N = 5
for _ in range(N):
    kilo_ops.kilo_forward(packed, r.view.num_models, aabb, res, pts, d, ray_idx, kind="bf16x3")
torch.cuda.synchronize()
L.nrf_debug_kf_stamps(out, 0)
names = ["job set-up", "(unused)", "tile top + position encoding (+ wait for the point)", "split of the encoding",
         "pts_linears 0 (24 MFMAs)", "relu + split", "pts_linears 1 (12 MFMAs)", "relu + sigma + split", "feature (12 MFMAs)",
         "split", "view layer (24 MFMAs, + wait for the rows)", "heads + store"]
tot = sum(out[:12])
tiles = (pts.shape[0] + 31) // 32
print(f"{pts.shape[0]} samples, >= {tiles} tiles; s_memtime ticks (shader clock) summed over 2048 waves, per launch")
for n, v in zip(names, out):
    print(f"  {n:52s} {v / N / 2048:11.0f} ticks per wave  {100 * v / tot:5.1f} %   {v / N / tiles:7.0f} ticks per tile")
