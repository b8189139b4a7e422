#!/usr/bin/env python3
"""One-off timing of the full 256^3 occupancy bake: 453 M teacher-MLP evaluations."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_amd import kilo_ops, ops  # noqa: E402
from nerf_amd.weights import make_state_dict_arrays  # noqa: E402

dev = torch.device("cuda:0")
res = [int(sys.argv[1])] * 3 if len(sys.argv) > 1 else [256] * 3
grids = {}
for kind in ("fp32", "bf16x3", "fp16x2"):
    pack = {"fp32": ops.pack_state_dict, "bf16x3": ops.pack_state_dict_bf16x3, "fp16x2": ops.pack_state_dict_fp16x2}[kind]
    packed = pack(make_state_dict_arrays(0, 30.0, 2.0), "model_fine", dev)
    kilo_ops.occ_bake(packed, [-1, -1, -1, 1, 1, 1], [32, 32, 32], 2.25, kind=kind)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    occ = grids[kind] = kilo_ops.occ_bake(packed, [-1, -1, -1, 1, 1, 1], res, 2.25, kind=kind)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n = res[0] * res[1] * res[2] * 27
    print(f"bake {res} ({kind} teacher): {dt:.3f} s, {n / dt / 1e6:.1f} M evals/s, {n * 1186816 / dt / 1e12:.1f} TFLOP/s, occupied {float(occ.float().mean()):.3f}")
for kind in ("bf16x3", "fp16x2"):
    print(f"cells that differ between the fp32 and the {kind} grid: {int((grids['fp32'] != grids[kind]).sum())} of {grids['fp32'].numel()}")
