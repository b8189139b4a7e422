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
ap.add_argument("--train-dtype", default="fp16x2", choices=["fp32", "bf16x3", "fp16x2"], help="--only train: TrainStep's forward_dtype")
ap.add_argument("--kilo-dtype", default="fp16x2", choices=["fp32", "fp16x2", "bf16x3"], help="--only kilo: KiloRenderer's mlp_dtype")
ap.add_argument("--only", default="all", choices=["all", "mlp", "composite", "sample", "bf16", "bf16x3", "fp16x2", "kilo", "train"])
a = ap.parse_args()

dev = torch.device("cuda:0")
sd = make_state_dict_arrays(0, 30.0, 2.0)

if a.only == "kilo":        # KiloNeRF frame (BASELINE configs[3]): march + route/sort + kilo_mlp_kernel + composite
    from nerf_amd import kilo_ops
    from nerf_amd.kilonerf import KiloNetwork, KiloRenderer, default_kilo_cfg
    from nerf_amd.weights import make_kilo_state_dict_arrays
    net = KiloNetwork()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in make_kilo_state_dict_arrays(0, 4096, 10.0, 0.5).items()}, strict=False)
    net.to(dev)
    c = (torch.arange(256, device=dev, dtype=torch.float32) + 0.5) / 256 * 2 - 1
    occ = ((c[:, None, None] ** 2 + c[None, :, None] ** 2 + c[None, None, :] ** 2).sqrt() < 0.75).to(torch.uint8)
    r = KiloRenderer(net, default_kilo_cfg(), occ, mlp_dtype=a.kilo_dtype)
    o, d = ops.raygen(pose_spherical(0.0), 800, 800, focal_from_angle(800), dev)
    for _ in range(a.reps + 1):
        out = r({"xyz": o[None], "viewdirs": d[None]}, is_training=False)
    torch.cuda.synchronize()
    print("done kilo", r.last_num_samples)
    sys.exit(0)

if a.only == "train":       # NeRF training step (BASELINE configs[2]): mlp_fwd<save>, dgrad, wgrad, reduce, adam
    from nerf_amd.network import Network
    from nerf_amd.train import TrainStep
    from nerf_amd.weights import make_state_dict
    net = Network()
    net.load_state_dict(make_state_dict(0, 30.0, 2.0))
    ts = TrainStep(net.to(dev), dev, forward_dtype=a.train_dtype)
    R = 4096
    g = torch.Generator(device=dev).manual_seed(0)
    pix = torch.randint(0, 640000, (R,), device=dev, generator=g)
    o, d = ops.raygen(pose_spherical(0.0), 800, 800, focal_from_angle(800), dev, pixel_index=pix)
    rnd = lambda n: torch.rand((R, n), device=dev, generator=g)  # noqa: E731
    gt, u, u2, uc, uf = rnd(3), rnd(128), rnd(128), rnd(64), rnd(128)
    for _ in range(a.reps + 1):
        ts.step(o, d, gt, u, u2, uc, uf)
    torch.cuda.synchronize()
    print("done train", float(ts.loss))
    sys.exit(0)
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
if a.only == "bf16":        # mlp_bf16_kernel, coarse + fine launch per rep
    bc, bf = ops.pack_state_dict_bf16(sd, "model", dev), ops.pack_state_dict_bf16(sd, "model_fine", dev)
    for _ in range(a.reps + 1):
        ops.mlp_forward_rays_bf16(bc, o, d, bins)
        ops.mlp_forward_rays_bf16(bf, o, d, z_f)
    torch.cuda.synchronize()
    print("done bf16")
    sys.exit(0)
if a.only == "bf16x3":      # mlp_bf16x3_m16_kernel, coarse + fine launch per rep
    xc, xf = ops.pack_state_dict_bf16x3(sd, "model", dev), ops.pack_state_dict_bf16x3(sd, "model_fine", dev)
    for _ in range(a.reps + 1):
        ops.mlp_forward_rays_bf16x3(xc, o, d, bins)
        ops.mlp_forward_rays_bf16x3(xf, o, d, z_f)
    torch.cuda.synchronize()
    print("done bf16x3")
    sys.exit(0)
if a.only == "fp16x2":      # mlp_fp16x2_kernel, coarse + fine launch per rep, event-timed
    hc, hf = ops.pack_state_dict_fp16x2(sd, "model", dev), ops.pack_state_dict_fp16x2(sd, "model_fine", dev)
    xf = ops.pack_state_dict_bf16x3(sd, "model_fine", dev)
    for name, fn, pk in (("fp16x2", ops.mlp_forward_rays_fp16x2, hf), ("bf16x3", ops.mlp_forward_rays_bf16x3, xf)):
        fn(pk, o, d, z_f)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            fn(pk, o, d, z_f)
        e1.record()
        torch.cuda.synchronize()
        print(f"{name}: fine pass {a.rays} rays x {z_f.shape[-1]}: {e0.elapsed_time(e1) / a.reps:.3f} ms")
    ops.mlp_forward_rays_fp16x2(hc, o, d, bins)
    torch.cuda.synchronize()
    print("done fp16x2")
    sys.exit(0)
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
