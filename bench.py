#!/usr/bin/env python3
"""Headline benchmark: rays/sec and ms per 800x800 frame (64 coarse + 128 fine samples/ray).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

A *step* is one full 800x800 test-view render (BASELINE.json configs[1]: NeRF coarse + fine,
fp32 MLP, random-init weights, synthetic orbit camera): ray generation -> coarse MLP ->
compositing -> importance resampling + merge -> fine MLP -> compositing, everything on the
GPU, inputs (pose, weights) resident before the timed region.  With N ranks every rank
renders its own view per step (views of the 200-view orbit are sharded round-robin, no
data-path collective): weak scaling, value = all rays of all ranks / max-over-ranks time.

The same JSON line carries
  roofline      the dominant kernel (mlp_fwd_kernel, MFMA-bound): algorithmic FLOPs per launch
                / its mean launch time, measured with HIP events on the launch stream inside
                the timed region; peak = fp32 MFMA 157.3 TFLOP/s (MI355X_MICROARCH.md)
  cpu_baseline  oracle/ (the CPU restatement of the reference) timed on this host's cores
                on a bounded sample of the same rays (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H = W = 800
N_COARSE, N_IMPORTANCE = 64, 128
MACS_PER_POINT = 593_408
FLOP_PER_POINT = 2 * MACS_PER_POINT
FP32_MFMA_PEAK_TFLOPS = 157.3                  # MI355X_MICROARCH.md, chip-level parameters
BF16_MFMA_PEAK_TFLOPS = 2500.0                 # MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--mode", default="render", choices=["render", "train", "kilo", "kilo_train", "testset", "frame"],
                    help="render: BASELINE configs[1] (default, the headline metric); train: configs[2], "
                         "4096-ray fwd+bwd+Adam step per rank with one RCCL gradient all-reduce; kilo: configs[3], KiloNeRF "
                         "800x800 frame with the sphere occupancy grid; kilo_train: KiloNeRF distillation step (SURVEY 8f rank 3); testset: configs[4], the 200-view test set "
                         "dealt round-robin over the ranks (strong scaling; a step = one pass over all views, a warm-up step = one "
                         "view per rank; use with --dtype bf16); frame: one view per step, its rays in contiguous slices over the ranks + one "
                         "all_gather (strong scaling, the latency form of the metric)")
    ap.add_argument("--views", type=int, default=200, help="testset mode: number of views in the set")
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16"],
                    help="render mode only: fp32 = BASELINE configs[1] (the headline metric); bf16 = the bf16 MLP of "
                         "configs[4], reported under its own metric name")
    ap.add_argument("--train-rays", type=int, default=4096)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def recorded_traffic():
    """HBM bytes per mlp_fwd_kernel launch from the committed rocprofv3 PMC passes of this same command
    (profiles/r01_mlp_traffic_pmc.json: FETCH_SIZE x2 + WRITE_SIZE, MI355X_MICROARCH.md corrections).  Counters cannot
    be read from inside the process, so this is the recorded figure, or None when the file is absent."""
    path = os.path.join(ROOT, "profiles", "r01_mlp_traffic_pmc.json")
    try:
        with open(path) as f:
            return float(json.load(f)["traffic_bytes_per_launch_mean"])
    except (OSError, KeyError, ValueError):
        return None


def recorded_composite_traffic():
    """Same for the fine compositing launch (profiles/r01_composite_traffic_pmc.json)."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_composite_traffic_pmc.json")) as f:
            return float(json.load(f)["traffic_bytes_fine_composite"])
    except (OSError, KeyError, ValueError):
        return None


def cpu_baseline(seconds: float):
    """Oracle render of 1024-ray slices of the bench view until ~`seconds` of CPU time."""
    from oracle import nerf_oracle as O
    from nerf_amd.weights import make_state_dict
    from nerf_amd.camera import pose_spherical, LEGO_CAMERA_ANGLE_X
    sd = make_state_dict(0, 30.0, 2.0)
    o, d = O.raygen(pose_spherical(0.0), H, W, LEGO_CAMERA_ANGLE_X)
    rng = np.random.Generator(np.random.PCG64(0))
    chunk, done, t0 = 1024, 0, time.perf_counter()
    with torch.no_grad():
        while True:
            r0 = 400 * W + done          # central rows: rays that cross the scene box
            u = rng.random((chunk, N_IMPORTANCE), dtype=np.float32)
            u2 = rng.random((chunk, N_IMPORTANCE), dtype=np.float32)
            O.render(sd, o[r0:r0 + chunk], d[r0:r0 + chunk], u=u, u2=u2)
            done += chunk
            el = time.perf_counter() - t0
            if el >= seconds or done >= 64 * chunk:
                break
    return {"value": done / el, "unit": "rays/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{done} rays of the same 800x800 view (rows 400+), coarse+fine, oracle/nerf_oracle.py "
                      f"on torch-CPU fp32, {el:.1f} s; host has {os.cpu_count()} logical CPUs",
            "ms_per_frame_extrapolated": 1e3 * H * W * el / done}


def train_cpu_baseline(seconds: float, rays_per_step: int = 256):
    """The training step of trainer.py:68-75 on the CPU: oracle forward (perturbed samples), torch autograd,
    clip_grad_value_(40), torch.optim.Adam(5e-4), on batches of `rays_per_step` random pixels until ~`seconds`."""
    from oracle import nerf_oracle as O
    from nerf_amd.weights import make_state_dict
    from nerf_amd.camera import pose_spherical, LEGO_CAMERA_ANGLE_X
    sd = {k: v.clone().requires_grad_(True) for k, v in make_state_dict(0, 30.0, 2.0).items()}
    opt = torch.optim.Adam(list(sd.values()), lr=5e-4)
    o, d = O.raygen(pose_spherical(0.0), H, W, LEGO_CAMERA_ANGLE_X)
    o, d = o.reshape(-1, 3), d.reshape(-1, 3)
    rng = np.random.Generator(np.random.PCG64(1))
    done, steps, t0 = 0, 0, time.perf_counter()
    while True:
        pix = rng.integers(0, H * W, rays_per_step)
        u, u2, uf = (rng.random((rays_per_step, N_IMPORTANCE), dtype=np.float32) for _ in range(3))
        uc = rng.random((rays_per_step, N_COARSE), dtype=np.float32)
        gt = torch.from_numpy(rng.random((rays_per_step, 3), dtype=np.float32))
        out = O.render(sd, o[pix], d[pix], u=u, u2=u2, u_coarse=uc, u_fine=uf)
        opt.zero_grad(set_to_none=True)
        O.loss(out, gt).backward()
        torch.nn.utils.clip_grad_value_(list(sd.values()), 40.0)
        opt.step()
        done += rays_per_step
        steps += 1
        el = time.perf_counter() - t0
        if el >= seconds or steps >= 256:
            break
    return {"value": done / el, "unit": "rays/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{steps} steps of {rays_per_step} random pixels (fwd + autograd + clip + Adam), oracle/nerf_oracle.py on "
                      f"torch-CPU fp32, {el:.1f} s; host has {os.cpu_count()} logical CPUs"}


def train_bench(args, net, dev, rank, world, dist):
    """BASELINE configs[2]: per rank 4096 random pixels of one view, perturb=1, fwd + 2xMSE + bwd +
    gradient all-reduce (mean) + clip 40 + Adam(5e-4).  Weak scaling: every rank has its own batch."""
    from nerf_amd import ops
    from nerf_amd.camera import test_set_poses, focal_from_angle
    from nerf_amd.train import TrainStep
    ts = TrainStep(net, dev)
    poses = test_set_poses(200)
    focal = focal_from_angle(W)
    R = args.train_rays
    gen = torch.Generator(device=dev)
    gen.manual_seed(4321 + rank)

    def step(k):
        pix = torch.randint(0, H * W, (R,), device=dev, generator=gen)
        rays_o, rays_d = ops.raygen(poses[(k * world + rank) % len(poses)], H, W, focal, dev, pixel_index=pix)
        gt = torch.rand((R, 3), device=dev, generator=gen)
        u_c = torch.rand((R, N_COARSE), device=dev, generator=gen)
        u = torch.rand((R, N_IMPORTANCE), device=dev, generator=gen)
        u2 = torch.rand((R, N_IMPORTANCE), device=dev, generator=gen)
        u_f = torch.rand((R, N_IMPORTANCE), device=dev, generator=gen)
        ts.step(rays_o, rays_d, gt, u, u2, u_c, u_f)

    def sync():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for k in range(args.warmup):
        step(k)
    sync()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(args.warmup + k)
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss = float(ts.loss)
    assert loss == loss
    if rank == 0:
        ms = 1e3 * elapsed / args.steps
        flop = 3.0 * R * (N_COARSE + N_COARSE + N_IMPORTANCE) * FLOP_PER_POINT   # fwd + dgrad + wgrad, SURVEY 8d
        line = {
            "metric": "training rays/sec (fwd+bwd+Adam step, 64 coarse + 128 fine samples/ray)",
            "value": world * R * args.steps / elapsed, "unit": "rays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic (random pixels of orbit views, random gt, random-init weights)",
            "config": {"workload": "BASELINE configs[2]: NeRF training step, %d-ray batch per rank, fwd+bwd with Adam, "
                                   "data-parallel RCCL all-reduce" % R, "rays_per_rank": R, "perturb": 1},
            "roofline": {"kernel": "whole step (mlp_fwd<save> + mlp_dgrad + wgrad GEMMs)", "bound": "mfma",
                         "achieved": flop / (ms * 1e-3) / 1e12, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": flop / (ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, "traffic": None,
                         "note": "algorithmic FLOPs of the step / wall time of the step, not one kernel"},
            "final_loss": loss}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = train_cpu_baseline(args.cpu_seconds)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def kilo_cpu_baseline(seconds: float, sd, frame_samples: float):
    """oracle/kilo_oracle.py (the reference's batched-bmm network over this repository's marching / compositing
    rule) on 128-ray slices of the bench view's central rows until ~`seconds` of CPU time."""
    from oracle import kilo_oracle as K
    from oracle import nerf_oracle as O
    from nerf_amd.camera import pose_spherical, LEGO_CAMERA_ANGLE_X
    c = (np.arange(256, dtype=np.float32) + 0.5) / 256 * 2 - 1
    occ = (np.sqrt(c[:, None, None] ** 2 + c[None, :, None] ** 2 + c[None, None, :] ** 2) < 0.75).astype(np.uint8)
    o, d = O.raygen(pose_spherical(0.0), H, W, LEGO_CAMERA_ANGLE_X)
    o, d = o.reshape(-1, 3), d.reshape(-1, 3)
    sdt = {k: torch.from_numpy(v) for k, v in sd.items()}
    chunk, done, samples, t0 = 128, 0, 0, time.perf_counter()
    with torch.no_grad():
        while True:
            r0 = (250 + (done // chunk) % 300) * W + 336       # 128 central pixels of rows 250 .. 549
            out = K.render(sdt, o[r0:r0 + chunk], d[r0:r0 + chunk], occ, [-1.0] * 3, [1.0] * 3, [16, 16, 16], 2.0, 6.0,
                           0.005, 0.01, True)
            samples += int(out["n_samples"])
            done += chunk
            el = time.perf_counter() - t0
            if el >= seconds or done >= 1024 * chunk:
                break
    return {"value": done / el, "unit": "rays/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{done} rays through the sphere (central 128 pixels of rows 250+ of the same view), "
                      f"oracle/kilo_oracle.py on torch-CPU fp32, {el:.1f} s; host has {os.cpu_count()} logical CPUs; "
                      f"{samples / done:.0f} samples per ray here (these rays all cross the occupied sphere), "
                      "43 per ray over the whole frame",
            "samples_per_s": samples / el,
            "ms_per_frame_extrapolated": 1e3 * frame_samples / max(samples / el, 1e-9),
            "extrapolation": "by evaluated samples (the frame's samples / the sample rate measured here)"}


def kilo_bench(args, dev, rank, world, dist):
    """BASELINE configs[3]: KiloNeRF, 4096 cells (W=32, D=2), occupancy = sphere |x| < 0.75 at 256^3, step 0.005,
    ERT 0.01, near 2 / far 6, 800x800 view per rank per step (weak scaling, no data-path collective)."""
    from nerf_amd import kilo_ops, ops
    from nerf_amd.camera import test_set_poses, focal_from_angle
    from nerf_amd.kilonerf import KiloNetwork, KiloRenderer, default_kilo_cfg
    from nerf_amd.weights import make_kilo_state_dict_arrays
    net = KiloNetwork()
    sd = make_kilo_state_dict_arrays(0, 4096, 10.0, 0.5)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    net.to(dev)
    c = (torch.arange(256, device=dev, dtype=torch.float32) + 0.5) / 256 * 2 - 1
    occ = ((c[:, None, None] ** 2 + c[None, :, None] ** 2 + c[None, None, :] ** 2).sqrt() < 0.75).to(torch.uint8)
    r = KiloRenderer(net, default_kilo_cfg(), occ)
    poses = test_set_poses(200)
    focal = focal_from_angle(W)
    R = H * W
    aabb, res = net.grid()
    packed = net.packed()
    ev = {k: [] for k in ("march", "mlp", "composite")}

    def E():
        e = torch.cuda.Event(enable_timing=True)
        e.record(torch.cuda.current_stream(dev))
        return e

    samples = []

    def step(k, timed):
        rays_o, rays_d = ops.raygen(poses[(k * world + rank) % len(poses)], H, W, focal, dev)
        e0 = E()
        pts, ray_idx, ts, te, offsets, counts = kilo_ops.kilo_march(rays_o, rays_d, r.occ_grid, aabb, r.near, r.far, r.step)
        e1 = E()
        raw = kilo_ops.kilo_forward(packed, net.num_models, aabb, res, pts, rays_d, ray_idx)
        e2 = E()
        out = kilo_ops.kilo_composite(raw, ts, te, offsets, counts, r.ert, True)
        e3 = E()
        if timed:
            ev["march"].append((e0, e1)), ev["mlp"].append((e1, e2)), ev["composite"].append((e2, e3))
            samples.append(int(pts.shape[0]))
        return out

    def sync():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for k in range(args.warmup):
        out = step(k, False)
    sync()
    t0 = time.perf_counter()
    for k in range(args.steps):
        out = step(args.warmup + k, True)
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert torch.isfinite(out[0]).all()
    if rank == 0:
        ms = {k: float(np.mean([a.elapsed_time(b) for a, b in v])) for k, v in ev.items()}
        n = float(np.mean(samples))
        flop = 12160.0 * n                                    # 6080 MACs per evaluated sample
        line = {
            "metric": "rays/sec (KiloNeRF 800x800 frame, occupancy-grid marching + early termination)",
            "value": world * R * args.steps / elapsed, "unit": "rays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic (orbit camera, random-init cells, sigma head x10 + 0.5, sphere occupancy)",
            "config": {"workload": "BASELINE configs[3]: KiloNeRF 16^3 cells W=32 D=2, sphere |x|<0.75 occupancy at 256^3, "
                                   "step 0.005, ERT 0.01, 800x800 view per rank per step", "rays_per_frame": R,
                       "samples_per_frame": n, "samples_per_hit_ray": n / max(1.0, float((out[2] > 0).sum()))},
            "roofline": {"kernel": "kilo_forward (route + counting sort + kilo_mlp_kernel)", "bound": "mfma",
                         "achieved": flop / (ms["mlp"] * 1e-3) / 1e12, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": flop / (ms["mlp"] * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, "traffic": None,
                         "samples_per_s": n / (ms["mlp"] * 1e-3)},
            "stage_ms": ms}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = kilo_cpu_baseline(args.cpu_seconds, sd, n)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def kilo_train_cpu_baseline(seconds: float, sd, rays_per_step: int = 256):
    """The KiloNeRF distillation step on the CPU: oracle march + the reference's batched network + the packed
    compositing rule in torch, autograd, clip 40, torch.optim.Adam over all 25.4 M parameters (dense, as in the
    reference), batches of `rays_per_step` random pixels until ~`seconds`."""
    from oracle import kilo_oracle as K
    from oracle import nerf_oracle as O
    from nerf_amd.camera import pose_spherical, LEGO_CAMERA_ANGLE_X
    c = (np.arange(256, dtype=np.float32) + 0.5) / 256 * 2 - 1
    occ = (np.sqrt(c[:, None, None] ** 2 + c[None, :, None] ** 2 + c[None, None, :] ** 2) < 0.75).astype(np.uint8)
    o, d = O.raygen(pose_spherical(0.0), H, W, LEGO_CAMERA_ANGLE_X)
    o, d = o.reshape(-1, 3), d.reshape(-1, 3)
    sdt = {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in sd.items() if v.dtype == np.float32 and v.ndim > 1}
    full = {k: (sdt[k] if k in sdt else torch.from_numpy(v)) for k, v in sd.items()}
    opt = torch.optim.Adam(list(sdt.values()), lr=5e-4)
    rng = np.random.Generator(np.random.PCG64(2))
    done, steps, samples, t0 = 0, 0, 0, time.perf_counter()
    while True:
        pix = rng.integers(0, H * W, rays_per_step)
        t_rgb = torch.from_numpy(rng.random((rays_per_step, 3), dtype=np.float32))
        t_alpha = torch.from_numpy(rng.random((rays_per_step, 1), dtype=np.float32))
        pts, ri, ts, te, off, cnt = K.march(o[pix], d[pix], occ, [-1.0] * 3, [1.0] * 3, 2.0, 6.0, 0.005)
        raw = K.network_forward(full, torch.from_numpy(pts), torch.from_numpy(d[pix][ri]), [-1.0] * 3, [1.0] * 3, [16, 16, 16])
        rgb, _depth, acc = K.composite_packed_torch(raw, ts, te, off, cnt, 0.01, True)
        opt.zero_grad(set_to_none=True)
        K.train_loss(rgb, acc, t_rgb, t_alpha, "distillation", 1e-6, full).backward()
        torch.nn.utils.clip_grad_value_(list(sdt.values()), 40.0)
        opt.step()
        done += rays_per_step
        steps += 1
        samples += len(pts)
        el = time.perf_counter() - t0
        if el >= seconds or steps >= 256:
            break
    return {"value": done / el, "unit": "rays/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{steps} steps of {rays_per_step} random pixels ({samples / max(steps, 1):.0f} samples per step; march + "
                      f"batched network + packed compositing + autograd + clip + dense Adam), oracle/kilo_oracle.py on torch-CPU, "
                      f"{el:.1f} s; host has {os.cpu_count()} logical CPUs"}


def kilo_train_bench(args, dev, rank, world, dist):
    """KiloNeRF distillation step (configs/kilonerf/lego.yaml: batch_size 4 x N_rays 1024 rays per rank):
    march + tiny MLPs + packed compositing, mse(rgb) + mse(alpha) + l2_reg colour regulariser, backward, gradient
    all-reduce (mean), clip 40, Adam over all 25.4 M parameters.  Weak scaling: every rank has its own batch."""
    from nerf_amd import ops
    from nerf_amd.camera import test_set_poses, focal_from_angle
    from nerf_amd.kilo_train import KiloTrainStep
    from nerf_amd.kilonerf import KiloNetwork, KiloRenderer, default_kilo_cfg
    from nerf_amd.weights import make_kilo_state_dict_arrays
    net = KiloNetwork()
    sd = make_kilo_state_dict_arrays(0, 4096, 10.0, 0.5)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    net.to(dev)
    c = (torch.arange(256, device=dev, dtype=torch.float32) + 0.5) / 256 * 2 - 1
    occ = ((c[:, None, None] ** 2 + c[None, :, None] ** 2 + c[None, None, :] ** 2).sqrt() < 0.75).to(torch.uint8)
    r = KiloRenderer(net, default_kilo_cfg(), occ)
    ts = KiloTrainStep(r, dev, lr=5e-4, clip_value=40.0, l2_reg=1e-6)
    poses = test_set_poses(200)
    focal = focal_from_angle(W)
    R = args.train_rays
    gen = torch.Generator(device=dev)
    gen.manual_seed(977 + rank)
    samples = []

    def step(k):
        pix = torch.randint(0, H * W, (R,), device=dev, generator=gen)
        rays_o, rays_d = ops.raygen(poses[(k * world + rank) % len(poses)], H, W, focal, dev, pixel_index=pix)
        t_rgb = torch.rand((R, 3), device=dev, generator=gen)
        t_alpha = torch.rand((R, 1), device=dev, generator=gen)
        ts.step(rays_o, rays_d, t_rgb, t_alpha, stage="distillation")
        samples.append(ts.num_samples)

    def sync():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for k in range(args.warmup):
        step(k)
    sync()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(args.warmup + k)
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss = float(ts.loss)
    assert loss == loss
    if rank == 0:
        ms = 1e3 * elapsed / args.steps
        n = float(np.mean(samples[args.warmup:]))
        adam_bytes = 28.0 * ts.flat.numel()                  # 16 B read + 12 B write per parameter
        line = {
            "metric": "training rays/sec (KiloNeRF distillation step: fwd+bwd+Adam)",
            "value": world * R * args.steps / elapsed, "unit": "rays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic (random pixels of orbit views, random targets, random-init cells, sphere occupancy)",
            "config": {"workload": "KiloNeRF distillation step (SURVEY 8f rank 3), %d-ray batch per rank, 4096 cells, "
                                   "dense Adam over 25.4 M parameters, data-parallel RCCL all-reduce of the sampled cells' "
                                   "gradient records" % R,
                       "rays_per_rank": R, "samples_per_step": n,
                       "cells_exchanged": (int((ts.touched != 0).sum()) if ts.touched is not None else None)},
            "roofline": {"kernel": "clip_adam_kernel over the flat parameter buffer (dominant: the step is optimiser-bound)",
                         "bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                         "note": "algorithmic bytes of the Adam pass = %.0f MB per step; see DESIGN.md section 7 for the "
                                 "per-kernel times (rocprofv3)" % (adam_bytes / 1e6)},
            "final_loss": loss}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = kilo_train_cpu_baseline(args.cpu_seconds, sd)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def testset_bench(args, net, dev, rank, world, dist):
    """BASELINE configs[4]: the 200 test views (theta = -180 ... 178.2 deg), (view) items dealt round-robin over the
    ranks (nerf_amd.parallel.view_schedule), no data-path collective.  Strong scaling: the set is fixed."""
    from nerf_amd import ops
    from nerf_amd.camera import test_set_poses, focal_from_angle
    from nerf_amd.config import default_cfg
    from nerf_amd.parallel import view_schedule
    from nerf_amd.renderer import Renderer
    renderer = Renderer(net, default_cfg(perturb=0), mlp_dtype=args.dtype)
    poses = test_set_poses(args.views)
    focal = focal_from_angle(W)
    R = H * W
    mine = [v for v, _ in view_schedule(len(poses), rank, world)]
    gen = torch.Generator(device=dev)
    gen.manual_seed(99 + rank)

    def render(view):
        rays_o, rays_d = ops.raygen(poses[view], H, W, focal, dev)
        u = torch.rand((R, N_IMPORTANCE), device=dev, generator=gen)
        u2 = torch.rand((R, N_IMPORTANCE), device=dev, generator=gen)
        return renderer.render_rays(rays_o, rays_d, u, u2)

    def sync():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    with torch.no_grad():
        for k in range(args.warmup):
            out = render(mine[k % len(mine)])
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            for view in mine:
                out = render(view)
        sync()
        elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert torch.isfinite(out["fine_rgb_map"]).all()
    if rank == 0:
        bf16 = args.dtype == "bf16"
        flop = float(len(poses)) * R * (N_COARSE + N_COARSE + N_IMPORTANCE) * FLOP_PER_POINT
        peak = (BF16_MFMA_PEAK_TFLOPS if bf16 else FP32_MFMA_PEAK_TFLOPS) * world
        achieved = flop * args.steps / elapsed / 1e12
        line = {
            "metric": "rays/sec (%d-view test set, 800x800, 64 coarse + 128 fine samples/ray%s)" % (len(poses), ", bf16 MLP" if bf16 else ""),
            "value": len(poses) * R * args.steps / elapsed, "unit": "rays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "ms_per_frame": 1e3 * elapsed / args.steps / len(poses),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "bf16" if bf16 else "f32",
            "data": "synthetic (orbit camera, random-init weights, alpha_linear x30 + 2)",
            "config": {"workload": "BASELINE configs[4]: %d-view test-set render, views dealt round-robin over %d rank(s), %s MLP"
                                   % (len(poses), world, args.dtype), "views": len(poses), "rays_per_frame": R,
                       "views_per_rank": len(mine)},
            "roofline": {"kernel": "whole job (MLP FLOPs of all views / wall time / (peak x ranks))", "bound": "mfma",
                         "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "traffic": None}}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def frame_bench(args, net, dev, rank, world, dist):
    """Latency form of BASELINE configs[1] / the metric's "ms per 800x800 frame at 1/2/4/8 MI355X": ONE view per step,
    its 640,000 rays split in contiguous slices over the ranks (Renderer.render_view_sharded), one all_gather of the
    per-ray outputs so every rank ends the step holding the whole frame.  Strong scaling: the frame is fixed."""
    from nerf_amd.camera import test_set_poses, focal_from_angle
    from nerf_amd.config import default_cfg
    from nerf_amd.renderer import Renderer
    renderer = Renderer(net, default_cfg(perturb=0), mlp_dtype=args.dtype)
    poses = test_set_poses(200)
    focal = focal_from_angle(W)
    R = H * W
    gen = torch.Generator(device=dev)
    gen.manual_seed(4321 + rank)

    def step(k):
        return renderer.render_view_sharded(poses[k % len(poses)], H, W, focal, rank, world, dev, gather=True, generator=gen)[0]

    def sync():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for k in range(args.warmup):
        out = step(k)
    sync()
    t0 = time.perf_counter()
    for k in range(args.steps):
        out = step(args.warmup + k)
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert out["fine_rgb_map"].shape == (R, 3) and torch.isfinite(out["fine_rgb_map"]).all()
    if rank == 0:
        bf16 = args.dtype == "bf16"
        flop = float(R) * (N_COARSE + N_COARSE + N_IMPORTANCE) * FLOP_PER_POINT
        peak = (BF16_MFMA_PEAK_TFLOPS if bf16 else FP32_MFMA_PEAK_TFLOPS) * world
        achieved = flop * args.steps / elapsed / 1e12
        line = {
            "metric": "rays/sec (one 800x800 frame ray-sharded over the GPUs, 64 coarse + 128 fine samples/ray%s)" % (", bf16 MLP" if bf16 else ""),
            "value": R * args.steps / elapsed, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "ms_per_frame": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "bf16" if bf16 else "f32",
            "data": "synthetic (orbit camera, random-init weights, alpha_linear x30 + 2)",
            "config": {"workload": "BASELINE configs[1] latency form: one 800x800 view per step, rays in contiguous slices over "
                                   "%d rank(s), outputs all-gathered on every rank, %s MLP" % (world, args.dtype),
                       "rays_per_frame": R, "rays_per_rank": -(-R // world), "gather_bytes_per_ray": 36},
            "roofline": {"kernel": "whole step (MLP FLOPs of the frame / wall time / (peak x ranks))", "bound": "mfma",
                         "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "traffic": None}}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    # One rank per GPU.  NRF_BENCH_BACKEND=gloo is a rehearsal switch for boxes with fewer GPUs than ranks (ranks then
    # share devices and the timing collectives run over gloo); the measured configuration is always nccl = RCCL.
    backend = os.environ.get("NRF_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group(backend="nccl", init_method="env://", device_id=dev)
        else:
            dist.init_process_group(backend=backend, init_method="env://")

    from nerf_amd import ops
    from nerf_amd.camera import test_set_poses, focal_from_angle
    from nerf_amd.config import default_cfg
    from nerf_amd.network import Network
    from nerf_amd.renderer import Renderer
    from nerf_amd.weights import make_state_dict

    if args.mode == "kilo":
        return kilo_bench(args, dev, rank, world, dist)
    if args.mode == "kilo_train":
        return kilo_train_bench(args, dev, rank, world, dist)
    net = Network()
    net.load_state_dict(make_state_dict(0, 30.0, 2.0))   # random init, density rescaled so rays terminate
    net.to(dev)
    if args.mode == "train":
        return train_bench(args, net, dev, rank, world, dist)
    if args.mode == "testset":
        return testset_bench(args, net, dev, rank, world, dist)
    if args.mode == "frame":
        return frame_bench(args, net, dev, rank, world, dist)
    renderer = Renderer(net, default_cfg(perturb=0), mlp_dtype=args.dtype)
    bf16 = args.dtype == "bf16"
    peak = BF16_MFMA_PEAK_TFLOPS if bf16 else FP32_MFMA_PEAK_TFLOPS
    poses = test_set_poses(200)
    focal = focal_from_angle(W)
    R = H * W
    events = {}
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)

    def step(k, ev):
        c2w = poses[(k * world + rank) % len(poses)]
        rays_o, rays_d = ops.raygen(c2w, H, W, focal, dev)
        u = torch.rand((R, N_IMPORTANCE), device=dev, generator=gen)
        u2 = torch.rand((R, N_IMPORTANCE), device=dev, generator=gen)
        return renderer.render_rays_staged(rays_o, rays_d, u, u2, events=ev)

    def sync():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    with torch.no_grad():
        for k in range(args.warmup):
            out = step(k, None)
        sync()
        t0 = time.perf_counter()
        for k in range(args.steps):
            out = step(args.warmup + k, events)
        sync()
        elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert torch.isfinite(out["fine_rgb_map"]).all()

    def mean_ms(name):
        return float(np.mean([a.elapsed_time(b) for a, b in events[name]]))

    ms_mlp_c, ms_mlp_f = mean_ms("mlp_coarse"), mean_ms("mlp_fine")
    ms_cmp_c, ms_cmp_f, ms_smp = mean_ms("composite_coarse"), mean_ms("composite_fine"), mean_ms("sample_fine")
    if rank == 0:
        ms_step = 1e3 * elapsed / args.steps
        flop_launch = 0.5 * R * (N_COARSE + N_COARSE + N_IMPORTANCE) * FLOP_PER_POINT   # mean of the 2 launches
        ms_launch = 0.5 * (ms_mlp_c + ms_mlp_f)
        achieved = flop_launch / (ms_launch * 1e-3) / 1e12
        cmp_bytes = R * ((N_COARSE + N_IMPORTANCE) * 20 + 20)
        line = {
            "metric": "rays/sec (full 800x800 frame, 64 coarse + 128 fine samples/ray)" + (", bf16 MLP" if bf16 else ""),
            "value": world * R * args.steps / elapsed,
            "unit": "rays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step,
            "ms_per_frame": ms_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16" if bf16 else "f32",
            "data": "synthetic (orbit camera, random-init weights, alpha_linear x30 + 2)",
            "config": {"workload": ("BASELINE configs[4] per-view form: full 800x800 test-view render, NeRF coarse+fine, bf16 MLP "
                                    "(bf16 weights + activations, fp32 accumulate), " if bf16 else
                                    "BASELINE configs[1]: full 800x800 lego test-view render, NeRF coarse+fine, fp32, ") +
                                   "one view per rank per step", "rays_per_frame": R, "n_coarse": N_COARSE,
                       "n_importance": N_IMPORTANCE, "sharding": f"views round-robin over {world} rank(s)"},
            "roofline": {"kernel": "mlp_bf16_kernel (fused points + PE + 8x256 MLP, bf16 MFMA)" if bf16 else
                                   "mlp_fwd_kernel<false> (fused points + PE + 8x256 MLP, fp32 MFMA)",
                         "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": None if bf16 else recorded_traffic(),
                         "traffic_note": "HBM bytes per launch (mean of coarse + fine) from profiles/r01_mlp_traffic_pmc.json; "
                                         "algorithmic bytes per launch %.4g" % (0.5 * R * ((N_COARSE + N_COARSE + N_IMPORTANCE) * 16
                                                                                        + 48 + (N_COARSE + N_IMPORTANCE) * 4)),
                         "flop_per_launch": flop_launch, "ms_per_launch": ms_launch,
                         "ms_coarse_launch": ms_mlp_c, "ms_fine_launch": ms_mlp_f},
            "roofline_composite": {"kernel": "composite_kernel<3> (fine pass)", "bound": "hbm",
                                   "achieved": cmp_bytes / (ms_cmp_f * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": cmp_bytes / (ms_cmp_f * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                   "traffic": recorded_composite_traffic(), "ms_per_launch": ms_cmp_f,
                                   "bytes_per_launch": cmp_bytes},
            "stage_ms": {"mlp_coarse": ms_mlp_c, "composite_coarse": ms_cmp_c, "sample_fine": ms_smp,
                         "mlp_fine": ms_mlp_f, "composite_fine": ms_cmp_f},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
