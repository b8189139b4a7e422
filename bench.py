#!/usr/bin/env python3
"""Headline benchmark: rays/sec and ms per 800x800 frame (64 coarse + 128 fine samples/ray).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Either form runs N ranks, one per GPU, on RCCL (backend "nccl"): started plainly with N > 1 this file spawns the
ranks itself before any GPU call (launch_ranks); under torch.distributed.run it takes the world from the environment
and insists that it equals --gpus.  `n_gpus` in the line is the process group's own world size.

A *step* is one full 800x800 test-view render (BASELINE.json configs[1]: NeRF coarse + fine,
fp32 MLP, random-init weights, synthetic orbit camera): ray generation -> coarse MLP ->
compositing -> importance resampling + merge -> fine MLP -> compositing, everything on the
GPU, inputs (pose, weights) resident before the timed region.  With N > 1 ranks the view's rays are
split in contiguous slices over the ranks and one all_gather assembles the frame on every rank:
value = rays of the frame / max-over-ranks time per frame (strong scaling, the metric's "ms per
800x800 frame at 1/2/4/8 MI355X"); the one-view-per-rank rate is carried as `weak_scaling`.

The MLP of the default run is mlp_bf16x3_m16_kernel: the reference's fp32 arithmetic (BASELINE configs[1] "fp32",
src/models/nerf/network.py:190) -- fp32 in, fp32 out, fp32 accumulate, every fp32 operand carried EXACTLY as three bf16
terms (3 x 8 = 24 significand bits) and each product as six bf16 MFMAs -- and `dtype` names it.  `--dtype fp32` runs the
fp32 MFMA kernel (also 24-bit operands); `--dtype fp16x2` is the narrower form (two fp16 terms = 22 significand bits,
three MFMAs per product: less than the reference's arithmetic, so never the headline).  The default line carries the
other two frames as `also.fp32_frame` and `also.fp16x2_frame`, and the training step / KiloNeRF frame in the 24-bit
dtype first (`also.train_step`, `also.kilo_frame`: bf16x3 both; `also.kilo_frame_fp32` is the fp32 MFMA kernel's frame)
with their fp16x2 forms beside them (`also.train_step_fp16x2`, `also.kilo_frame_fp16x2`).

The same JSON line carries
  roofline      the dominant kernel (MFMA-bound): algorithmic FLOPs of the fp32 problem per launch
                / its mean launch time, measured with HIP events on the launch stream inside
                the timed region; peak = fp16 MFMA 2500 / 3 TFLOP/s for fp16x2 (three MFMAs per product), bf16 MFMA
                2500 / 6 for bf16x3 (six), fp32 MFMA 157.3 TFLOP/s for --dtype fp32 (MI355X_MICROARCH.md)
  cpu_baseline  oracle/ (the CPU restatement of the reference) timed on this host's cores
                on a bounded sample of the same rays (rank 0, N=1 only)
  also          (N=1) the bf16x3 and fp32 MFMA frames, BASELINE configs[4] per-view form (bf16 MLP), configs[2] (training step)
                and configs[3] (KiloNeRF frame), measured in the same process after the headline.
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
DEFAULT_DTYPE = "bf16x3"
# MFMA ceiling on the FLOPs of the fp32 problem: the fp16 / bf16 matrix peak divided by the MFMAs one fp32 product takes
PEAK_TFLOPS = {"fp32": FP32_MFMA_PEAK_TFLOPS, "bf16": BF16_MFMA_PEAK_TFLOPS, "bf16x3": BF16_MFMA_PEAK_TFLOPS / 6,
               "fp16x2": BF16_MFMA_PEAK_TFLOPS / 3}   # MI355X_MICROARCH.md: fp16 and bf16 dense MFMA peaks are equal
DTYPE_LABEL = {"fp32": "f32", "bf16": "bf16",
               "bf16x3": "bf16x3 (f32 operands as 3 bf16 terms, f32 accumulate; f32 elsewhere)",
               "fp16x2": "fp16x2 (f32 operands as 2 fp16 terms = 22 significand bits, f32 accumulate; f32 elsewhere)"}
MLP_KERNEL = {"fp32": "mlp_fwd_kernel<false> (fused points + PE + 8x256 MLP, fp32 MFMA)",
              "bf16": "mlp_bf16_kernel (fused points + PE + 8x256 MLP, bf16 MFMA 16x16x32)",
              "bf16x3": "mlp_bf16x3_m16_kernel (fused points + PE + 8x256 MLP, 6 bf16 MFMAs 16x16x32 per fp32 product; peak = bf16 peak / 6)",
              "fp16x2": "mlp_fp16x2_kernel (fused points + PE + 8x256 MLP, 3 fp16 MFMAs per fp32 product; peak = fp16 peak / 3)"}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1,
                    help="number of ranks = GPUs.  Started plainly (no RANK / WORLD_SIZE in the environment) with N > 1, "
                         "bench.py spawns the N ranks itself before anything touches a GPU; under torch.distributed.run it "
                         "must equal WORLD_SIZE")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--mode", default="render", choices=["render", "train", "kilo", "kilo_train", "testset", "frame"],
                    help="render: BASELINE configs[1] (default, the headline metric; with N > 1 ranks the headline is ONE view "
                         "per step ray-sharded over the ranks and the one-view-per-rank rate is carried as `weak_scaling`); "
                         "train: configs[2], 4096-ray fwd+bwd+Adam step per rank with one RCCL gradient all-reduce; kilo: "
                         "configs[3], KiloNeRF 800x800 frame with the sphere occupancy grid; kilo_train: KiloNeRF distillation "
                         "step (SURVEY 8f rank 3); testset: configs[4], the 200-view test set dealt round-robin over the ranks "
                         "(strong scaling; a step = one pass over all views, a warm-up step = one view per rank; use with "
                         "--dtype bf16); frame: the ray-sharded headline of `render` alone, without per-stage events")
    ap.add_argument("--views", type=int, default=200, help="testset mode: number of views in the set")
    ap.add_argument("--dtype", default=DEFAULT_DTYPE, choices=["fp16x2", "bf16x3", "fp32", "bf16"],
                    help="bf16x3 (default) = BASELINE configs[1] in the reference's fp32 arithmetic: every MLP operand carried "
                         "exactly as three bf16 terms (all 24 significand bits), six bf16 MFMAs per product, fp32 accumulate; "
                         "fp32 = the fp32 MFMA kernel; fp16x2 = two fp16 terms per operand (22 significand bits, three MFMAs "
                         "per product: narrower than the reference's arithmetic, reported beside the headline only); bf16 = "
                         "the bf16 MLP of configs[4], reported under its own metric name.  --mode train runs its matrix work "
                         "on the kernels of the same --dtype (bf16x3, fp32 or fp16x2); --mode kilo runs the tiny MLPs on the "
                         "fp32 MFMA kernel unless --dtype fp16x2")
    ap.add_argument("--kilo-occ", default="bricks", choices=["bricks", "bytes"],
                    help="--mode kilo: the march's count pass on the bit-brick copy of the occupancy grid (default, what "
                         "KiloRenderer does) or on the uint8 grid")
    ap.add_argument("--kilo-dirs", default="per_ray", choices=["per_ray", "per_point"],
                    help="kilo mode: direction encodings once per ray from a table (default) or re-evaluated per point as the "
                         "reference does (A/B switch; bit-identical outputs)")
    ap.add_argument("--train-rays", type=int, default=4096)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-also", action="store_true",
                    help="render mode, 1 GPU: skip the `also` block (bf16 frame, training step, KiloNeRF frame measured in "
                         "the same process after the headline)")
    return ap.parse_args(argv)


def _recorded(path: str, family: str, pick):
    """A figure from a committed rocprofv3 PMC profile -- only while the profile was taken from the kernel sources that
    are in the tree now (nerf_amd/source_hash.py: the profile stores their hash); (None, why) otherwise."""
    from nerf_amd.source_hash import kernel_source_sha256
    try:
        with open(os.path.join(ROOT, "profiles", path)) as f:
            prof = json.load(f)
    except (OSError, ValueError):
        return None, f"profiles/{path} absent"
    if prof.get("kernel_source_sha256") != kernel_source_sha256(family):
        return None, (f"profiles/{path} was recorded from other kernel sources than the ones in the tree (stale): "
                      "re-run tools/run_bench_traffic.sh")
    try:
        return float(pick(prof)), f"profiles/{path} (same kernel sources as this build)"
    except (KeyError, ValueError, TypeError):
        return None, f"profiles/{path} unreadable"


TRAFFIC_PROFILE = {"fp32": "r03_fp32_traffic_pmc.json", "bf16x3": "r03_bf16x3_traffic_pmc.json",
                   "bf16": "r03_bf16_traffic_pmc.json", "fp16x2": "r03_fp16x2_traffic_pmc.json"}


def recorded_traffic(dtype: str = "fp32"):
    """HBM bytes per MLP launch from the committed rocprofv3 PMC passes of this same command (FETCH_SIZE x2 + WRITE_SIZE,
    MI355X_MICROARCH.md corrections; tools/run_bench_traffic.sh collects them, tools/bench_traffic_digest.py digests).
    Counters cannot be read from inside the process, so this is the recorded figure -> (bytes or None, provenance)."""
    return _recorded(TRAFFIC_PROFILE[dtype], dtype, lambda p: p["traffic_bytes_per_launch_mean"])


def recorded_kilo_traffic(kind: str):
    """HBM bytes of the KiloNeRF frame's tiny-MLP stage (route + cell scan + scatter + direction table + kilo_mlp kernel,
    one launch each per frame) from the committed rocprofv3 PMC passes over the same frame (tools/run_pmc.sh ... kilo;
    FETCH_SIZE x2 + WRITE_SIZE as for the other kernels)."""
    def pick(prof):
        stage = [v["derived"]["hbm_bytes"] for k, v in prof["kernels"].items()
                 if k.startswith(("kilo_route_kernel", "kilo_cell_scan_kernel", "kilo_scatter_kernel", "kilo_dir_table_kernel", "kilo_mlp_"))]
        if len(stage) != 5:
            raise KeyError("stage kernels")
        return sum(stage)
    return _recorded({"fp32": "r03_kilo_pmc.json", "fp16x2": "r03_kilo_h2_pmc.json", "bf16x3": "r03_kilo_x3_pmc.json"}[kind], "kilo", pick)


def recorded_composite_traffic():
    """Same for the fine compositing launch."""
    return _recorded("r03_composite_traffic_pmc.json", "composite", lambda p: p["traffic_bytes_fine_composite"])


def cpu_baseline(seconds: float, dev=None, dtype="fp32"):
    """Oracle render of 1024-ray slices of the bench view until ~`seconds` of CPU time.  With `dev`, the first slices
    are then (untimed) rendered by the HIP path with the same uniforms and compared: the end-to-end flip report of
    SURVEY section 7 (fine-sample bin indices that differ because the two coarse passes differ in the last bits) and
    the rgb / depth / acc distance, in the line as `cpu_baseline.parity_check`."""
    from oracle import nerf_oracle as O
    from nerf_amd.weights import make_state_dict
    from nerf_amd.camera import pose_spherical, LEGO_CAMERA_ANGLE_X
    sd = make_state_dict(0, 30.0, 2.0)
    o, d = O.raygen(pose_spherical(0.0), H, W, LEGO_CAMERA_ANGLE_X)
    rng = np.random.Generator(np.random.PCG64(0))
    chunk, done, t0 = 1024, 0, time.perf_counter()
    kept = []
    with torch.no_grad():
        while True:
            r0 = 400 * W + done          # central rows: rays that cross the scene box
            u = rng.random((chunk, N_IMPORTANCE), dtype=np.float32)
            u2 = rng.random((chunk, N_IMPORTANCE), dtype=np.float32)
            out = O.render(sd, o[r0:r0 + chunk], d[r0:r0 + chunk], u=u, u2=u2, return_intermediates=dev is not None)
            if dev is not None and len(kept) < 4:
                kept.append((r0, u, u2, out))
            done += chunk
            el = time.perf_counter() - t0
            if el >= seconds or done >= 64 * chunk:
                break
    line = {"value": done / el, "unit": "rays/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{done} rays of the same 800x800 view (rows 400+), coarse+fine, oracle/nerf_oracle.py "
                      f"on torch-CPU fp32, {el:.1f} s; host has {os.cpu_count()} logical CPUs",
            "ms_per_frame_extrapolated": 1e3 * H * W * el / done}
    if kept:
        line["parity_check"] = parity_check(sd, o, d, kept, dev, dtype)
    return line


def parity_check(sd, o, d, kept, dev, dtype="fp32"):
    """HIP path vs the oracle's kept slices (the oracle is the checker here, never the thing measured)."""
    from nerf_amd import ops
    from nerf_amd.renderer import near_far_table
    bins = torch.from_numpy(near_far_table(2.0, 6.0, N_COARSE)).to(dev)
    pack = {"bf16x3": ops.pack_state_dict_bf16x3, "fp16x2": ops.pack_state_dict_fp16x2}.get(dtype, ops.pack_state_dict)
    mlp = {"bf16x3": ops.mlp_forward_rays_bf16x3, "fp16x2": ops.mlp_forward_rays_fp16x2}.get(dtype, ops.mlp_forward_rays)
    pc, pf = pack(sd, "model", dev), pack(sd, "model_fine", dev)
    flips = samples = 0
    err = {"fine_rgb_map": 0.0, "fine_depth_map": 0.0, "alpha_map": 0.0}
    for r0, u, u2, ref in kept:
        n = u.shape[0]
        ro, rd = torch.from_numpy(o[r0:r0 + n]).to(dev), torch.from_numpy(d[r0:r0 + n]).to(dev)
        raw = mlp(pc, ro, rd, bins)
        _, _, w, _ = ops.composite(raw, bins, True, True)
        z, _, idx = ops.sample_fine(w, bins, bins, torch.from_numpy(u).to(dev), torch.from_numpy(u2).to(dev), want_debug=True)
        rgb, depth, _, acc = ops.composite(mlp(pf, ro, rd, z), z, True, False)
        flips += int((idx.cpu().numpy().astype(np.int64) != ref["bin_indices"].numpy()).sum())
        samples += idx.numel()
        for k, got in (("fine_rgb_map", rgb), ("fine_depth_map", depth), ("alpha_map", acc)):
            err[k] = max(err[k], float(np.abs(got.cpu().numpy() - ref[k].numpy().reshape(got.shape)).max()))
    return {"rays": samples // N_IMPORTANCE, "fine_sample_indices": samples, "index_flips": flips,
            "flip_rate": flips / max(samples, 1), "max_abs_err": err,
            "note": "HIP coarse MLP -> HIP weights -> HIP indices vs oracle indices, same uniforms; stage-wise (same weights "
                    "in) the indices are bit-exact (tests/test_gpu_parity.py), these flips come from last-bit differences "
                    "of the coarse weights"}


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
    # --dtype bf16x3 (the default): the step's matrix work (two forward passes, the backward's data-gradient chain, the
    # weight-gradient GEMMs) on the three-term bf16 kernels (fp32-class, same saved activations); the alpha / rgb head
    # rows stay on the fp32 MFMA / VALU.  --dtype fp32: everything on the fp32 MFMA.
    x3 = args.dtype in ("bf16x3", "fp16x2")
    h2 = args.dtype == "fp16x2"                  # forward passes, data-gradient chain and weight-gradient GEMMs on the fp16x2 kernels
    ts = TrainStep(net, dev, forward_dtype=args.dtype if x3 else "fp32")
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
        peak = BF16_MFMA_PEAK_TFLOPS / 3 if h2 else BF16_MFMA_PEAK_TFLOPS / 6 if x3 else FP32_MFMA_PEAK_TFLOPS
        line = {
            "metric": "training rays/sec (fwd+bwd+Adam step, 64 coarse + 128 fine samples/ray)",
            "value": world * R * args.steps / elapsed, "unit": "rays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (matrix work fp16x2: f32 operands as 2 fp16 terms = 22 significand bits, 3 fp16 MFMAs per product, f32 "
                     "accumulate; the backward at power-of-two scales chosen on the device and undone exactly)" if h2 else
                     "f32 (matrix work bf16x3: f32 operands as 3 bf16 terms, 6 bf16 MFMAs per product, f32 accumulate)" if x3 else "f32",
            "data": "synthetic (random pixels of orbit views, random gt, random-init weights)",
            "config": {"workload": "BASELINE configs[2]: NeRF training step, %d-ray batch per rank, fwd+bwd with Adam, "
                                   "data-parallel RCCL all-reduce" % R, "rays_per_rank": R, "perturb": 1,
                       "forward_kernel": "mlp_fp16x2_kernel<save>" if h2 else "mlp_bf16x3_kernel<save>" if x3 else "mlp_fwd_kernel<save>"},
            "roofline": {"kernel": "whole step (%s)" % ("mlp_fp16x2<save> + mlp_dgrad_h2 + wgrad_h2p GEMMs" if h2 else
                                                        "mlp_bf16x3<save> + mlp_dgrad_x3 + wgrad_x3 GEMMs" if x3 else "mlp_fwd<save> + mlp_dgrad + wgrad GEMMs"),
                         "bound": "mfma", "achieved": flop / (ms * 1e-3) / 1e12,
                         "peak": peak, "unit": "TFLOP/s", "frac": flop / (ms * 1e-3) / 1e12 / peak,
                         "traffic": None,
                         "note": "algorithmic FLOPs of the step / wall time of the step, not one kernel; peak = " +
                                 ("fp16 MFMA / 3 (three MFMAs per fp32 product); the "
                                  "step also moves ~42 GB of saved activations and dY rows through HBM (DESIGN 3.5)" if h2 else
                                  "bf16 MFMA / 6 (six MFMAs per fp32 product); the step also moves ~42 GB of saved activations "
                                  "and dY rows through HBM (DESIGN 3.5)" if x3 else "fp32 MFMA")},
            "final_loss": loss}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = train_cpu_baseline(args.cpu_seconds)
        return line
    return None


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
    # the tiny MLPs' arithmetic: fp32 MFMA kernel, the same 24-bit operands as three bf16 terms (bf16x3), or fp16x2 (22-bit)
    kind = args.dtype if args.dtype in ("fp16x2", "bf16x3") else "fp32"
    r = KiloRenderer(net, default_kilo_cfg(), occ, mlp_dtype=kind)
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
    # the occupancy grid as bit bricks (built once per grid, as KiloRenderer does); --kilo-occ bytes walks the uint8 grid
    bricks = r.occ_bricks() if getattr(args, "kilo_occ", "bricks") == "bricks" else None

    def step(k, timed):
        rays_o, rays_d = ops.raygen(poses[(k * world + rank) % len(poses)], H, W, focal, dev)
        e0 = E()
        pts, ray_idx, ts, te, offsets, counts = kilo_ops.kilo_march(rays_o, rays_d, r.occ_grid, aabb, r.near, r.far, r.step,
                                                                    bricks=bricks)
        e1 = E()
        raw = kilo_ops.kilo_forward(packed, net.num_models, aabb, res, pts, rays_d, ray_idx, kind=kind,
                                    encode_dirs_per_ray=getattr(args, "kilo_dirs", "per_ray") == "per_ray")
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
            "scaling": "weak", "vs_baseline": None, "dtype": DTYPE_LABEL[kind],
            "data": "synthetic (orbit camera, random-init cells, sigma head x10 + 0.5, sphere occupancy)",
            "config": {"workload": "BASELINE configs[3]: KiloNeRF 16^3 cells W=32 D=2, sphere |x|<0.75 occupancy at 256^3, "
                                   "step 0.005, ERT 0.01, 800x800 view per rank per step", "rays_per_frame": R,
                       "samples_per_frame": n, "samples_per_hit_ray": n / max(1.0, float((out[2] > 0).sum()))},
            "roofline": {"kernel": "kilo_forward (route + counting sort + %s)" % {
                "fp32": "kilo_mlp_kernel", "fp16x2": "kilo_mlp_h2_kernel: 3 fp16 MFMAs per fp32 product; peak = fp16 peak / 3",
                "bf16x3": "kilo_mlp_x3_kernel: 6 bf16 MFMAs per fp32 product; peak = bf16 peak / 6"}[kind], "bound": "mfma",
                         "achieved": flop / (ms["mlp"] * 1e-3) / 1e12, "peak": PEAK_TFLOPS[kind], "unit": "TFLOP/s",
                         "frac": flop / (ms["mlp"] * 1e-3) / 1e12 / PEAK_TFLOPS[kind], "traffic": recorded_kilo_traffic(kind)[0],
                         "traffic_note": "HBM bytes of the stage's kernels per frame (far from the HBM bound): " + recorded_kilo_traffic(kind)[1],
                         "samples_per_s": n / (ms["mlp"] * 1e-3)},
            "stage_ms": ms}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = kilo_cpu_baseline(args.cpu_seconds, sd, n)
        return line
    return None


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
                                   "Adam over 25.4 M parameters (cells without gradient history skipped: their update is "
                                   "the identity), data-parallel RCCL all-reduce of the sampled cells' gradient records" % R,
                       "rays_per_rank": R, "samples_per_step": n,
                       "cells_exchanged": (int((ts.touched != 0).sum()) if ts.touched is not None else None)},
            # forward 96 + backward 240 MFMAs of 32x32x2 per 32-sample tile: 12,288 + 30,720 FLOP per
            # sample; the step's other kernels are byte movers
            "roofline": {"kernel": "whole step: tiny-MLP FLOPs (kilo_mlp_kernel 96 + kilo_bwd_kernel 240 MFMAs per 32-sample "
                                   "tile) / wall time; kilo_bwd_kernel is a third of the step (DESIGN 7.1, profiles/)",
                         "bound": "mfma", "achieved": n * (12288 + 30720) / (ms * 1e-3) / 1e12, "peak": FP32_MFMA_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": n * (12288 + 30720) / (ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                         "traffic": None,
                         "note": "a latency- and byte-bound step, not an MFMA-bound one: dense Adam would move %.0f MB per "
                                 "step; see DESIGN.md section 7.1 for the per-kernel times (rocprofv3)" % (adam_bytes / 1e6)},
            "final_loss": loss}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = kilo_train_cpu_baseline(args.cpu_seconds, sd)
        return line
    return None


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
        peak = PEAK_TFLOPS[args.dtype] * world
        achieved = flop * args.steps / elapsed / 1e12
        line = {
            "metric": "rays/sec (%d-view test set, 800x800, 64 coarse + 128 fine samples/ray%s)" % (len(poses), ", bf16 MLP" if bf16 else ""),
            "value": len(poses) * R * args.steps / elapsed, "unit": "rays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "ms_per_frame": 1e3 * elapsed / args.steps / len(poses),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": DTYPE_LABEL[args.dtype],
            "data": "synthetic (orbit camera, random-init weights, alpha_linear x30 + 2)",
            "config": {"workload": "BASELINE configs[4]: %d-view test-set render, views dealt round-robin over %d rank(s), %s MLP"
                                   % (len(poses), world, args.dtype), "views": len(poses), "rays_per_frame": R,
                       "views_per_rank": len(mine)},
            "roofline": {"kernel": "whole job (MLP FLOPs of all views / wall time / (peak x ranks))", "bound": "mfma",
                         "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "traffic": None}}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
        return line
    return None


def render_bench(args, net, dev, rank, world, dist, dtype="fp32", steps=None, warmup=None, staged=True, cpu=True):
    """The headline (BASELINE configs[1], or configs[4]'s per-view form with dtype bf16).  A step = ONE 800x800 view:
    with one rank the whole frame on that GPU; with N ranks the view's 640,000 rays in contiguous slices over the ranks
    (north star: "rays/image tiles are sharded"), every rank generating its own rays from the pose, and one all_gather
    of 36 B per ray so every rank ends the step holding the whole frame (strong scaling: ms per frame at N GPUs).
    With N > 1 the one-view-per-rank rate (no collective) is measured after it and carried as `weak_scaling`."""
    from nerf_amd import ops
    from nerf_amd.camera import test_set_poses, focal_from_angle
    from nerf_amd.config import default_cfg
    from nerf_amd.parallel import render_frame_sharded
    from nerf_amd.renderer import Renderer
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    renderer = Renderer(net, default_cfg(perturb=0), mlp_dtype=dtype)
    bf16, x3, h2 = dtype == "bf16", dtype == "bf16x3", dtype == "fp16x2"
    # bf16x3 / fp16x2: six / three MFMAs stand for one fp32 product, so the ceiling on the FLOPs of the fp32 problem is
    # the matrix peak / 6 or / 3
    peak = PEAK_TFLOPS[dtype]
    poses = test_set_poses(200)
    focal = focal_from_angle(W)
    R = H * W
    events = {}
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    keys = ("coarse_rgb_map", "coarse_depth_map", "fine_rgb_map", "fine_depth_map", "alpha_map")

    def rows(c2w, ev):
        def render(first, count):
            rays_o, rays_d = ops.raygen(c2w, H, W, focal, dev, first_ray=first, n_rays=count)
            u = torch.rand((count, N_IMPORTANCE), device=dev, generator=gen)
            u2 = torch.rand((count, N_IMPORTANCE), device=dev, generator=gen)
            if staged:
                out = renderer.render_rays_staged(rays_o, rays_d, u, u2, events=ev)
            else:
                out = renderer.render_rays(rays_o, rays_d, u, u2)
            return {k: out[k] for k in keys}
        return render

    def frame_step(k, ev):          # one view, ray-sharded over the ranks, assembled on every rank
        return render_frame_sharded(rows(poses[k % len(poses)], ev), R, rank, world, gather=True)[0]

    def view_step(k, ev):           # one view per rank, no collective
        return rows(poses[(k * world + rank) % len(poses)], ev)(0, R)

    def sync():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def timed(step, ev):
        with torch.no_grad():
            for k in range(warmup):
                out = step(k, None)
            sync()
            t0 = time.perf_counter()
            for k in range(steps):
                out = step(warmup + k, ev)
            sync()
            elapsed = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        assert out["fine_rgb_map"].shape == (R, 3) and torch.isfinite(out["fine_rgb_map"]).all()
        return elapsed

    elapsed = timed(frame_step, events if staged else None)
    weak = None
    if world > 1:
        el_w = timed(view_step, None)
        weak = {"value": world * R * steps / el_w, "unit": "rays/s", "ms_per_step": 1e3 * el_w / steps, "scaling": "weak",
                "note": "every rank renders its own 800x800 view per step (views of the 200-view orbit dealt round-robin), "
                        "no data-path collective"}
    if rank != 0:
        return None
    ms_step = 1e3 * elapsed / steps
    mine = -(-R // world)                                      # rays of rank 0's slice (the largest)
    line = {
        "metric": "rays/sec (full 800x800 frame, 64 coarse + 128 fine samples/ray)" +
                  (", bf16 MLP" if bf16 else ""),
        "value": R * steps / elapsed,
        "unit": "rays/s",
        "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": ms_step,
        "ms_per_frame": ms_step,
        "higher_is_better": True,
        "scaling": "weak" if world == 1 else "strong",
        "vs_baseline": None,
        "dtype": DTYPE_LABEL[dtype],
        "data": "synthetic (orbit camera, random-init weights, alpha_linear x30 + 2)",
        "config": {"workload": ("BASELINE configs[4] per-view form: full 800x800 test-view render, NeRF coarse+fine, bf16 MLP "
                                "(bf16 weights + activations, fp32 accumulate), " if bf16 else
                                "BASELINE configs[1] with the MLP's fp32 operands carried as three bf16 terms each (six bf16 "
                                "MFMAs per product, fp32 accumulate; fp32 in, fp32 out, fp32 everywhere else), " if x3 else
                                "BASELINE configs[1] with the MLP's fp32 operands carried as two fp16 terms each (22 significand "
                                "bits, three fp16 MFMAs per product, fp32 accumulate; fp32 in, fp32 out, fp32 everywhere "
                                "else), " if h2 else
                                "BASELINE configs[1]: full 800x800 lego test-view render, NeRF coarse+fine, fp32, ") +
                               ("one view per step on one GPU" if world == 1 else
                                "one view per step, its rays in contiguous slices over %d ranks, outputs all-gathered on "
                                "every rank" % world),
                   "rays_per_frame": R, "n_coarse": N_COARSE, "n_importance": N_IMPORTANCE, "rays_per_rank": mine,
                   "sharding": ("whole frame on one GPU" if world == 1 else
                                f"contiguous ray slices of one view over {world} ranks + one all_gather of 36 B/ray")},
    }
    if weak is not None:
        line["weak_scaling"] = weak
    if staged:
        def mean_ms(name):
            return float(np.mean([a.elapsed_time(b) for a, b in events[name]]))

        ms_mlp_c, ms_mlp_f = mean_ms("mlp_coarse"), mean_ms("mlp_fine")
        ms_cmp_c, ms_cmp_f, ms_smp = mean_ms("composite_coarse"), mean_ms("composite_fine"), mean_ms("sample_fine")
        flop_launch = 0.5 * mine * (N_COARSE + N_COARSE + N_IMPORTANCE) * FLOP_PER_POINT   # mean of the 2 launches
        ms_launch = 0.5 * (ms_mlp_c + ms_mlp_f)
        achieved = flop_launch / (ms_launch * 1e-3) / 1e12
        cmp_bytes = mine * ((N_COARSE + N_IMPORTANCE) * 20 + 20)
        whole = world == 1
        line["roofline"] = {
            "kernel": MLP_KERNEL[dtype],
            "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
            "traffic": recorded_traffic(dtype)[0] if whole else None,
            "traffic_note": "HBM bytes per launch (mean of coarse + fine) from rocprofv3 PMC passes of this command at 1 GPU: "
                            "%s; algorithmic bytes per launch %.4g"
                            % (recorded_traffic(dtype)[1], 0.5 * mine * ((N_COARSE + N_COARSE + N_IMPORTANCE) * 16 + 48 + (N_COARSE + N_IMPORTANCE) * 4)),
            "flop_per_launch": flop_launch, "ms_per_launch": ms_launch,
            "ms_coarse_launch": ms_mlp_c, "ms_fine_launch": ms_mlp_f, "measured_on": "rank 0, HIP events on the launch stream"}
        line["roofline_composite"] = {
            "kernel": "composite_kernel<3> (fine pass)", "bound": "hbm",
            "achieved": cmp_bytes / (ms_cmp_f * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": cmp_bytes / (ms_cmp_f * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "traffic": recorded_composite_traffic()[0] if whole else None, "traffic_note": recorded_composite_traffic()[1],
            "ms_per_launch": ms_cmp_f,
            "bytes_per_launch": cmp_bytes}
        line["stage_ms"] = {"mlp_coarse": ms_mlp_c, "composite_coarse": ms_cmp_c, "sample_fine": ms_smp,
                            "mlp_fine": ms_mlp_f, "composite_fine": ms_cmp_f}
    else:
        flop = float(R) * (N_COARSE + N_COARSE + N_IMPORTANCE) * FLOP_PER_POINT
        achieved = flop * steps / elapsed / 1e12
        line["roofline"] = {"kernel": "whole step (MLP FLOPs of the frame / wall time / (peak x ranks))", "bound": "mfma",
                            "achieved": achieved, "peak": peak * world, "unit": "TFLOP/s", "frac": achieved / (peak * world),
                            "traffic": None}
    if cpu and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(args.cpu_seconds, None if bf16 else dev, dtype)
    return line


def also_block(args, net, dev):
    """BASELINE configs[4] (per-view form), configs[2] and configs[3] under the same clock as the headline: measured in
    this process right after it, one GPU, no CPU legs.  Each entry is a cut of the line its own --mode prints."""
    def cut(line):
        return {"metric": line["metric"], "value": line["value"], "unit": line["unit"], "ms_per_step": line["ms_per_step"],
                "steps": line["steps"], "warmup": line["warmup"], "dtype": line["dtype"],
                "workload": line["config"]["workload"],
                "roofline": {k: line["roofline"].get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac")}}

    sub = argparse.Namespace(**vars(args))
    sub.no_cpu_baseline = True
    out = {}
    for other in ("fp16x2", "bf16x3", "fp32"):     # the headline's other fp32-class kernels, same view sequence
        if other != args.dtype:
            out[other + "_frame"] = cut(render_bench(sub, net, dev, 0, 1, None, dtype=other, steps=3, warmup=1, cpu=False))
            torch.cuda.empty_cache()
    out["bf16_frame"] = cut(render_bench(sub, net, dev, 0, 1, None, dtype="bf16", steps=5, warmup=1, cpu=False))
    torch.cuda.empty_cache()
    # configs[2] / configs[3] in 24-bit-operand arithmetic first (the reference's: bf16x3 matrix work for the training
    # step and the KiloNeRF tiny MLPs), their narrower fp16x2 forms beside them
    wide = "fp32" if args.dtype == "fp32" else "bf16x3"
    for key, dtype in (("train_step", wide), ("train_step_fp16x2", "fp16x2")):
        sub.steps, sub.warmup, sub.dtype = 20, 3, dtype
        out[key] = cut(train_bench(sub, copy_network(net, dev), dev, 0, 1, None))
        torch.cuda.empty_cache()
    # (KiloNeRF: the three-term bf16 kernel is the faster of the two 24-bit forms; the fp32 MFMA kernel's frame beside it)
    for key, dtype in (("kilo_frame", "bf16x3"), ("kilo_frame_fp32", "fp32"), ("kilo_frame_fp16x2", "fp16x2")):
        sub.steps, sub.warmup, sub.dtype = 20, 3, dtype
        out[key] = cut(kilo_bench(sub, dev, 0, 1, None))
        torch.cuda.empty_cache()
    return out


def copy_network(net, dev):
    """TrainStep re-points a network's parameters at its flat buffer: give it a copy, not the bench's network."""
    from nerf_amd.network import Network
    twin = Network()
    twin.load_state_dict(net.state_dict())
    return twin.to(dev)


# ---- ranks ------------------------------------------------------------------------------------------------------------
def launch_ranks(args, argv) -> int:
    """`python bench.py --gpus N` started plainly: this process touches no GPU (torch.cuda.device_count() does not
    initialise one here) and starts N children of this same file, one rank per GPU, with the env:// rendezvous
    variables torch.distributed.run would set (the reference's launch contract: train.py:116-120 reads LOCAL_RANK and
    calls init_process_group("nccl", "env://")).  Rank 0's stdout carries the JSON line.  A rank that fails ends the
    others (by their PIDs) and its exit code is ours."""
    import socket
    import subprocess
    n = args.gpus
    backend = os.environ.get("NRF_BENCH_BACKEND", "nccl")
    if backend == "nccl" and not DRY_RUN and torch.cuda.device_count() < n:
        raise SystemExit(f"bench.py --gpus {n}: this node shows {torch.cuda.device_count()} GPU(s); RCCL needs one GPU per "
                         "rank (NRF_BENCH_BACKEND=gloo rehearses more ranks than GPUs)")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.abspath(__file__)] + list(argv), env=env))
    code = 0
    live = list(procs)
    while live:
        time.sleep(0.05)
        for p in list(live):
            rc = p.poll()
            if rc is None:
                continue
            live.remove(p)
            if rc != 0 and code == 0:
                code = rc
                for q in live:
                    q.terminate()
    return code


DRY_RUN = os.environ.get("NRF_BENCH_DRY_RUN", "") == "1"


def dry_run(args, rank, world, dist):
    """NRF_BENCH_DRY_RUN=1: the launch / rendezvous / barrier / max-over-ranks plumbing with an empty step, for hosts
    without a GPU (tests/test_bench_launch.py).  It measures nothing and says so: value is null."""
    def sync():
        if dist is not None:
            dist.barrier()

    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pass
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank != 0:
        return None
    return {"metric": "dry run (no GPU work: launcher and rendezvous only)", "value": None, "unit": "rays/s",
            "n_gpus": dist.get_world_size() if dist is not None else 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": None, "dry_run": True, "mode": args.mode,
            "backend": dist.get_backend() if dist is not None else None}


def main():
    argv = sys.argv[1:]
    args = parse(argv)
    if "RANK" not in os.environ and "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args, argv))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; start it plainly (it spawns its own ranks) "
                         "or give torch.distributed.run the same --nproc-per-node")
    # One rank per GPU on RCCL (backend "nccl").  NRF_BENCH_BACKEND=gloo is a rehearsal switch for boxes with fewer GPUs
    # than ranks (ranks then share devices and the collectives are staged through the host); it is never the measured
    # configuration and the line says which backend ran.
    backend = os.environ.get("NRF_BENCH_BACKEND", "nccl")
    dist = None
    if DRY_RUN:
        if world > 1:
            import torch.distributed as dist
            dist.init_process_group(backend="gloo", init_method="env://")
        line = dry_run(args, rank, world, dist)
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
        if backend != "nccl":
            local %= torch.cuda.device_count()
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
        if world > 1:
            import torch.distributed as dist
            if backend == "nccl":
                dist.init_process_group(backend="nccl", init_method="env://", device_id=dev)
            else:
                dist.init_process_group(backend=backend, init_method="env://")
            world = dist.get_world_size()                       # what the process group itself reports
        line = run_mode(args, dev, rank, world, dist)
        if line is not None and dist is not None:
            line["backend"] = dist.get_backend() + (" (= RCCL over xGMI)" if dist.get_backend() == "nccl" else
                                                    " (rehearsal: ranks share GPUs, collectives staged through the host)")
    if line is not None:
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def run_mode(args, dev, rank, world, dist):
    from nerf_amd.network import Network
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
        return render_bench(args, net, dev, rank, world, dist, dtype=args.dtype, staged=False)
    line = render_bench(args, net, dev, rank, world, dist, dtype=args.dtype)
    if line is not None and world == 1 and args.dtype != "bf16" and not args.no_also:
        line["also"] = also_block(args, net, dev)
    return line


if __name__ == "__main__":
    main()
