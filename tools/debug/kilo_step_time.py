#!/usr/bin/env python3
"""Host-side time per wrapper call of one KiloTrainStep.step (perf_counter around every kilo_ops / ops function it uses;
kilo_march contains the step's one host sync, so its figure includes waiting for the GPU to drain the previous step)."""
import collections
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nerf_amd import kilo_ops, ops, kilo_train  # noqa: E402
from nerf_amd.camera import focal_from_angle, test_set_poses  # noqa: E402
from nerf_amd.kilonerf import KiloNetwork, KiloRenderer, default_kilo_cfg  # noqa: E402
from nerf_amd.weights import make_kilo_state_dict_arrays  # noqa: E402

acc = collections.defaultdict(float)


def wrap(mod, name):
    f = getattr(mod, name)

    def g(*a, **k):
        t = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            acc[mod.__name__.split(".")[-1] + "." + name] += time.perf_counter() - t
    setattr(mod, name, g)


for n in ("kilo_march", "kilo_forward", "kilo_composite", "kilo_composite_backward", "kilo_backward", "kilo_route", "l2_reg",
          "kilo_pack_weights", "kilo_pack_weights_transposed"):
    wrap(kilo_ops, n)
for n in ("mse_grad", "clip_adam", "raygen"):
    wrap(ops, n)
wrap(kilo_train, "allreduce_gradients")


class TimedLib:          # the C calls alone (ctypes marshalling included)
    def __init__(self, real):
        self._real = real

    def __getattr__(self, name):
        f = getattr(self._real, name)

        def g(*a):
            t = time.perf_counter()
            try:
                return f(*a)
            finally:
                acc["C:" + name] += time.perf_counter() - t
        return g


_empty = torch.empty
slow = collections.Counter()


def timed_empty(*a, **k):
    t = time.perf_counter()
    try:
        return _empty(*a, **k)
    finally:
        dt = time.perf_counter() - t
        acc["torch.empty"] += dt
        if dt > 50e-6:
            slow[str(a[0])[:40]] += 1


torch.empty = timed_empty
_real_lib = kilo_ops.lib()
_timed = TimedLib(_real_lib)
kilo_ops.lib = lambda: _timed

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
R, N, WARM = 4096, 200, 20
tot = 0.0
for k in range(N + WARM):
    if k == WARM:
        torch.cuda.synchronize()
        acc.clear()
        t0 = time.perf_counter()
    pix = torch.randint(0, 640000, (R,), device=dev, generator=gen)
    o, d = ops.raygen(poses[k % 200], 800, 800, focal, dev, pixel_index=pix)
    t_rgb = torch.rand((R, 3), device=dev, generator=gen)
    t_alpha = torch.rand((R, 1), device=dev, generator=gen)
    ts.step(o, d, t_rgb, t_alpha, stage="distillation")
torch.cuda.synchronize()
el = time.perf_counter() - t0
print(f"step {1e3 * el / N:.3f} ms; host time inside wrappers, us per step:")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"  {k:40s} {1e6 * v / N:8.1f}")
print("torch.empty calls over 50 us:", dict(slow), "reserved MB", torch.cuda.memory_reserved() >> 20, "mallocs",
      torch.cuda.memory_stats().get("num_device_alloc"))
print(f"  {'(other host code)':40s} {1e6 * (el - sum(acc.values())) / N:8.1f}")
