#!/usr/bin/env python3
"""Where a wave of kilo_bwd_kernel spends its time: a -DKB_STAMP build of kilo_bwd.hip sums s_memtime ticks (100 MHz)
per phase over all waves.  usage: kilo_bwd_phases.py path/to/stamped/libnerf_amd.so"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import nerf_amd._lib as _lib  # noqa: E402

_lib.LIB_PATH = os.path.abspath(sys.argv[1])
_lib.SYMBOLS = tuple(_lib.SYMBOLS)
import torch  # noqa: E402,F401

L = _lib.lib()
out = (C.c_ulonglong * 8)()
L.nrf_debug_kb_stamps.argtypes = [C.c_void_p, C.c_int]
sys.argv = [sys.argv[0], sys.argv[1]]
exec(open(os.path.join(ROOT, "tools", "debug", "kilo_bwd_time.py")).read().split("grads = None")[0])   # the batch
grads = None
for it in range(3):
    grads = kilo_ops.kilo_backward(packed, packed_t, 4096, pts, d, ray_idx, g_raw, ws, grads=grads)  # noqa: F821
torch.cuda.synchronize()
L.nrf_debug_kb_stamps(out, 1)
N = 10
for it in range(N):
    kilo_ops.kilo_backward(packed, packed_t, 4096, pts, d, ray_idx, g_raw, ws, grads=grads)  # noqa: F821
torch.cuda.synchronize()
L.nrf_debug_kb_stamps(out, 0)
names = ["job fetch", "cell search", "record staging", "first gather", "forward + dgrad (per tile)", "wgrad (per tile)",
         "job record"]
tot = sum(out[:7])
print(f"ticks of 10 ns summed over 1024 waves, per launch; tiles {int(((cnt + 31) // 32).sum())}")  # noqa: F821
for n, v in zip(names, out):
    print(f"  {n:30s} {v / N / 1024 * 0.01:8.2f} us per wave  {100 * v / tot:5.1f} %")
