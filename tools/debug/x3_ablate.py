"""Times nrf_mlp_forward_rays_bf16x3 from arbitrary builds of the library in one process (A/B of kernel variants) and
checks that every build returns the first build's bits.  usage: x3_ablate.py [--kind fp16x2] <lib.so> [<lib.so> ...]
(--kind fp16x2: nrf_mlp_forward_rays_fp16x2 from builds made by tools/debug/h2_ab.sh)"""
import ctypes as C
import os
import sys
import time

import torch

P = C.c_void_p
dev = torch.device("cuda:0")
R, S = 65536, 192
g = torch.Generator().manual_seed(0)
o = (torch.tensor([0.0, 0.0, 4.0]) + 0.1 * torch.randn(R, 3, generator=g)).to(dev)
d = torch.randn(R, 3, generator=g)
d[:, 2] = -d[:, 2].abs() - 1
d = d.to(dev)
z = torch.sort(2 + 4 * torch.rand(R, S, generator=g), -1).values.to(dev)
shapes = [(256, 63)] + [(256, 256)] * 4 + [(256, 319)] + [(256, 256)] * 3 + [(1, 256), (128, 283), (3, 128)]
ws = [torch.randn(s, device=dev, generator=torch.Generator(device=dev).manual_seed(i)) * 0.05 for i, s in enumerate(shapes)]
bs = [torch.zeros(s[0], device=dev) for s in shapes]
first = None
paths = sys.argv[1:]
kind = "bf16x3"
if paths and paths[0] == "--kind":
    kind, paths = paths[1], paths[2:]
for path in paths:
    L = C.CDLL(os.path.abspath(path))
    getattr(L, f"nrf_packed_{kind}_bytes").restype = C.c_int64
    packed = torch.zeros(getattr(L, f"nrf_packed_{kind}_bytes")(), dtype=torch.uint8, device=dev)
    wp = (P * 12)(*[w.data_ptr() for w in ws])
    bp = (P * 12)(*[b.data_ptr() for b in bs])
    assert getattr(L, f"nrf_pack_weights_{kind}")(wp, bp, P(packed.data_ptr()), None) == 0
    raw = torch.empty(R, S, 4, device=dev)

    def run():
        st = getattr(L, f"nrf_mlp_forward_rays_{kind}")(P(packed.data_ptr()), P(o.data_ptr()), P(d.data_ptr()), P(z.data_ptr()),
                                           C.c_int64(S), C.c_int64(R), C.c_int(S), P(raw.data_ptr()), None)
        assert st == 0, st
    run()
    torch.cuda.synchronize()
    if first is None:
        first = raw.clone()
    same = bool(torch.equal(raw, first))
    best = 1e9
    for _ in range(3):
        t = time.perf_counter()
        run()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t)
    print(f"{path.split('/')[-1]:32s} {best * 1e3:8.2f} ms  {R * S * 1186816 / best / 1e12:7.1f} TFLOP/s  same bits as first: {same}", flush=True)
