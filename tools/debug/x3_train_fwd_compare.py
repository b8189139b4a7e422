"""The bf16x3 training forward of two builds of the library side by side (A/B of kernel forms): raw against the inference
forward of the same build, run-to-run determinism, and the saved rows / sign words of build B against build A's.
usage: x3_train_fwd_compare.py <libA.so> <libB.so>"""
import ctypes as C
import os
import sys

import torch

P = C.c_void_p
dev = torch.device("cuda:0")
R, S = 2048, 192
g = torch.Generator().manual_seed(0)
o = (torch.tensor([0.0, 0.0, 4.0]) + 0.1 * torch.randn(R, 3, generator=g)).to(dev)
d = torch.randn(R, 3, generator=g)
d[:, 2] = -d[:, 2].abs() - 1
d = d.to(dev)
z = torch.sort(2 + 4 * torch.rand(R, S, generator=g), -1).values.to(dev)
shapes = [(256, 63)] + [(256, 256)] * 4 + [(256, 319)] + [(256, 256)] * 3 + [(1, 256), (128, 283), (3, 128)]
ws = [torch.randn(s, device=dev, generator=torch.Generator(device=dev).manual_seed(i)) * 0.08 for i, s in enumerate(shapes)]
bs = [torch.randn(s[0], device=dev, generator=torch.Generator(device=dev).manual_seed(100 + i)) * 0.05 for i, s in enumerate(shapes)]
res = []
for path in sys.argv[1:3]:
    L = C.CDLL(os.path.abspath(path))
    for f in ("nrf_packed_bf16x3_bytes", "nrf_mlp_acts_bytes", "nrf_mlp_masks_bytes"):
        getattr(L, f).restype = C.c_int64
    packed = torch.zeros(L.nrf_packed_bf16x3_bytes(), dtype=torch.uint8, device=dev)
    wp = (P * 12)(*[w.data_ptr() for w in ws])
    bp = (P * 12)(*[b.data_ptr() for b in bs])
    assert L.nrf_pack_weights_bf16x3(wp, bp, P(packed.data_ptr()), None) == 0
    na, nm = L.nrf_mlp_acts_bytes(C.c_int64(R), C.c_int(S)), L.nrf_mlp_masks_bytes(C.c_int64(R), C.c_int(S))
    outs = []
    for rep in range(2):
        raw = torch.empty(R, S, 4, device=dev)
        acts = torch.zeros(na // 4, device=dev)
        masks = torch.zeros(nm // 4, dtype=torch.int32, device=dev)
        st = L.nrf_mlp_forward_rays_train_bf16x3(P(packed.data_ptr()), P(o.data_ptr()), P(d.data_ptr()), P(z.data_ptr()), C.c_int64(S),
                                                 C.c_int64(R), C.c_int(S), P(raw.data_ptr()), P(acts.data_ptr()), P(masks.data_ptr()), None)
        assert st == 0, st
        torch.cuda.synchronize()
        outs.append((raw, acts, masks))
    inf = torch.empty(R, S, 4, device=dev)
    assert L.nrf_mlp_forward_rays_bf16x3(P(packed.data_ptr()), P(o.data_ptr()), P(d.data_ptr()), P(z.data_ptr()), C.c_int64(S), C.c_int64(R),
                                         C.c_int(S), P(inf.data_ptr()), None) == 0
    torch.cuda.synchronize()
    print(os.path.basename(path), "train raw == inference raw:", bool(torch.equal(outs[0][0], inf)),
          "| run-to-run raw / acts / masks identical:", bool(torch.equal(outs[0][0], outs[1][0])), bool(torch.equal(outs[0][1], outs[1][1])),
          bool(torch.equal(outs[0][2], outs[1][2])), flush=True)
    res.append(outs[0])
(ra, aa, ma), (rb, ab, mb) = res
print("B vs A: raw max |d|", float((ra - rb).abs().max()), " acts max |d|", float((aa - ab).abs().max()), " acts max |A|", float(aa.abs().max()),
      " sign bits that differ", int(torch.bitwise_xor(ma, mb).to(torch.int64).bitwise_and(0xFFFFFFFF).cpu().apply_(lambda v: bin(v).count("1")).sum()),
      "of", ma.numel() * 32)
