"""The bf16x3 data-gradient chain of two builds of the library side by side (A/B of kernel forms): the dY rows the chain
leaves in the workspace and the flat gradients, per layer, of build B against build A's on the same saved forward (A's).
usage: x3_dgrad_compare.py <libA.so> <libB.so> [n_samples [n_rays]]"""
import ctypes as C
import os
import sys

import torch

P = C.c_void_p
dev = torch.device("cuda:0")
R, S = int(sys.argv[4]) if len(sys.argv) > 4 else 1024, int(sys.argv[3]) if len(sys.argv) > 3 else 64
g = torch.Generator().manual_seed(0)
o = (torch.tensor([0.0, 0.0, 4.0]) + 0.1 * torch.randn(R, 3, generator=g)).to(dev)
d = torch.randn(R, 3, generator=g)
d[:, 2] = -d[:, 2].abs() - 1
d = d.to(dev)
z = torch.sort(2 + 4 * torch.rand(R, S, generator=g), -1).values.to(dev)
shapes = [(256, 63)] + [(256, 256)] * 4 + [(256, 319)] + [(256, 256)] * 3 + [(1, 256), (128, 283), (3, 128)]
ws = [torch.randn(s, device=dev, generator=torch.Generator(device=dev).manual_seed(i)) * 0.08 for i, s in enumerate(shapes)]
bs = [torch.randn(s[0], device=dev, generator=torch.Generator(device=dev).manual_seed(100 + i)) * 0.05 for i, s in enumerate(shapes)]
g_raw = (torch.randn(R, S, 4, generator=g) * torch.tensor([1.0, 1.0, 1.0, 0.05])).to(dev)
KDY = 2464
ROWS = [("dY0..7", 0, 2048), ("dFeat", 2048, 2304), ("dYv", 2304, 2432), ("d raw", 2432, 2436)]
res, saved = [], None
for path in sys.argv[1:3]:
    L = C.CDLL(os.path.abspath(path))
    for f in ("nrf_packed_bf16x3_bytes", "nrf_packed_floats", "nrf_packed_bwd_bf16x3_bytes", "nrf_mlp_acts_bytes", "nrf_mlp_masks_bytes",
              "nrf_mlp_backward_workspace_bytes"):
        getattr(L, f).restype = C.c_int64
    wp = (P * 12)(*[w.data_ptr() for w in ws])
    bp = (P * 12)(*[b.data_ptr() for b in bs])
    px = torch.zeros(L.nrf_packed_bf16x3_bytes(), dtype=torch.uint8, device=dev)
    assert L.nrf_pack_weights_bf16x3(wp, bp, P(px.data_ptr()), None) == 0
    p32 = torch.zeros(L.nrf_packed_floats(), dtype=torch.float32, device=dev)
    assert L.nrf_pack_weights(wp, bp, P(p32.data_ptr()), None) == 0
    pt = torch.zeros(L.nrf_packed_bwd_bf16x3_bytes(), dtype=torch.uint8, device=dev)
    assert L.nrf_pack_weights_transposed_bf16x3(wp, P(pt.data_ptr()), None) == 0
    if saved is None:
        na, nm = L.nrf_mlp_acts_bytes(C.c_int64(R), C.c_int(S)), L.nrf_mlp_masks_bytes(C.c_int64(R), C.c_int(S))
        raw = torch.empty(R, S, 4, device=dev)
        acts = torch.zeros(na // 4, device=dev)
        masks = torch.zeros(nm // 4, dtype=torch.int32, device=dev)
        assert L.nrf_mlp_forward_rays_train_bf16x3(P(px.data_ptr()), P(o.data_ptr()), P(d.data_ptr()), P(z.data_ptr()), C.c_int64(S),
                                                   C.c_int64(R), C.c_int(S), P(raw.data_ptr()), P(acts.data_ptr()), P(masks.data_ptr()), None) == 0
        saved = (acts, masks)
    acts, masks = saved
    nw = L.nrf_mlp_backward_workspace_bytes(C.c_int64(R), C.c_int(S))
    wsb = torch.zeros(nw // 4, device=dev)
    flat = torch.zeros(595844, device=dev)
    st = L.nrf_mlp_backward_bf16x3(P(p32.data_ptr()), P(pt.data_ptr()), P(g_raw.data_ptr()), P(acts.data_ptr()), P(masks.data_ptr()),
                                   C.c_int64(R), C.c_int(S), P(wsb.data_ptr()), C.c_int64(nw), P(flat.data_ptr()), None)
    assert st == 0, st
    torch.cuda.synchronize()
    tiles = R * S // 32
    res.append((wsb[:tiles * KDY * 32].view(tiles, KDY, 32).clone(), flat))
(da, fa), (db, fb) = res
for name, r0, r1 in ROWS:
    a, b = da[:, r0:r1], db[:, r0:r1]
    diff = (a - b).abs()
    bad = (diff > 1e-5 * float(a.abs().max())).nonzero()
    print(f"{name:8s} max |A| {float(a.abs().max()):.3e}  max |A-B| {float(diff.max()):.3e}  entries off by > 1e-5 of max: {bad.shape[0]} of {a.numel()}",
          ("first (tile, row, point): " + str(bad[:6].tolist())) if bad.shape[0] else "")
a, b = da[:, 2304:2432], db[:, 2304:2432]
bad = ((a - b).abs() > 1e-5 * float(a.abs().max())).nonzero()
if bad.shape[0]:
    rows = torch.bincount(bad[:, 1], minlength=128).tolist()
    print("dYv rows with differences (row: count):", {i: c for i, c in enumerate(rows) if c})
    print("dYv points with differences:", torch.bincount(bad[:, 2], minlength=32).tolist())
    print("tiles with differences:", int(torch.unique(bad[:, 0]).numel()), "first", torch.unique(bad[:, 0])[:12].tolist())
    va, vb = a[bad[:, 0], bad[:, 1], bad[:, 2]], b[bad[:, 0], bad[:, 1], bad[:, 2]]
    print("A zero there:", int((va == 0).sum()), " B zero there:", int((vb == 0).sum()), " examples A/B:", list(zip(va[:8].tolist(), vb[:8].tolist())))
    t0 = int(bad[0, 0])
    gw = g_raw.view(-1, 32, 4)[t0, :, 3]
    dd = (b - a)[t0, 94]
    print("tile", t0, "row 94: (B - A) / d sigma per point:", [round(float(x), 5) for x in (dd / gw)[16:].tolist()])
    G = g_raw.view(-1, 32, 4)[t0, 16:, :].double().cpu()
    sol = torch.linalg.lstsq(G, dd[16:].double().cpu().unsqueeze(1)).solution.squeeze(1)
    print("least squares (B - A) = G . x: x =", [round(float(v), 6) for v in sol], " residual", float((G @ sol - dd[16:].double().cpu()).abs().max()))
    print("rgb weights of feature 94:", [round(float(ws[11][c, 94]), 6) for c in range(3)], " of 78:", [round(float(ws[11][c, 78]), 6) for c in range(3)])
    print("alpha weights near feature 94:", [round(float(x), 5) for x in ws[9][0, 88:96].tolist()])
print("flat gradient: max |A| %.3e  max |A-B| %.3e" % (float(fa.abs().max()), float((fa - fb).abs().max())))
