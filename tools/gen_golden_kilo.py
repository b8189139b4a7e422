#!/usr/bin/env python3
"""Golden fixtures for the KiloNeRF rows from the REAL reference (build container only).

Same recipe as tools/gen_golden.py with configs/kilonerf/lego.yaml (one config per process: the
reference's cfg is a global built at import).  Covers what is pinned: cell routing
(kilonerf/network.py:246-253, 314-334), the batched tiny MLPs (:123-167, :223-311) and the
colour regulariser (:357-384).  Sampling / packed rendering live inside nerfacc (absent, version
unpinned) and have no fixture.
"""
import os
import sys
import types

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True
os.chdir(REF)
sys.path.insert(0, REF)
sys.argv = ["gen_golden_kilo", "--cfg_file", "configs/kilonerf/lego.yaml"]
sys.modules["nerfacc"] = types.ModuleType("nerfacc")

import numpy as np  # noqa: E402
import torch  # noqa: E402

from src.config import cfg  # noqa: E402,F401
from src.models.kilonerf.network import Network  # noqa: E402

from nerf_amd.weights import make_kilo_state_dict_arrays  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")

net = Network()
sd = {k: torch.from_numpy(v) for k, v in make_kilo_state_dict_arrays(7, 4096, 20.0, 1.0).items()}
missing = net.load_state_dict(sd, strict=False)
assert set(missing.missing_keys) <= {"aabb_min", "aabb_max", "grid_resolution"}, missing
net.eval()

g = torch.Generator().manual_seed(31)
pts = (torch.rand(4000, 3, generator=g) * 2.6 - 1.3).float()          # inside and outside the [-1,1]^3 box
special = torch.tensor([[-1, -1, -1], [1, 1, 1], [0, 0, 0], [0.999999, -0.999999, 0.125], [-1.5, 2.0, 0.3],
                        [0.125, 0.25, 0.375], [-0.875, 0.875, -0.0625], [1.0, -1.0, 1.0],
                        [np.nextafter(np.float32(0.125), np.float32(0)), 0.125, np.nextafter(np.float32(0.125), np.float32(1))]],
                       dtype=torch.float32)
# points on / next to every cell boundary along x
edges = torch.linspace(-1, 1, 17)
near_edges = torch.stack([torch.cat([edges, torch.nextafter(edges, torch.tensor(2.0)), torch.nextafter(edges, torch.tensor(-2.0))]),
                          torch.zeros(51), torch.full((51,), 0.3)], -1).float()
pts = torch.cat([pts, special, near_edges])
with torch.no_grad():
    idx3 = net._get_index(pts)
    flat = idx3[:, 0] * 256 + idx3[:, 1] * 16 + idx3[:, 2]
np.savez_compressed(os.path.join(OUT, "kilo_route.npz"), pts=pts.numpy(), index3=idx3.numpy(), flat=flat.numpy())
print("kilo_route", pts.shape, int(flat.min()), int(flat.max()))

P = 768
p2 = (torch.rand(P, 3, generator=g) * 2 - 1).float()
p2[:256] = p2[0] * 0.01 + torch.tensor([0.3, -0.4, 0.55]) + 0.05 * torch.rand(256, 3, generator=g)   # many points in few cells
d2 = torch.randn(P, 3, generator=g).float()
with torch.no_grad():
    rgb, sigma = net(p2, d2)                       # (P,1,3), (P,1,1)
    R, S = 24, 32
    p3 = p2[: R * S].view(R, S, 3)
    rgb3, sigma3 = net(p3, d2[:R])                 # dirs per ray
    reg = net.get_color_reg_loss()
np.savez_compressed(os.path.join(OUT, "kilo_network.npz"), pts=p2.numpy(), dirs=d2.numpy(), rgb=rgb.numpy(),
                    sigma=sigma.numpy(), rgb_rays=rgb3.numpy(), sigma_rays=sigma3.numpy(), color_reg_loss=float(reg),
                    weight_seed=7, density_gain=20.0, density_bias=1.0)
print("kilo_network", rgb.shape, sigma.shape, float(reg), float(sigma.mean()))

# ---- parameter gradients of the tiny MLPs (autograd through BatchedKiloNeRF.forward, network.py:123-167) ----
# loss = sum(cat([rgb, sigma]) * G) for a fixed upstream gradient G, so param.grad = (d raw / d param)^T G.
# Stored: the full gradient rows of a few cells (the crowded ones + some with a single point) and float64
# sums / absolute sums per tensor over ALL cells.
net.zero_grad()
for p_ in net.parameters():
    p_.requires_grad_(True)
G = torch.randn(P, 4, generator=g).float()
rgb_g, sigma_g = net(p2, d2)
loss = (torch.cat([rgb_g.reshape(P, 3), sigma_g.reshape(P, 1)], -1) * G).sum()
loss.backward()
with torch.no_grad():
    i3 = net._get_index(p2)
    flat2 = (i3[:, 0] * 256 + i3[:, 1] * 16 + i3[:, 2]).numpy()
cells, cnt = np.unique(flat2, return_counts=True)
order = np.argsort(-cnt)
pick = np.concatenate([cells[order[:4]], cells[order[-4:]]]).astype(np.int64)
from nerf_amd.weights import KILO_TENSORS  # noqa: E402
named = dict(net.named_parameters())
out = {"grad_raw": G.numpy(), "cells": pick, "cell_counts": np.array([int((flat2 == c).sum()) for c in pick])}
for name, _, _ in KILO_TENSORS:
    gr = named["model." + name].grad
    out["grad." + name] = gr[pick].numpy()
    out["sum." + name] = np.float64(gr.double().sum())
    out["abssum." + name] = np.float64(gr.double().abs().sum())
np.savez_compressed(os.path.join(OUT, "kilo_network_grad.npz"), **out)
print("kilo_network_grad cells", pick, out["cell_counts"], {k: float(v) for k, v in out.items() if k.startswith("abssum.")})

# ---- non-finite coordinates through the reference's _get_index (network.py:314-334) ----
# +-inf survive the clamp (:329-331: clamp(min=0) then min(res - 1)); NaN passes both clamps unchanged and `.long()` of
# NaN is whatever the CPU's cvttss2si returns (INT64_MIN on x86), which the reference then uses as an index and fails.
# The fixture records exactly what the reference returned here, so the HIP kernel's choices are pinned where the
# reference has a defined answer (+-inf) and documented where it has none (NaN -> cell 0 in kilo_route_kernel).
inf = float("inf")
nf = torch.tensor([[inf, 0.0, 0.0], [-inf, 0.0, 0.0], [0.3, inf, -inf], [inf, inf, inf], [-inf, -inf, -inf],
                   [0.0, -0.2, inf], [float("nan"), 0.0, 0.0], [0.5, float("nan"), float("nan")]], dtype=torch.float32)
with torch.no_grad():
    nf3 = net._get_index(nf)
np.savez_compressed(os.path.join(OUT, "kilo_route_nonfinite.npz"), pts=nf.numpy(), index3=nf3.numpy())
print("kilo_route_nonfinite", nf3.tolist())
