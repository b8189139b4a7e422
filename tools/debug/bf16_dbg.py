import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from nerf_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = sys.argv[1]
from nerf_amd import ops
from nerf_amd.weights import make_state_dict, make_state_dict_arrays
from nerf_amd.renderer import near_far_table
from oracle import nerf_oracle as O
dev = torch.device("cuda:0")
sda = make_state_dict_arrays(0, 30.0, 2.0)
sd = make_state_dict(0, 30.0, 2.0)
pb = ops.pack_state_dict_bf16(sda, "model", dev)
pf = ops.pack_state_dict(sda, "model", dev)
g = torch.Generator().manual_seed(0)
R = 300
o = torch.tensor([0.0, 0.0, 4.0]) + 0.1 * torch.randn(R, 3, generator=g)
d = torch.randn(R, 3, generator=g); d[:, 2] = -d[:, 2].abs() - 1
for S in (64,):
    z = torch.sort(2 + 4 * torch.rand(R, S, generator=g), -1).values
    raw = ops.mlp_forward_rays_bf16(pb, o.to(dev), d.to(dev), z.to(dev)).cpu()
    raw32 = ops.mlp_forward_rays(pf, o.to(dev), d.to(dev), z.to(dev)).cpu()
    pts, dhat = O.points(o, d, z)
    ref_b = O.network_forward(sd, pts, dhat, bf16=True)
    ref = O.network_forward(sd, pts, dhat)
    print(S, "finite", bool(torch.isfinite(raw).all()),
          "| gpu-bf16 vs oracle-bf16: rgb", float((raw[..., :3] - ref_b[..., :3]).abs().max()), "sigma", float((raw[..., 3] - ref_b[..., 3]).abs().max()),
          "| gpu-bf16 vs fp32 ref: rgb", float((raw[..., :3] - ref[..., :3]).abs().max()), "sigma", float((raw[..., 3] - ref[..., 3]).abs().max()),
          "| oracle-bf16 vs fp32 ref: rgb", float((ref_b[..., :3] - ref[..., :3]).abs().max()), "sigma", float((ref_b[..., 3] - ref[..., 3]).abs().max()),
          "| fp32 gpu vs ref", float((raw32 - ref).abs().max()), flush=True)
