"""Gradients of ONE step by the fused route (TrainStep) and by the autograd route (Renderer) on the same parameters and
batch: they run the same kernels and must agree bit for bit.  usage: x3_routes_compare.py [dtype]   (library: nerf_amd/lib)"""
import sys

import torch

from nerf_amd import ops
from nerf_amd.camera import focal_from_angle, pose_spherical
from nerf_amd.config import default_cfg
from nerf_amd.network import Network
from nerf_amd.renderer import Renderer
from nerf_amd.train import TrainStep
from nerf_amd.weights import make_state_dict

dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
dev = torch.device("cuda:0")
R, NC, NI, W = 256, 64, 128, 800
g = torch.Generator().manual_seed(77)
pix = (torch.randint(250, 550, (R,), generator=g) * W + torch.randint(250, 550, (R,), generator=g)).to(torch.int64)
rays_o, rays_d = ops.raygen(pose_spherical(45.0), W, W, focal_from_angle(W), dev, pixel_index=pix.to(dev))
gt = torch.rand(R, 3, generator=g).to(dev)
args = [torch.rand(R, n, generator=g).to(dev) for n in (NI, NI, NC, NI)]
sd0 = make_state_dict(0, 30.0, 2.0)
nets = []
for _ in range(2):
    n = Network()
    n.load_state_dict(sd0)
    nets.append(n.to(dev))
ts = TrainStep(nets[0], dev, forward_dtype=dtype)
for rep in range(2):
    ts.forward_backward(rays_o, rays_d, gt, *args)
    torch.cuda.synchronize()
    fused = ts.grad.clone()
    if rep == 0:
        first = fused
print("fused route run-to-run identical:", bool(torch.equal(first, fused)))
r = Renderer(nets[1], default_cfg(perturb=1), mlp_dtype=dtype)
out = r.render_rays_train(rays_o, rays_d, *args)
loss = ((out["coarse_rgb_map"] - gt) ** 2).mean() + ((out["fine_rgb_map"] - gt) ** 2).mean()
loss.backward()
auto = torch.cat([p.grad.reshape(-1) for p in r._model_params("model") + r._model_params("model_fine")])
print("loss fused / autograd:", float(ts.loss), float(loss))
d = (auto - fused).abs()
print("autograd vs fused gradients: identical", bool(torch.equal(auto, fused)), " max |d|", float(d.max()), " max |g|", float(fused.abs().max()),
      " entries that differ", int((d > 0).sum()), "of", d.numel())
