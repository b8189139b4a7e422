"""Per-tensor comparison of the 4096-ray training gradients: backward_dtype fp16x2 against bf16x3 (same fp16x2 forward)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from nerf_amd import ops
from nerf_amd.camera import pose_spherical, focal_from_angle
from nerf_amd.network import Network
from nerf_amd.train import TrainStep
from nerf_amd.weights import make_state_dict

dev = torch.device("cuda:0")
res = {}
for bd in ("bf16x3", "fp16x2"):
    net = Network()
    net.load_state_dict(make_state_dict(0, 30.0, 2.0))
    ts = TrainStep(net.to(dev), dev, forward_dtype="fp16x2", backward_dtype=bd)
    R = 4096
    g = torch.Generator(device=dev).manual_seed(0)
    pix = torch.randint(0, 640000, (R,), device=dev, generator=g)
    o, d = ops.raygen(pose_spherical(0.0), 800, 800, focal_from_angle(800), dev, pixel_index=pix)
    rnd = lambda n: torch.rand((R, n), device=dev, generator=g)  # noqa: E731
    gt, u, u2, uc, uf = rnd(3), rnd(128), rnd(128), rnd(64), rnd(128)
    ts.forward_backward(o, d, gt, u, u2, uc, uf)
    torch.cuda.synchronize()
    res[bd] = ts.grad.clone()
n = ops.model_param_floats()
for k, name in enumerate(("model", "model_fine")):
    a = ops.split_flat_grad(res["bf16x3"][k * n:(k + 1) * n])
    b = ops.split_flat_grad(res["fp16x2"][k * n:(k + 1) * n])
    for ln, (aw, ab), (bw, bb) in zip(ops.LAYER_NAMES, a, b):
        ew = float((aw - bw).abs().max() / aw.abs().max().clamp_min(1e-30))
        eb = float((ab - bb).abs().max() / ab.abs().max().clamp_min(1e-30))
        print(f"{name}.{ln:18s} w rel {ew:.3e} finite {bool(torch.isfinite(bw).all())}   b rel {eb:.3e} finite {bool(torch.isfinite(bb).all())}")
