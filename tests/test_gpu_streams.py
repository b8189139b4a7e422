"""The C ABI enqueues on the stream it is given (include/nerf_amd.h, Conventions); the Python layer passes torch's CURRENT
stream.  The reference runs everything on the default stream, so nothing else exercises a side stream: here the
render, the fused training step and the KiloNeRF frame run inside ``torch.cuda.stream(side)`` -- inputs produced on that
stream too, nothing synchronised in between -- and must give the bits of the default-stream run.  A kernel or memset issued
on the wrong stream would race with its producers and show up as a difference (or a fault)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _net(dev, seed=0):
    from nerf_amd.network import Network
    from nerf_amd.weights import make_state_dict
    net = Network()
    net.load_state_dict(make_state_dict(seed, 30.0, 2.0))
    return net.to(dev)


def _inputs(dev, n, seed):
    from nerf_amd import ops
    from nerf_amd.camera import focal_from_angle, pose_spherical
    g = torch.Generator(device=dev).manual_seed(seed)
    pix = torch.randint(0, 800 * 800, (n,), device=dev, generator=g)
    o, d = ops.raygen(pose_spherical(10.0), 800, 800, focal_from_angle(800), dev, pixel_index=pix)
    u = [torch.rand((n, k), device=dev, generator=g) for k in (128, 128, 64, 128)]
    gt = torch.rand((n, 3), device=dev, generator=g)
    return o, d, u, gt


@pytest.mark.parametrize("dtype", ["fp32", "bf16x3", "bf16"])
def test_render_on_a_side_stream(dtype):
    from nerf_amd.config import default_cfg
    from nerf_amd.renderer import Renderer
    dev = torch.device("cuda:0")
    r = Renderer(_net(dev), default_cfg(perturb=0), mlp_dtype=dtype)
    o, d, u, _ = _inputs(dev, 3000, 1)
    with torch.no_grad():
        want = r.render_rays(o, d, u[0], u[1])
    torch.cuda.synchronize()
    side = torch.cuda.Stream(dev)
    with torch.cuda.stream(side), torch.no_grad():
        o2, d2, u2, _ = _inputs(dev, 3000, 1)                # produced on the side stream, consumed right away
        r2 = Renderer(_net(dev), default_cfg(perturb=0), mlp_dtype=dtype)    # packs its weight streams on the side stream
        got = r2.render_rays(o2, d2, u2[0], u2[1])
    side.synchronize()
    for k in ("coarse_rgb_map", "fine_rgb_map", "fine_depth_map", "alpha_map"):
        assert torch.equal(got[k], want[k]), k


def test_training_step_on_a_side_stream():
    from nerf_amd.train import TrainStep
    dev = torch.device("cuda:0")
    o, d, u, gt = _inputs(dev, 512, 2)
    ts = TrainStep(_net(dev), dev, forward_dtype="bf16x3")
    for _ in range(2):
        ts.step(o, d, gt, u[0], u[1], u[2], u[3])
    torch.cuda.synchronize()
    want, loss = ts.flat.clone(), float(ts.loss)
    side = torch.cuda.Stream(dev)
    with torch.cuda.stream(side):
        o2, d2, u2, gt2 = _inputs(dev, 512, 2)
        ts2 = TrainStep(_net(dev), dev, forward_dtype="bf16x3")
        for _ in range(2):
            ts2.step(o2, d2, gt2, u2[0], u2[1], u2[2], u2[3])
    side.synchronize()
    assert torch.equal(ts2.flat, want)                          # parameters after two steps: bit for bit
    # (the loss word is summed with one float atomic per wavefront, nrf_mse_grad: the order may differ in the last bit)
    assert abs(float(ts2.loss) - loss) <= 1e-6 * max(1.0, abs(loss))


def test_kilonerf_frame_and_training_step_on_a_side_stream():
    from nerf_amd.kilo_train import KiloTrainStep
    from nerf_amd.kilonerf import KiloNetwork, KiloRenderer, default_kilo_cfg
    from nerf_amd.weights import make_kilo_state_dict_arrays
    dev = torch.device("cuda:0")
    sd = {k: torch.from_numpy(v) for k, v in make_kilo_state_dict_arrays(0, 4096, 10.0, 0.5).items()}
    c = (torch.arange(256, device=dev, dtype=torch.float32) + 0.5) / 256 * 2 - 1
    occ = ((c[:, None, None] ** 2 + c[None, :, None] ** 2 + c[None, None, :] ** 2).sqrt() < 0.75).to(torch.uint8)

    def run():
        net = KiloNetwork()
        net.load_state_dict(sd, strict=False)
        r = KiloRenderer(net.to(dev), default_kilo_cfg(), occ)
        o, d, _, gt = _inputs(dev, 2048, 3)
        with torch.no_grad():
            out = r({"xyz": o[None], "viewdirs": d[None]}, is_training=False)
        step = KiloTrainStep(r, dev, l2_reg=1e-6)
        step.step(o, d, gt, gt[:, :1].contiguous(), stage="distillation")
        return out["rgb_map"].clone(), float(step.loss), step.num_samples

    rgb, loss, n = run()
    torch.cuda.synchronize()
    side = torch.cuda.Stream(dev)
    with torch.cuda.stream(side):
        rgb2, loss2, n2 = run()
    side.synchronize()
    assert n2 == n and n > 0 and torch.equal(rgb2, rgb)
    assert abs(loss2 - loss) <= 1e-6 * max(1.0, abs(loss))     # (the loss word is summed with float atomics: order may differ)
