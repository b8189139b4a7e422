"""GPU parity of the training step's backward kernels against the reference's autograd results
(tests/golden/train_camera32.npz) and against torch autograd on the CPU oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def test_composite_backward_vs_oracle_autograd(golden, dev):
    from nerf_amd import ops
    from oracle import nerf_oracle as O
    for name, zkey, rkey in (("forward_camera64", "z_fine", "raw_fine"), ("forward_dense32", "t_bins", "raw_coarse"),
                             ("forward_plumbing64", "z_fine", "raw_fine")):
        fx = golden(name)
        raw = torch.from_numpy(fx[rkey]).clone().requires_grad_(True)
        R, S = raw.shape[0], raw.shape[1]
        z = torch.from_numpy(np.ascontiguousarray(np.broadcast_to(fx[zkey], (R, S))))
        rgb, depth, w, acc = O.composite(raw, z, True)
        g = torch.Generator().manual_seed(1)
        g_rgb, g_depth, g_acc = torch.randn(R, 3, generator=g), torch.randn(R, generator=g), torch.randn(R, 1, generator=g)
        ((rgb * g_rgb).sum() + (depth * g_depth).sum() + (acc * g_acc).sum()).backward()
        got = ops.composite_backward(_t(fx[rkey], dev), _t(z.numpy(), dev), True, g_rgb.to(dev), g_depth.to(dev),
                                     g_acc.to(dev)).cpu().numpy()
        ref = raw.grad.numpy()
        assert np.isfinite(got).all()
        assert np.abs(got - ref).max() < 2e-5 * max(1.0, np.abs(ref).max()), name


def _setup(fx, dev, mlp_dtype="fp32"):
    from nerf_amd.network import Network
    from nerf_amd.renderer import Renderer
    from nerf_amd.weights import make_state_dict
    from nerf_amd.config import default_cfg
    net = Network()
    net.load_state_dict(make_state_dict(int(fx["weight_seed"]), float(fx["density_gain"]), float(fx["density_bias"])))
    net.to(dev)
    r = Renderer(net, default_cfg(perturb=1), mlp_dtype=mlp_dtype)
    r._bins = _t(fx["t_bins"], dev)
    return net, r


@pytest.mark.parametrize("mlp_dtype", ["fp32", "bf16x3", "fp16x2"])
def test_training_step_gradients_match_reference(golden, dev, mlp_dtype):
    """Gradients of the REFERENCE's own training step (tests/golden/train_camera32.npz: autograd through the real
    reference renderer) through Renderer's autograd route -- on the fp32 MFMA kernels and on the three-term bf16 kernels
    (forward, data-gradient chain and weight-gradient GEMMs), same bars."""
    fx = golden("train_camera32")
    net, r = _setup(fx, dev, mlp_dtype)
    out = r.render_rays_train(_t(fx["xyz"], dev), _t(fx["viewdirs"], dev), _t(fx["u"], dev), _t(fx["u2"], dev),
                              _t(fx["u_coarse"], dev), _t(fx["u_fine"], dev))
    gt = _t(fx["gt_rgb"], dev)
    mse = torch.nn.MSELoss(reduction="mean")
    loss = mse(out["coarse_rgb_map"], gt) + mse(out["fine_rgb_map"], gt)       # trainers/nerf.py:25-27
    assert abs(float(loss) - float(fx["loss"])) < 1e-5
    loss.backward()
    named = dict(net.named_parameters())
    for n, ref in zip(list(fx["grad_names"]), fx["grad_norms"]):
        got = float(named[n].grad.double().norm())
        assert abs(got - ref) <= 2e-3 * max(ref, 1e-7), (n, got, ref)
    for pre in ("model", "model_fine"):
        m = getattr(net, pre)
        checks = {
            f"g.{pre}.alpha_linear.weight": m.alpha_linear.weight.grad,
            f"g.{pre}.alpha_linear.bias": m.alpha_linear.bias.grad,
            f"g.{pre}.rgb_linear.weight": m.rgb_linear.weight.grad,
            f"g.{pre}.rgb_linear.bias": m.rgb_linear.bias.grad,
            f"g.{pre}.pts_linears.0.bias": m.pts_linears[0].bias.grad,
            f"g.{pre}.pts_linears.0.weight": m.pts_linears[0].weight.grad,
            f"g.{pre}.pts_linears.5.weight[:, :72]": m.pts_linears[5].weight.grad[:, :72],
            f"g.{pre}.pts_linears.7.weight[:8]": m.pts_linears[7].weight.grad[:8],
            f"g.{pre}.views_linears.0.weight[:, 248:]": m.views_linears[0].weight.grad[:, 248:],
            f"g.{pre}.feature_linear.bias": m.feature_linear.bias.grad,
        }
        for k, got in checks.items():
            ref = fx[k]
            err = float(np.abs(got.cpu().numpy() - ref).max())
            assert err <= 2e-4 * max(float(np.abs(ref).max()), 1e-7), (k, err, float(np.abs(ref).max()))


def test_renderer_forward_under_autograd_matches_inference(golden, dev):
    fx = golden("train_camera32")
    net, r = _setup(fx, dev)
    args = (_t(fx["xyz"], dev), _t(fx["viewdirs"], dev), _t(fx["u"], dev), _t(fx["u2"], dev),
            _t(fx["u_coarse"], dev), _t(fx["u_fine"], dev))
    a = r.render_rays_train(*args)
    with torch.no_grad():
        b = r.render_rays(*args)
    for k in ("coarse_rgb_map", "fine_rgb_map", "fine_depth_map", "alpha_map"):
        assert torch.equal(a[k].detach(), b[k]), k
    assert a["fine_rgb_map"].requires_grad


def test_fused_train_step_matches_reference_adam(golden, dev):
    """TrainStep (flat buffers, fused clip + Adam) reproduces the reference's loss, gradients and the
    parameters after one torch.optim.Adam step (tests/golden/train_camera32.npz `adam.*`)."""
    from nerf_amd.network import Network
    from nerf_amd.train import TrainStep
    from nerf_amd.weights import make_state_dict
    fx = golden("train_camera32")
    net = Network()
    net.load_state_dict(make_state_dict(int(fx["weight_seed"]), float(fx["density_gain"]), float(fx["density_bias"])))
    net.to(dev)
    ts = TrainStep(net, dev)
    ts.bins = _t(fx["t_bins"], dev)
    ts.step(_t(fx["xyz"], dev), _t(fx["viewdirs"], dev), _t(fx["gt_rgb"], dev), _t(fx["u"], dev), _t(fx["u2"], dev),
            _t(fx["u_coarse"], dev), _t(fx["u_fine"], dev))
    assert abs(float(ts.loss) - float(fx["loss"])) < 1e-5
    for pre in ("model", "model_fine"):
        m = getattr(net, pre)
        for key, p in ((f"adam.{pre}.alpha_linear.weight", m.alpha_linear.weight),
                       (f"adam.{pre}.pts_linears.0.bias", m.pts_linears[0].bias),
                       (f"adam.{pre}.rgb_linear.weight", m.rgb_linear.weight)):
            # first Adam step moves every parameter by ~lr * sign(g): compare the MOVE, not the value
            err = np.abs(p.detach().cpu().numpy() - fx[key]).max()
            assert err < 2e-5, (key, float(err))      # lr = 5e-4: 4 % of one step where g ~ eps-noise, else exact


def test_clip_adam_matches_torch_over_steps(dev):
    from nerf_amd import ops
    g = torch.Generator(device="cpu").manual_seed(0)
    n = 100003
    p0 = torch.randn(n, generator=g)
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=5e-4, eps=1e-8, weight_decay=0.0)
    p = p0.clone().to(dev)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for step in range(1, 6):
        grad = torch.randn(n, generator=g) * (100.0 if step == 2 else 1.0)     # step 2 exercises the clip at 40
        ref.grad = grad.clone()
        torch.nn.utils.clip_grad_value_([ref], 40)
        opt.step()
        ops.clip_adam(p, grad.to(dev), m, v, step)
        assert np.abs(p.cpu().numpy() - ref.detach().numpy()).max() < 2e-6, step


def test_fused_step_optimizer_state_round_trips_through_torch_adam(golden, dev):
    """Checkpoint interop of the fused step (net_utils.py:288-340: 'optim' = torch.optim.Adam.state_dict() over
    net.parameters()): after three fused steps the exported state, loaded into a REAL torch.optim.Adam over a twin
    network, makes torch's fourth step land where the fused fourth step lands; and the state loads back bit for bit."""
    from nerf_amd.network import Network
    from nerf_amd.train import TrainStep
    from nerf_amd.weights import make_state_dict
    fx = golden("train_camera32")
    sd0 = make_state_dict(int(fx["weight_seed"]), float(fx["density_gain"]), float(fx["density_bias"]))
    net = Network()
    net.load_state_dict(sd0)
    ts = TrainStep(net.to(dev), dev)
    ts.bins = _t(fx["t_bins"], dev)
    args = [_t(fx[k], dev) for k in ("xyz", "viewdirs", "gt_rgb", "u", "u2", "u_coarse", "u_fine")]
    for _ in range(3):
        ts.step(*args)
    state = ts.optimizer_state_dict()
    assert len(state["param_groups"]) == 1 and len(state["param_groups"][0]["params"]) == 48 and len(state["state"]) == 48
    assert all(float(s["step"]) == 3.0 for s in state["state"].values())
    twin = Network()
    twin.load_state_dict({k: v.detach().cpu().clone() for k, v in net.state_dict().items()})
    twin.to(dev)
    opt = torch.optim.Adam(twin.parameters(), lr=5e-4, eps=1e-8)
    opt.load_state_dict(state)                                   # the reference's resume (load_model)
    ts.forward_backward(*args)                                   # gradients of step 4 at the current weights
    flat = {p.data_ptr(): ts.grad[(p.data_ptr() - ts.flat.data_ptr()) // 4:][:p.numel()].view_as(p) for p in net.parameters()}
    for p, q in zip(net.parameters(), twin.parameters()):
        q.grad = flat[p.data_ptr()].clone()
    torch.nn.utils.clip_grad_value_(list(twin.parameters()), 40.0)
    opt.step()
    ts.step(*args)
    for (name, p), q in zip(net.named_parameters(), twin.parameters()):
        assert float((p - q).abs().max()) < 2e-6, name           # the bar of test_clip_adam_matches_torch_over_steps
    # ... and back: a fresh fused step resumes from the torch optimiser's state, bit for bit
    net2 = Network()
    net2.load_state_dict(sd0)
    ts2 = TrainStep(net2.to(dev), dev)
    ts2.load_optimizer_state_dict(opt.state_dict())
    state4 = ts.optimizer_state_dict()
    back = ts2.optimizer_state_dict()
    assert ts2.t == 4 and ts.t == 4
    for i in range(48):
        for k in ("exp_avg", "exp_avg_sq"):
            assert torch.equal(back["state"][i][k], opt.state_dict()["state"][i][k]), (i, k)
            assert float((state4["state"][i][k] - back["state"][i][k]).abs().max()) <= 1e-5 * float(state4["state"][i][k].abs().max() + 1e-30), (i, k)   # fused kernel vs torch: another association of the same products
