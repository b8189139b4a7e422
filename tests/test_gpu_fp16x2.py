"""GPU parity of the fp16x2 MLP (nrf_pack_weights_fp16x2 / nrf_mlp_forward_rays_fp16x2): fp32 operands split into two
fp16 terms (22 significand bits), three MFMAs per product, fp32 accumulation.  The claim is "fp32-class", so the bars are
the fp32 kernel's own (tests/test_gpu_parity.py): raw outputs as close to a float64 evaluation as the fp32 REFERENCE
fixture is (x3 slack) or within 2e-5 of it outright, and rgb / depth / acc end to end within 1e-4 abs of the reference
fixtures.  Beyond fp16's range the kernel must fail loudly (non-finite output), never quietly."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

FWD = ["forward_plumbing64", "forward_camera64", "forward_dense32", "forward_empty16"]


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _sd(fx, arrays=False):
    from nerf_amd.weights import make_state_dict, make_state_dict_arrays
    f = make_state_dict_arrays if arrays else make_state_dict
    return f(int(fx["weight_seed"]), float(fx["density_gain"]), float(fx["density_bias"]))


def _f64(fx, model, z):
    from oracle import nerf_oracle as O
    zt = torch.from_numpy(np.ascontiguousarray(np.broadcast_to(z, (fx["xyz"].shape[0], z.shape[-1]))))
    pts, dhat = O.points(torch.from_numpy(fx["xyz"]), torch.from_numpy(fx["viewdirs"]), zt)
    return (O.network_forward(_sd(fx), pts, dhat, model, dtype=torch.float64).numpy(),
            O.network_forward(_sd(fx), pts, dhat, model, split2_f16=True).numpy())


@pytest.mark.parametrize("name", FWD)
def test_mlp_fp16x2_is_fp32_class(golden, dev, name):
    from nerf_amd import ops
    fx = golden(name)
    o, d = _t(fx["xyz"], dev), _t(fx["viewdirs"], dev)
    for prefix, model, zkey, refkey in (("model", "", "t_bins", "raw_coarse"), ("model_fine", "fine", "z_fine", "raw_fine")):
        raw = ops.mlp_forward_rays_fp16x2(ops.pack_state_dict_fp16x2(_sd(fx, True), prefix, dev), o, d, _t(fx[zkey], dev)).cpu().numpy()
        f64, emu = _f64(fx, model, fx[zkey])
        ref = fx[refkey]
        assert np.isfinite(raw).all()
        err_ref, err_gpu, err_emu = (float(np.abs(x - f64).max()) for x in (ref, raw, emu))
        rms_ref, rms_gpu = (float(np.sqrt(np.mean((x - f64) ** 2))) for x in (ref, raw))
        print(f"{name} {prefix}: |ref-f64| {err_ref:.2e} (rms {rms_ref:.2e})  |fp16x2-f64| {err_gpu:.2e} (rms {rms_gpu:.2e})  "
              f"|oracle split2_f16 - f64| {err_emu:.2e}")
        assert err_gpu <= max(3.0 * err_ref, 2e-5 * max(1.0, float(np.abs(ref).max()))), (err_gpu, err_ref)
        assert rms_gpu <= 1.5 * rms_ref, (rms_gpu, rms_ref)


@pytest.mark.parametrize("name", FWD)
def test_render_fp16x2_end_to_end(golden, dev, name):
    from nerf_amd.config import default_cfg
    from nerf_amd.network import Network
    from nerf_amd.renderer import Renderer
    fx = golden(name)
    net = Network()
    net.load_state_dict(_sd(fx))
    r = Renderer(net.to(dev), default_cfg(perturb=0), mlp_dtype="fp16x2")
    r._bins = _t(fx["t_bins"], dev)
    args = (_t(fx["xyz"], dev), _t(fx["viewdirs"], dev), _t(fx["u"], dev), _t(fx["u2"], dev))
    with torch.no_grad():
        out = r.render_rays(*args)
        staged = r.render_rays_staged(*args)
    for k in ("coarse_rgb_map", "fine_rgb_map", "alpha_map", "coarse_depth_map", "fine_depth_map"):
        err = float(np.abs(out[k].cpu().numpy() - fx[k]).max())
        assert err < 1e-4, (k, err)                                            # the north star's fp32 tolerance
        assert torch.equal(out[k], staged[k]), k


def test_fp16x2_points_and_ragged(dev):
    """Flat batches of any size (Network.forward, network.py:184-186) and the fp32 kernel as the yardstick."""
    from nerf_amd import ops
    from nerf_amd.weights import make_state_dict_arrays
    sd = make_state_dict_arrays(2, 30.0, 1.0)
    ph, pf = ops.pack_state_dict_fp16x2(sd, "model", dev), ops.pack_state_dict(sd, "model", dev)
    rng = np.random.default_rng(0)
    for n in (1, 31, 32, 33, 1000, 70001):
        pts = _t(rng.uniform(-1.2, 1.2, size=(n, 3)).astype(np.float32), dev)
        dirs = _t(rng.normal(size=(n, 3)).astype(np.float32), dev)
        a, b = ops.mlp_forward_points_fp16x2(ph, pts, dirs), ops.mlp_forward_points(pf, pts, dirs)
        assert a.shape == (n, 4) and torch.isfinite(a).all()
        assert float((a - b).abs().max()) < 2e-5 * max(1.0, float(b.abs().max())), n
    L = ops.lib()
    assert L.nrf_mlp_forward_rays_fp16x2(None, None, None, None, 0, 4, 48, None, None) == -2     # 48 % 32 != 0
    assert L.nrf_mlp_forward_rays_fp16x2(None, None, None, None, 0, 0, 64, None, None) == 0      # empty batch
    assert L.nrf_mlp_forward_rays_fp16x2(None, None, None, None, 0, 4, 64, None, None) == -1


def test_fp16x2_small_and_large_operands(dev):
    """The split keeps its precision over fp16's whole normal range because the low term is stored scaled by 2^11:
    a network scaled down (weights x 2^-6 in one layer, x 2^6 in the next: the same function) must agree with the
    fp32 kernel as well as the unscaled one does; a weight beyond fp16's range must give a non-finite output."""
    from nerf_amd import ops
    from nerf_amd.weights import make_state_dict_arrays
    rng = np.random.default_rng(1)
    pts = _t(rng.uniform(-1.2, 1.2, size=(4096, 3)).astype(np.float32), dev)
    dirs = _t(rng.normal(size=(4096, 3)).astype(np.float32), dev)
    sd = make_state_dict_arrays(2, 30.0, 1.0)
    base = ops.mlp_forward_points(ops.pack_state_dict(sd, "model", dev), pts, dirs)
    scaled = dict(sd)
    for name, s in (("pts_linears.2", 2.0 ** -10), ("pts_linears.3", 2.0 ** 10)):
        scaled[f"model.{name}.weight"] = sd[f"model.{name}.weight"] * np.float32(s)
    # relu(s * x) = s * relu(x): scale pts_linears.2's bias with it, and pts_linears.3 undoes the factor exactly
    scaled["model.pts_linears.2.bias"] = sd["model.pts_linears.2.bias"] * np.float32(2.0 ** -10)
    out = ops.mlp_forward_points_fp16x2(ops.pack_state_dict_fp16x2(scaled, "model", dev), pts, dirs)
    err = float((out - base).abs().max())
    print(f"fp16x2 with a 2^-10-scaled hidden layer vs fp32 kernel: {err:.2e}")
    assert err < 2e-5 * max(1.0, float(base.abs().max()))
    big = dict(sd)
    w = sd["model.pts_linears.4.weight"].copy()
    w[3, 5] = 7.0e4                                   # outside fp16: hi term = inf
    big["model.pts_linears.4.weight"] = w
    from nerf_amd._lib import NrfError
    with pytest.raises(NrfError, match="fp16's range"):
        ops.pack_state_dict_fp16x2(big, "model", dev)
    out = ops.mlp_forward_points_fp16x2(ops.pack_state_dict_fp16x2(big, "model", dev, check_range=False), pts, dirs)
    assert torch.isnan(out).all()                     # the kernel refuses the whole network
    # activations beyond the range poison their own point only: one huge coordinate in a batch of ordinary ones
    far = pts.clone()
    far[17, 0] = 1.0e5
    ph = ops.pack_state_dict_fp16x2(sd, "model", dev)
    out = ops.mlp_forward_points_fp16x2(ph, far, dirs)
    ok = torch.ones(far.shape[0], dtype=torch.bool, device=dev)
    ok[17] = False
    assert torch.isnan(out[17]).all() and torch.isfinite(out[ok]).all()
    assert torch.equal(out[ok], ops.mlp_forward_points_fp16x2(ph, pts, dirs)[ok])
    # ... and so does a hidden activation beyond it (h2 scaled by 2^20 through two layers, undone by the third)
    hot = dict(sd)
    f = lambda e: np.float32(2.0 ** e)
    hot["model.pts_linears.1.weight"] = sd["model.pts_linears.1.weight"] * f(6)
    hot["model.pts_linears.1.bias"] = sd["model.pts_linears.1.bias"] * f(6)
    hot["model.pts_linears.2.weight"] = sd["model.pts_linears.2.weight"] * f(14)
    hot["model.pts_linears.2.bias"] = sd["model.pts_linears.2.bias"] * f(20)
    hot["model.pts_linears.3.weight"] = sd["model.pts_linears.3.weight"] * f(-20)
    out = ops.mlp_forward_points_fp16x2(ops.pack_state_dict_fp16x2(hot, "model", dev), pts, dirs)
    h2 = float(torch.isnan(out).all(-1).float().mean())
    print(f"hidden activations scaled by 2^20: {h2:.3f} of the points refused (NaN), the rest finite: "
          f"{bool(torch.isfinite(out[~torch.isnan(out).any(-1)]).all())}")
    assert h2 > 0.5 and bool((torch.isnan(out).all(-1) | torch.isfinite(out).all(-1)).all())
