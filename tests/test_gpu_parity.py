"""GPU parity: every C-ABI entry point against the golden fixtures generated from the real
reference (tests/golden, tools/gen_golden.py) and against the CPU oracle on fresh inputs.

Tolerances: integer/index work and the sampling stage are bit-exact; fp32 MLP outputs are
compared at 2e-5 * max(1, |raw|) (fp32 MFMA is an fmaf chain, the reference a CPU sgemm);
rgb/depth/acc at the north star's 1e-4 abs.
"""
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


def _packed(fx, dev, prefix):
    from nerf_amd import ops
    from nerf_amd.weights import make_state_dict_arrays
    sd = make_state_dict_arrays(int(fx["weight_seed"]), float(fx.get("density_gain", 1.0)),
                                float(fx.get("density_bias", 0.0)))
    return ops.pack_state_dict(sd, prefix, dev)


def _raw_tol(ref):
    return 2e-5 * max(1.0, float(np.abs(ref).max()))


def _check_raw(raw, ref, f64):
    """fp32 evaluation order differs (MFMA fmaf chain vs CPU sgemm), and the synthetic weights scale
    alpha_linear by up to 200x, so the bar is set by a float64 evaluation of the same inputs: we must
    be as close to it as the reference is (x3 slack), or within 2e-5 of the reference outright."""
    err_ref = float(np.abs(ref - f64).max())
    err_gpu = float(np.abs(raw - f64).max())
    assert err_gpu <= max(3.0 * err_ref, _raw_tol(ref)), (err_gpu, err_ref)


def _f64_rays(fx, prefix, z):
    from oracle import nerf_oracle as O
    from nerf_amd.weights import make_state_dict
    sd = make_state_dict(int(fx["weight_seed"]), float(fx["density_gain"]), float(fx["density_bias"]))
    zt = torch.from_numpy(np.ascontiguousarray(np.broadcast_to(z, (fx["xyz"].shape[0], z.shape[-1]))))
    pts, dhat = O.points(torch.from_numpy(fx["xyz"]), torch.from_numpy(fx["viewdirs"]), zt)
    return O.network_forward(sd, pts, dhat, "fine" if prefix == "model_fine" else "", dtype=torch.float64).numpy()


def test_library_loads(dev):
    from nerf_amd import _lib
    assert _lib.lib().nrf_version() >= 100


def test_raygen_bit_exact(golden, dev):
    from nerf_amd import ops
    fx = golden("raygen")
    H, W = int(fx["H"]), int(fx["W"])
    focal = 0.5 * W / np.tan(0.5 * float(fx["camera_angle_x"]))
    xyz, d = ops.raygen(fx["c2w"], H, W, focal, dev, pixel_index=_t(fx["ray_index"].astype(np.int64), dev))
    assert np.array_equal(xyz.cpu().numpy(), fx["xyz"])
    d = d.cpu().numpy()
    assert np.abs(d - fx["viewdirs"]).max() <= 1.2e-7          # reference dgemm may differ in the last fp64 bit
    assert (d != fx["viewdirs"]).mean() < 0.01
    # contiguous range == listed pixels
    xyz2, d2 = ops.raygen(fx["c2w"], H, W, focal, dev, first_ray=799 * 800, n_rays=800)
    sel = fx["ray_index"] >= 799 * 800
    assert np.array_equal(d2.cpu().numpy()[fx["ray_index"][sel] - 799 * 800], d[sel])


def test_mlp_points(golden, dev):
    from nerf_amd import ops
    fx = golden("network_points")
    for prefix, key in (("model", "raw_coarse"), ("model_fine", "raw_fine")):
        packed = _packed(fx, dev, prefix)
        raw = ops.mlp_forward_points(packed, _t(fx["pts"], dev), _t(fx["dirs"], dev)).cpu().numpy()
        assert raw.shape == fx[key].shape
        assert np.abs(raw - fx[key]).max() < _raw_tol(fx[key]), key
    # ragged tail: 96 points = 3 tiles; 70 points exercises the partial last tile
    raw70 = ops.mlp_forward_points(packed, _t(fx["pts"][:70], dev), _t(fx["dirs"][:70], dev)).cpu().numpy()
    assert np.abs(raw70 - fx["raw_fine"][:70]).max() < _raw_tol(fx["raw_fine"])


@pytest.mark.parametrize("name", FWD)
def test_mlp_rays(golden, dev, name):
    from nerf_amd import ops
    fx = golden(name)
    o, d = _t(fx["xyz"], dev), _t(fx["viewdirs"], dev)
    raw = ops.mlp_forward_rays(_packed(fx, dev, "model"), o, d, _t(fx["t_bins"], dev)).cpu().numpy()
    _check_raw(raw, fx["raw_coarse"], _f64_rays(fx, "model", fx["t_bins"]))
    raw = ops.mlp_forward_rays(_packed(fx, dev, "model_fine"), o, d, _t(fx["z_fine"], dev)).cpu().numpy()
    _check_raw(raw, fx["raw_fine"], _f64_rays(fx, "model_fine", fx["z_fine"]))


def test_composite_edge(golden, dev):
    from nerf_amd import ops
    fx = golden("composite_edge")
    rgb, depth, w, acc = ops.composite(_t(fx["raw"], dev), _t(fx["z"], dev), True)
    assert np.abs(w.cpu().numpy() - fx["weights"]).max() < 1e-6
    assert np.abs(rgb.cpu().numpy() - fx["rgb_map"]).max() < 1e-5
    assert np.abs(depth.cpu().numpy() - fx["depth_map"]).max() < 1e-5
    assert np.abs(acc.cpu().numpy() - fx["acc"]).max() < 1e-5
    assert np.all(rgb.cpu().numpy()[0] == 1.0) and acc.cpu().numpy()[0, 0] == 0.0   # sigma<=0: pure white


@pytest.mark.parametrize("name", FWD)
def test_composite(golden, dev, name):
    from nerf_amd import ops
    fx = golden(name)
    rgb, depth, w, acc = ops.composite(_t(fx["raw_coarse"], dev), _t(fx["t_bins"], dev), True)
    assert np.abs(w.cpu().numpy() - fx["w_coarse"]).max() < 1e-6
    assert np.abs(rgb.cpu().numpy() - fx["coarse_rgb_map"]).max() < 1e-5
    assert np.abs(depth.cpu().numpy() - fx["coarse_depth_map"]).max() < 2e-5
    rgb, depth, w, acc = ops.composite(_t(fx["raw_fine"], dev), _t(fx["z_fine"], dev), True, want_weights=False)
    assert w is None
    assert np.abs(rgb.cpu().numpy() - fx["fine_rgb_map"]).max() < 1e-5
    assert np.abs(depth.cpu().numpy() - fx["fine_depth_map"]).max() < 2e-5
    assert np.abs(acc.cpu().numpy() - fx["alpha_map"]).max() < 1e-5


@pytest.mark.parametrize("name", FWD)
def test_sample_fine_bit_exact(golden, dev, name):
    from nerf_amd import ops
    fx = golden(name)
    bins = _t(fx["t_bins"], dev)
    z, fine, idx = ops.sample_fine(_t(fx["w_coarse"], dev), bins, bins, _t(fx["u"], dev), _t(fx["u2"], dev),
                                   want_debug=True)
    assert np.array_equal(idx.cpu().numpy().astype(np.int64), fx["bin_indices"])
    assert np.array_equal(fine.cpu().numpy(), fx["fine_samples"])
    assert np.array_equal(z.cpu().numpy(), fx["z_fine"])


def test_sample_fine_edge_bit_exact(golden, dev):
    from nerf_amd import ops
    fx = golden("importance_edge")
    bins = _t(fx["t_bins"], dev)
    z, fine, idx = ops.sample_fine(_t(fx["w"], dev), bins, bins, _t(fx["u"], dev), _t(fx["u2"], dev), want_debug=True)
    assert np.array_equal(fine.cpu().numpy(), fx["fine_samples"])
    assert np.all(fine.cpu().numpy()[0] == 6.0)
    assert np.array_equal(z.cpu().numpy(), np.sort(np.concatenate(
        [np.broadcast_to(fx["t_bins"], (8, 64)), fx["fine_samples"]], -1), -1))


def test_perturbed_sampling_bit_exact(golden, dev):
    from nerf_amd import ops
    from oracle import nerf_oracle as O
    fx = golden("train_camera32")
    bins = _t(fx["t_bins"], dev)
    zc = ops.perturb_samples(bins, _t(fx["u_coarse"], dev))
    assert np.array_equal(zc.cpu().numpy(), fx["z_coarse"])
    z, fine, idx = ops.sample_fine(_t(fx["w_coarse"], dev), bins, zc, _t(fx["u"], dev), _t(fx["u2"], dev),
                                   u_fine=_t(fx["u_fine"], dev), want_debug=True)
    assert np.array_equal(fine.cpu().numpy(), fx["fine_samples"])
    assert np.array_equal(z.cpu().numpy(), O.merge_sort(fx["z_coarse"], fx["fine_samples"]))


def _render(fx, dev, perturbed=False):
    from nerf_amd.network import Network
    from nerf_amd.renderer import Renderer
    from nerf_amd.weights import make_state_dict
    from nerf_amd.config import default_cfg
    net = Network()
    net.load_state_dict(make_state_dict(int(fx["weight_seed"]), float(fx["density_gain"]), float(fx["density_bias"])))
    net.to(dev)
    r = Renderer(net, default_cfg(perturb=1 if perturbed else 0))
    r._bins = _t(fx["t_bins"], dev)       # the reference's own near/far table (differs by <= 1 ulp on 5 entries)
    with torch.no_grad():
        return r.render_rays(_t(fx["xyz"], dev), _t(fx["viewdirs"], dev), _t(fx["u"], dev), _t(fx["u2"], dev),
                             _t(fx["u_coarse"], dev) if perturbed else None,
                             _t(fx["u_fine"], dev) if perturbed else None)


@pytest.mark.parametrize("name", FWD)
def test_render_end_to_end(golden, dev, name):
    fx = golden(name)
    out = _render(fx, dev)
    for k, tol in [("coarse_rgb_map", 1e-4), ("coarse_depth_map", 1e-4), ("fine_rgb_map", 1e-4),
                   ("fine_depth_map", 1e-4), ("alpha_map", 1e-4)]:
        got = out[k].cpu().numpy()
        assert got.shape == fx[k].shape, k
        assert np.abs(got - fx[k]).max() < tol, (k, float(np.abs(got - fx[k]).max()))
    assert out["rgb_map"] is out["fine_rgb_map"] and out["acc_map"] is out["alpha_map"]


def test_render_training_sampling_end_to_end(golden, dev):
    fx = golden("train_camera32")
    out = _render(fx, dev, perturbed=True)
    for k in ("coarse_rgb_map", "fine_rgb_map", "fine_depth_map", "alpha_map"):
        assert np.abs(out[k].cpu().numpy() - fx[k]).max() < 1e-4, k


def test_renderer_batch_api(golden, dev):
    """Renderer.render(batch) with the reference's batch dict; deterministic under manual_seed."""
    from nerf_amd.network import Network
    from nerf_amd.renderer import Renderer
    from nerf_amd.weights import make_state_dict
    from nerf_amd.config import default_cfg
    fx = golden("forward_camera64")
    net = Network()
    net.load_state_dict(make_state_dict(1, 40.0, 3.5))
    net.to(dev)
    r = Renderer(net, default_cfg())
    batch = {"xyz": _t(fx["xyz"], dev)[None], "viewdirs": _t(fx["viewdirs"], dev)[None],
             "gt_rgb": _t(fx["gt_rgb"], dev)[None], "H": 800, "W": 800}
    with torch.no_grad():
        torch.manual_seed(7)
        a = r.render(batch, is_training=False)
        torch.manual_seed(7)
        b = r(batch, is_training=False)
    for k in ("coarse_rgb_map", "coarse_depth_map", "fine_rgb_map", "fine_depth_map", "alpha_map"):
        assert torch.equal(a[k], b[k])
    assert a["fine_rgb_map"].shape == (64, 3) and a["fine_depth_map"].shape == (64,) and a["alpha_map"].shape == (64, 1)
    # coarse pass does not depend on the uniforms: must match the reference
    assert np.abs(a["coarse_rgb_map"].cpu().numpy() - fx["coarse_rgb_map"]).max() < 1e-4
    with pytest.raises(Exception):
        r.render({"xyz": batch["xyz"].cpu(), "viewdirs": batch["viewdirs"].cpu(), "gt_rgb": batch["gt_rgb"].cpu()}, False)


def test_mlp_large_vs_oracle(dev):
    """4096 rays x 64 samples on the GPU vs the CPU oracle on a random subset of rays."""
    from nerf_amd import ops
    from nerf_amd.weights import make_state_dict, make_state_dict_arrays
    from nerf_amd.camera import pose_spherical, focal_from_angle
    from oracle import nerf_oracle as O
    xyz, d = ops.raygen(pose_spherical(77.0), 800, 800, focal_from_angle(800), dev, first_ray=300 * 800, n_rays=4096)
    bins = torch.from_numpy(O.t_bins(2.0, 6.0, 64)).to(dev)
    raw = ops.mlp_forward_rays(ops.pack_state_dict(make_state_dict_arrays(5, 30.0, 2.0), "model", dev), xyz, d, bins)
    sel = np.arange(0, 4096, 97)
    o_c, d_c = xyz.cpu()[sel], d.cpu()[sel]
    pts, dhat = O.points(o_c, d_c, bins.cpu()[None].expand(len(sel), 64))
    ref = O.network_forward(make_state_dict(5, 30.0, 2.0), pts, dhat).numpy()
    f64 = O.network_forward(make_state_dict(5, 30.0, 2.0), pts, dhat, dtype=torch.float64).numpy()
    _check_raw(raw.cpu().numpy()[sel], ref, f64)


def test_error_codes(dev):
    from nerf_amd import ops, _lib
    z = torch.zeros(48, device=dev)
    with pytest.raises(_lib.NrfError):
        ops.mlp_forward_rays(torch.zeros(ops.packed_floats(), device=dev), torch.zeros(4, 3, device=dev),
                             torch.ones(4, 3, device=dev), z)          # 48 samples: not a multiple of 32
    with pytest.raises(_lib.NrfError):
        ops.composite(torch.zeros(4, 96, 4, device=dev), torch.zeros(96, device=dev))    # not a multiple of 64
    assert ops.mlp_forward_points(torch.zeros(ops.packed_floats(), device=dev), torch.zeros(0, 3, device=dev),
                                  torch.zeros(0, 3, device=dev)).shape == (0, 4)        # empty input is fine


def test_distillation_records(dev, tmp_path):
    """gen_dist_data.py's per-view record (keys, shapes, file names) from the HIP renderer, two 'ranks'."""
    from nerf_amd import distill
    from nerf_amd.camera import focal_from_angle, test_set_poses
    from nerf_amd.config import default_cfg
    from nerf_amd.network import Network
    from nerf_amd.renderer import Renderer
    from nerf_amd.weights import make_state_dict
    net = Network()
    net.load_state_dict(make_state_dict(0, 30.0, 2.0))
    r = Renderer(net.to(dev), default_cfg(perturb=0))
    poses = test_set_poses(200)[:3]
    p0 = distill.generate(r, poses, 16, 16, focal_from_angle(16), str(tmp_path), dev, rank=0, world=2)
    torch.manual_seed(5)                       # the importance-sampling uniforms are drawn per render call
    p1 = distill.generate(r, poses, 16, 16, focal_from_angle(16), str(tmp_path), dev, rank=1, world=2)
    assert [p.split("/")[-1] for p in p0] == ["dist_0000.pt", "dist_0002.pt"] and [p.split("/")[-1] for p in p1] == ["dist_0001.pt"]
    rec = torch.load(p1[0], weights_only=True)
    assert set(rec) == {"xyz", "viewdirs", "teacher_rgb", "teacher_alpha"}
    assert rec["xyz"].shape == (256, 3) and rec["teacher_rgb"].shape == (256, 3) and rec["teacher_alpha"].shape == (256, 1)
    assert not rec["teacher_rgb"].is_cuda and torch.isfinite(rec["teacher_rgb"]).all()
    # same numbers as rendering that view directly
    o, d = rec["xyz"].to(dev), rec["viewdirs"].to(dev)
    torch.manual_seed(5)
    out = r.render({"xyz": o[None], "viewdirs": d[None], "gt_rgb": o[None]}, is_training=False)
    assert torch.equal(out["fine_rgb_map"].cpu(), rec["teacher_rgb"])


def test_checkpoint_render_and_psnr(dev, tmp_path):
    """a checkpoint in the reference's format -> fresh Network -> HIP render -> evaluator PSNR."""
    from nerf_amd import checkpoint as ckpt, ops
    from nerf_amd.camera import focal_from_angle, test_set_poses
    from nerf_amd.config import default_cfg
    from nerf_amd.network import Network
    from nerf_amd.renderer import Renderer
    from nerf_amd.weights import make_state_dict
    a = Network()
    a.load_state_dict(make_state_dict(2, 30.0, 2.0))
    ckpt.save_model(a, str(tmp_path), 199, last=True)
    b = Network().to(dev)
    assert ckpt.load_network(b, str(tmp_path), map_location=dev) == 200
    H = W = 16
    o, d = ops.raygen(test_set_poses(200)[7], H, W, focal_from_angle(W), dev)
    gt = torch.rand(H * W, 3, device=dev)
    batch = {"xyz": o[None], "viewdirs": d[None], "gt_rgb": gt[None], "H": torch.tensor(H), "W": torch.tensor(W)}
    outs = []
    for net in (a.to(dev), b):
        torch.manual_seed(11)
        outs.append(Renderer(net, default_cfg(perturb=0)).render(batch, is_training=False))
    assert torch.equal(outs[0]["fine_rgb_map"], outs[1]["fine_rgb_map"])
    ev = ckpt.Evaluator()
    ev.evaluate(outs[1], batch)
    want = -10 * np.log10(np.mean((outs[1]["fine_rgb_map"].detach().cpu().numpy() - gt.cpu().numpy()) ** 2))
    assert abs(ev.summarize()["mean_psnr"] - want) < 1e-4


@pytest.mark.parametrize("n_imp,n_rays", [(128, 200000), (64, 20000), (192, 20000)])
def test_sample_fine_bit_exact_at_scale(dev, n_imp, n_rays):
    """Bit-exactness of the resampling + merge sort on many random rays: the fp64 wave-scan CDF with its
    rounding-boundary guard must agree with the sequential running sum of the reference everywhere (a guard that
    never fired would show up as ~1 wrong prefix per 5 M), and the register sort with np.sort."""
    from nerf_amd import ops
    from oracle import nerf_oracle as O
    rng = np.random.default_rng(n_imp)
    bins = O.t_bins(2.0, 6.0, 64)
    w = rng.uniform(size=(n_rays, 64)).astype(np.float32) ** 8                 # peaky, like real coarse weights
    w[rng.uniform(size=(n_rays, 64)) < 0.5] = 0
    w[:100] = 0                                                                 # all-zero rows: every sample = far
    w[100:200, 1:] = 0                                                          # one-hot rows
    w[200:300] = np.float32(1e-30) * w[200:300]                                 # tiny weights
    u = rng.uniform(size=(n_rays, n_imp)).astype(np.float32)
    u2 = rng.uniform(size=(n_rays, n_imp)).astype(np.float32)
    u[300:310, :4] = [0.0, np.nextafter(np.float32(1), np.float32(0)), 0.5, 0.25]
    zc = ops.perturb_samples(_t(bins, dev), _t(rng.uniform(size=(n_rays, 64)).astype(np.float32), dev))
    z, fine, idx = ops.sample_fine(_t(w, dev), _t(bins, dev), zc, _t(u, dev), _t(u2, dev), want_debug=True)
    z, fine, idx, zc = z.cpu().numpy(), fine.cpu().numpy(), idx.cpu().numpy(), zc.cpu().numpy()
    for b in range(0, n_rays, 16384):
        sl = slice(b, min(b + 16384, n_rays))
        f_ref, i_ref = O.importance_sampling(w[sl], bins, u[sl], u2[sl])
        assert np.array_equal(idx[sl].astype(np.int64), i_ref)
        assert np.array_equal(fine[sl], f_ref)
        assert np.array_equal(z[sl], O.merge_sort(zc[sl], f_ref))


def test_occupancy_grid_and_occupancy_forward_vs_oracle(dev):
    """The reference renderer's second path (volume_renderer.py:50-139, 156-197): vertex occupancy bake against the
    oracle's torch restatement, then occupancy-grid rendering of the coarse network against the oracle under this
    repository's sampling rule (nerfacc absent: parity unpinned)."""
    from nerf_amd.camera import LEGO_CAMERA_ANGLE_X, pose_spherical
    from nerf_amd.config import default_cfg
    from nerf_amd.network import Network
    from nerf_amd.renderer import Renderer
    from nerf_amd.weights import make_state_dict
    from oracle import nerf_oracle as O
    sd = make_state_dict(5, 30.0, -1.0)                      # density bias -1: about half of the box is empty
    net = Network()
    net.load_state_dict(sd)
    cfg = default_cfg(perturb=0, grid_resolution=[2, 2, 2], occ_factor=12,
                      aabb={"min": [-1.0, -1.0, -1.0], "max": [1.0, 1.0, 1.0]})
    r = Renderer(net.to(dev), cfg)
    assert r.occ_grid_resolution == [24, 24, 24]
    thr = 0.01
    occ = r.build_occ_grid(thr).cpu().numpy()
    ref_occ, sig = O.build_occ_grid(sd, [-1.0] * 3, [1.0] * 3, [24, 24, 24], thr, want_sigma=True)
    clear = np.abs(sig - thr) > 2e-4 * 30                    # fp32 rounding next to the threshold may flip a voxel
    assert clear.mean() > 0.97 and np.array_equal(occ.astype(bool)[clear], ref_occ[clear])
    assert 0.05 < occ.mean() < 0.95
    r.occ_grid = _t(ref_occ.astype(np.uint8), dev)           # same grid on both sides from here on
    r.render_step_size = 0.01                                # keeps the CPU oracle to ~20 k samples
    o, d = O.raygen(pose_spherical(20.0, -30.0, 4.031), 12, 12, LEGO_CAMERA_ANGLE_X * 0.9)
    o, d = o.reshape(-1, 3).astype(np.float32), d.reshape(-1, 3).astype(np.float32)
    out = r.render_occupancy({"xyz": _t(o, dev)[None], "viewdirs": _t(d, dev)[None]})
    ref = O.render_occupancy(sd, o, d, ref_occ.astype(np.uint8), [-1.0] * 3, [1.0] * 3, 2.0, 6.0, 0.01, r.early_stop_eps, True)
    assert r.last_num_samples == ref["n_samples"] > 2000
    assert out["fine_rgb_map"].shape == (1, 144, 3) and out["alpha_map"].shape == (1, 144, 1)
    for k, tol in (("fine_rgb_map", 1e-4), ("alpha_map", 1e-4), ("fine_depth_map", 5e-4)):
        assert np.abs(out[k].cpu().numpy() - ref[k]).max() < tol, k
    # forward() takes this path for inference when the switch is on, and the classic path otherwise
    r.occupancy_forward = True
    batch = {"xyz": _t(o, dev)[None], "viewdirs": _t(d, dev)[None], "gt_rgb": _t(o, dev)[None]}
    with torch.no_grad():
        again = r(batch, is_training=False)
    assert torch.equal(again["fine_rgb_map"], out["fine_rgb_map"]) and "coarse_rgb_map" not in again
    r.occupancy_forward = False
    with torch.no_grad():
        assert "coarse_rgb_map" in r(batch, is_training=False)
