"""KiloNeRF oracle against the fixtures made from the real reference (tools/gen_golden_kilo.py),
plus CPU self-consistency of the own-spec (unpinned) marching / packed-compositing rules."""
import numpy as np
import torch

from oracle import kilo_oracle as K
from nerf_amd.weights import KILO_PARAMS_PER_CELL, make_kilo_state_dict_arrays

AABB_MIN, AABB_MAX, RES = [-1.0, -1.0, -1.0], [1.0, 1.0, 1.0], [16, 16, 16]


def test_route_matches_reference_bit_exact(golden):
    fx = golden("kilo_route")
    idx3, flat = K.route(fx["pts"], AABB_MIN, AABB_MAX, RES)
    assert np.array_equal(idx3, fx["index3"])
    assert np.array_equal(flat, fx["flat"])
    assert flat.min() == 0 and flat.max() == 4095


def test_route_non_finite_points_vs_reference(golden):
    """+-inf coordinates have a defined answer in the reference (clamped to the first / last cell, network.py:329-331)
    and the oracle reproduces it; NaN does not: the reference's `.long()` of NaN is INT64_MIN (recorded in the fixture)
    and its next line indexes with it and raises.  The oracle -- and kilo_route_kernel -- send a NaN coordinate to
    cell 0 instead: a documented difference, not a pinned one."""
    fx = golden("kilo_route_nonfinite")
    idx3, _ = K.route(fx["pts"], AABB_MIN, AABB_MAX, RES)
    finite = ~np.isnan(fx["pts"])
    assert np.array_equal(idx3[finite], fx["index3"][finite])
    assert (fx["index3"][~finite] == np.iinfo(np.int64).min).all()         # what the reference does with NaN
    assert (idx3[~finite] == 0).all()


def _sd(fx):
    return make_kilo_state_dict_arrays(int(fx["weight_seed"]), 4096, float(fx["density_gain"]), float(fx["density_bias"]))


def test_network_matches_reference(golden):
    fx = golden("kilo_network")
    sd = _sd(fx)
    raw = K.network_forward(sd, torch.from_numpy(fx["pts"]), torch.from_numpy(fx["dirs"]), AABB_MIN, AABB_MAX, RES).numpy()
    assert np.abs(raw[:, :3] - fx["rgb"].reshape(-1, 3)).max() < 2e-5
    assert np.abs(raw[:, 3] - fx["sigma"].reshape(-1)).max() < 2e-4      # sigma head scaled by 20
    R, S = fx["rgb_rays"].shape[:2]
    raw3 = K.network_forward(sd, torch.from_numpy(fx["pts"][:R * S].reshape(R, S, 3)), torch.from_numpy(fx["dirs"][:R]),
                             AABB_MIN, AABB_MAX, RES).numpy()
    assert np.abs(raw3[:, :3] - fx["rgb_rays"].reshape(-1, 3)).max() < 2e-5
    assert np.abs(raw3[:, 3] - fx["sigma_rays"].reshape(-1)).max() < 2e-4


def test_color_reg_loss_matches_reference(golden):
    fx = golden("kilo_network")
    assert abs(K.color_reg_loss(_sd(fx)) - float(fx["color_reg_loss"])) < 1e-3 * float(fx["color_reg_loss"])
    assert KILO_PARAMS_PER_CELL == 6212


def _sphere_occ(n=32, radius=0.75):
    c = (np.arange(n) + 0.5) / n * 2 - 1
    x, y, z = np.meshgrid(c, c, c, indexing="ij")
    return (np.sqrt(x * x + y * y + z * z) < radius).astype(np.uint8)


def test_march_rules():
    occ = _sphere_occ()
    o = np.array([[0, 0, 4.0], [0, 0, 4.0], [3.0, 3.0, 3.0], [0.0, 0.0, 0.0]], dtype=np.float32)
    d = np.array([[0, 0, -1.0], [0, 1.0, 0], [0, 0, -1.0], [0.6, 0.0, 0.8]], dtype=np.float32)
    pts, ri, ts, te, off, cnt = K.march(o, d, occ, AABB_MIN, AABB_MAX, 2.0, 6.0, 0.01)
    assert cnt[1] == 0 and cnt[2] == 0                       # misses the box
    assert cnt[0] > 0 and off[0] == 0 and off[1] == cnt[0]
    # ray 0 travels through the sphere along -z: samples are inside it, ordered, step-sized
    assert np.all(np.linalg.norm(pts[:cnt[0]], axis=1) < 0.75 + 2.0 / 32 * np.sqrt(3))
    assert np.all(np.diff(ts[:cnt[0]]) > 0)
    assert np.all(te - ts <= 0.01 + 2e-6) and np.all(te > ts)
    assert abs(cnt[0] * 0.01 - 1.5) < 0.15                   # chord of the voxelised sphere
    # ray 3 starts inside the box: clipped to t >= near = 2 -> outside the box, nothing left
    assert cnt[3] == 0
    assert np.array_equal(ri, np.repeat(np.arange(4), cnt).astype(np.int32))


def test_composite_packed_rules():
    rng = np.random.default_rng(0)
    cnt = np.array([5, 0, 70, 1], dtype=np.int32)
    off = (np.cumsum(cnt) - cnt).astype(np.int32)
    n = int(cnt.sum())
    raw = rng.normal(size=(n, 4)).astype(np.float32)
    raw[:, 3] = np.abs(raw[:, 3]) * 30
    ts = np.concatenate([2 + 0.01 * np.arange(c) for c in cnt]).astype(np.float32)
    te = (ts + np.float32(0.01)).astype(np.float32)
    rgb, depth, acc = K.composite_packed(raw, ts, te, off, cnt, 0.0, True)
    rgb0, depth0, acc0 = K.composite_packed(raw, ts, te, off, cnt, 0.01, True)
    assert np.allclose(rgb[1], 1.0) and acc[1] == 0 and depth[1] == 0          # empty ray -> background
    assert np.all(acc <= 1 + 1e-12) and np.all(acc0 <= acc + 1e-12)
    assert np.all(acc - acc0 < 0.01 + 1e-12)                                   # ERT drops < eps of opacity
    assert np.abs(rgb - rgb0).max() < 0.011
    # single sample ray: closed form
    a = 1 - np.exp(-max(raw[-1, 3], 0) * 0.01)
    assert abs(acc[3] - a) < 1e-6
    assert abs(depth[3] - 0.5 * (ts[-1] + te[-1])) < 1e-5


def test_bake_points_layout():
    """_evaluate_grid's points: cell order C (i,j,k), 27 offsets in meshgrid 'ij' order, centre = corner + cell/2."""
    pts = K.bake_points(AABB_MIN, AABB_MAX, [4, 2, 8]).numpy()
    assert pts.shape == (64, 27, 3)
    cell = np.array([0.5, 1.0, 0.25], dtype=np.float32)
    assert np.allclose(pts[0, 13], np.array([-1, -1, -1]) + cell / 2)            # sub-point 13 = the centre
    assert np.allclose(pts[0, 0], pts[0, 13] - 0.25 * cell) and np.allclose(pts[0, 26], pts[0, 13] + 0.25 * cell)
    assert np.allclose(pts[1, 13] - pts[0, 13], [0, 0, 0.25])                     # k is the fastest cell index
    assert np.allclose(pts[0, 1] - pts[0, 0], [0, 0, 0.25 * 0.25])                # z offset is the fastest sub index
    assert np.allclose(pts[63, 13], np.array([1, 1, 1]) - cell / 2)


def test_network_gradients_match_reference_autograd(golden):
    """Autograd through the oracle's restatement == autograd through the real reference network (fixture)."""
    from nerf_amd.weights import KILO_TENSORS
    fx, gx = golden("kilo_network"), golden("kilo_network_grad")
    sd = {k: torch.from_numpy(v).requires_grad_(True) for k, v in _sd(fx).items()}
    raw = K.network_forward(sd, torch.from_numpy(fx["pts"]), torch.from_numpy(fx["dirs"]), AABB_MIN, AABB_MAX, RES)
    (raw * torch.from_numpy(gx["grad_raw"])).sum().backward()
    cells = gx["cells"]
    for name, _, _ in KILO_TENSORS:
        g = sd["model." + name].grad
        ref = gx["grad." + name]
        scale = max(1.0, float(np.abs(ref).max()))
        assert np.abs(g[cells].numpy() - ref).max() < 2e-5 * scale, name
        assert abs(float(g.double().sum()) - float(gx["sum." + name])) < 1e-5 * float(gx["abssum." + name]), name
        assert abs(float(g.double().abs().sum()) - float(gx["abssum." + name])) < 1e-5 * float(gx["abssum." + name]), name


def test_composite_torch_matches_numpy_rule_and_finite_differences():
    rng = np.random.default_rng(5)
    counts = np.array([0, 7, 70, 1, 130], dtype=np.int32)
    offsets = (np.cumsum(counts) - counts).astype(np.int32)
    n = int(counts.sum())
    raw = rng.normal(size=(n, 4)).astype(np.float64)
    raw[:, 3] = raw[:, 3] * 30 + 10
    ts = np.concatenate([2.0 + 0.005 * np.arange(c) for c in counts])
    te = ts + 0.005
    rgb, depth, acc = K.composite_packed(raw, ts, te, offsets, counts, 0.01, True)
    t = torch.from_numpy(raw).requires_grad_(True)
    rgb_t, depth_t, acc_t = K.composite_packed_torch(t, ts, te, offsets, counts, 0.01, True)
    assert np.abs(rgb_t.detach().numpy() - rgb).max() < 1e-12
    assert np.abs(depth_t.detach().numpy() - depth).max() < 1e-12
    assert np.abs(acc_t.detach().numpy() - acc).max() < 1e-12
    wr, wd, wa = rng.normal(size=rgb.shape), rng.normal(size=depth.shape), rng.normal(size=acc.shape)
    ((rgb_t * torch.from_numpy(wr)).sum() + (depth_t * torch.from_numpy(wd)).sum() + (acc_t * torch.from_numpy(wa)).sum()).backward()
    g = t.grad.numpy()

    def f(x):
        a, b, c = K.composite_packed(x, ts, te, offsets, counts, 0.01, True)
        return (a * wr).sum() + (b * wd).sum() + (c * wa).sum()

    for i in rng.choice(n, 12, replace=False):
        for ch in (0, 3):
            e = np.zeros_like(raw)
            e[i, ch] = 1e-6
            fd = (f(raw + e) - f(raw - e)) / 2e-6
            assert abs(fd - g[i, ch]) < 1e-5 * max(1.0, abs(fd)), (i, ch, fd, g[i, ch])
