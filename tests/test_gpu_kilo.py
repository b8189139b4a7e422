"""GPU parity of the KiloNeRF entry points (nrf_kilo_*) through the C ABI.

Routing is bit-exact against the reference fixture; the tiny-MLP outputs are checked against the
reference fixture with the float64 yardstick; marching is bit-exact against the oracle's own rule
and packed compositing within 1e-4 (those two rules are this repo's own: parity unpinned, see
oracle/kilo_oracle.py)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

AABB = [-1.0, -1.0, -1.0, 1.0, 1.0, 1.0]
RES = [16, 16, 16]


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _sd(seed=7, gain=20.0, bias=1.0):
    from nerf_amd.weights import make_kilo_state_dict_arrays
    return make_kilo_state_dict_arrays(seed, 4096, gain, bias)


@pytest.fixture(scope="module", params=["fp32", "fp16x2", "bf16x3"])
def packed(dev, request):
    """(records, kind): the fp32 MFMA records, the two-fp16-term records of nrf_kilo_forward_fp16x2 and the three-bf16-term
    records of nrf_kilo_forward_bf16x3 -- one set of tests and one set of bars for the three kernels."""
    from nerf_amd import kilo_ops
    return kilo_ops.kilo_pack_state_dict(_sd(), dev, kind=request.param), request.param


def test_route_bit_exact(golden, dev):
    from nerf_amd import kilo_ops
    fx = golden("kilo_route")
    cell = kilo_ops.kilo_route(_t(fx["pts"], dev), AABB, RES).cpu().numpy()
    assert np.array_equal(cell.astype(np.int64), fx["flat"])


def test_route_non_finite_vs_reference_fixture(golden, dev):
    """+-inf: bit-exact with the reference (tests/golden/kilo_route_nonfinite.npz, made by the real _get_index);
    NaN: the reference yields INT64_MIN and then fails to index, the kernel routes to cell 0 (documented difference)."""
    from nerf_amd import kilo_ops
    fx = golden("kilo_route_nonfinite")
    cell = kilo_ops.kilo_route(_t(fx["pts"], dev), AABB, RES).cpu().numpy().astype(np.int64)
    i3 = np.where(np.isnan(fx["pts"]), 0, fx["index3"])
    assert np.array_equal(cell, i3[:, 0] * 256 + i3[:, 1] * 16 + i3[:, 2])


def test_route_nan_inf_and_empty(dev):
    from nerf_amd import kilo_ops
    from oracle import kilo_oracle as K
    p = np.array([[np.nan, 0, 0], [np.inf, -np.inf, 0.5], [1e30, -1e30, 0], [0, 0, 0]], dtype=np.float32)
    cell = kilo_ops.kilo_route(_t(p, dev), AABB, RES).cpu().numpy()
    assert np.array_equal(cell, K.route(p, AABB[:3], AABB[3:], RES)[1])
    assert kilo_ops.kilo_route(torch.empty((0, 3), device=dev), AABB, RES).numel() == 0


def _check(raw, ref, f64, tol):
    err_ref = float(np.abs(ref - f64).max())
    err_gpu = float(np.abs(raw - f64).max())
    assert err_gpu <= max(3.0 * err_ref, tol), (err_gpu, err_ref)


def test_tiny_mlp_matches_reference(golden, dev, packed):
    from nerf_amd import kilo_ops
    from oracle import kilo_oracle as K
    fx = golden("kilo_network")
    packed, kind = packed
    raw = kilo_ops.kilo_forward(packed, 4096, AABB, RES, _t(fx["pts"], dev), _t(fx["dirs"], dev), kind=kind).cpu().numpy()
    ref = np.concatenate([fx["rgb"].reshape(-1, 3), fx["sigma"].reshape(-1, 1)], -1)
    f64 = K.network_forward(_sd(), torch.from_numpy(fx["pts"]), torch.from_numpy(fx["dirs"]), AABB[:3], AABB[3:], RES,
                            dtype=torch.float64).numpy()
    _check(raw[:, :3], ref[:, :3], f64[:, :3], 2e-5)
    _check(raw[:, 3], ref[:, 3], f64[:, 3], 2e-4)
    # (R, S, 3) points with one direction per ray == dir_index form of the entry point
    R, S = fx["rgb_rays"].shape[:2]
    di = torch.arange(R, device=dev, dtype=torch.int32).repeat_interleave(S)
    raw3 = kilo_ops.kilo_forward(packed, 4096, AABB, RES, _t(fx["pts"][:R * S], dev), _t(fx["dirs"][:R], dev), di,
                                 kind=kind).cpu().numpy()
    ref3 = np.concatenate([fx["rgb_rays"].reshape(-1, 3), fx["sigma_rays"].reshape(-1, 1)], -1)
    assert np.abs(raw3[:, :3] - ref3[:, :3]).max() < 5e-5
    assert np.abs(raw3[:, 3] - ref3[:, 3]).max() < 5e-4


@pytest.mark.parametrize("n,spread", [(1, 1.0), (31, 1.0), (33, 0.02), (5000, 0.05), (200000, 1.2)])
def test_tiny_mlp_matches_oracle_sizes(dev, packed, n, spread):
    """Ragged tile tails, one hot cell (several jobs of one cell), all cells, points outside the box."""
    from nerf_amd import kilo_ops
    from oracle import kilo_oracle as K
    rng = np.random.default_rng(n)
    pts = (rng.uniform(-1, 1, size=(n, 3)) * spread + (0.31 if spread < 1 else 0.0)).astype(np.float32)
    dirs = rng.normal(size=(n, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    packed, kind = packed
    raw = kilo_ops.kilo_forward(packed, 4096, AABB, RES, _t(pts, dev), _t(dirs, dev), kind=kind).cpu().numpy()
    sub = slice(0, min(n, 20000))
    ref = K.network_forward(_sd(), torch.from_numpy(pts[sub]), torch.from_numpy(dirs[sub]), AABB[:3], AABB[3:], RES).numpy()
    f64 = K.network_forward(_sd(), torch.from_numpy(pts[sub]), torch.from_numpy(dirs[sub]), AABB[:3], AABB[3:], RES,
                            dtype=torch.float64).numpy()
    assert np.isfinite(raw).all()
    _check(raw[sub, :3], ref[:, :3], f64[:, :3], 2e-5)
    _check(raw[sub, 3], ref[:, 3], f64[:, 3], 2e-4)
    # order independence: a permuted batch gives the permuted result, bit for bit
    perm = rng.permutation(n)
    raw_p = kilo_ops.kilo_forward(packed, 4096, AABB, RES, _t(pts[perm], dev), _t(dirs[perm], dev), kind=kind).cpu().numpy()
    assert np.array_equal(raw_p, raw[perm])


def test_per_ray_direction_table_is_bit_identical_to_per_point_encoding(dev, packed):
    """nrf_kilo_encode_dirs + nrf_kilo_forward_encoded (direction features once per ray, read from a table) against
    nrf_kilo_forward / _fp16x2 with the same dir_index (features re-evaluated per point, as the reference does,
    network.py:243): the same device functions produce the operands, so raw must agree bit for bit -- ragged tails,
    rays shared between tiles and cells, a ray nobody samples, both kernels."""
    from nerf_amd import kilo_ops
    packed, kind = packed
    rng = np.random.default_rng(12)
    n_rays = 700
    dirs = rng.normal(size=(n_rays, 3)).astype(np.float32) * rng.uniform(0.2, 3.0, size=(n_rays, 1)).astype(np.float32)
    for n in (1, 37, 20011):
        pts = rng.uniform(-1.1, 1.1, size=(n, 3)).astype(np.float32)
        di = np.sort(rng.integers(0, n_rays - 1, size=n)).astype(np.int32)          # ray n_rays - 1 has no sample
        a = kilo_ops.kilo_forward(packed, 4096, AABB, RES, _t(pts, dev), _t(dirs, dev), _t(di, dev), kind=kind)
        b = kilo_ops.kilo_forward(packed, 4096, AABB, RES, _t(pts, dev), _t(dirs, dev), _t(di, dev), kind=kind,
                                  encode_dirs_per_ray=False)
        c = kilo_ops.kilo_forward(packed, 4096, AABB, RES, _t(pts, dev), _t(dirs[di], dev), kind=kind)   # per-point directions
        assert torch.isfinite(a).all() and torch.equal(a, b) and torch.equal(a, c), (kind, n)
    if kind == "fp16x2":   # the range guard travels with the table: a direction beyond fp16 refuses its ray's points only
        pts = rng.uniform(-1, 1, size=(4096, 3)).astype(np.float32)
        di = np.sort(rng.integers(0, n_rays, size=4096)).astype(np.int32)
        clean = kilo_ops.kilo_forward(packed, 4096, AABB, RES, _t(pts, dev), _t(dirs, dev), _t(di, dev), kind=kind)
        bad = dirs.copy()
        bad[di[2000], 2] = 1.0e5
        out = kilo_ops.kilo_forward(packed, 4096, AABB, RES, _t(pts, dev), _t(bad, dev), _t(di, dev), kind=kind)
        hit = torch.from_numpy(di == di[2000]).to(dev)
        assert torch.isnan(out[hit]).all() and torch.equal(out[~hit], clean[~hit])


def test_tiny_mlp_fp16x2_range_guard(dev):
    """Beyond fp16's range the fp16x2 kernel answers NaN, for the point (a huge coordinate) or for the cell (a huge
    weight), and leaves every other point's bits alone."""
    from nerf_amd import kilo_ops
    rng = np.random.default_rng(4)
    pts = rng.uniform(-1, 1, size=(4096, 3)).astype(np.float32)
    dirs = rng.normal(size=(4096, 3)).astype(np.float32)
    sd = _sd()
    clean = kilo_ops.kilo_forward(kilo_ops.kilo_pack_state_dict(sd, dev, kind="fp16x2"), 4096, AABB, RES, _t(pts, dev),
                                  _t(dirs, dev), kind="fp16x2")
    assert torch.isfinite(clean).all()
    far = pts.copy()
    far[100, 1] = 1.0e5          # routed to a border cell (the reference clamps the index) and refused there
    out = kilo_ops.kilo_forward(kilo_ops.kilo_pack_state_dict(sd, dev, kind="fp16x2"), 4096, AABB, RES, _t(far, dev),
                                _t(dirs, dev), kind="fp16x2")
    ok = torch.ones(4096, dtype=torch.bool, device=dev)
    ok[100] = False
    assert torch.isnan(out[100]).all() and torch.equal(out[ok], clean[ok])
    cell = int(kilo_ops.kilo_route(_t(pts, dev), AABB, RES)[7])
    big = {k: v.copy() for k, v in sd.items()}
    big["model.pts_linears_weights.1"][cell, 3, 5] = 7.0e4
    out = kilo_ops.kilo_forward(kilo_ops.kilo_pack_state_dict(big, dev, kind="fp16x2"), 4096, AABB, RES, _t(pts, dev),
                                _t(dirs, dev), kind="fp16x2")
    in_cell = kilo_ops.kilo_route(_t(pts, dev), AABB, RES) == cell
    assert bool(in_cell[7]) and torch.isnan(out[in_cell]).all() and torch.equal(out[~in_cell], clean[~in_cell])


def test_tiny_mlp_bf16x3_has_fp32_range_and_accuracy(dev):
    """The three-term bf16 kernel is the fp32 kernel's arithmetic: the inputs that make the fp16x2 kernel answer NaN (a
    coordinate of 1e5, a weight of 7e4) give finite values within the fp32 kernel's own distance from float64, and on
    ordinary inputs its distance from float64 is no larger than 1.5x the fp32 MFMA kernel's."""
    from nerf_amd import kilo_ops
    from oracle import kilo_oracle as K
    rng = np.random.default_rng(4)
    pts = rng.uniform(-1, 1, size=(4096, 3)).astype(np.float32)
    dirs = rng.normal(size=(4096, 3)).astype(np.float32)
    sd = _sd()
    cell = int(kilo_ops.kilo_route(_t(pts, dev), AABB, RES)[7])
    big = {k: v.copy() for k, v in sd.items()}
    big["model.pts_linears_weights.1"][cell, 3, 5] = 7.0e4
    far = pts.copy()
    far[100, 1] = 1.0e5
    for state, p in ((sd, pts), (big, pts), (sd, far)):
        out = {k: kilo_ops.kilo_forward(kilo_ops.kilo_pack_state_dict(state, dev, kind=k), 4096, AABB, RES, _t(p, dev),
                                        _t(dirs, dev), kind=k).cpu().numpy() for k in ("fp32", "bf16x3")}
        f64 = K.network_forward(state, torch.from_numpy(p), torch.from_numpy(dirs), AABB[:3], AABB[3:], RES,
                                dtype=torch.float64).numpy()
        assert np.isfinite(out["bf16x3"]).all()
        scale = np.maximum(np.abs(f64), 1.0)
        e32, ex3 = float((np.abs(out["fp32"] - f64) / scale).max()), float((np.abs(out["bf16x3"] - f64) / scale).max())
        print("relative distance from float64: fp32 kernel %.2e, bf16x3 kernel %.2e" % (e32, ex3))
        assert ex3 <= max(1.5 * e32, 1e-6), (e32, ex3)


def _sphere_occ(n, radius=0.75):
    c = (np.arange(n) + 0.5) / n * 2 - 1
    x, y, z = np.meshgrid(c, c, c, indexing="ij")
    return (np.sqrt(x * x + y * y + z * z) < radius).astype(np.uint8)


def _camera_rays(n_side, seed=0):
    from nerf_amd.camera import LEGO_CAMERA_ANGLE_X, pose_spherical
    from oracle import nerf_oracle as O
    c2w = pose_spherical(30.0 + seed, -30.0, 4.0)
    o, d = O.raygen(c2w, n_side, n_side, LEGO_CAMERA_ANGLE_X)
    d = d / np.linalg.norm(d, axis=-1, keepdims=True)
    return o.reshape(-1, 3).astype(np.float32), d.reshape(-1, 3).astype(np.float32)


def test_march_bit_exact_vs_oracle(dev):
    from nerf_amd import kilo_ops
    from oracle import kilo_oracle as K
    occ = _sphere_occ(64)
    o, d = _camera_rays(40)
    # add axis-parallel and missing rays
    o = np.concatenate([o, [[0, 0, 4], [0.2, 5, 0.1], [3, 3, 3]]]).astype(np.float32)
    d = np.concatenate([d, [[0, 0, -1], [0, -1, 0], [0, 0, 1]]]).astype(np.float32)
    ref = K.march(o, d, occ, AABB[:3], AABB[3:], 2.0, 6.0, 0.005)
    out = kilo_ops.kilo_march(_t(o, dev), _t(d, dev), _t(occ, dev), AABB, 2.0, 6.0, 0.005)
    names = ["pts", "ray_idx", "t_start", "t_end", "offsets", "counts"]
    assert int(ref[5].sum()) > 10000
    for nm, a, b in zip(names, out, ref):
        assert np.array_equal(a.cpu().numpy(), b), nm
    # the write pass that walks the grid again (no masks handed over) gives the same bits
    again = kilo_ops.kilo_march(_t(o, dev), _t(d, dev), _t(occ, dev), AABB, 2.0, 6.0, 0.005, reuse_masks=False)
    for nm, a, b in zip(names, again, out):
        assert torch.equal(a, b), nm


@pytest.mark.parametrize("step,box", [(0.0007, AABB), (0.05, AABB), (3.0, AABB),
                                      (0.004, [-1.3, -0.9, -1.0, 1.1, 1.2, 0.7]), (0.01, [-2.0, -0.5, -1.0, 2.0, 0.5, 1.0])])
def test_march_mask_handover_other_step_sizes(dev, step, box):
    """More than 64 chunks per ray (step 7e-4: 5715 steps), a few steps per ray, a single step; boxes whose edges
    are not powers of two (the kernel then divides instead of multiplying by the exact reciprocal) or differ."""
    from nerf_amd import kilo_ops
    from oracle import kilo_oracle as K
    occ = _sphere_occ(32, 0.9)
    o, d = _camera_rays(12, seed=5)
    ref = K.march(o, d, occ, box[:3], box[3:], 2.0, 6.0, step)
    assert int(ref[5].sum()) > 0
    for reuse in (True, False):
        out = kilo_ops.kilo_march(_t(o, dev), _t(d, dev), _t(occ, dev), box, 2.0, 6.0, step, reuse_masks=reuse)
        for nm, a, b in zip(["pts", "ray_idx", "t_start", "t_end", "offsets", "counts"], out, ref):
            assert np.array_equal(a.cpu().numpy(), b), (nm, reuse)


@pytest.mark.parametrize("res,box", [((32, 32, 32), AABB), ((30, 17, 9), [-1.3, -0.9, -1.0, 1.1, 1.2, 0.7])])
def test_march_on_bit_bricks_is_bit_identical(dev, res, box):
    """nrf_occ_pack_bricks + nrf_kilo_march_count_bricks: the count pass on one 64-bit word per 4 x 4 x 4 voxels gives the
    uint8 grid's samples bit for bit -- also for a random (not smooth) grid and edges that are no multiples of four."""
    from nerf_amd import kilo_ops
    rng = np.random.default_rng(11)
    occ = (rng.random(res) < 0.3).astype(np.uint8)
    o, d = _camera_rays(24, seed=3)
    g = _t(occ, dev)
    bricks = kilo_ops.occ_pack_bricks(g)
    bx, by, bz = [(r + 3) // 4 for r in res]
    words = bricks.cpu().numpy().view(np.uint64)[:bx * by * bz].reshape(bx, by, bz)
    pad = np.zeros((bx * 4, by * 4, bz * 4), dtype=np.uint64)
    pad[:res[0], :res[1], :res[2]] = occ
    want = np.zeros((bx, by, bz), dtype=np.uint64)
    for dx in range(4):
        for dy in range(4):
            for dz in range(4):
                want |= pad[dx::4, dy::4, dz::4] << np.uint64(dx * 16 + dy * 4 + dz)
    assert np.array_equal(words, want)
    a = kilo_ops.kilo_march(_t(o, dev), _t(d, dev), g, box, 2.0, 6.0, 0.005)
    b = kilo_ops.kilo_march(_t(o, dev), _t(d, dev), g, box, 2.0, 6.0, 0.005, bricks=bricks)
    assert int(a[5].sum()) > 0
    for nm, x, y in zip(["pts", "ray_idx", "t_start", "t_end", "offsets", "counts"], a, b):
        assert torch.equal(x, y), nm
    with pytest.raises(kilo_ops.NrfError):
        kilo_ops.kilo_march(_t(o, dev), _t(d, dev), g, box, 2.0, 6.0, 0.005, bricks=bricks[:-1])


def test_march_empty_grid_and_no_rays(dev):
    from nerf_amd import kilo_ops
    o, d = _camera_rays(8)
    occ = np.zeros((16, 16, 16), dtype=np.uint8)
    pts, ri, ts, te, off, cnt = kilo_ops.kilo_march(_t(o, dev), _t(d, dev), _t(occ, dev), AABB, 2.0, 6.0, 0.01)
    assert pts.shape == (0, 3) and int(cnt.sum()) == 0 and int(off.sum()) == 0
    rgb, depth, acc = kilo_ops.kilo_composite(torch.empty((0, 4), device=dev), ts, te, off, cnt, 0.01, True)
    assert torch.all(rgb == 1) and torch.all(acc == 0) and torch.all(depth == 0)
    e = torch.empty((0, 3), device=dev)
    out = kilo_ops.kilo_march(e, e, _t(occ, dev), AABB, 2.0, 6.0, 0.01)
    assert out[0].shape == (0, 3) and out[4].numel() == 0


@pytest.mark.parametrize("ert,white", [(0.0, True), (0.01, True), (0.01, False)])
def test_composite_packed_vs_oracle(dev, ert, white):
    from nerf_amd import kilo_ops
    from oracle import kilo_oracle as K
    rng = np.random.default_rng(5)
    cnt = rng.integers(0, 300, size=500).astype(np.int32)
    cnt[:4] = [0, 1, 64, 65]
    off = (np.cumsum(cnt) - cnt).astype(np.int32)
    n = int(cnt.sum())
    raw = rng.normal(size=(n, 4)).astype(np.float32)
    raw[:, 3] = (raw[:, 3] * 40 + 10).astype(np.float32)
    ts = np.concatenate([2 + 0.005 * np.arange(c) for c in cnt]).astype(np.float32)
    te = (ts + np.float32(0.005)).astype(np.float32)
    rgb, depth, acc = kilo_ops.kilo_composite(_t(raw, dev), _t(ts, dev), _t(te, dev), _t(off, dev), _t(cnt, dev), ert, white)
    r_rgb, r_depth, r_acc = K.composite_packed(raw, ts, te, off, cnt, ert, white)
    assert np.abs(rgb.cpu().numpy() - r_rgb).max() < 1e-4
    assert np.abs(acc.cpu().numpy() - r_acc).max() < 1e-4
    assert np.abs(depth.cpu().numpy() - r_depth).max() < 1e-4 * 6


@pytest.mark.parametrize("mlp_dtype", ["fp32", "fp16x2", "bf16x3"])
def test_kilo_renderer_vs_oracle(dev, mlp_dtype):
    """KiloRenderer(net)(batch) end to end (march -> route/sort -> tiny MLPs -> packed composite) against the
    oracle's render on a 24x24 view of the sphere grid; un-normalised camera directions as the reference passes."""
    from nerf_amd.camera import LEGO_CAMERA_ANGLE_X, pose_spherical
    from nerf_amd.kilonerf import KiloNetwork, KiloRenderer, default_kilo_cfg
    from oracle import kilo_oracle as K
    from oracle import nerf_oracle as O
    sd = _sd(3, 10.0, 0.5)
    net = KiloNetwork()
    missing = net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    assert set(missing.missing_keys) <= {"aabb_min", "aabb_max", "grid_resolution"} and not missing.unexpected_keys
    net = net.to(dev)
    occ = _sphere_occ(256)
    cfg = default_kilo_cfg()
    r = KiloRenderer(net, cfg, _t(occ, dev), mlp_dtype=mlp_dtype)
    o, d = O.raygen(pose_spherical(20.0, -30.0, 4.031), 24, 24, LEGO_CAMERA_ANGLE_X * 0.9)   # ~160 of 576 rays hit the sphere
    o, d = o.reshape(-1, 3).astype(np.float32), d.reshape(-1, 3).astype(np.float32)
    out = r({"xyz": _t(o, dev)[None], "viewdirs": _t(d, dev)[None]}, is_training=False)
    ref = K.render(sd, o, d, occ, AABB[:3], AABB[3:], RES, 2.0, 6.0, 0.005, 0.01, True)
    assert r.last_num_samples == ref["n_samples"] > 20000
    assert out["rgb_map"].shape == (1, 576, 3) and out["depth_map"].shape == (1, 576, 1) and out["alpha_map"].shape == (1, 576, 1)
    acc = ref["alpha_map"].reshape(-1)
    assert (acc > 0.5).sum() > 50 and (acc == 0).sum() > 50 and ((acc > 0.2) & (acc < 0.8)).sum() > 50
    for k in ("rgb_map", "alpha_map"):
        assert np.abs(out[k].cpu().numpy() - ref[k]).max() < 1e-4, k
    assert np.abs(out["depth_map"].cpu().numpy() - ref["depth_map"]).max() < 5e-4
    # network.forward signature: (R,S,3) + (R,3) -> rgb (R,S,3), sigma (R,S,1)
    p = torch.rand(6, 5, 3, device=dev) * 2 - 1
    rgb, sigma = net(p, _t(d[:6], dev))
    assert rgb.shape == (6, 5, 3) and sigma.shape == (6, 5, 1)
    ref_raw = K.network_forward(sd, p.cpu(), torch.from_numpy(d[:6]), AABB[:3], AABB[3:], RES).numpy()
    assert np.abs(torch.cat([rgb, sigma], -1).reshape(-1, 4).cpu().numpy() - ref_raw).max() < 1e-3
    assert abs(float(net.get_color_reg_loss().detach()) - K.color_reg_loss(sd)) < 1e-3 * K.color_reg_loss(sd)


@pytest.mark.parametrize("kind", ["fp32", "fp16x2", "bf16x3"])
def test_occupancy_bake_vs_oracle(dev, kind):
    """nrf_occ_bake / nrf_occ_bake_fp16x2 / nrf_occ_bake_bf16x3 (27 sub-points per cell through the teacher's fine MLP, any sigma > threshold) against the
    oracle's restatement of _evaluate_grid on a 12x10x14 grid; cells whose deciding sigma sits within 2e-4 of the
    threshold may legitimately differ (fp32 MLP rounding) and are excluded."""
    from nerf_amd import kilo_ops, ops
    from nerf_amd.weights import make_state_dict, make_state_dict_arrays
    from oracle import kilo_oracle as K
    res, thr = [12, 10, 14], 2.25      # random-init teacher: max sigma per cell spans 1.7 .. 2.8
    pack = {"fp32": ops.pack_state_dict, "fp16x2": ops.pack_state_dict_fp16x2, "bf16x3": ops.pack_state_dict_bf16x3}[kind]
    packed = pack(make_state_dict_arrays(0, 30.0, 2.0), "model_fine", dev)
    occ = kilo_ops.occ_bake(packed, AABB, res, thr, chunk_cells=500, kind=kind).cpu().numpy()      # 4 chunks, ragged tail
    ref, sig = K.bake(make_state_dict(0, 30.0, 2.0), AABB[:3], AABB[3:], res, thr, want_sigma=True)
    assert occ.shape == tuple(res)
    sure = (np.abs(sig - thr).min(-1) > 2e-4).reshape(res)
    assert sure.mean() > 0.9
    assert np.array_equal(occ[sure], ref[sure])
    assert 0.05 < ref.mean() < 0.95                      # a non-trivial grid


def test_bake_points_match_torch_formula(dev):
    """The in-kernel point generator against torch.linspace-based points (bit-exact on the power-of-two lego box,
    <= 1 ulp on an odd box)."""
    import ctypes as C
    from nerf_amd._lib import lib
    from oracle import kilo_oracle as K
    for aabb, res in [(AABB, [16, 16, 16]), ([-1.3, -0.7, 0.1, 2.9, 1.1, 3.3], [7, 5, 3])]:
        n = res[0] * res[1] * res[2]
        ref = K.bake_points(aabb[:3], aabb[3:], res).reshape(-1, 3).numpy()
        # drive the generator through the public entry point: a teacher whose sigma is the x coordinate is overkill,
        # so read the points back from the workspace the bake leaves behind (pts are its first n*27*3 floats)
        from nerf_amd import ops
        from nerf_amd.weights import make_state_dict_arrays
        packed = ops.pack_state_dict(make_state_dict_arrays(0, 1.0, 0.0), "model_fine", dev)
        ws = torch.empty(int(lib().nrf_occ_bake_workspace_bytes(n)), dtype=torch.uint8, device=dev)
        occ = torch.zeros(n, dtype=torch.uint8, device=dev)
        a = np.asarray(aabb, dtype=np.float32)
        r = np.asarray(res, dtype=np.int32)
        st = lib().nrf_occ_bake(C.c_void_p(packed.data_ptr()), a.ctypes.data_as(C.c_void_p), r.ctypes.data_as(C.c_void_p),
                                C.c_float(0.0), 0, n, C.c_void_p(occ.data_ptr()), C.c_void_p(ws.data_ptr()), ws.numel(), None)
        assert st == 0
        torch.cuda.synchronize()
        pts = ws[: n * 27 * 12].view(torch.float32).view(-1, 3).cpu().numpy()
        if aabb == AABB:
            assert np.array_equal(pts, ref)
        else:
            assert np.abs(pts - ref).max() <= 2.5e-7 * 3.3


def test_renderer_with_reference_style_network_module(dev):
    """KiloRenderer on a module that only has the reference's parameter / buffer names (what the drop-in sees when
    the reference keeps its own src.models.kilonerf.network.Network): same bits as with KiloNetwork, and a
    parameter update is picked up (version-keyed repack)."""
    from test_dropin import _RefStyleKiloNet
    from nerf_amd.kilonerf import KiloNetwork, KiloRenderer, default_kilo_cfg
    sd = _sd(3, 10.0, 0.5)
    a = KiloNetwork()
    a.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    b = _RefStyleKiloNet(4096)
    b.grid_resolution.copy_(torch.tensor([16, 16, 16]))
    missing = b.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    assert not missing.unexpected_keys and set(missing.missing_keys) <= {"aabb_min", "aabb_max", "grid_resolution"}
    occ = _t(_sphere_occ(256), dev)
    ra, rb = KiloRenderer(a.to(dev), default_kilo_cfg(), occ), KiloRenderer(b.to(dev), default_kilo_cfg(), occ)
    o = torch.tensor([[0.0, 0.0, 4.0]], device=dev).repeat(64, 1) + 0.01 * torch.randn(64, 3, device=dev)
    d = torch.tensor([[0.0, 0.0, -1.0]], device=dev).repeat(64, 1) + 0.05 * torch.randn(64, 3, device=dev)
    batch = {"xyz": o[None], "viewdirs": d[None]}
    oa, ob = ra(batch, is_training=False), rb(batch, is_training=False)
    assert ra.last_num_samples == rb.last_num_samples > 1000
    for k in ("rgb_map", "depth_map", "alpha_map"):
        assert torch.equal(oa[k], ob[k]), k
    with torch.no_grad():
        b.model.rgb_linear_biases.add_(0.5)
    oc = rb(batch, is_training=False)
    assert not torch.equal(oc["rgb_map"], ob["rgb_map"])


def test_march_random_configurations_both_write_paths(dev):
    """Seeded sweep over boxes (power-of-two edges or not), grid shapes, steps, near/far and occupancy densities:
    the mask hand-over and the grid-walking write pass agree bit for bit, counts match the oracle on a ray subset."""
    from nerf_amd import kilo_ops
    from oracle import kilo_oracle as K
    rng = np.random.default_rng(2024)
    o_all, d_all = _camera_rays(24, seed=9)
    for trial in range(10):
        res = tuple(int(x) for x in rng.choice([8, 12, 16, 33, 64], size=3))
        occ = (rng.uniform(size=res) < rng.choice([0.02, 0.3, 0.9])).astype(np.uint8)
        if trial % 2:
            half = rng.choice([0.5, 1.0, 2.0], size=3)                       # power-of-two edges
            box = [*(-half), *half]
        else:
            lo = -rng.uniform(0.6, 1.4, size=3)
            box = [*lo, *(lo + rng.uniform(1.1, 2.7, size=3))]
        box = [float(x) for x in box]
        step = float(rng.choice([0.003, 0.011, 0.07]))
        near, far = float(rng.uniform(0.5, 2.5)), float(rng.uniform(4.5, 7.0))
        a = kilo_ops.kilo_march(_t(o_all, dev), _t(d_all, dev), _t(occ, dev), box, near, far, step, reuse_masks=True)
        b = kilo_ops.kilo_march(_t(o_all, dev), _t(d_all, dev), _t(occ, dev), box, near, far, step, reuse_masks=False)
        for x, y in zip(a, b):
            assert torch.equal(x, y), trial
        sub = rng.permutation(len(o_all))[:48]
        ref = K.march(o_all[sub], d_all[sub], occ, box[:3], box[3:], near, far, step)
        assert np.array_equal(a[5].cpu().numpy()[sub], ref[5]), trial
        # the samples of the subset's rays, in ray order, are the oracle's
        off, cnt = a[4].cpu().numpy(), a[5].cpu().numpy()
        idx = np.concatenate([np.arange(off[r], off[r] + cnt[r]) for r in sub]) if cnt[sub].sum() else np.zeros(0, int)
        assert np.array_equal(a[0].cpu().numpy()[idx], ref[0]), trial
        assert np.array_equal(a[2].cpu().numpy()[idx], ref[2]) and np.array_equal(a[3].cpu().numpy()[idx], ref[3]), trial


@pytest.mark.parametrize("box", [[-1.0, -1.0, -1.0, 1.0, 1.0, 1.0], [-2.0, -0.5, -1.0, 2.0, 0.5, 1.0],
                                 [-1.3, -0.9, -1.0, 1.1, 1.2, 0.7], [-1.0, -1.0, -1.0, 1.0, 1.0, 0.5]])
def test_route_other_boxes_vs_oracle(dev, box):
    """Boxes other than the fixture's [-1, 1]^3, with equal, unequal, power-of-two and other edges: the oracle's
    (= the reference formula's) cell for points on, inside and outside the box."""
    from nerf_amd import kilo_ops
    from oracle import kilo_oracle as K
    rng = np.random.default_rng(5)
    lo, hi = np.array(box[:3], np.float32), np.array(box[3:], np.float32)
    pts = (lo + (hi - lo) * rng.uniform(-0.1, 1.1, size=(20000, 3))).astype(np.float32)
    grid = (np.arange(17, dtype=np.float32) / 16)[:, None] * (hi - lo) + lo          # points on the cell faces
    pts = np.concatenate([pts, np.stack(np.meshgrid(grid[:, 0], grid[:, 1], grid[:, 2], indexing="ij"), -1).reshape(-1, 3)])
    res = [16, 16, 16]
    got = kilo_ops.kilo_route(_t(pts, dev), box, res).cpu().numpy()
    _idx3, want = K.route(pts, box[:3], box[3:], res)
    assert np.array_equal(got, np.asarray(want).astype(np.int32))
