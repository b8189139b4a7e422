"""Parity at BASELINE.json's FULL sizes (800x800 = 640,000 rays, 64 + 128 samples) through size-independent
properties: the oracle cannot render a frame in test time (400 s), so the full-size runs are checked for
  * ray independence, bit for bit: any slice of the frame rendered on its own (other grid size, other tile ->
    workgroup mapping of the persistent kernels) gives the same bits as that slice of the whole-frame launch;
  * an oracle spot check: 48 rays scattered over the frame against oracle/nerf_oracle.py to the 1e-4 bar;
  * range / order invariants of every stage (sorted merged depths inside [near, far], 0 <= acc <= 1, colours
    in [0, 1] on the white background, coarse bins contained in the merged samples);
  * KiloNeRF: the same slice property for the packed path, sample count = sum of the slices' counts.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

H = W = 800
KEYS = ("coarse_rgb_map", "coarse_depth_map", "fine_rgb_map", "fine_depth_map", "alpha_map")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def frame(dev):
    """One full fp32 frame (the bench's configuration) + everything needed to re-render parts of it."""
    from nerf_amd import ops
    from nerf_amd.camera import focal_from_angle, test_set_poses
    from nerf_amd.config import default_cfg
    from nerf_amd.network import Network
    from nerf_amd.renderer import Renderer
    from nerf_amd.weights import make_state_dict
    net = Network()
    net.load_state_dict(make_state_dict(0, 30.0, 2.0))
    r = Renderer(net.to(dev), default_cfg(perturb=0))
    o, d = ops.raygen(test_set_poses(200)[3], H, W, focal_from_angle(W), dev)
    g = torch.Generator(device=dev).manual_seed(7)
    u = torch.rand((H * W, 128), device=dev, generator=g)
    u2 = torch.rand((H * W, 128), device=dev, generator=g)
    with torch.no_grad():
        out = r.render_rays(o, d, u, u2)
    torch.cuda.synchronize()
    return r, o, d, u, u2, out


def test_full_frame_ranges(frame):
    r, o, d, u, u2, out = frame
    for k in KEYS:
        assert out[k].shape[0] == H * W and torch.isfinite(out[k]).all(), k
    acc = out["alpha_map"]
    assert float(acc.min()) >= 0 and float(acc.max()) <= 1 + 1e-5
    for k in ("coarse_rgb_map", "fine_rgb_map"):
        assert float(out[k].min()) >= -1e-5 and float(out[k].max()) <= 1 + 1e-5      # white background: convex mix
    assert float(out["fine_depth_map"].min()) >= 0 and float(out["fine_depth_map"].max()) <= 6 + 1e-4
    assert float(acc.max()) > 0.5 and float(out["fine_rgb_map"].std()) > 0           # a rendered image, not a constant


@pytest.mark.parametrize("first,count", [(0, 4096), (123457, 65536), (640000 - 1000, 1000), (320000, 31)])
def test_slices_of_the_frame_are_bit_identical(frame, first, count):
    r, o, d, u, u2, out = frame
    sl = slice(first, first + count)
    with torch.no_grad():
        part = r.render_rays(o[sl].contiguous(), d[sl].contiguous(), u[sl].contiguous(), u2[sl].contiguous())
    for k in KEYS:
        assert torch.equal(part[k], out[k][sl]), k


def test_full_frame_oracle_spot_check(frame):
    from nerf_amd.weights import make_state_dict
    from oracle import nerf_oracle as O
    r, o, d, u, u2, out = frame
    idx = torch.linspace(0, H * W - 1, 48).long()
    idx[:8] = torch.tensor([0, 799, 320400, 320401, 400 * 800 + 399, 639999, 200 * 800 + 400, 600 * 800 + 400])
    sd = make_state_dict(0, 30.0, 2.0)
    di = idx.to(o.device)
    with torch.no_grad():
        ref = O.render(sd, o[di].cpu().numpy(), d[di].cpu().numpy(), u=u[di].cpu().numpy(), u2=u2[di].cpu().numpy(),
                       bins_table=r._bins.cpu().numpy())
    for k in KEYS:
        want = ref[k].numpy()
        got = out[k][di].cpu().numpy().reshape(want.shape)
        assert np.abs(got - want).max() < 1e-4, k


def test_full_size_resampling_invariants(frame, dev):
    from nerf_amd import ops
    r, o, d, u, u2, out = frame
    bins = r._bins
    g = torch.Generator(device=dev).manual_seed(3)
    w = torch.rand((H * W, 64), device=dev, generator=g) ** 6
    z = ops.sample_fine(w, bins, bins, u, u2)
    assert z.shape == (H * W, 192)
    assert bool((z[:, 1:] >= z[:, :-1]).all())                                        # sorted
    assert float(z.min()) >= 2.0 and float(z.max()) <= 6.0                            # inside [near, far]
    # the 64 coarse depths are a sub-multiset of every row: count of entries <= each bin is at least its rank
    cnt = (z[:4096, :, None] <= bins[None, None, :]).sum(1)
    assert bool((cnt >= torch.arange(1, 65, device=dev)[None]).all())
    # value checksum is permutation invariant: sum of merged row = sum of bins + sum of that row's fine samples
    z2, fine, _ = ops.sample_fine(w[:8192], bins, bins, u[:8192], u2[:8192], want_debug=True)
    assert torch.equal(z2, z[:8192])
    assert torch.allclose(z2.double().sum(1), bins.double().sum() + fine.double().sum(1), rtol=0, atol=1e-9)


@pytest.mark.parametrize("mlp_dtype", ["fp32", "bf16x3"])
def test_kilonerf_full_frame_slices(dev, mlp_dtype):
    from nerf_amd import ops
    from nerf_amd.camera import focal_from_angle, test_set_poses
    from nerf_amd.kilonerf import KiloNetwork, KiloRenderer, default_kilo_cfg
    from nerf_amd.weights import make_kilo_state_dict_arrays
    net = KiloNetwork()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in make_kilo_state_dict_arrays(0, 4096, 10.0, 0.5).items()},
                        strict=False)
    c = (torch.arange(256, device=dev, dtype=torch.float32) + 0.5) / 256 * 2 - 1
    occ = ((c[:, None, None] ** 2 + c[None, :, None] ** 2 + c[None, None, :] ** 2).sqrt() < 0.75).to(torch.uint8)
    r = KiloRenderer(net.to(dev), default_kilo_cfg(), occ, mlp_dtype=mlp_dtype)
    o, d = ops.raygen(test_set_poses(200)[11], H, W, focal_from_angle(W), dev)
    whole = r({"xyz": o[None], "viewdirs": d[None]}, is_training=False)
    n_whole = r.last_num_samples
    assert n_whole > 20_000_000
    acc = whole["alpha_map"]
    assert float(acc.min()) >= 0 and float(acc.max()) <= 1 + 1e-5 and torch.isfinite(whole["rgb_map"]).all()
    total = 0
    for first, count in [(0, 200000), (200000, 240000), (440000, 200000)]:
        sl = slice(first, first + count)
        part = r({"xyz": o[sl][None].contiguous(), "viewdirs": d[sl][None].contiguous()}, is_training=False)
        total += r.last_num_samples
        # per-sample values do not depend on which other rays share the launch (sort order inside a cell only
        # permutes independent points), so the packed path is slice-invariant bit for bit as well
        for k in ("rgb_map", "depth_map", "alpha_map"):
            assert torch.equal(part[k][0], whole[k][0, sl]), k
    assert total == n_whole
    if mlp_dtype == "bf16x3":   # the three-term bf16 tiny MLPs against the fp32 MFMA ones on the same 640,000 rays
        ref = KiloRenderer(net, default_kilo_cfg(), occ, mlp_dtype="fp32")({"xyz": o[None], "viewdirs": d[None]}, is_training=False)
        for k, tol in (("rgb_map", 2e-5), ("alpha_map", 2e-5), ("depth_map", 1e-4)):
            err = float((whole[k] - ref[k]).abs().max())
            print(f"KiloNeRF frame, bf16x3 vs fp32 tiny MLPs, {k}: {err:.2e}")
            assert err < tol, (k, err)


def test_sharded_view_slices_reassemble(frame, dev):
    """Renderer.render_view_sharded on world = 1 (no process group) equals the plain frame for the same uniforms'
    generator; and rank slices rendered without the gather are the matching slices of the ray-generated frame."""
    from nerf_amd import ops
    from nerf_amd.camera import focal_from_angle, test_set_poses
    from nerf_amd.parallel import shard_range
    r = frame[0]
    pose, focal = test_set_poses(200)[3], focal_from_angle(W)
    g = torch.Generator(device=dev).manual_seed(11)
    whole, span = r.render_view_sharded(pose, H, W, focal, 0, 1, dev, generator=g)
    assert span == (0, H * W) and whole["rgb_map"].shape == (H * W, 3)
    o, d = ops.raygen(pose, H, W, focal, dev)
    for rank in (0, 3):
        b, e = shard_range(H * W, rank, 4)
        g = torch.Generator(device=dev).manual_seed(100 + rank)
        part, sp = r.render_view_sharded(pose, H, W, focal, rank, 4, dev, gather=False, generator=g)
        assert sp == (b, e)
        g = torch.Generator(device=dev).manual_seed(100 + rank)
        u = torch.rand((e - b, 128), device=dev, generator=g)
        u2 = torch.rand((e - b, 128), device=dev, generator=g)
        with torch.no_grad():
            ref = r.render_rays(o[b:e].contiguous(), d[b:e].contiguous(), u, u2)
        for k in KEYS:
            assert torch.equal(part[k], ref[k]), k


# ---- bf16 MLP at full size (BASELINE configs[4]: the 200-view set is 200 of these frames) -----------------------------
@pytest.fixture(scope="module")
def frame_bf16(dev):
    from nerf_amd import ops
    from nerf_amd.camera import focal_from_angle, test_set_poses
    from nerf_amd.config import default_cfg
    from nerf_amd.network import Network
    from nerf_amd.renderer import Renderer
    from nerf_amd.weights import make_state_dict
    net = Network()
    net.load_state_dict(make_state_dict(0, 30.0, 2.0))
    r = Renderer(net.to(dev), default_cfg(perturb=0), mlp_dtype="bf16")
    o, d = ops.raygen(test_set_poses(200)[57], H, W, focal_from_angle(W), dev)
    g = torch.Generator(device=dev).manual_seed(8)
    u = torch.rand((H * W, 128), device=dev, generator=g)
    u2 = torch.rand((H * W, 128), device=dev, generator=g)
    with torch.no_grad():
        out = r.render_rays(o, d, u, u2)
    torch.cuda.synchronize()
    return r, o, d, u, u2, out


@pytest.mark.parametrize("first,count", [(0, 4096), (200001, 65536), (640000 - 1000, 1000), (320000, 31)])
def test_bf16_slices_of_the_frame_are_bit_identical(frame_bf16, first, count):
    r, o, d, u, u2, out = frame_bf16
    for k in KEYS:
        assert out[k].shape[0] == H * W and torch.isfinite(out[k]).all(), k
    sl = slice(first, first + count)
    with torch.no_grad():
        part = r.render_rays(o[sl].contiguous(), d[sl].contiguous(), u[sl].contiguous(), u2[sl].contiguous())
    for k in KEYS:
        assert torch.equal(part[k], out[k][sl]), k


def test_bf16_full_frame_oracle_spot_check(frame_bf16):
    """48 rays of the bf16 frame against (a) the oracle with the kernel's rounding points (bf16 weights and layer inputs,
    fp32 accumulate: oracle.render(bf16=True)) and (b) the fp32 oracle = the reference, at the stated bf16 tolerance of
    tests/test_gpu_bf16.py (rgb / acc 3e-3, depth 1e-2)."""
    from nerf_amd.weights import make_state_dict
    from oracle import nerf_oracle as O
    r, o, d, u, u2, out = frame_bf16
    idx = torch.linspace(0, H * W - 1, 48).long()
    idx[:8] = torch.tensor([0, 799, 320400, 320401, 400 * 800 + 399, 639999, 200 * 800 + 400, 600 * 800 + 400])
    sd = make_state_dict(0, 30.0, 2.0)
    di = idx.to(o.device)
    args = dict(u=u[di].cpu().numpy(), u2=u2[di].cpu().numpy(), bins_table=r._bins.cpu().numpy())
    with torch.no_grad():
        ref16 = O.render(sd, o[di].cpu().numpy(), d[di].cpu().numpy(), bf16=True, **args)
        ref32 = O.render(sd, o[di].cpu().numpy(), d[di].cpu().numpy(), **args)
    for k in KEYS:
        got = out[k][di].cpu().numpy()
        tol = 1e-2 if "depth" in k else 3e-3
        e16 = float(np.abs(got - ref16[k].numpy().reshape(got.shape)).max())
        e32 = float(np.abs(got - ref32[k].numpy().reshape(got.shape)).max())
        print(f"bf16 frame spot check {k}: vs bf16 oracle {e16:.2e}, vs fp32 oracle {e32:.2e}")
        assert e32 < tol, (k, e32)
        assert e16 < tol, (k, e16)


# ---- the split-operand MLPs at full size: the fp32 headline's kernels (bf16x3: three bf16 terms per operand; fp16x2: two
# fp16 terms; both fp32-class) ---------------------------------------------------------------------------------------
@pytest.fixture(scope="module", params=["bf16x3", "fp16x2"])
def frame_x3(dev, request):
    from nerf_amd import ops
    from nerf_amd.camera import focal_from_angle, test_set_poses
    from nerf_amd.config import default_cfg
    from nerf_amd.network import Network
    from nerf_amd.renderer import Renderer
    from nerf_amd.weights import make_state_dict
    net = Network()
    net.load_state_dict(make_state_dict(0, 30.0, 2.0))
    r = Renderer(net.to(dev), default_cfg(perturb=0), mlp_dtype=request.param)
    o, d = ops.raygen(test_set_poses(200)[3], H, W, focal_from_angle(W), dev)      # the fp32 frame's view and uniforms
    g = torch.Generator(device=dev).manual_seed(7)
    u = torch.rand((H * W, 128), device=dev, generator=g)
    u2 = torch.rand((H * W, 128), device=dev, generator=g)
    with torch.no_grad():
        out = r.render_rays(o, d, u, u2)
    torch.cuda.synchronize()
    return r, o, d, u, u2, out


@pytest.mark.parametrize("first,count", [(0, 4096), (123457, 65536), (640000 - 1000, 1000), (320000, 31)])
def test_x3_slices_of_the_frame_are_bit_identical(frame_x3, first, count):
    r, o, d, u, u2, out = frame_x3
    for k in KEYS:
        assert out[k].shape[0] == H * W and torch.isfinite(out[k]).all(), k
    sl = slice(first, first + count)
    with torch.no_grad():
        part = r.render_rays(o[sl].contiguous(), d[sl].contiguous(), u[sl].contiguous(), u2[sl].contiguous())
    for k in KEYS:
        assert torch.equal(part[k], out[k][sl]), k


def test_x3_full_frame_oracle_spot_check(frame_x3):
    """The fp32 frame's 48-ray spot check at the fp32 bar (1e-4 abs) against the fp32 oracle = the reference."""
    from nerf_amd.weights import make_state_dict
    from oracle import nerf_oracle as O
    r, o, d, u, u2, out = frame_x3
    idx = torch.linspace(0, H * W - 1, 48).long()
    idx[:8] = torch.tensor([0, 799, 320400, 320401, 400 * 800 + 399, 639999, 200 * 800 + 400, 600 * 800 + 400])
    sd = make_state_dict(0, 30.0, 2.0)
    di = idx.to(o.device)
    with torch.no_grad():
        ref = O.render(sd, o[di].cpu().numpy(), d[di].cpu().numpy(), u=u[di].cpu().numpy(), u2=u2[di].cpu().numpy(),
                       bins_table=r._bins.cpu().numpy())
    for k in KEYS:
        want = ref[k].numpy()
        got = out[k][di].cpu().numpy().reshape(want.shape)
        err = float(np.abs(got - want).max())
        print(f"{r.mlp_dtype} frame spot check {k}: {err:.2e}")
        assert err < 1e-4, (k, err)


def test_x3_frame_against_the_fp32_frame(frame, frame_x3):
    """The two kernels on the same view and uniforms, all 640,000 rays: away from the rays whose fine samples fell in
    another bin (last-bit differences of the coarse weights) the maps agree to the fp32 bar."""
    out32, out3 = frame[5], frame_x3[5]
    worst = {}
    for k in KEYS:
        diff = (out32[k].reshape(H * W, -1) - out3[k].reshape(H * W, -1)).abs().amax(1)
        worst[k] = float(diff.max())
        frac = float((diff > 1e-4).float().mean())
        print(f"{frame_x3[0].mlp_dtype} vs fp32 frame {k}: max {worst[k]:.2e}, rays over 1e-4: {frac:.2e}")
        assert frac < 2e-4, (k, frac)
    assert worst["coarse_rgb_map"] < 1e-4 and worst["coarse_depth_map"] < 1e-4     # no resampling before the coarse maps
