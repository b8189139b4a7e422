#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference (build container only).

Usage:  python tools/gen_golden.py            (from /root/repo; needs /root/reference)

The reference cannot travel to the GPU box, so its outputs are frozen here as small data
fixtures (inputs + expected outputs, no reference source).  Recipe:
cwd=/root/reference, argv set before ``import src.config``, empty ``nerfacc`` stub module
(volume_renderer.py:6 imports it at top level; the classic path never touches it), Renderer
built with ``__new__`` so the nerfacc estimator of ``__init__`` is skipped.

Weights are NOT stored: they come from nerf_amd.weights.make_state_dict(seed, gain, bias)
(numpy PCG64) and are loaded into the reference with ``load_state_dict``.
"""
import os
import sys
import types
import json

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True

os.chdir(REF)
sys.path.insert(0, REF)
sys.argv = ["gen_golden", "--cfg_file", "configs/nerf/lego.yaml"]
sys.modules["nerfacc"] = types.ModuleType("nerfacc")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

from src.config import cfg  # noqa: E402,F401
from src.models.nerf.network import Network  # noqa: E402
from src.models.nerf.renderer.volume_renderer import Renderer  # noqa: E402

from nerf_amd.weights import make_state_dict  # noqa: E402
from nerf_amd.camera import pose_spherical, LEGO_CAMERA_ANGLE_X  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def make_renderer(net, perturb):
    r = Renderer.__new__(Renderer)
    nn.Module.__init__(r)
    r.net = net
    r.N_samples, r.N_importance = 64, 128
    r.white_bkgd, r.perturb, r.raw_noise_std = 1, perturb, 0
    r.near, r.far, r.chunk_size = 2, 6, 4096
    r.coarse_samples = r.fine_samples = r.t_bins = None
    return r


def record_network(r):
    calls = []
    orig = r._run_network

    def rec(*a, **k):
        out = orig(*a, **k)
        calls.append(out.detach().clone())
        return out
    r._run_network = rec
    return calls


def random_rays(R, seed):
    g = torch.Generator().manual_seed(seed)
    o = torch.tensor([0.0, 0.0, 4.0]) + 0.1 * torch.randn(R, 3, generator=g)
    d = torch.randn(R, 3, generator=g)
    gt = torch.rand(R, 3, generator=g)
    return o.float(), d.float(), gt.float()


def camera_rays(R, seed):
    """R random pixels of the theta=30 deg view of the synthetic orbit (fp64 -> fp32)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    H = W = 800
    c2w = pose_spherical(30.0)
    focal = 0.5 * W / np.tan(0.5 * LEGO_CAMERA_ANGLE_X)
    u = rng.integers(0, W, R)
    v = rng.integers(0, H, R)
    d = np.stack([(u - W / 2) / focal, -(v - H / 2) / focal, -np.ones_like(u)], -1)
    d = (c2w[:3, :3] @ d.T).T
    o = np.broadcast_to(c2w[:3, 3], d.shape)
    gt = rng.random((R, 3))
    return (torch.tensor(o, dtype=torch.float32), torch.tensor(d, dtype=torch.float32),
            torch.tensor(gt, dtype=torch.float32))


def t2n(d):
    return {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in d.items()}


def forward_case(name, rays, wseed, gain, bias, rand_seed):
    o, d, gt = rays
    R = o.shape[0]
    net = Network()
    net.load_state_dict(make_state_dict(wseed, gain, bias))
    net.eval()
    r = make_renderer(net, perturb=0)
    calls = record_network(r)
    batch = {"xyz": o[None], "viewdirs": d[None], "gt_rgb": gt[None]}
    with torch.no_grad():
        rgb_c, depth_c, w_c, acc_c = r.render_coarse(batch, False)
        torch.manual_seed(rand_seed)
        rgb_f, depth_f, acc_f = r.render_fine(batch, w_c, False)
        # the two draws importance_sampling made (volume_renderer.py:379, :389)
        torch.manual_seed(rand_seed)
        u = torch.rand((R, 128))
        u2 = torch.rand(R, 128)
        # bin indices: same library calls as :374-384 on the same weights
        pdf = w_c / (torch.sum(w_c, dim=-1, keepdim=True) + 1e-10)
        cdf = torch.cat([torch.zeros(R, 1), torch.cumsum(pdf, -1)], -1)
        idx = torch.clamp(torch.searchsorted(cdf, u, right=True) - 1, 0, 63)
        lo = torch.gather(r.t_bins, 1, idx)
        hi = torch.gather(r.t_bins, 1, (idx + 1).clamp(max=63))
        assert torch.equal(lo + (hi - lo) * u2, r.fine_samples), "u/u2 replay mismatch"
        z_fine = torch.sort(torch.cat([r.coarse_samples, r.fine_samples], -1), -1)[0]
    fx = dict(xyz=o, viewdirs=d, gt_rgb=gt, t_bins=r.t_bins[0], raw_coarse=calls[0], raw_fine=calls[1],
              coarse_rgb_map=rgb_c, coarse_depth_map=depth_c, w_coarse=w_c, acc_coarse=acc_c,
              u=u, u2=u2, bin_indices=idx, cdf=cdf, fine_samples=r.fine_samples, z_fine=z_fine,
              fine_rgb_map=rgb_f, fine_depth_map=depth_f, alpha_map=acc_f,
              weight_seed=wseed, density_gain=gain, density_bias=bias)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **t2n(fx))
    print(name, "coarse acc mean", float(acc_c.mean()), "fine rgb mean", float(rgb_f.mean()))


def train_case(name, rays, wseed, gain, bias, rand_seed):
    """perturb=1, is_training=True, loss + backward (trainers/nerf.py:19-27)."""
    o, d, gt = rays
    R = o.shape[0]
    net = Network()
    net.load_state_dict(make_state_dict(wseed, gain, bias))
    net.train()
    r = make_renderer(net, perturb=1)
    calls = record_network(r)
    batch = {"xyz": o[None], "viewdirs": d[None], "gt_rgb": gt[None]}
    torch.manual_seed(rand_seed)
    rgb_c, depth_c, w_c, acc_c = r.render_coarse(batch, True)
    rgb_f, depth_f, acc_f = r.render_fine(batch, w_c, True)
    mse = nn.MSELoss(reduction="mean")
    loss = mse(rgb_c, gt[None]) + mse(rgb_f, gt[None])
    loss.backward()
    torch.manual_seed(rand_seed)
    u_coarse = torch.rand(R, 64)       # add_perturbation in render_coarse (:348)
    u = torch.rand((R, 128))           # :379
    u2 = torch.rand(R, 128)            # :389
    u_fine = torch.rand(R, 128)        # add_perturbation in render_fine (:348)
    fx = dict(xyz=o, viewdirs=d, gt_rgb=gt, t_bins=r.t_bins[0], u_coarse=u_coarse, u=u, u2=u2, u_fine=u_fine,
              z_coarse=r.coarse_samples, fine_samples=r.fine_samples,
              coarse_rgb_map=rgb_c, coarse_depth_map=depth_c, w_coarse=w_c,
              fine_rgb_map=rgb_f, fine_depth_map=depth_f, alpha_map=acc_f, loss=loss,
              raw_coarse=calls[0], raw_fine=calls[1],
              weight_seed=wseed, density_gain=gain, density_bias=bias)
    names, norms = [], []
    for k, p in net.named_parameters():
        names.append(k)
        norms.append(float(p.grad.double().norm()))
    fx["grad_names"] = np.array(names)
    fx["grad_norms"] = np.array(norms, dtype=np.float64)
    for pre in ("model", "model_fine"):
        m = getattr(net, pre)
        fx[f"g.{pre}.alpha_linear.weight"] = m.alpha_linear.weight.grad
        fx[f"g.{pre}.alpha_linear.bias"] = m.alpha_linear.bias.grad
        fx[f"g.{pre}.rgb_linear.weight"] = m.rgb_linear.weight.grad
        fx[f"g.{pre}.rgb_linear.bias"] = m.rgb_linear.bias.grad
        fx[f"g.{pre}.pts_linears.0.bias"] = m.pts_linears[0].bias.grad
        fx[f"g.{pre}.pts_linears.0.weight"] = m.pts_linears[0].weight.grad
        fx[f"g.{pre}.pts_linears.5.weight[:, :72]"] = m.pts_linears[5].weight.grad[:, :72]
        fx[f"g.{pre}.pts_linears.7.weight[:8]"] = m.pts_linears[7].weight.grad[:8]
        fx[f"g.{pre}.views_linears.0.weight[:, 248:]"] = m.views_linears[0].weight.grad[:, 248:]
        fx[f"g.{pre}.feature_linear.bias"] = m.feature_linear.bias.grad
    # one Adam step (src/train/optimizer.py:9-29 defaults: lr 5e-4, eps 1e-8, wd 0) after clip 40
    # (src/train/trainers/trainer.py:74-75)
    params = [p for p in net.parameters() if p.requires_grad]
    torch.nn.utils.clip_grad_value_(params, 40)
    opt = torch.optim.Adam(params, lr=5e-4, weight_decay=0.0, eps=1e-8)
    opt.step()
    for pre in ("model", "model_fine"):
        m = getattr(net, pre)
        fx[f"adam.{pre}.alpha_linear.weight"] = m.alpha_linear.weight.detach()
        fx[f"adam.{pre}.pts_linears.0.bias"] = m.pts_linears[0].bias.detach()
        fx[f"adam.{pre}.rgb_linear.weight"] = m.rgb_linear.weight.detach()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **t2n(fx))
    print(name, "loss", float(loss))


def pe_case():
    net = Network()
    g = torch.Generator().manual_seed(5)
    x = torch.cat([torch.randn(24, 3, generator=g) * 2.0,
                   torch.tensor([[0.0, 0.0, 0.0], [1.0, -1.0, 6.0], [-5.5, 3.25, 0.001],
                                 [1e-8, -1e-8, 2.5], [6.0, 6.0, 6.0], [-6.0, 0.5, 3.14159265]])]).float()
    dd = torch.randn(30, 3, generator=g)
    dd = (dd / dd.norm(dim=-1, keepdim=True)).float()
    np.savez_compressed(os.path.join(OUT, "posenc.npz"), x=x.numpy(), d=dd.numpy(),
                        ex=net.embed_fn(x).numpy(), ed=net.embeddirs_fn(dd).numpy())
    print("posenc", net.embed_fn(x).shape, net.embeddirs_fn(dd).shape)


def network_points_case():
    """Network.forward on a flat (P,3) batch with per-point dirs (network.py:182-188)."""
    net = Network()
    net.load_state_dict(make_state_dict(3, 1.0, 0.0))
    g = torch.Generator().manual_seed(9)
    p = (torch.rand(96, 3, generator=g) * 2 - 1).float()
    dd = torch.randn(96, 3, generator=g).float()
    with torch.no_grad():
        out_c = net(p, dd)
        out_f = net(p, dd, "fine")
    np.savez_compressed(os.path.join(OUT, "network_points.npz"), pts=p.numpy(), dirs=dd.numpy(),
                        raw_coarse=out_c.numpy(), raw_fine=out_f.numpy(), weight_seed=3)
    print("network_points", out_c.shape)


def importance_edge_case():
    """importance_sampling on crafted weights with crafted uniforms (torch.rand patched)."""
    net = Network()
    r = make_renderer(net, 0)
    R = 8
    s = torch.linspace(0., 1., steps=64)
    r.t_bins = (2 * torch.ones(R, 1)) * (1 - s) + (6 * torch.ones(R, 1)) * s
    g = torch.Generator().manual_seed(11)
    w = torch.zeros(R, 64)
    w[1, 10] = 0.7                                 # single spike
    w[2] = torch.rand(64, generator=g)             # generic
    w[3] = torch.rand(64, generator=g) ** 8        # peaky
    w[4, 63] = 1.0                                 # all mass in last bin
    w[5, 0] = 1e-12                                # below the 1e-10 floor
    w[6] = 1.0 / 64                                # uniform
    w[7, 20:24] = torch.tensor([0.25, 0.25, 0.25, 0.25])   # exact quarter steps
    u = torch.rand(R, 128, generator=g)
    u[:, 0] = 0.0
    u[:, 1] = float(np.nextafter(np.float32(1.0), np.float32(0.0)))
    u[:, 2] = 0.25
    u[:, 3] = 0.5
    u[:, 4] = 0.75
    u2 = torch.rand(R, 128, generator=g)
    u2[:, 5] = 0.0
    seq = [u, u2]
    orig = torch.rand
    torch.rand = lambda *a, **k: seq.pop(0)
    try:
        fs = r.importance_sampling(w)
    finally:
        torch.rand = orig
    np.savez_compressed(os.path.join(OUT, "importance_edge.npz"), w=w.numpy(), t_bins=r.t_bins[0].numpy(),
                        u=u.numpy(), u2=u2.numpy(), fine_samples=fs.numpy())
    print("importance_edge", fs.shape, float(fs[0].min()), float(fs[0].max()))


def composite_edge_case():
    """render_coarse on crafted raw network outputs (the network is replaced by a table)."""
    net = Network()
    r = make_renderer(net, 0)
    R = 6
    g = torch.Generator().manual_seed(13)
    raw = torch.randn(R, 64, 4, generator=g)
    raw[0, :, 3] = -1.0                     # sigma <= 0 everywhere: acc 0, pure white
    raw[1, 0, 3] = 1e4                      # opaque first sample
    raw[2, :, 3] = 30.0 * torch.rand(64, generator=g)
    raw[3, :, 3] = 0.0
    raw[3, 63, 3] = 1e-3                    # only the last (delta=1e10) sample is hit
    raw[4, :, :3] = 20.0 * torch.randn(64, 3, generator=g)     # saturated sigmoid
    raw[5, :, 3] = 200.0                    # transmittance underflows fast
    r._run_network = lambda pts, vd, *a: raw
    o, d, gt = random_rays(R, 17)
    batch = {"xyz": o[None], "viewdirs": d[None], "gt_rgb": gt[None]}
    with torch.no_grad():
        rgb, depth, w, acc = r.render_coarse(batch, False)
    np.savez_compressed(os.path.join(OUT, "composite_edge.npz"), raw=raw.numpy(), z=r.coarse_samples.numpy(),
                        rgb_map=rgb.numpy(), depth_map=depth.numpy(), weights=w.numpy(), acc=acc.numpy())
    print("composite_edge", rgb[0], acc[:, 0])


def raygen_case():
    """Run the reference Dataset on a throw-away synthetic scene (1 frame, blank image)."""
    import tempfile
    from PIL import Image
    from src.datasets.nerf.blender import Dataset
    c2w = pose_spherical(57.0, -30.0)
    with tempfile.TemporaryDirectory() as tmp:
        root = os.path.join(tmp, "lego")
        os.makedirs(os.path.join(root, "test"))
        Image.fromarray(np.zeros((8, 8, 4), np.uint8)).save(os.path.join(root, "test", "r_0.png"))
        with open(os.path.join(root, "transforms_test.json"), "w") as f:
            json.dump({"camera_angle_x": LEGO_CAMERA_ANGLE_X,
                       "frames": [{"file_path": "./test/r_0", "rotation": 0.0,
                                   "transform_matrix": c2w.tolist()}]}, f)
        ds = Dataset(data_root=root, split="test", input_ratio=1.0, H=800, W=800, cams=[0, -1, 1])
        b = ds[0]
    xyz, vd = b["xyz"].numpy(), b["viewdirs"].numpy()
    rng = np.random.Generator(np.random.PCG64(21))
    sel = np.unique(np.concatenate([np.arange(0, 16), np.arange(799 * 800, 800 * 800),
                                    np.arange(400 * 800 + 392, 400 * 800 + 408),
                                    rng.integers(0, 640000, 480)]))
    np.savez_compressed(os.path.join(OUT, "raygen.npz"), c2w=c2w, camera_angle_x=LEGO_CAMERA_ANGLE_X,
                        H=800, W=800, ray_index=sel, xyz=xyz[sel], viewdirs=vd[sel])
    print("raygen", xyz.shape, len(sel))


if __name__ == "__main__":
    torch.set_grad_enabled(True)
    pe_case()
    raygen_case()
    network_points_case()
    importance_edge_case()
    composite_edge_case()
    forward_case("forward_plumbing64", random_rays(64, 1), wseed=0, gain=1.0, bias=0.0, rand_seed=100)
    forward_case("forward_camera64", camera_rays(64, 2), wseed=1, gain=40.0, bias=3.5, rand_seed=101)
    forward_case("forward_empty16", camera_rays(16, 5), wseed=1, gain=40.0, bias=0.0, rand_seed=104)   # sigma<0 everywhere
    forward_case("forward_dense32", camera_rays(32, 3), wseed=2, gain=200.0, bias=2.0, rand_seed=102)
    train_case("train_camera32", camera_rays(32, 4), wseed=1, gain=40.0, bias=3.5, rand_seed=103)
