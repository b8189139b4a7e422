"""A training TRAJECTORY, not one step: ten consecutive TrainStep.step calls against the oracle's autograd +
clip_grad_value_(40) + torch.optim.Adam run over the same ten batches.

Reference semantics: the loop of src/train/trainers/trainer.py:43-109 (per batch: forward, loss, backward, clip, Adam
step), the loss of src/train/trainers/nerf.py:19-27, and the per-epoch learning-rate schedule of
src/utils/optimizer/lr_scheduler.py:75-79 (``base_lr * gamma ** (epoch / decay_epochs)``, stepped once per epoch:
train.py) -- here driven through ``nerf_amd.train.exponential_lr`` across an epoch boundary in the middle of the run.
One-step tests cannot see what a trajectory shows: weights that move (re-packed streams every step), Adam moments that
accumulate, the bias corrections of steps 2.., and for fp16x2 a loss scale chosen anew from every batch.

256 fixed rays (central window of one view), perturb = 1 with fresh uniforms per step; the oracle's ten steps take ~5 s
of CPU.  Bars: loss of every step within 1e-5 of the oracle's (measured: 1e-6).  Parameters after step 10: Adam's
update is lr * m / (sqrt(v) + eps), a NORMALISED step, so an entry whose gradient is known to a relative 1e-3 -- the
fp32 summation noise of these 65 k-term sums, see the float64 yardstick of tests/test_gpu_train_fullsize.py -- moves
differently by lr * 1e-3 per step in two equally correct fp32 implementations, and an entry at the noise floor by a full
+-lr.  The yardstick is therefore the reference against ITSELF: the oracle's ten steps run a second time with the 256
rays in another order (the loss is a mean over rays: same mathematics, another fp32 summation order).  The distance
hip <-> oracle must not exceed ``YARDSTICK_X[dtype]`` times the distance oracle <-> oracle-permuted at the median, the 99 %,
99.9 % quantiles and the maximum (entries whose first gradient is at least 1e-3 of its tensor's max), and no entry may be
further apart than the two trajectories' possible travel."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

R, NC, NI, STEPS = 256, 64, 128, 10
H = W = 800
EPOCH_OF_STEP = [0] * 5 + [100] * 5          # an epoch boundary after step 5: lr 5e-4 -> 5e-4 * 0.1 ** (100 / 500) = 3.15e-4
# hip <-> oracle may be this many times oracle <-> oracle-permuted.  3 = the factor of tests/test_gpu_parity.py's float64
# yardstick, for the two routes that carry the reference's 24-bit operands (measured: 2.3x at the median, 1.9x at the
# 99.9 % quantile, 1.2x at the maximum, fp32 MFMA and bf16x3 alike).  The fp16x2 route (22-bit operands, device-chosen loss
# scales) is measurably noisier over a trajectory -- 3.7x at the median, 3.5x at the maximum -- and gets the bound it
# meets, stated here: it is the narrower arithmetic and is not the credited configuration (bench.py: `also.*_fp16x2`).
YARDSTICK_X = {"fp32": 3.0, "bf16x3": 3.0, "fp16x2": 5.0}


@pytest.fixture(scope="module")
def run():
    from nerf_amd import ops
    from nerf_amd.camera import focal_from_angle, pose_spherical
    from nerf_amd.train import exponential_lr
    from nerf_amd.weights import make_state_dict
    from oracle import nerf_oracle as O
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(77)
    pix = (torch.randint(250, 550, (R,), generator=g) * W + torch.randint(250, 550, (R,), generator=g)).to(torch.int64)
    rays_o, rays_d = ops.raygen(pose_spherical(45.0), H, W, focal_from_angle(W), dev, pixel_index=pix.to(dev))
    gt = torch.rand(R, 3, generator=g)
    batches = [{"u": torch.rand(R, NI, generator=g), "u2": torch.rand(R, NI, generator=g),
                "u_coarse": torch.rand(R, NC, generator=g), "u_fine": torch.rand(R, NI, generator=g)} for _ in range(STEPS)]
    lrs = [exponential_lr(5e-4, e) for e in EPOCH_OF_STEP]
    assert lrs[0] == 5e-4 and abs(lrs[-1] - 5e-4 * 0.1 ** 0.2) < 1e-12
    sd0 = make_state_dict(0, 30.0, 2.0)
    o, d = rays_o.cpu(), rays_d.cpu()

    def oracle_trajectory(order):
        sd = {k: v.clone().requires_grad_(True) for k, v in sd0.items()}
        opt = torch.optim.Adam(list(sd.values()), lr=5e-4, eps=1e-8, weight_decay=0.0)
        losses, first_grads = [], None
        for b, lr in zip(batches, lrs):
            for grp in opt.param_groups:          # scheduler.step() at the epoch boundary (train.py)
                grp["lr"] = lr
            out = O.render(sd, o[order], d[order], u=b["u"][order].numpy(), u2=b["u2"][order].numpy(),
                           u_coarse=b["u_coarse"][order].numpy(), u_fine=b["u_fine"][order].numpy())
            opt.zero_grad(set_to_none=True)
            loss = O.loss(out, gt[order])
            loss.backward()
            if first_grads is None:
                first_grads = {k: v.grad.clone() for k, v in sd.items()}
            torch.nn.utils.clip_grad_value_(list(sd.values()), 40.0)
            opt.step()
            losses.append(float(loss.detach()))
        return losses, {k: v.detach().clone() for k, v in sd.items()}, first_grads

    losses, final, first_grads = oracle_trajectory(torch.arange(R))
    losses_p, final_p, _ = oracle_trajectory(torch.randperm(R, generator=g))     # the reference against itself
    solid = {k: (g1.abs() > 1e-3 * g1.abs().max()) for k, g1 in first_grads.items()}
    return {"dev": dev, "rays_o": rays_o, "rays_d": rays_d, "gt": gt, "batches": batches, "lrs": lrs, "sd0": sd0,
            "losses": losses, "final": final, "solid": solid, "losses_permuted": losses_p,
            "yardstick": _quantiles(torch.cat([(final_p[k] - final[k]).abs()[solid[k]].reshape(-1) for k in final]))}


def _quantiles(d: torch.Tensor):
    pick = d[torch.randperm(d.numel(), generator=torch.Generator().manual_seed(0))[:2000000]]
    return [float(torch.quantile(pick, q)) for q in (0.5, 0.99, 0.999)] + [float(d.max())]


@pytest.mark.parametrize("dtype", ["fp32", "bf16x3", "fp16x2"])
def test_ten_step_trajectory_matches_oracle_adam(run, dtype):
    from nerf_amd import ops
    from nerf_amd.network import Network
    from nerf_amd.train import TrainStep
    dev = run["dev"]
    net = Network()
    net.load_state_dict(run["sd0"])
    net.to(dev)
    ts = TrainStep(net, dev, forward_dtype=dtype, range_check_every=4)
    ts.log_scales = [] if dtype == "fp16x2" else None
    gt = run["gt"].to(dev)
    losses = []
    for b, lr in zip(run["batches"], run["lrs"]):
        ts.lr = lr                               # once per epoch in a trainer; the value is what matters here
        ts.step(run["rays_o"], run["rays_d"], gt, b["u"].to(dev), b["u2"].to(dev), b["u_coarse"].to(dev), b["u_fine"].to(dev))
        losses.append(float(ts.loss))
    worst_loss = max(abs(a - b) for a, b in zip(losses, run["losses"]))
    print(f"{dtype}: loss per step hip / oracle: " + " ".join(f"{a:.6f}/{b:.6f}" for a, b in zip(losses, run["losses"])))
    assert worst_loss < 1e-5, (worst_loss, losses, run["losses"])
    assert abs(losses[0] - losses[-1]) > 1e-4                      # the weights did move
    if dtype == "fp16x2":
        scales = []
        for blk_c, blk_f in ts.log_scales:
            scales.append(tuple(ops.loss_scale_exponent(int(blk[0].item()) & 0xFFFFFFFF) for blk in (blk_c, blk_f)))
        print("fp16x2 loss-scale exponents S (d raw * 2^S; coarse chain, fine chain) per step: " + " ".join(map(str, scales)))
        assert len(scales) == STEPS and all(0 < s <= 100 for pair in scales for s in pair)   # d raw ~ 1e-3 / R: scaled UP
    named = dict(net.named_parameters())
    travel = 2 * sum(run["lrs"]) + 1e-6
    diffs, total = [], 0
    for k, ref in run["final"].items():
        diff = (named[k].detach().cpu() - ref).abs()
        assert torch.isfinite(diff).all(), k
        assert float(diff.max()) <= travel, k
        diffs.append(diff[run["solid"][k]].reshape(-1))
        total += ref.numel()
        assert float((ref - run["sd0"][k]).abs().max()) > 1e-4, k    # every tensor trained
    d = torch.cat(diffs)
    got, yard = _quantiles(d), run["yardstick"]
    names = ("median", "99 %", "99.9 %", "max")
    print(f"{dtype}: parameters after {STEPS} steps over the {d.numel()} of {total} entries whose first gradient is resolved:")
    print("   |hip - oracle|              " + "  ".join(f"{n} {v:.1e}" for n, v in zip(names, got)))
    print("   |oracle permuted - oracle|  " + "  ".join(f"{n} {v:.1e}" for n, v in zip(names, yard)) +
          f"   (its losses differ by <= {max(abs(a - b) for a, b in zip(run['losses'], run['losses_permuted'])):.1e})")
    assert d.numel() > 0.5 * total
    for n, a, b in zip(names, got, yard):
        assert a <= YARDSTICK_X[dtype] * b + 1e-7, (n, a, b)


@pytest.mark.parametrize("dtype", ["fp32", "bf16x3"])
def test_autograd_route_follows_the_fused_route(run, dtype):
    """The route the reference's own trainer takes (trainers/nerf.py:19-27 + trainer.py:68-75: the drop-in Renderer under
    autograd, loss.backward(), clip_grad_value_(40), torch.optim.Adam.step()) against the fused TrainStep over the same
    five batches: the same HIP kernels compute the same forward and gradients, so the losses agree to fp32 rounding and
    the parameters to the fused Adam kernel's distance from torch's (2e-6 per step).  Every step re-packs the weight
    streams from parameters the optimiser has just changed."""
    from nerf_amd.config import default_cfg
    from nerf_amd.network import Network
    from nerf_amd.renderer import Renderer
    from nerf_amd.train import TrainStep
    dev = run["dev"]
    gt = run["gt"].to(dev)
    nets = []
    for _ in range(2):
        n = Network()
        n.load_state_dict(run["sd0"])
        nets.append(n.to(dev))
    ts = TrainStep(nets[0], dev, forward_dtype=dtype)
    r = Renderer(nets[1], default_cfg(perturb=1), mlp_dtype=dtype)
    opt = torch.optim.Adam(nets[1].parameters(), lr=5e-4, eps=1e-8)
    for k, (b, lr) in enumerate(zip(run["batches"][:5], run["lrs"][:5])):
        args = [b[n_].to(dev) for n_ in ("u", "u2", "u_coarse", "u_fine")]
        ts.lr = lr
        ts.step(run["rays_o"], run["rays_d"], gt, *args)
        for g in opt.param_groups:
            g["lr"] = lr
        out = r.render_rays_train(run["rays_o"], run["rays_d"], *args)
        loss = ((out["coarse_rgb_map"] - gt) ** 2).mean() + ((out["fine_rgb_map"] - gt) ** 2).mean()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        torch.nn.utils.clip_grad_value_(list(nets[1].parameters()), 40.0)
        opt.step()
        assert abs(float(loss.detach()) - float(ts.loss)) < 2e-6, (k, float(loss.detach()), float(ts.loss))
    d = torch.cat([(p - q).abs().reshape(-1) for p, q in zip(nets[0].parameters(), nets[1].parameters())]).cpu()
    got = _quantiles(d)
    print(f"{dtype}: autograd route vs fused route after 5 steps, |parameter difference|: median {got[0]:.1e}  99 % {got[1]:.1e}  "
          f"99.9 % {got[2]:.1e}  max {got[3]:.1e}")
    # Same gradients kernels, two Adam implementations (2e-6 apart per step on O(1) parameters): nearly every entry agrees
    # to that; the tail is entries whose gradient changed because a 1e-7 parameter difference after step 1 flipped a
    # fine-sample bin or a relu sign in a later forward -- Adam's normalised step turns that into up to lr per step.
    # One flipped bin moves ONE ray's contribution, 1 / 256 of every entry's gradient: lr * 0.4 % = 2e-6 per later step
    # for every entry.  Whether the five steps contain such a flip depends on the kernels' last bits: with the forward on
    # the 32x32x16 kernel they did not (median 2.6e-8, 99 % 3.1e-6), on the 16x16x32 kernel one occurs (6.5e-7, 2.1e-5) --
    # while the one-step gradients of the two routes agree to 1e-9 of 1e-2 with either kernel
    # (tools/debug/x3_routes_compare.py) and the forward's saved rows / sign words agree between the kernels to 3e-6 / 109
    # bits of 9e8 (tools/debug/x3_train_fwd_compare.py).  The bars allow for one flip.
    assert got[0] <= 2e-6 and got[1] <= 5e-5, got
    assert got[3] <= 2 * sum(run["lrs"][:5]), got
