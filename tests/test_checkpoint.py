"""Checkpoint / evaluator interop: our modules carry the reference's state-dict layout
(tests/golden/state_dict_layout.json, from the real reference networks), the reference's file format and
file-selection rule round-trip, and PSNR follows evaluators/nerf.py:22-25."""
import json
import os

import numpy as np
import pytest
import torch

from nerf_amd import checkpoint as ckpt
from nerf_amd.kilonerf import KiloNetwork
from nerf_amd.network import Network
from nerf_amd.weights import make_state_dict

LAYOUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "state_dict_layout.json")


@pytest.mark.parametrize("which,cls", [("nerf", Network), ("kilonerf", KiloNetwork)])
def test_state_dict_layout_matches_reference(which, cls):
    want = json.load(open(LAYOUT))[which]
    got = {k: {"shape": list(v.shape), "dtype": str(v.dtype)} for k, v in cls().state_dict().items()}
    assert got == want


def test_reference_format_round_trip(tmp_path):
    d = str(tmp_path / "trained_model")
    assert ckpt.resolve_checkpoint(d) is None and ckpt.load_network(Network(), d) == 0
    a = Network()
    a.load_state_dict(make_state_dict(3))
    for epoch in (9, 19, 29):
        ckpt.save_model(a, d, epoch)
    assert ckpt.resolve_checkpoint(d).endswith("29.pth")             # largest epoch when there is no latest.pth
    assert ckpt.resolve_checkpoint(d, epoch=9).endswith("9.pth")
    b = Network()
    b.load_state_dict(make_state_dict(4))
    ckpt.save_model(b, d, 34, last=True)
    assert ckpt.resolve_checkpoint(d).endswith("latest.pth")          # latest.pth wins (net_utils.py:365-366)
    raw = torch.load(ckpt.resolve_checkpoint(d), weights_only=True)
    assert set(raw) == {"net", "optim", "scheduler", "recorder", "epoch"} and raw["epoch"] == 34
    c = Network()
    assert ckpt.load_network(c, d) == 35
    for k, v in b.state_dict().items():
        assert torch.equal(c.state_dict()[k], v), k
    assert ckpt.load_network(c, d, epoch=19) == 20
    for k, v in a.state_dict().items():
        assert torch.equal(c.state_dict()[k], v), k
    for epoch in range(40, 46):                                       # only five numbered files are kept (:342-346)
        ckpt.save_model(a, d, epoch)
    assert len([n for n in os.listdir(d) if n != "latest.pth"]) == 5
    assert not os.path.exists(os.path.join(d, "9.pth"))


class Recorder:                                                       # an arbitrary pickled object, as a recorder might be
    pass


def test_unsafe_checkpoint_is_refused_not_unpickled(tmp_path):
    p = str(tmp_path / "latest.pth")
    torch.save({"net": Network().state_dict(), "recorder": Recorder(), "epoch": 1}, p)
    with pytest.raises(RuntimeError, match="weights_only"):
        ckpt.load_network(Network(), p)


def test_psnr_and_evaluator():
    rng = np.random.default_rng(0)
    H, W = 6, 5
    pred, gt = rng.uniform(size=(H * W, 3)).astype(np.float32), rng.uniform(size=(H * W, 3)).astype(np.float32)
    want = -10 * np.log(np.mean((pred.reshape(H, W, 3) - gt.reshape(H, W, 3)) ** 2)) / np.log(10)
    assert abs(ckpt.psnr(torch.from_numpy(pred), torch.from_numpy(gt)) - want) < 1e-5
    ev = ckpt.Evaluator()
    batch = {"H": torch.tensor(H), "W": torch.tensor(W), "gt_rgb": torch.from_numpy(gt)[None]}
    ev.evaluate({"fine_rgb_map": torch.from_numpy(pred)}, batch)
    ev.evaluate({"rgb_map": torch.from_numpy(pred)[None]}, batch)    # KiloNeRF's output dict (evaluators/kilonerf.py:55)
    s = ev.summarize()
    assert abs(s["mean_psnr"] - want) < 1e-5 and ev.summarize() == {}


def test_exponential_lr_matches_torch_scheduler_formula():
    """lr_scheduler.py:75-79 restated on a real torch optimiser: lr_e = base * gamma ** (e / decay_epochs)."""
    from nerf_amd.train import exponential_lr
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.Adam([p], lr=5e-4)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda e: 0.1 ** (e / 500))
    for epoch in range(0, 1200, 97):
        want = 5e-4 * 0.1 ** (epoch / 500)
        assert abs(exponential_lr(5e-4, epoch) - want) < 1e-15
    for epoch in range(3):
        assert abs(opt.param_groups[0]["lr"] - exponential_lr(5e-4, epoch)) < 1e-12
        opt.step()
        sched.step()
    assert abs(exponential_lr(5e-4, 100, 100, 0.9) - 4.5e-4) < 1e-12
