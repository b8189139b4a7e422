"""`renderer.estimator` (nerf_amd/estimator.py): the object trainer.py:58-66 drives every iteration.  CPU logic only
here (a stand-in owner, a closed-form density); the trainer's call sequence against the real drop-in renderer on the
HIP kernels is tests/test_gpu_trainer_sequence.py."""
import pytest
import torch

from nerf_amd.estimator import GridEstimator


class _Owner:
    def __init__(self, res):
        self.training = True
        self.occ_grid = None
        self.res = res

    def set_occ_grid(self, g):
        assert list(g.shape) == self.res
        self.occ_grid = g.to(torch.uint8).contiguous()


def _sphere_sigma(x):                         # density 5 inside |x| < 0.5, 0 outside
    return (x.norm(dim=-1) < 0.5).float() * 5.0


def test_warmup_update_marks_the_dense_cells():
    torch.manual_seed(0)
    res = [16, 16, 16]
    o = _Owner(res)
    e = GridEstimator(o, [-1, -1, -1, 1, 1, 1], res)
    calls = []

    def fn(x):
        calls.append(x.shape[0])
        assert float(x.min()) >= -1 and float(x.max()) <= 1
        return _sphere_sigma(x)

    assert e.update_every_n_steps(step=0, occ_eval_fn=fn) is True
    assert calls == [16 ** 3]                                      # warm-up: every cell, one jittered point each
    c = (torch.arange(16, dtype=torch.float32) + 0.5) / 16 * 2 - 1
    r = (c[:, None, None] ** 2 + c[None, :, None] ** 2 + c[None, None, :] ** 2).sqrt()
    half_diag = (3 ** 0.5) / 16
    g = o.occ_grid.bool()
    assert g[r < 0.5 - half_diag].all() and not g[r > 0.5 + half_diag].any()
    assert e.binaries.shape == (1, 16, 16, 16) and e.binaries.dtype == torch.bool
    assert e.update_every_n_steps(step=5, occ_eval_fn=fn) is False and len(calls) == 1     # only every n = 16 steps
    assert e.update_every_n_steps(step=16, occ_eval_fn=fn) is True


def test_after_warmup_samples_a_quarter_plus_the_occupied_cells_and_decays():
    torch.manual_seed(1)
    res = [8, 8, 8]
    o = _Owner(res)
    e = GridEstimator(o, [-1, -1, -1, 1, 1, 1], res)
    e.update_every_n_steps(step=0, occ_eval_fn=_sphere_sigma)
    occupied = int(o.occ_grid.sum())
    assert occupied > 0
    seen = []
    e.update_every_n_steps(step=512, occ_eval_fn=lambda x: (seen.append(x.shape[0]), torch.zeros(x.shape[0]))[1])
    assert seen == [8 ** 3 // 4 + min(occupied, 8 ** 3 // 4)]
    pos = e.occs[e.occs > 0]
    assert float(pos.min()) == pytest.approx(5.0 * 0.95)           # EMA of a re-sampled cell: max(old * decay, new = 0)
    assert float(pos.max()) <= 5.0                                 # cells this round did not draw keep their value
    # the density vanished: every cell is drawn with probability >= 1/4 per refresh, so 400 refreshes decay them all
    for k in range(400):
        e.update_every_n_steps(step=528 + 16 * k, occ_eval_fn=lambda x: torch.zeros(x.shape[0]))
    assert float(e.occs.max()) < 1.0


def test_only_while_training_and_only_when_active():
    o = _Owner([4, 4, 4])
    e = GridEstimator(o, [-1, -1, -1, 1, 1, 1], [4, 4, 4], active=lambda: False)
    assert e.update_every_n_steps(step=0, occ_eval_fn=lambda x: 1 / 0) is False       # inactive: nothing evaluated
    o.training = False
    with pytest.raises(RuntimeError, match="training"):
        e.update_every_n_steps(step=0, occ_eval_fn=_sphere_sigma)


def test_binaries_setter_feeds_the_owner():
    o = _Owner([4, 4, 4])
    e = GridEstimator(o, [-1, -1, -1, 1, 1, 1], [4, 4, 4])
    g = torch.zeros(4, 4, 4, dtype=torch.bool)
    g[1, 2, 3] = True
    e.binaries = g[None]                                             # volumn_renderer.py:54
    assert o.occ_grid.dtype == torch.uint8 and int(o.occ_grid.sum()) == 1 and bool(e.binaries[0, 1, 2, 3])
