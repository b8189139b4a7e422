"""KiloNeRF data parallelism on the GPU: the compact per-cell gradient exchange (nrf_kilo_cells_gather / _scatter,
nerf_amd.parallel.allreduce_cell_gradients) against plain torch indexing, and -- two ranks over gloo sharing the
one GPU of the test box -- against the dense all-reduce of the flat gradient buffer (DDP's semantics,
src/train/trainers/trainer.py:14-21)."""
import os
import queue
import time
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def _tensors(dev, cells=64, seed=0):
    from nerf_amd.weights import KILO_TENSORS
    g = torch.Generator(device="cpu").manual_seed(seed)
    return [torch.randn((cells,) + tuple(shape), generator=g).to(dev) for _, shape, _ in KILO_TENSORS]


def test_cells_gather_scatter_roundtrip(dev):
    from nerf_amd import kilo_ops
    from nerf_amd._lib import NrfError, lib
    ts = _tensors(dev)
    rec_floats = int(lib().nrf_kilo_record_floats())
    assert rec_floats == sum(int(np.prod(t.shape[1:])) for t in ts) == 6212
    cells = torch.tensor([5, 0, 63, 17], dtype=torch.int32, device=dev)
    rec = kilo_ops.kilo_cells_gather(ts, cells)
    want = torch.cat([t.reshape(t.shape[0], -1)[cells.long()] for t in ts], 1)
    assert rec.shape == (4, rec_floats) and torch.equal(rec, want)
    # scatter into zeroed tensors: listed cells get their slices back, every other cell stays as it was
    out = [torch.full_like(t, -7.0) for t in ts]
    kilo_ops.kilo_cells_scatter(rec, cells, out)
    for t, o in zip(ts, out):
        assert torch.equal(o[cells.long()], t[cells.long()])
        rest = torch.ones(t.shape[0], dtype=torch.bool, device=dev)
        rest[cells.long()] = False
        assert torch.all(o[rest] == -7.0)
    # empty list is a no-op; argument checks
    empty = torch.empty(0, dtype=torch.int32, device=dev)
    assert kilo_ops.kilo_cells_gather(ts, empty).shape == (0, rec_floats)
    with pytest.raises(NrfError):
        kilo_ops.kilo_cells_gather(ts[:11], cells)
    with pytest.raises(NrfError):
        kilo_ops.kilo_cells_gather(ts, cells.long())
    with pytest.raises(NrfError):
        kilo_ops.kilo_cells_scatter(rec[:, :100].contiguous(), cells, out)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nerf_amd.kilo_train import KiloTrainStep
        from tests.test_gpu_kilo_backward import _net_and_renderer, _rays, _t
        dev = torch.device("cuda:0")
        o, d = _rays(10)
        rng = np.random.default_rng(10 + rank)                 # every rank its own rays and targets
        sel = rng.permutation(len(o))[:60]
        o, d = o[sel], d[sel]
        t_rgb = rng.uniform(size=(len(o), 3)).astype(np.float32)
        t_alpha = rng.uniform(size=(len(o), 1)).astype(np.float32)
        flats, grads, touched = [], [], None
        for sparse in (False, True):
            _sd, _net, r, _occ = _net_and_renderer(dev, seed=6)  # identical initial parameters on every rank
            step = KiloTrainStep(r, dev, lr=5e-4, l2_reg=1e-6, sparse_allreduce=sparse)
            assert step.world == world
            step.step(_t(o, dev), _t(d, dev), _t(t_rgb, dev), _t(t_alpha, dev), stage="distillation")
            flats.append(step.flat.clone()), grads.append(step.grad.clone())
            if sparse:
                touched = int((step.touched != 0).sum())
        # same summed gradient (the regulariser is associated differently: (g0 + r) + (g1 + r) vs (g0 + g1) + 2 r)
        gd, gs = grads
        scale = float(gd.abs().max())
        ok_grad = bool((gd - gs).abs().max() <= 1e-6 * scale)
        moved = (flats[0] - flats[1]).abs().max()
        ok_param = bool(moved <= 2e-7)
        q.put((rank, ok_grad, ok_param, touched, float((gd - gs).abs().max()), float(moved)))
    finally:
        dist.destroy_process_group()


def test_two_ranks_compact_exchange_equals_dense_allreduce(dev):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() * 7 + 3) % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res, deadline = [], time.time() + 150
    try:
        while len(res) < len(procs):
            try:
                res.append(q.get(timeout=1.0))
            except queue.Empty:
                dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
                assert not dead, f"a worker died with exit code {dead} before reporting"
                assert time.time() < deadline, "workers did not report within 150 s"
        for p in procs:
            p.join(timeout=60)
    finally:
        for p in procs:              # a failed or hung run leaves no worker behind (exact handles, no patterns)
            if p.is_alive():
                p.terminate()
                p.join(timeout=10)
    res.sort()
    for rank, ok_grad, ok_param, touched, dg, dp in res:
        assert ok_grad and ok_param, (rank, dg, dp)
        assert 0 < touched < 4096
