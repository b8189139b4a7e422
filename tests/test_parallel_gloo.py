"""N > 1 path on CPU: 2 ranks over gloo assemble one frame from ray shards (no GPU needed)."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from nerf_amd.parallel import render_frame_sharded, shard_range, view_schedule


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 640000, 640001):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def test_view_schedule_round_robin():
    items = [view_schedule(200, r, 8, tiles_per_view=2) for r in range(8)]
    assert sorted(sum(items, [])) == [(v, t) for v in range(200) for t in range(2)]
    assert max(map(len, items)) - min(map(len, items)) <= 1


def _fake_render(first, count):
    i = torch.arange(first, first + count, dtype=torch.float32)
    return {"fine_rgb_map": torch.stack([i, 2 * i, 3 * i], -1), "fine_depth_map": i * 0.5,
            "alpha_map": (i % 7)[:, None]}


def _worker(rank, world, port, n_rays, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        out, span = render_frame_sharded(_fake_render, n_rays, rank, world, gather=True)
        ref = _fake_render(0, n_rays)
        ok = all(torch.equal(out[k], ref[k]) for k in ref) and span == shard_range(n_rays, rank, world)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_rays", [1001, 64])
def test_two_rank_frame_assembly(n_rays):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + n_rays) % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_rays, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


def _grad_worker(rank, world, port, q):
    """The training steps' gradient exchange: per-rank flat gradients -> identical sums on every rank and the
    1/world factor the fused clip+Adam kernel applies (here applied on the CPU to check the update)."""
    from nerf_amd.parallel import allreduce_gradients
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 1191688                                             # both NeRF models, SURVEY 8a row 12
        g = torch.Generator().manual_seed(100 + rank)
        local = torch.randn(n, generator=g)
        flat = local.clone()
        scale = allreduce_gradients(flat)
        others = [torch.randn(n, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)]
        mean = sum(others) / world
        ok = scale == 1.0 / world and torch.allclose(flat * scale, mean, atol=1e-6)
        # the reference order: mean over ranks, then clip_grad_value_(40), then Adam (trainer.py:73-75)
        p = torch.zeros(n, requires_grad=True)
        p.grad = (flat * scale * 30).clone()
        torch.nn.utils.clip_grad_value_([p], 40.0)
        torch.optim.Adam([p], lr=5e-4).step()
        gathered = [torch.zeros(n) for _ in range(world)]
        dist.all_gather(gathered, p.detach())
        ok = ok and all(torch.equal(gathered[0], t) for t in gathered)      # replicas stay bit-identical
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_two_rank_gradient_mean():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


def test_allreduce_gradients_without_process_group_is_identity():
    from nerf_amd.parallel import allreduce_gradients
    g = torch.arange(5, dtype=torch.float32)
    assert allreduce_gradients(g) == 1.0 and torch.equal(g, torch.arange(5, dtype=torch.float32))


def test_frame_assembly_single_rank_passthrough():
    out, span = render_frame_sharded(_fake_render, 10, 0, 1)
    assert span == (0, 10) and torch.equal(out["fine_depth_map"], _fake_render(0, 10)["fine_depth_map"])


def _cell_worker(rank, world, port, q):
    """allreduce_cell_gradients' orchestration on CPU: the two HIP copies are replaced by their torch-indexing
    equivalents (tests/test_gpu_kilo_ddp.py checks the kernels against exactly these), everything else -- union of
    the touched masks, record all-reduce, scatter, the returned scale -- is the product code."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nerf_amd import kilo_ops, parallel
        kilo_ops.kilo_cells_gather = lambda ts, cells, out=None: torch.cat(
            [t.reshape(t.shape[0], -1)[cells.long()] for t in ts], 1).contiguous()

        def scatter(rec, cells, ts):
            c0 = 0
            for t in ts:
                w = t[0].numel()
                t.reshape(t.shape[0], -1)[cells.long()] = rec[:, c0:c0 + w]
                c0 += w
        kilo_ops.kilo_cells_scatter = scatter
        ncell = 32
        g = torch.Generator().manual_seed(100 + rank)
        touched = torch.zeros(ncell, dtype=torch.int32)
        touched[torch.randperm(ncell, generator=g)[:10]] = 1
        shapes = [(ncell, 3, 2), (ncell, 5)]
        grads = [torch.randn(s, generator=g) * touched.view(-1, *([1] * (len(s) - 1))) for s in shapes]
        dense = [t.clone() for t in grads]
        for t in dense:
            dist.all_reduce(t)
        scale = parallel.allreduce_cell_gradients(grads, touched)
        ok = scale == 0.5 and all(torch.equal(a, b) for a, b in zip(grads, dense))
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_two_rank_cell_gradient_exchange_matches_dense_sum():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() * 3 + 11) % 2000
    procs = [ctx.Process(target=_cell_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


def test_cell_gradient_exchange_single_process_is_identity():
    from nerf_amd.parallel import allreduce_cell_gradients
    g = [torch.ones(4, 2)]
    assert allreduce_cell_gradients(g, torch.ones(4, dtype=torch.int32)) == 1.0 and torch.all(g[0] == 1)


# ---- eight ranks: the node the scaling bench runs on (8 x MI355X), rehearsed on the CPU -------------------------------
def _eight_rank_worker(rank, world, port, q):
    """Everything `bench.py --gpus 8` does between its ranks, on gloo: (a) one 800x800 view ray-sharded over 8 ranks and
    assembled on every rank, for a ray count the ranks divide (640,000: 80,000 each) and one they do not (640,001: the
    padding path of gather_rows); (b) the 200-view test set dealt by view_schedule, every view rendered exactly once,
    25 per rank; (c) three data-parallel steps -- flat-gradient all-reduce, mean, clip, Adam -- with the replicas
    bit-identical at the end."""
    from nerf_amd.parallel import allreduce_gradients, gather_rows
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ok = {}
        # (a)
        for n_rays in (640000, 640001):
            out, span = render_frame_sharded(_fake_render, n_rays, rank, world, gather=True)
            ref = _fake_render(0, n_rays)
            ok[f"frame_{n_rays}"] = (span == shard_range(n_rays, rank, world)
                                     and all(out[k].shape == ref[k].shape and torch.equal(out[k], ref[k]) for k in ref))
        b, e = shard_range(640001, rank, world)
        ok["uneven_slices"] = (e - b) == (80001 if rank == 0 else 80000)
        # (b) every rank "renders" its views (a fake renderer: one row per view carrying the view number and the rank)
        mine = [v for v, _ in view_schedule(200, rank, world)]
        ok["views_per_rank"] = len(mine) == 25 and mine == list(range(rank, 200, world))
        rows = {"view": torch.tensor(mine, dtype=torch.float32), "rank": torch.full((len(mine),), float(rank))}
        allv = gather_rows(rows, 200, rank, world)                   # rank r's rows land at [25 r, 25 r + 25)
        ok["every_view_once"] = (sorted(allv["view"].tolist()) == [float(v) for v in range(200)]
                                 and all(int(v) % world == int(r) for v, r in zip(allv["view"].tolist(), allv["rank"].tolist())))
        # (c)
        n = 1191688                                                  # both NeRF models, SURVEY 8a row 12
        p = torch.zeros(n, requires_grad=True)
        opt = torch.optim.Adam([p], lr=5e-4)
        for step in range(3):
            g = torch.Generator().manual_seed(1000 * step + rank)    # every rank its own batch -> its own gradient
            flat = torch.randn(n, generator=g) * 50
            scale = allreduce_gradients(flat)
            ok[f"scale_{step}"] = scale == 1.0 / world
            p.grad = flat * scale                                    # trainer.py:73-75: mean, then clip, then Adam
            torch.nn.utils.clip_grad_value_([p], 40.0)
            opt.step()
        gathered = [torch.zeros(n) for _ in range(world)]
        dist.all_gather(gathered, p.detach())
        ok["replicas_identical"] = all(torch.equal(gathered[0], t) for t in gathered) and bool(p.detach().abs().max() > 0)
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_eight_rank_rehearsal_frame_views_and_gradient_steps():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    world = 8
    procs = [ctx.Process(target=_eight_rank_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == list(range(world))
    for rank, ok in res.items():
        assert all(ok.values()), (rank, {k: v for k, v in ok.items() if not v})
