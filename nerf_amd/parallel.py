"""Ray / view sharding across the GPUs of one node.

Rays are independent, so inference needs no data-path collective: every rank generates its own
rays from the pose and renders its slice with replicated weights.  The only communication is
the optional assembly of one frame on every rank (one ``all_gather`` of 36 B per ray, 20 B for the fine outputs alone, over
RCCL / xGMI; ``gloo`` in the CPU tests).  The reference has no counterpart: it renders a view
on a single device (``run.py:45-93``).
"""
from __future__ import annotations

import torch


def shard_range(n: int, rank: int, world: int):
    """Contiguous, balanced slice [begin, end) of ``n`` items for ``rank`` (first ranks get the extra)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    base, extra = divmod(n, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def view_schedule(n_views: int, rank: int, world: int, tiles_per_view: int = 1):
    """(view, tile) work items of ``rank``: items are dealt round-robin so all ranks stay busy across
    view boundaries (200-view test set over 8 GPUs: BASELINE config 5)."""
    items = [(v, t) for v in range(n_views) for t in range(tiles_per_view)]
    return items[rank::world]


def _all_reduce(t: torch.Tensor, op, group=None) -> None:
    import torch.distributed as dist
    if t.is_cuda and dist.get_backend(group) == "gloo":              # rehearsal / CPU-only transport: stage through host
        host = t.cpu()
        dist.all_reduce(host, op=op, group=group)
        t.copy_(host)
    else:
        dist.all_reduce(t, op=op, group=group)


def allreduce_gradients(flat_grad: torch.Tensor, group=None) -> float:
    """Data-parallel gradient exchange of the training steps (``trainer.py:14-21`` DDP semantics: every rank has
    its own batch, parameters move by the MEAN gradient): ONE sum all-reduce of the flat gradient buffer (RCCL
    over xGMI on the GPU box, gloo in the CPU tests).  Returns the factor the optimiser kernel must apply to the
    summed gradient (1 / world); the division itself is fused into ``nrf_clip_adam`` (before the +-40 clamp, as
    DDP averages before ``clip_grad_value_``)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return 1.0
    world = dist.get_world_size(group)
    if world > 1:
        _all_reduce(flat_grad, dist.ReduceOp.SUM, group)
    return 1.0 / world


def allreduce_cell_gradients(grads, touched: torch.Tensor, group=None) -> float:
    """Data-parallel exchange of KiloNeRF gradients in compact form ("sparse gradient
    all-reduce").  ``grads``: the 12 batched (cells, ...) gradient tensors; ``touched`` (cells,) int32, non-zero
    where THIS rank had samples -- every other cell's data gradient is exactly zero here.  The ranks agree on the
    union of touched cells (one 16 KB MAX all-reduce), gather those cells' slices into (n, 6212) records
    (``nrf_kilo_cells_gather``), sum-all-reduce the records and scatter them back: n / cells of the 102 MB a dense
    exchange moves over xGMI (a quarter for the sphere scene of the bench).  Cells no rank touched keep their local
    zeros, which is also their sum.  Returns 1 / world for the optimiser kernel, like ``allreduce_gradients``.
    Terms that are identical on every rank and not zero outside the touched cells (the colour regulariser) must be
    added AFTER this call, scaled by world (``KiloTrainStep.step`` does).  On return ``touched`` holds the UNION over
    the ranks (0 / 1): the cells whose gradient may be non-zero now."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return 1.0
    world = dist.get_world_size(group)
    if world > 1:
        exchange_cell_gradients(grads, touched, group)
    return 1.0 / world


def exchange_cell_gradients(grads, touched: torch.Tensor, group=None) -> None:
    """The collectives of ``allreduce_cell_gradients`` (no short cut at world 1, for the same reason as
    ``gather_rows``)."""
    import torch.distributed as dist
    from . import kilo_ops
    union = (touched != 0).to(torch.int32)
    _all_reduce(union, dist.ReduceOp.MAX, group)
    touched.copy_(union)
    cells = union.nonzero().flatten().to(torch.int32)                # one host sync: the record count
    if cells.numel():
        rec = kilo_ops.kilo_cells_gather(grads, cells)
        _all_reduce(rec, dist.ReduceOp.SUM, group)
        kilo_ops.kilo_cells_scatter(rec, cells, grads)


def render_frame_sharded(render_rows, n_rays: int, rank: int, world: int, group=None, gather: bool = True):
    """Render rays ``shard_range(n_rays, rank, world)`` with ``render_rows(first_ray, count) -> dict`` of
    per-ray tensors ((count, C) or (count,)), and, if ``gather``, assemble all ranks' slices in ray
    order on every rank.  Returns (dict, (begin, end))."""
    begin, end = shard_range(n_rays, rank, world)
    local = render_rows(begin, end - begin)
    if not gather or world == 1:
        return local, (begin, end)
    return gather_rows(local, n_rays, rank, world, group), (begin, end)


def gather_rows(local, n_rays: int, rank: int, world: int, group=None):
    """The collective of ``render_frame_sharded``: ``local`` holds this rank's slice of every per-ray output; returns
    the outputs of all ``n_rays`` rays in ray order, on every rank.  (No short cut at world 1: tests/test_gpu_rccl.py
    runs it through a one-rank RCCL group, the only kind a one-GPU box can form.)"""
    import torch.distributed as dist
    begin, end = shard_range(n_rays, rank, world)
    # ONE collective per frame: the per-ray columns of every output (9 floats per ray for rgb x 2, depth x 2, acc) are
    # packed side by side; all_gather wants equal shapes, so every rank pads to the largest slice.
    per = (n_rays + world - 1) // world
    keys = list(local)
    cols = [local[k].reshape(local[k].shape[0], -1) for k in keys]
    dtype, device = cols[0].dtype, cols[0].device
    if any(c.dtype != dtype for c in cols):
        raise TypeError("render_frame_sharded: all outputs must share one dtype to travel in one all_gather")
    widths = [c.shape[1] for c in cols]
    packed = torch.zeros((per, sum(widths)), dtype=dtype, device=device)
    packed[: end - begin] = torch.cat(cols, 1)
    shape = (world * per, packed.shape[1])                       # concatenated form: gloo accepts no stacked output
    if packed.is_cuda and dist.get_backend(group) == "gloo":     # rehearsal transport: stage through the host
        host = torch.empty(shape, dtype=dtype)
        dist.all_gather_into_tensor(host, packed.cpu(), group=group)
        gathered = host.to(device)
    else:
        gathered = torch.empty(shape, dtype=dtype, device=device)
        dist.all_gather_into_tensor(gathered, packed, group=group)
    gathered = gathered.view(world, per, -1)
    rows = torch.cat([gathered[r, : shard_range(n_rays, r, world)[1] - shard_range(n_rays, r, world)[0]]
                      for r in range(world)], 0)
    out, c0 = {}, 0
    for k, w in zip(keys, widths):
        out[k] = rows[:, c0:c0 + w].reshape((n_rays,) + tuple(local[k].shape[1:]))
        c0 += w
    return out
