"""Ray / view sharding across the GPUs of one node.

Rays are independent, so inference needs no data-path collective: every rank generates its own
rays from the pose and renders its slice with replicated weights.  The only communication is
the optional assembly of one frame on every rank (one ``all_gather`` of 20 B per ray over
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
        if flat_grad.is_cuda and dist.get_backend(group) == "gloo":   # rehearsal / CPU-only transport: stage through host
            host = flat_grad.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            flat_grad.copy_(host)
        else:
            dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group)
    return 1.0 / world


def render_frame_sharded(render_rows, n_rays: int, rank: int, world: int, group=None, gather: bool = True):
    """Render rays ``shard_range(n_rays, rank, world)`` with ``render_rows(first_ray, count) -> dict`` of
    per-ray tensors ((count, C) or (count,)), and, if ``gather``, assemble all ranks' slices in ray
    order on every rank.  Returns (dict, (begin, end))."""
    begin, end = shard_range(n_rays, rank, world)
    local = render_rows(begin, end - begin)
    if not gather or world == 1:
        return local, (begin, end)
    import torch.distributed as dist
    per = (n_rays + world - 1) // world          # all_gather wants equal shapes: pad to the largest slice
    out = {}
    for key, t in local.items():
        flat = t.reshape(t.shape[0], -1)
        pad = torch.zeros((per, flat.shape[1]), dtype=flat.dtype, device=flat.device)
        pad[: flat.shape[0]] = flat
        parts = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(parts, pad, group=group)
        pieces = []
        for r in range(world):
            b, e = shard_range(n_rays, r, world)
            pieces.append(parts[r][: e - b])
        full = torch.cat(pieces, 0)
        out[key] = full.reshape((n_rays,) + tuple(t.shape[1:]))
    return out, (begin, end)
