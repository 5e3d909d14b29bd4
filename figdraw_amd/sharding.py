"""Multi-GPU sharding of the path: independent frames, or row stripes of one frame.

No data-path collective is needed in either mode (stripes re-render the vertical blur halo themselves,
`fdh_set_stripe`); the only exchange is ONE gather of RGBA8 rows / frames to a destination rank, over RCCL
(`backend="nccl"` on ROCm) in production and gloo in the CPU tests.
"""
from __future__ import annotations

from typing import List, Optional, Tuple


def stripe_rows(height: int, world: int, rank: int, align: int = 8) -> Tuple[int, int]:
    """Rows [y0, y1) of rank `rank`: contiguous, tile-aligned (8 rows), covering [0, height) exactly once.
    (The library's fdh_stripe_rows is the same rule in C: tests/test_abi_and_sharding.py compares the two.)"""
    tiles = (height + align - 1) // align
    base, extra = divmod(tiles, world)
    t0 = rank * base + min(rank, extra)
    t1 = t0 + base + (1 if rank < extra else 0)
    return min(t0 * align, height), min(t1 * align, height)


def frame_of_rank(step: int, world: int, rank: int) -> int:
    """Frame-parallel mode: global frame index rendered by `rank` at its local `step`."""
    return step * world + rank


def gather_stripes(stripe, height: int, dst: int = 0, group=None):
    """Gather variable-height row stripes (torch uint8 tensors [rows, W, 4]) into one [height, W, 4] frame on `dst`.
    Stripes are padded to the tallest one so a single fixed-size gather suffices."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    rows = [stripe_rows(height, world, r) for r in range(world)]
    tallest = max(b - a for a, b in rows)
    pad = torch.zeros((tallest,) + tuple(stripe.shape[1:]), dtype=stripe.dtype, device=stripe.device)
    pad[: stripe.shape[0]] = stripe
    outs: Optional[List] = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, outs, dst=dst, group=group)
    if rank != dst:
        return None
    frame = torch.empty((height,) + tuple(stripe.shape[1:]), dtype=stripe.dtype, device=stripe.device)
    for r, (a, b) in enumerate(rows):
        frame[a:b] = outs[r][: b - a]
    return frame
