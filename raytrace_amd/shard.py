"""Sharding of one render over several GPUs, one process per GPU.

Every pixel sample is independent (render.go:186-217), so a frame is cut into bands of 16 image
rows (the reference's tile height, render.go:302) dealt round-robin to the ranks: band b belongs to
rank b % world.  Each rank keeps the whole scene in its own HBM and accumulates radiance sums for
its rows into a full-size float64 film that is zero elsewhere; one sum-reduce onto rank 0 (RCCL on
GPUs, gloo on CPU tensors) assembles the frame.  The library applies the same row rule when it is
given band_rank / band_world in rt_render_opts.
"""
import numpy as np

BAND_ROWS = 16


def owns_row(row: int, rank: int, world: int) -> bool:
    return (row // BAND_ROWS) % world == rank


def band_rows(dy: int, rank: int, world: int) -> np.ndarray:
    """Image rows (0 = top) rendered by `rank`."""
    rows = np.arange(dy)
    return rows[(rows // BAND_ROWS) % world == rank]


def combine_films(film, dst: int = 0):
    """Sums the per-rank films onto rank `dst`.  The ranks' rows are disjoint and the rest of every
    film is zero, so the sum is exact and independent of the reduction order."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(film, dst=dst, op=dist.ReduceOp.SUM)
    return film


def combine_counters(values):
    """(sum over ranks, max over ranks) of a 1-D float64 tensor of per-rank counters / timings."""
    import torch.distributed as dist
    total, worst = values.clone(), values.clone()
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(total, op=dist.ReduceOp.SUM)
        dist.all_reduce(worst, op=dist.ReduceOp.MAX)
    return total, worst
