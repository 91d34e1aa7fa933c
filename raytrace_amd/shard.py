"""Sharding of one render over several GPUs, one process per GPU.

Every pixel sample is independent (render.go:186-217), so a frame is cut into bands of 16 image
rows (the reference's tile height, render.go:302) dealt round-robin to the ranks: band b belongs to
rank b % world.  Each rank keeps the whole scene in its own HBM and accumulates radiance sums for
its rows into a full-size float64 film that is zero elsewhere; one gather of every rank's own rows
onto rank 0 (RCCL on GPUs, gloo on CPU tensors) assembles the frame: each peer sends 1/world of the
film straight to rank 0 over its own xGMI link, instead of the (world-1)/world of the whole film
that a ring reduce would move through every link.  The library applies the same row rule when it is
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
    """Assembles the frame on rank `dst`: every rank contributes the rows of its own bands of its
    (Dy, Dx, 3) film; rows a rank does not own are never read.  Exact (rows are copied, not added)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return film
    rank, world = dist.get_rank(), dist.get_world_size()
    dy = film.shape[0]
    rows = [torch.from_numpy(band_rows(dy, r, world)).to(film.device) for r in range(world)]
    most = max(len(r) for r in rows)  # gather needs equal-sized pieces: pad the short ones
    mine = torch.zeros((most,) + tuple(film.shape[1:]), dtype=film.dtype, device=film.device)
    mine[:len(rows[rank])] = film.index_select(0, rows[rank])
    # gloo (CPU rehearsals of the multi-rank path) moves host tensors only
    via_host = dist.get_backend() == "gloo" and mine.is_cuda
    send = mine.cpu() if via_host else mine
    if rank == dst:
        pieces = [torch.empty_like(send) for _ in range(world)]
        dist.gather(send, pieces, dst=dst)
        for r in range(world):
            if r != dst:
                film.index_copy_(0, rows[r], pieces[r][:len(rows[r])].to(film.device))
    else:
        dist.gather(send, None, dst=dst)
    return film


def combine_counters(values):
    """(sum over ranks, max over ranks) of a 1-D float64 tensor of per-rank counters / timings."""
    import torch.distributed as dist
    total, worst = values.clone(), values.clone()
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        if dist.get_backend() == "gloo" and values.is_cuda:
            total, worst = total.cpu(), worst.cpu()
        dist.all_reduce(total, op=dist.ReduceOp.SUM)
        dist.all_reduce(worst, op=dist.ReduceOp.MAX)
    return total, worst
