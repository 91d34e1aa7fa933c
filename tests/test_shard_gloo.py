"""The N > 1 path on CPU: two processes (gloo), each renders its bands of the frame, one gather of
the ranks' own rows assembles the film on rank 0.  The CPU oracle stands in for the GPU renderer: it applies the same
band rule (band b -> rank b % world) when given band_rank / band_world, so the test pins the rule,
the assembly of the frame from the ranks' rows and the counter combination that bench.py relies on."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DX, DY, SPP = 48, 72, 3


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _scene():
    from raytrace_amd import scenes
    return scenes.scan(level=3, dx=DX, dy=DY, spp=SPP, tex_size=32, lit=True)


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import binding as ob
    from raytrace_amd import shard
    sc = _scene()
    with ob.OracleScene(sc) as orc:
        _, st, _, film = orc.render(sc.Camera[0], sc.RenderOptions, n_threads=2, want_film=True, band_rank=rank, band_world=world)
    mine = shard.band_rows(DY, rank, world)
    others = np.setdiff1d(np.arange(DY), mine)
    assert not film[others].any(), "a rank wrote rows outside its bands"
    t = torch.from_numpy(film)
    shard.combine_films(t)
    counters = torch.tensor([float(st.total_rays), float(rank + 1)], dtype=torch.float64)
    total, worst = shard.combine_counters(counters)
    if rank == 0:
        np.savez(out_path, film=t.numpy(), total_rays=total[0].item(), worst=worst[1].item())
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_assemble_the_single_process_film(tmp_path):
    from oracle import binding as ob
    sc = _scene()
    with ob.OracleScene(sc) as orc:
        _, st, _, whole = orc.render(sc.Camera[0], sc.RenderOptions, n_threads=2, want_film=True)
    out = str(tmp_path / "rank0.npz")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = np.load(out)
    # sample streams are keyed by (pixel, sample, path), not by process: the sharded frame is the same frame
    assert np.array_equal(got["film"], whole)
    assert got["total_rays"] == st.total_rays
    assert got["worst"] == 2.0
