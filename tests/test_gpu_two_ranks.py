"""The one-process-per-GPU path end to end on real GPUs... of which the test box has one: `bench.py --gpus 2` starts its two
ranks itself (no torchrun around it), both on GPU 0 (`--same-gpu`, gloo for the gather because RCCL wants a device per rank).
Each rank renders its 16-row bands inside the library, rank 0 gathers the rows and compares the radiance sums with the CPU
oracle (the bench's check): what `bench.py --gpus 2` does on two GPUs, with the HIP renderer as the renderer
(tests/test_shard_gloo.py pins the sharding logic with the oracle standing in)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SMALL = ["--steps", "1", "--warmup", "0", "--level", "5", "--dx", "480", "--dy", "272", "--spp", "64", "--spp-step", "64", "--tex", "256",
         "--no-culling-leg", "--no-cold-frame", "--cpu-seconds", "1"]


def _bench(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", **(env_extra or {}))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    return out


def test_gpus_2_starts_two_ranks_itself_and_reproduces_the_oracle_sums():
    out = _bench(["--gpus", "2", "--backend", "gloo", "--same-gpu"] + SMALL)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1  # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0
    assert d["config"]["ranks_in_collective"] == 2
    chk = d["check"]
    assert chk["ok"] and chk["pixels"] > 0 and chk["samples_per_pixel"] == 64
    assert chk["max_rel_diff_of_sums"] < 1e-12, chk
    assert chk["nonzero_fraction"] > 0.005  # the sample is every 8th tile row and column; most of it is background
    assert 0 < d["roofline"]["frac"] <= 1.0 and d["roofline"]["bound"] == "valu"


def test_one_rank_line_carries_check_and_roofline():
    out = _bench(SMALL)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["check"]["ok"] and d["check"]["film"] == "the timed steps' own film"
    assert 0 < d["roofline"]["frac"] <= 1.0
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0


def test_gpus_flag_that_disagrees_with_the_launcher_fails_loudly():
    out = _bench(["--gpus", "4"] + SMALL, {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert out.returncode != 0 and "must agree" in out.stderr
