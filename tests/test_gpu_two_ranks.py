"""The one-process-per-GPU path end to end on real GPUs... of which the test box has one: two torchrun ranks share GPU 0
(`--same-gpu`, gloo for the gather because RCCL wants a device per rank).  Each rank renders its 16-row bands inside the library,
rank 0 gathers the rows and compares the radiance sums with the CPU oracle (`--check`): what `bench.py --gpus 2` does on two
GPUs, with the HIP renderer as the renderer (tests/test_shard_gloo.py pins the sharding logic with the oracle standing in)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_on_one_gpu_reproduce_the_oracle_sums():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29531", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--same-gpu", "--check",
           "--steps", "1", "--warmup", "0", "--level", "5", "--dx", "480", "--dy", "272", "--spp", "64", "--spp-step", "64", "--tex", "256",
           "--no-cpu-baseline", "--no-culling-leg", "--no-cold-frame"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["value"] > 0
    chk = d["check"]
    assert chk["pixels"] > 0 and chk["samples_per_pixel"] == 64
    assert chk["max_rel_diff_of_sums"] < 1e-12, chk
    assert chk["nonzero_fraction"] > 0.005  # the crop is every 16th tile row and column; most of it is background
