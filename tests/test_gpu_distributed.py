"""The N > 1 path on the GPU box (one card, so both ranks share it and the exchange runs over gloo; the 8-GPU node with
RCCL is the driver's): SURVEY.md 8e's parity check -- the all-reduced flat gradient equals the mean of the G independent
single-GPU shard gradients -- with the HIP backward, and `python bench.py --gpus 2` typed as is."""
import json
import multiprocessing as mp
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ctx():
    return mp.get_context("forkserver")          # started clean in conftest.pytest_configure


@pytest.mark.parametrize("world", [2, 4])
def test_allreduced_gradient_is_the_mean_of_the_shard_gradients(world):
    """world 4 = as many ranks as one card may carry beside the pytest process (the box allows six GPU processes): the
    rank arithmetic beyond two (shards, scale 1/world, replicas in step) with the HIP backward."""
    from _dist_gpu_worker import rank_main
    ctx = _ctx()
    q = ctx.Queue()
    port = 29600 + os.getpid() % 1000 + world
    procs = [ctx.Process(target=rank_main, args=(r, world, port, "gloo", q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(60)
    for rank, status, err, drift, n in res:
        assert status == "ok", err
        assert n == 14997
        # the weight gradients are added with float atomics: 1e-6 of the gradient's scale between two runs of one shard
        assert err < 5e-6, f"rank {rank}: all-reduced gradient differs from the mean of the shard gradients by {err:.2e}"
        assert drift == 0.0, f"replicas drifted apart by {drift:.2e} after two exchanged steps"


def test_bench_two_ranks_as_typed():
    """`python bench.py --gpus 2 ...` with no launcher: it starts its ranks itself; rank 0 prints the one JSON line."""
    from _dist_gpu_worker import run_command
    ctx = _ctx()
    q = ctx.Queue()
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--no-cpu-baseline",
           "--plots", "4", "--points", "8192"]
    p = ctx.Process(target=run_command, args=(cmd, {"SN2_BENCH_ONE_DEVICE": "1", "SN2_BENCH_BACKEND": "gloo"}, q))
    p.start()
    rc, out, err = q.get(timeout=600)
    p.join(60)
    assert rc == 0, err
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["scaling"] == "weak"
    assert d["loss"] == d["loss"] and d["value"] > 0          # finite loss
