"""Host-side mirror of the reference interface (CPU only): constructor, state-dict keys/shapes, layout helpers,
checkpoint helpers, and the loud failure of the product path without a HIP device."""
import os

import pytest
import torch

from conftest import golden_state_dict, load_golden
from stratanet2_vegetation_coverage_maps_amd import PointNet2, project_to_plotwise_coverages
from stratanet2_vegetation_coverage_maps_amd._lib import StrataHipError
from stratanet2_vegetation_coverage_maps_amd.synthetic import make_args, make_batch


def test_state_dict_keys_shapes_and_default_init_match_reference():
    ref = golden_state_dict(load_golden("c1_ref_defaults"))
    torch.manual_seed(0)
    m = PointNet2(make_args())
    sd = m.state_dict()
    assert list(sd.keys()) == list(ref.keys())
    for k in sd:
        assert sd[k].shape == ref[k].shape, k
        if ".0.weight" in k or ".0.bias" in k or k.startswith("lin"):
            assert torch.equal(sd[k], ref[k]), k          # same RNG stream as the reference constructor
    assert sum(p.numel() for p in m.parameters()) == 14997
    m.load_state_dict(ref)                                # a reference checkpoint loads


def test_layout_helpers():
    m = PointNet2(make_args(subsample_size=5))
    x = torch.arange(2 * 3 * 5, dtype=torch.float32).view(2, 3, 5)
    lf = m.get_long_form(x)
    assert lf.shape == (10, 3)
    assert torch.equal(lf, torch.cat(list(x), 1).transpose(1, 0))
    assert torch.equal(m.get_batch_format(lf), x)


def test_checkpoint_roundtrip_and_early_stopping(tmp_path):
    args = make_args(stats_path=str(tmp_path), patience_in_epochs=2, epoch_to_start_early_stop=1)
    m = PointNet2(args)
    assert m.stop_early(0.5, 1, args) is False and m.best_metric_epoch == 1
    assert os.path.exists(os.path.join(str(tmp_path), "PCC_model_full.pt"))
    assert m.stop_early(0.6, 2, args) is False
    assert m.stop_early(0.7, 3, args) is True and m.stopped_early
    m2 = PointNet2(args).load_best_state(args)
    assert m2.best_metric_value == 0.5
    for a, b in zip(m.state_dict().values(), m2.state_dict().values()):
        assert torch.equal(a, b)
    args.current_fold_id = 3
    m.save_state(args)
    assert os.path.exists(os.path.join(str(tmp_path), "PCC_model_fold_n=3.pt"))


def test_product_path_fails_loudly_without_a_device():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    args = make_args(subsample_size=64)
    m = PointNet2(args)
    data = make_batch(1, 64)
    with pytest.raises(StrataHipError):
        m(data)
    with pytest.raises(StrataHipError):
        project_to_plotwise_coverages(torch.zeros(64, 4), data["cloud"], args)


def test_no_oracle_import_in_product():
    import stratanet2_vegetation_coverage_maps_amd as pkg
    root = os.path.dirname(pkg.__file__)
    for fn in os.listdir(root):
        if fn.endswith(".py"):
            txt = open(os.path.join(root, fn)).read()
            assert "import oracle" not in txt and "from oracle" not in txt, fn


def _run_clean(cmd, env, timeout=120):
    """Run a command and hand back (rc, stdout, stderr).  When this pytest process already holds the GPU (an unfiltered run
    on a GPU box: the test_gpu_* modules come first) the child is started from the clean forkserver of conftest.py -- never a
    fork + exec of a process that has initialised the GPU."""
    import subprocess
    import torch
    if torch.cuda.is_initialized():
        import multiprocessing as mp
        from _dist_gpu_worker import run_command
        ctx = mp.get_context("forkserver")
        q = ctx.Queue()
        p = ctx.Process(target=run_command, args=(cmd, env, q))
        p.start()
        rc, out, err = q.get(timeout=timeout)
        p.join(30)
        return rc, out, err
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    e.update(env)
    r = subprocess.run(cmd, env=e, capture_output=True, text=True, timeout=timeout)
    return r.returncode, r.stdout, r.stderr


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus N` without a launcher: N child processes, rank/world/rendezvous set as torch.distributed.run
    would; a failing rank makes the launcher exit non-zero.  (SN2_BENCH_LAUNCH_CHECK: the ranks only report and exit.)"""
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rc, out, err = _run_clean([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--steps", "2"],
                              {"SN2_BENCH_LAUNCH_CHECK": "1"})
    assert rc == 0, err
    got = sorted((json.loads(l) for l in out.splitlines() if l.startswith("{")), key=lambda d: d["RANK"])
    assert [d["RANK"] for d in got] == ["0", "1", "2"] and all(d["WORLD_SIZE"] == "3" for d in got)
    assert all(d["LOCAL_RANK"] == d["RANK"] and d["MASTER_ADDR"] == "127.0.0.1" for d in got)
    assert len({d["MASTER_PORT"] for d in got}) == 1
    # every rank gets its share of the host cores (SN2 launcher: OMP_NUM_THREADS and, where the cpuset allows, an affinity mask)
    assert all(int(d["OMP_NUM_THREADS"]) >= 1 for d in got)
    masks = [set(d["affinity"]) for d in got]
    if all(d["affinity_pinned"] for d in got):
        assert not (masks[0] & masks[1]) and not (masks[1] & masks[2])
    rc, out, err = _run_clean([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"],
                              {"SN2_BENCH_LAUNCH_CHECK": "fail"})
    assert rc == 3


def test_bench_gives_a_timed_region_its_share_of_geometry():
    """bench.py: the batches a geometry pass covers divide the timed step count (so the region launches exactly `steps` batches'
    worth of FPS / ball query / 3-NN), and the phase puts the region's last pass G - 1 steps before its end; TrainPipeline
    turns the phase into how far ahead the passes run.  (CPU: arithmetic only.)"""
    import importlib
    bench = importlib.import_module("bench")
    assert bench.pipe_group_for(200) == 10 and bench.pipe_group_for(20) == 10 and bench.pipe_group_for(100) == 10
    assert bench.pipe_group_for(50) == 10 and bench.pipe_group_for(21) == 7 and bench.pipe_group_for(24) == 8
    assert bench.pipe_group_for(13) == 10                                                                        # a prime: no divisor
    for warmup, steps in ((5, 20), (20, 200), (10, 100), (6, 50), (0, 8)):
        G = bench.pipe_group_for(steps)
        ph = bench.pipe_phase_for(G, warmup, steps)
        assert 0 <= ph < G
        # passes are issued at the end of the steps that complete batch numbers = ph (mod G): the last one inside the region
        issued = [d for d in range(warmup + 1, warmup + steps + 1) if d % G == ph]
        assert len(issued) == steps // G and warmup + steps - issued[-1] == G - 1
    assert bench.host_cpu_share() >= 1


def test_pipeline_phase_sets_how_far_ahead_the_passes_run():
    from stratanet2_vegetation_coverage_maps_amd.pipeline import TrainPipeline
    G, depth = 5, 3

    class _Model:                                  # (no device work in the constructor beyond buffer allocation: stub it)
        def alloc_geometry_pair(self, B, N, dev, group=2):
            return object(), tuple(object() for _ in range(group))

    slots = [{"xyz": torch.zeros(2, 3, 8), "fps_start": torch.zeros(2, 2, dtype=torch.int32)} for _ in range(G * depth + G)]
    import stratanet2_vegetation_coverage_maps_amd.hip_ops as ops
    real = ops.shared_stream
    ops.shared_stream = lambda dev, name: None     # CPU: no streams
    try:
        ev = torch.cuda.Event
        torch.cuda.Event = lambda *a, **k: None
        try:
            for phase in range(G):
                p = TrainPipeline(_Model(), None, None, slots, depth=depth, group=G, phase=phase)
                assert p.phase == phase and p.ahead == G * depth - phase
                # the issue rule of TrainPipeline._run_ahead, replayed on the host
                issued, at = 0, []
                while issued + G <= 0 + p.ahead:
                    issued += G
                for done in range(1, 4 * G + 1):
                    while issued + G <= done + p.ahead:
                        issued += G
                        at.append(done)
                assert all(d % G == phase for d in at) and len(at) >= 3
        finally:
            torch.cuda.Event = ev
    finally:
        ops.shared_stream = real
