#!/usr/bin/env python3
"""bench.py -- plots/s of one full training step of the PointNet2 hot path on synthetic 32k-point plots.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus N ...            (no launcher: bench.py starts its N ranks itself, one process per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], SURVEY.md 8d "C2 ref-arch"): 16 plots per GPU x 32 768 points, the reference
architecture with ratio1 = 1024/N, r1 = 1 m, ratio2 = 0.25, r2 = 2 m, fp32, inputs resident in HBM.
A step = zero_grad -> PointNet2.forward -> project_to_plotwise_coverages -> loss (abs + 0.1 NLL + 0.04 entropy)
-> backward -> [one RCCL all-reduce of the flat 60 KB gradient when N > 1] -> Adam step
(the step of /root/reference/learning/train.py:52-66).  Plots are sharded over ranks (weak scaling), no other collective.

Prints ONE JSON line on rank 0 with the contract's fields plus
  "roofline"     for the dominant entry point of the step (HIP-event timed inside the timed region),
  "kernels"      per-entry-point ms/step and algorithmic GB/s from one instrumented step after the timed region,
  "cpu_baseline" the oracle (CPU restatement of the reference path) timed on this host on a bounded sample.
"""
import argparse
import json
import os
import sys
import time


def _gpu_numa_nodes():
    """NUMA node of every GPU in HIP device order, read from sysfs (KFD topology -> PCI location -> numa_node) without
    touching the GPU; [] when the box does not expose it."""
    import glob
    out = []
    try:
        nodes = sorted(glob.glob("/sys/class/kfd/kfd/topology/nodes/*"), key=lambda d: int(os.path.basename(d)))
        for nd in nodes:
            props = dict(l.split()[:2] for l in open(os.path.join(nd, "properties")) if len(l.split()) >= 2)
            if int(props.get("simd_count", "0")) == 0:
                continue                                   # a CPU node
            loc, dom = int(props["location_id"]), int(props.get("domain", "0"))
            bdf = f"{dom:04x}:{(loc >> 8) & 0xFF:02x}:{(loc >> 3) & 0x1F:02x}.{loc & 7}"
            out.append(int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read()))
    except (OSError, ValueError, KeyError):
        return []
    return out


def _parse_cpulist(txt):
    cpus = set()
    for part in txt.strip().split(","):
        if part:
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
    return cpus


def rank_cpu_share(local_rank, local_world):
    """The host cores of one rank: its share of the cores of the NUMA node its GPU hangs off (so that the launch thread, the
    pinned staging buffers and the CPU baseline stay local to the GPU), or -- when sysfs does not say -- a contiguous
    1/local_world slice of the cores this process may use.  -> (sorted core list, "numa<k>" | "slice")."""
    allowed = sorted(os.sched_getaffinity(0))
    numa = _gpu_numa_nodes()
    if len(numa) >= local_world and all(n >= 0 for n in numa[:local_world]):
        try:
            mine = numa[local_rank]
            node_cpus = sorted(_parse_cpulist(open(f"/sys/devices/system/node/node{mine}/cpulist").read()) & set(allowed))
            peers = [r for r in range(local_world) if numa[r] == mine]
            share = len(node_cpus) // len(peers)
            if share >= 1:
                k = peers.index(local_rank)
                return node_cpus[k * share:(k + 1) * share], f"numa{mine}"
        except OSError:
            pass
    share = len(allowed) // max(1, local_world)
    if share >= 1:
        return allowed[local_rank * share:(local_rank + 1) * share], "slice"
    return allowed, "shared"


def host_cpu_share():
    """Cores this process may really use: its affinity mask, cut to the container's CPU quota (cgroup v2 `cpu.max` / v1
    `cpu.cfs_quota_us`) where there is one.  torch sizes its intra-op thread pool by the machine's core count (128 on the GPU
    box, of which one GPU's job gets 16): a parallel 21 MB `copy_` then burns the quota in a burst and the whole process is
    throttled until the next 100 ms period -- seen as 85-95 ms stalls in every third step of the eager loop
    (scripts/profile_dropin_host.py)."""
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except (OSError, ValueError):
            pass
    return n


def pin_rank(local_rank, local_world):
    """Give this rank its own cores (`rank_cpu_share`) and size OpenMP to them; called before torch is imported.  With one
    rank nothing is pinned (the whole allowance is the rank's)."""
    if local_world <= 1:
        return sorted(os.sched_getaffinity(0)), "all"
    cpus, how = rank_cpu_share(local_rank, local_world)
    try:
        os.sched_setaffinity(0, cpus)
    except OSError:
        how = "unpinned"
    os.environ["OMP_NUM_THREADS"] = str(max(1, len(cpus)))
    return cpus, how


def _spawn_ranks_if_needed():
    """`python bench.py --gpus N` typed as is (no launcher): start the N ranks as fresh child processes -- one per GPU,
    the same environment torch.distributed.run would give them -- wait, and exit with their worst return code.  Runs
    before torch or the HIP library are imported: this process never touches a GPU, so starting children is safe.
    Every rank (started here or by torch.distributed.run) then pins itself to its share of the host cores (`pin_rank`)."""
    if "RANK" in os.environ or "WORLD_SIZE" in os.environ:
        lw = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
        cpus, how = pin_rank(int(os.environ.get("LOCAL_RANK", "0")), lw)
        os.environ["SN2_BENCH_PINNED"] = f"{how}:{len(cpus)}"
        if os.environ.get("SN2_BENCH_LAUNCH_CHECK"):       # tests/test_host_api.py: what environment did the rank get?
            d = {k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "OMP_NUM_THREADS")}
            d["affinity"], d["affinity_pinned"] = sorted(os.sched_getaffinity(0)), how in ("slice",) or how.startswith("numa")
            print(json.dumps(d), flush=True)
            sys.exit(0 if os.environ["SN2_BENCH_LAUNCH_CHECK"] != "fail" or os.environ["RANK"] != "1" else 3)
        return
    p = argparse.ArgumentParser(add_help=False)
    p.add_argument("--gpus", type=int, default=1)
    n = p.parse_known_args()[0].gpus
    if n <= 1:
        return
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    while procs:
        time.sleep(0.2)
        for pr in list(procs):
            code = pr.poll()
            if code is None:
                continue
            procs.remove(pr)
            if code != 0:
                rc = rc or code
                for other in procs:          # a rank died: the others would wait in a collective forever
                    other.terminate()
    sys.exit(rc)


# stdout carries EXACTLY ONE line, the JSON record: everything else a library may print there (RCCL prints a version banner to
# stdout when its first communicator comes up) goes to stderr -- file descriptor 1 is pointed at stderr for the life of the
# process and the record is written to a duplicate of the original stdout
_RECORD_OUT = None


def emit_record(obj):
    out = _RECORD_OUT if _RECORD_OUT is not None else sys.stdout
    print(json.dumps(obj), file=out, flush=True)


if __name__ == "__main__":
    _spawn_ranks_if_needed()
    sys.stdout.flush()
    _RECORD_OUT = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
from stratanet2_vegetation_coverage_maps_amd import PointNet2, project_to_2d_rasters, project_to_plotwise_coverages  # noqa: E402
from stratanet2_vegetation_coverage_maps_amd import hip_ops as ops  # noqa: E402
from stratanet2_vegetation_coverage_maps_amd import losses  # noqa: E402
from stratanet2_vegetation_coverage_maps_amd.optim import FlatAdam, flatten_parameters, shard_of_rank  # noqa: E402
from stratanet2_vegetation_coverage_maps_amd.synthetic import make_args, make_batch  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
PLOTS_PER_GPU = 16
N_POINTS = 32768
M1 = 1024


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def workload_args(device):
    return make_args(cuda=device, subsample_size=N_POINTS, ratio1=M1 / N_POINTS, r1=1.0, ratio2=0.25, r2=2.0)


def algorithmic_bytes(key, B, N, m1, m2, e1, e2, D=20):
    """Compulsory HBM bytes of ONE launch of an entry point (DESIGN.md 'algorithmic bytes')."""
    t = {
        f"sn2_fps:N={N}": 12 * N * B + 32 * m1 * B,
        f"sn2_fps:N={m1}": 12 * m1 * B + 32 * m2 * B,
        f"sn2_ball_query:N={N}": 12 * (N + m1) * B + 4 * e1 + 4 * m1 * B,
        f"sn2_ball_query:N={m1}": 12 * (m1 + m2) * B + 4 * e2 + 4 * m2 * B,
        # 3-NN of the N points among the m1 centroids (x,y-grid search): positions in, 3 x (index, weight) out
        f"sn2_three_nn_xy:T={N}": 12 * (m1 + N) * B + 24 * N * B,
        f"sn2_three_nn_xy:T={m1}": 12 * (m2 + m1) * B + 24 * m1 * B,
        f"sn2_three_nn:T={N}": 12 * (m1 + N) * B + 24 * N * B,
        f"sn2_three_nn:T={m1}": 12 * (m2 + m1) * B + 24 * m1 * B,
        f"sn2_three_nn:T={m2}": 12 * (1 + m2) * B + 24 * m2 * B,
        # inverted 3-NN index (source -> (row, weight) list): the three tables (rows N, m1, m2) in, the same entries out in
        # source order + offsets/counts and the sources' Morton order; ONE figure for the entry point (its three calls per
        # step are summed by the timer)
        "sn2_interp_index": 3 * (24 + 8) * (N + m1 + m2) * B + 24 * (m1 + m2 + 1) * B,
        # work items of the SA passes: counts in, 4 ints per centroid out (both levels)
        "sn2_sa_order": (4 + 16) * (m1 + m2) * B,
        "sn2_pack_rows": (44 + 48) * N * B,
        # SA1: per pass 4 B index + 48 B gathered row per message (the gather is L2 traffic once the 1.5 MB/plot
        # table is resident; counted here as the algorithmic upper bound), 2 forward passes / 2 backward passes
        "sn2_sa_forward:cf=8": 2 * 52 * e1 + 3 * 64 * m1 * B,
        "sn2_sa_backward:cf=8": 2 * 52 * e1 + 2 * 64 * m1 * B,
        "sn2_sa_forward:cf=16": 84 * e2 + 3 * 128 * m2 * B,
        "sn2_sa_backward:cf=16": 84 * e2 + 64 * e2 + 2 * 128 * m2 * B,
        # FP1: 24 B knn + 32 B skip + 3 x 144 B gathered rows (L2) read, 144 B written per point
        "sn2_fp_forward:34+8->34": (24 + 32 + 144) * N * B,
        # FP1 backward (source-side form): h + dy rows once (288 B; the BN sums come from the head's gradients), skip 32 B,
        # d pre-activation written once and gathered back through the inverted index (2 x 136 B: a 128-byte line + the pair of
        # channels 32, 33; the index entries 24 B);
        # the per-source work (G, dsrc, dW_A over B*m1 rows) is < 1 % of that
        "sn2_fp_backward:34+8->34": (288 + 32 + 136 + 136 + 24) * N * B,
        "sn2_head_forward": (144 + 32) * N * B,
        "sn2_head_backward": (144 + 32 + 144) * N * B,
        "sn2_plot_project_forward": (8 + 8 + 16 + 4) * N * B + 24 * D * D * B,
        "sn2_plot_project_backward": (4 + 16) * N * B + 12 * D * D * B,   # pix read, one row written per point; arg table
    }
    return t.get(key)


# entry point -> the kernel that dominates it (names as rocprofv3 prints them, see profiles/*_pmc_traffic.json)
DOMINANT_KERNEL = {
    "sn2_fps:N=32768": "fps_cluster_kernel<8, 8, 8, true>", "sn2_ball_query:N=32768": "ball_query_grid_kernel",
    "sn2_sa_forward:cf=8": "sa_mfma_fwd_kernel<8, 2, 16, 16, 1, false, true>", "sn2_sa_backward:cf=8": "sa_mfma_bwd_kernel<8, 2, 16, 16, 2, false>",
    "sn2_fp_forward:34+8->34": "fp_fwd_rows2_kernel<34, 8, 34, false>",
    "sn2_fp_backward:34+8->34": "fp_bwd_rows_kernel<34, 8, 34, 512, false>",
    "sn2_head_forward": "head_fwd_mfma_kernel<false>", "sn2_head_backward": "head_bwd_mfma_kernel<false>",
    "sn2_three_nn:T=32768": "three_nn_grid_kernel", "sn2_pack_rows": "pack_rows_kernel",
}


# entry point -> all device kernels it launches (for the PMC traffic of the whole entry point, where it has several)
ENTRY_KERNELS = {
    "sn2_fp_backward:34+8->34": ["fp_bwd_rows_kernel<34, 8, 34, 512, false>", "fp_bwd_src_chunk_kernel<34, 8, 34, false>",
                                 "fp_bwd_src_merge_dw_kernel<34, 8, 34>", "fp_bwd_bn_kernel<34>"],
    "sn2_fp_forward:34+8->34": ["fp_src_table_kernel<34, 8, 34>", "fp_fwd_rows2_kernel<34, 8, 34, false>"],
}


def pmc_traffic(entry):
    """HBM bytes per launch of the entry point (the sum over its kernels) from the committed PMC passes (rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE in separate runs; FETCH doubled per the gfx950 correction).  bench.py cannot collect
    counters itself; None when no profile is present."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
    ks = ENTRY_KERNELS.get(entry) or ([DOMINANT_KERNEL[entry]] if entry in DOMINANT_KERNEL else [])
    if not files or not ks:
        return None, None
    d = json.load(open(files[-1]))
    if DOMINANT_KERNEL.get(entry) not in d:
        return None, None
    return sum(d[k]["hbm_bytes_per_launch_corrected"] for k in ks if k in d), os.path.basename(files[-1])


def rocprof_kernel_avg_ms(entry):
    """Average duration of the entry point's dominant device kernel in the committed rocprofv3 --kernel-trace --stats summary
    (profiles/*_kernel_stats.csv), for comparison with the live HIP-event time of the whole entry point."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_kernel_stats.csv")))
    k = DOMINANT_KERNEL.get(entry)
    if not files or k is None:
        return None
    for r in csv.DictReader(open(files[-1])):
        if k in r["Name"]:
            return round(float(r["AverageNs"]) * 1e-6, 4)
    return None


def measured_peaks(dev):
    """The chip's peaks MEASURED in this run, beside the datasheet figures the roofline fractions are quoted against (SURVEY.md
    8d: "to be confirmed by a measured stream/MFMA microbenchmark in the same run"): a stream copy and a
    read-only stream over 1 GiB (csrc/misc.hip: stream_probe_kernel, 16 B per lane and load, four loads in flight), and
    independent chains of one matrix instruction (mfma_probe_kernel) for fp32 and bf16 operands; HIP events on the launch
    stream, best of 5 launches after a warm-up."""
    from stratanet2_vegetation_coverage_maps_amd import _lib
    import ctypes
    lib = _lib.load()
    st = ops._stream()
    n = 1 << 28                                           # floats: 1 GiB per buffer
    src = torch.empty(n, dtype=torch.float32, device=dev).fill_(1.0)
    dst = torch.empty(n, dtype=torch.float32, device=dev)
    sink = torch.zeros(4096, dtype=torch.float32, device=dev)

    def best_ms(fn, reps=5):
        fn()
        torch.cuda.synchronize()
        t = []
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn()
            b.record()
            torch.cuda.synchronize()
            t.append(a.elapsed_time(b))
        return min(t)

    out = {}
    ms = best_ms(lambda: _lib.check(lib.sn2_debug_stream_probe(src.data_ptr(), dst.data_ptr(), n, 0, sink.data_ptr(), st), "stream copy"))
    out["hbm_copy_GBps"] = round(2 * 4 * n / (ms * 1e-3) / 1e9, 1)
    ms = best_ms(lambda: _lib.check(lib.sn2_debug_stream_probe(src.data_ptr(), None, n, 1, sink.data_ptr(), st), "stream read"))
    out["hbm_read_GBps"] = round(4 * n / (ms * 1e-3) / 1e9, 1)
    del src, dst
    names = {0: "mfma_f32_16x16x4_TFLOPs", 1: "mfma_f32_32x32x2_TFLOPs", 2: "mfma_bf16_16x16x32_TFLOPs", 3: "mfma_bf16_32x32x16_TFLOPs"}
    for mode, name in names.items():
        fl = ctypes.c_double()
        iters = 4000 if mode < 2 else 16000
        ms = best_ms(lambda: _lib.check(lib.sn2_debug_mfma_probe(mode, iters, sink.data_ptr(), ctypes.byref(fl), st), "mfma probe"))
        out[name] = round(fl.value / (ms * 1e-3) / 1e12, 1)
    out["mfma_f32_TFLOPs"] = max(out[names[0]], out[names[1]])
    out["mfma_bf16_TFLOPs"] = max(out[names[2]], out[names[3]])
    out["datasheet"] = {"hbm_GBps": HBM_PEAK_GBS, "mfma_f32_TFLOPs": 157.3, "mfma_bf16_TFLOPs": 2500.0,
                        "source": "/opt/skills/guides/MI355X_MICROARCH.md"}
    out["what"] = ("measured in this run: 1 GiB stream copy (read + write) / read-only stream; chains of independent matrix "
                   "instructions, 8 workgroups x 4 waves per CU; best of 5 launches, HIP events")
    torch.cuda.empty_cache()
    return out


def dominant_kernel_alone(step_fn, dev, reps=5, opt=None):
    """Live duration of the ONE device kernel the roofline record names (`fp_bwd_rows_kernel`: the row pass of the per-point
    layer's backward) -- the entry point `sn2_fp_backward:34+8->34` launches three kernels, and its HIP-event time is theirs
    together.  A diagnostic switch (sn2_debug_fp1_backward_parts) makes the entry point launch the row pass only; HIP events
    around the entry point over `reps` eager steps then time that kernel alone (the steps' gradients are not used)."""
    from stratanet2_vegetation_coverage_maps_amd import _lib
    lib = _lib.load()
    keep = None if opt is None else [t.clone() for t in (opt.flat, opt.exp_avg, opt.exp_avg_sq, opt.step_words)]
    lib.sn2_debug_fp1_backward_parts(1)
    try:
        with ops.timing({"sn2_fp_backward:34+8->34"}) as t:
            for _ in range(reps):
                step_fn()
        r = t.summary().get("sn2_fp_backward:34+8->34")
    finally:
        lib.sn2_debug_fp1_backward_parts(7)
        if keep is not None:                 # (those steps' gradients were incomplete: put the optimiser's state back)
            for dst, src in zip((opt.flat, opt.exp_avg, opt.exp_avg_sq, opt.step_words), keep):
                dst.copy_(src)
    return None if not r else r[1] / r[0]


def cpu_baseline():
    """The oracle (CPU restatement of the reference path, kd-tree neighbour search, all host cores) on a bounded
    sample of the same workload: the metric's own batch (16 plots of 32 768 points, ~2.7 s per step on 16 threads), 1 warm-up
    step, then timed steps until ~12 s of CPU work are done (at least 3, at most 20)."""
    from oracle import losses as olosses, network, projection
    # the GPU box gives one GPU's job a share of 16 cores whatever os.cpu_count() says: more OpenMP threads than that
    # only spin against each other
    ncores = max(1, min(host_cpu_share(), 16))
    torch.set_num_threads(ncores)
    B = PLOTS_PER_GPU
    args = workload_args(None)
    d = make_batch(B, N_POINTS)
    sd = network.init_state_dict(0)
    keys = network.param_keys(sd)
    params = [sd[k].requires_grad_(True) for k in keys]
    opt = torch.optim.Adam(params, lr=1e-3, weight_decay=1e-3)
    times = []
    for it in range(21):
        t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        cov, proba, _ = network.forward(sd, d["cloud"], d["xyz"], args, training=True, use_kdtree=True)
        pred = projection.project_to_plotwise_coverages(cov, d["cloud"], args)
        loss, _ = olosses.total_loss(pred, proba, d["coverages"], d["pdf_all"], args.m, args.e)
        loss.backward()
        opt.step()
        times.append(time.perf_counter() - t0)
        log(f"cpu_baseline step {it}: {times[-1]:.2f} s")
        if it >= 3 and (times[-1] > 60 or sum(times[1:]) > 12.0):
            break
    t = sum(times[1:]) / len(times[1:])
    return {"value": round(B / t, 3), "unit": "plots/s", "cores": ncores, "kind": "port",
            "sample": f"{B} plots x {N_POINTS} pts, same step (fwd+P2+loss+bwd+Adam), mean of {len(times) - 1} steps "
                      f"({sum(times[1:]):.1f} s) after 1 warm-up, "
                      f"torch {torch.get_num_threads()} threads + scipy cKDTree"}


# batches one geometry pass of the pipelined loop covers (stratanet2_vegetation_coverage_maps_amd/pipeline.py: `group`).  FPS is one
# workgroup per plot and a fixed number of sequential rounds, so a pass over 8 batches takes as long as a pass over one, and
# what its workgroups cost the feature pass beside them is the TIME they are resident, not the CUs they hold (DESIGN.md
# section 4): 1 batch per pass 0.96 ms/step, 2: 0.831, 4: 0.792, 8: 0.776, 16: 0.806 (128 plots: a third of the chip's CUs).
# Round 5: with the per-batch products of a pass built by grouped launches (18 small launches per pass instead of 18 per batch)
# larger groups pay again: 200 steps 0.7216 / ~0.717 / 0.7156 / 0.7595 at G = 8 / 10 / 12 / 16, the driver's 20 steps 0.741 /
# 0.740 / 0.7285 at G = 4 / 5 / 10 -- 10 divides both step counts, so both commands run the same mode.
PIPE_GROUP = int(os.environ.get("SN2_PIPE_GROUP", "10"))


def pipe_group_for(steps):
    """Batches per geometry pass for a timed region of `steps` steps: PIPE_GROUP when it divides `steps`, else the best divisor
    of `steps` near it -- so that the region launches EXACTLY `steps` batches' worth of FPS / ball query / 3-NN (round 3's
    driver-timed line ran 20 steps with 8 batches per pass: three passes = 24 batches of geometry for 20 feature passes)."""
    if "SN2_PIPE_GROUP" in os.environ or steps % PIPE_GROUP == 0:
        return PIPE_GROUP
    for g in (8, 5, 12, 4, 6, 7, 3, 2):
        if steps % g == 0:
            return g
    return PIPE_GROUP


def pipe_phase_for(group, warmup, steps):
    """TrainPipeline.phase: passes are issued at the end of the steps that complete batch numbers = phase (mod group).  Chosen so
    that the LAST pass launched inside the timed region is launched group - 1 steps before the region ends: the region ends
    with every stream drained, and a pass launched at its very last step would be waited for in full (1.4 ms = 0.07 ms per
    step of a 20-step region) although the loop, left running, overlaps it.  Steady-state throughput does not depend on it."""
    return (warmup + steps + 1) % group


_ONE_RANK_GROUP = [False]


def make_exchange(kind, dev, world):
    """The gradient exchange of a training leg -> (rccl communicator or None, force torch's all_reduce at world 1, description).
    "rccl": `ncclAllReduce` through the direct binding (stratanet2_vegetation_coverage_maps_amd/rccl.py) on the step's own stream,
    INSIDE the slot's hipGraph, at any world size (world 1: a one-rank communicator -- the collective call still runs);
    "torch": torch.distributed.all_reduce between the backward graph and the Adam graph (at world 1 over a one-rank process
    group, so that the call really happens); "none": no exchange (only valid at world 1);
    "auto": none at world 1; at world > 1 rccl if every rank's self-test passes (agreed on through torch's group), else torch."""
    from stratanet2_vegetation_coverage_maps_amd import rccl
    dist = torch.distributed
    if kind == "auto":
        if world == 1:
            return None, False, "none (single GPU)"
        # stage by stage, with an agreement through torch's group between the stages: every RCCL collective is entered by all
        # ranks or by none (rccl.negotiate_comm)
        comm, why = rccl.negotiate_comm(dev, graph=True, log=log)
        if comm is not None:
            return comm, False, f"rccl {rccl.version()} ncclAllReduce inside the step's hipGraph (direct binding; self-test passed on all {world} ranks)"
        log(f"direct RCCL exchange not used ({why}): torch.distributed.all_reduce between two graphs")
        return None, False, f"torch.distributed.all_reduce (nccl backend) between the backward graph and the Adam graph [direct RCCL refused: {why}]"
    if kind == "rccl":
        comm = rccl.comm_from_torch_group(dev)
        rccl.self_test(comm, graph=True)
        return comm, False, f"rccl {rccl.version()} ncclAllReduce inside the step's hipGraph (direct binding, {comm.world} rank(s))"
    if kind == "torch":
        if world == 1 and not dist.is_initialized():
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
            _ONE_RANK_GROUP[0] = True
        return None, True, "torch.distributed.all_reduce (nccl backend) between the backward graph and the Adam graph"
    if world > 1:
        raise SystemExit("--exchange none needs --gpus 1")
    return None, False, "none (single GPU)"


def build_training(dev, local_rank, rank, world, B, n_points, arch, dtype, n_slots, exchange="auto"):
    """Model, optimiser, resident input slots and the feature-step closure of one training workload (the step of
    /root/reference/learning/train.py:52-66 minus exchange + optimiser)."""
    from types import SimpleNamespace
    args = make_args(cuda=local_rank, subsample_size=n_points, ratio1=M1 / n_points, r1=1.0, ratio2=0.25, r2=2.0)
    args.mma_dtype = "bf16" if dtype == "bf16" else "fp32"
    torch.manual_seed(0)                       # identical initial weights on every rank
    if arch == "3sa":
        from stratanet2_vegetation_coverage_maps_amd.point_net2_3sa import PointNet2ThreeSA
        args.ratio3, args.r3 = 0.25, 4.0
        model = PointNet2ThreeSA(args).train()
        model.set_mma_dtype(args.mma_dtype)
    else:
        model = PointNet2(args).train()
    n_fps = 3 if arch == "3sa" else 2
    model.p2_diam_pix = args.diam_pix        # geometry passes also compute the projection's pixel ids (functions of x, y only)
    flatten_parameters(model)
    comm, force, exchange_desc = make_exchange(exchange, dev, world)
    # (no exchange between backward and update: the Adam kernel folds the gradient's images itself, one launch less per step)
    opt = FlatAdam(model, lr=1e-3, weight_decay=1e-3, world_size=world, comm=comm,     # config.py:84,97
                   fold_gradient_images=(world == 1 and comm is None and not force))
    opt.force_exchange = force
    # batch j of rank r = plots [(j*world + r)*B, +B) of the seeded set
    slots = []
    for j in range(n_slots):
        host = make_batch(B, n_points, first_plot=j * world * B + shard_of_rank(rank, B)[0])
        slots.append({"cloud": host["cloud"].to(dev), "xyz": host["xyz"].to(dev),
                      "fps_start": torch.zeros(n_fps, B, dtype=torch.int32, device=dev),
                      "gt": host["coverages"].to(dev), "pdf": host["pdf_all"].to(dev)})

    seed = torch.ones((), dtype=torch.float64, device=dev)      # d loss / d loss, made once: `loss.backward()` fills a new one per step

    def feature_step(inp, geo=None):
        """zero_grad -> forward -> plot-wise projection -> loss -> backward (everything but exchange + Adam)"""
        opt.zero_grad(set_to_none=True)
        cd = {"cloud": inp["cloud"], "xyz": inp["xyz"], "fps_start": inp["fps_start"]}
        if geo is not None:
            cd["geometry"] = geo
        cov, proba = model(cd)
        # projection + loss (learning/train.py:54-62): one autograd node over three launches when the geometry pass left the
        # pixel ids (else project_to_plotwise_coverages + total_loss: seven)
        loss, _, _ = losses.projected_total_loss(cov, proba, inp["cloud"], inp["gt"], inp["pdf"], args, geometry=geo, model=model)
        loss.backward(gradient=seed)
        return loss

    return SimpleNamespace(args=args, model=model, opt=opt, slots=slots, feature_step=feature_step, n_fps=n_fps,
                           exchange=exchange_desc, split=True if force else None)


def step_model_figures(B, n_points, m1, m2, e1, e2):
    """compulsory HBM bytes (2.2 x forward) and dense-layer flops (3 x forward - 352 E1) of ONE training step
    of the reference architecture at the measured message counts."""
    E1p, E2p = e1 / B, e2 / B                                                    # messages per plot
    fwd_flops = 864 * E1p + 1216 * E2p + 4480 * m2 + 12288 * m2 + 5440 * m1 + 2856 * n_points + 1248 * n_points
    fwd_bytes = 368 * n_points + 8 * (E1p + E2p) + 16 * (m1 + m2) + 8 * (16 * m1 + 32 * m2 + 64 + 64 * m2 + 34 * m1) + 48 * m1 + 16 * m2
    return B * 2.2 * fwd_bytes, B * (3 * fwd_flops - 352 * E1p)


def secondary_train_leg(dev, arch, B, n_points, dtype, steps, warmup, depth=3, exchange="none"):
    """One more training configuration through the SAME software-pipelined loop as the headline (pipe_group_for(steps) batches per
    geometry pass, one hipGraph per slot), compact: ms/step, plots/s and the whole step against both roofs.  Single GPU, inputs resident."""
    from stratanet2_vegetation_coverage_maps_amd.pipeline import TrainPipeline
    G = pipe_group_for(steps)
    w = build_training(dev, dev.index or 0, 0, 1, B, n_points, arch, dtype, G * depth + G, exchange=exchange)
    pipe = TrainPipeline(w.model, w.opt, w.feature_step, w.slots, depth=depth, group=G, phase=pipe_phase_for(G, warmup, steps),
                         split_exchange=w.split)
    pipe.capture()
    pipe.prime()
    for _ in range(warmup):
        pipe.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = pipe.step()
    pipe.drain()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    with torch.no_grad():
        _, _, saved = w.model._forward_impl(w.slots[0]["xyz"], w.slots[0]["cloud"], w.slots[0]["fps_start"], False)
        e1, e2 = int(saved.tot1.item()), int(saved.tot2.item())
        m1, m2 = saved.M1, saved.M2
    by, fl = step_model_figures(B, n_points, m1, m2, e1, e2)       # the reference architecture's terms (3sa: a lower bound)
    peak = {"f32": 157.3e12, "bf16": 2.5e15}[dtype]
    out = {"arch": arch, "plots_per_gpu": B, "points_per_plot": n_points, "dtype": dtype, "steps": steps, "warmup": warmup,
           "batches_per_geometry_pass": G, "exchange": w.exchange, "ms_per_step": round(ms, 4), "plots_per_s": round(B / (ms * 1e-3), 2), "loss": round(float(loss.item()), 6),
           "messages_sa1": e1, "messages_sa2": e2,
           "roofline": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "achieved": round(by / (ms * 1e-3) / 1e9, 2),
                        "frac": round(by / (ms * 1e-3) / (HBM_PEAK_GBS * 1e9), 4), "compulsory_bytes": int(by),
                        "dense_flops": int(fl), "mfma_frac": round(fl / (ms * 1e-3) / peak, 4),
                        "what": "whole step: SURVEY.md 8d compulsory bytes / dense flops of the reference architecture's "
                                "layers at the measured message counts over ms_per_step"}}
    if w.opt.comm is not None:
        w.opt.comm.destroy()
    del pipe, w
    torch.cuda.empty_cache()
    return out


def exchange_leg(dev, steps=100, warmup=10):
    """The metric's workload on ONE GPU with the gradient exchange really executed: (a) `ncclAllReduce` through
    the direct RCCL binding, captured INSIDE each slot's hipGraph (one graph per step: the launch sequence of an N-GPU run),
    (b) torch.distributed.all_reduce over a one-rank nccl process group between the backward graph and the Adam graph (the
    fallback's launch sequence), (c) no exchange.  Same loop, same kernels; only the exchange differs."""
    out = {}
    for kind in ("none", "rccl", "torch"):
        r = secondary_train_leg(dev, "ref", PLOTS_PER_GPU, 32768, "f32", steps, warmup, exchange=kind)
        out[kind] = {"ms_per_step": r["ms_per_step"], "plots_per_s": r["plots_per_s"], "exchange": r["exchange"], "loss": r["loss"]}
    if _ONE_RANK_GROUP[0]:
        torch.distributed.destroy_process_group()
        _ONE_RANK_GROUP[0] = False
    out["what"] = ("C2 ref-arch on one GPU, world size 1: the same pipelined loop with the exchange step executed by RCCL through "
                   "the direct binding inside the graph / by torch's process group between two graphs / not at all")
    return out


def inference_leg(dev, plots=2048, points=10000, batch=512, repeat=3, prefetch=3, cpu_sample_plots=8, dtype="f32"):
    """BASELINE configs[3] (SURVEY.md 8d "C4"): parcel inference -- 2048 overlapping 10 m plots x 10 000 points tiling one
    parcel, eval forward + fixed-grid max rasters + the ordered weighted mosaic merge (predict.py:96-141), inputs resident.
    512 plots per launch, three geometry passes in flight (measured in round 4: 256 / 4 gives 50.5k plots/s, 512 / 2, 3, 4: 52.9k /
    52.6k / 51.8k, 1024 / 2: 52.7k -- the loop is bound by the chip time of its full-chip kernels, not by the passes' latency).
    Median of `repeat` whole-parcel runs; roofline on SURVEY 8d's 368 B/point forward figure; the oracle's eval forward +
    rasters on a bounded sample of the same plots as the CPU baseline."""
    from stratanet2_vegetation_coverage_maps_amd import inference
    args = make_args(cuda=dev.index or 0, subsample_size=points)           # reference defaults: ratios .25/.25, r sqrt2/sqrt8
    args.mma_dtype = "bf16" if dtype == "bf16" else "fp32"
    torch.manual_seed(0)
    model = PointNet2(args).eval()
    cols, stride = 64, 5.0                                                 # plot centres every 5 m: each pixel sees ~12 plots
    rows = (plots + cols - 1) // cols
    batches, sample = [], None
    for s in range(0, plots, batch):
        nb = min(batch, plots - s)
        d = make_batch(nb, points, first_plot=s)
        if sample is None:
            sample = {"cloud": d["cloud"][:cpu_sample_plots].clone(), "xyz": d["xyz"][:cpu_sample_plots].clone()}
        k = torch.arange(s, s + nb)
        c = torch.stack([10.0 + stride * (k % cols), 10.0 + stride * (k // cols)], 1).double()
        batches.append({"cloud": d["cloud"].to(dev), "xyz": d["xyz"].to(dev), "plot_center": c,
                        "fps_start": torch.zeros(2, nb, dtype=torch.int64)})
    H, W = int(20 + stride * (rows - 1)), int(20 + stride * (cols - 1))

    def run():
        mos = inference.ParcelMosaic(0.0, float(H), H, W, args, dev)
        n = inference.predict_parcel(model, batches, mos, args, prefetch=prefetch)
        return mos, n

    run()                                                        # warm-up (allocator, lazy module load)
    torch.cuda.synchronize()
    times = []
    for _ in range(repeat):
        torch.cuda.synchronize()
        t = time.perf_counter()
        mos, n = run()
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t)
    med = sorted(times)[len(times) // 2]
    cover = float((~torch.isnan(mos.result()[0])).float().mean())
    by = 368 * points * n                                        # compulsory forward bytes per point
    out = {"workload": f"C4: {plots} plots x {points} pts, B={batch} per launch, reference defaults (ratios .25/.25, r sqrt2/sqrt8), "
                       f"eval forward + rasters + ordered mosaic merge, geometry prefetch {prefetch}, inputs resident",
           "seconds_per_parcel": round(med, 5), "plots_per_s": round(n / med, 1), "runs_s": [round(t, 5) for t in times],
           "statistic": f"median of {repeat}", "parcel_pix": [H, W], "covered_frac": round(cover, 3), "dtype": dtype,
           "roofline": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "achieved": round(by / med / 1e9, 2),
                        "frac": round(by / med / (HBM_PEAK_GBS * 1e9), 5), "compulsory_bytes": int(by),
                        "what": "whole parcel: 368 B/point (SURVEY.md 8d forward figure) x points x plots over seconds_per_parcel"}}
    if dtype != "f32":
        out["workload"] += "; bfloat16 operands on the matrix cores (SA levels, SA3/FP3/FP2) and bfloat16 per-point rows -- NOT the reference's precision"
        del batches, model
        torch.cuda.empty_cache()
        return out
    # CPU baseline: the oracle's eval forward + fixed-grid rasters on the first plots of the same parcel
    from oracle import network, projection
    ncores = max(1, min(len(os.sched_getaffinity(0)), 16))
    torch.set_num_threads(ncores)
    sd = network.init_state_dict(0)
    ctimes = []
    with torch.no_grad():
        for it in range(4):
            t0 = time.perf_counter()
            cov, _, _ = network.forward(sd, sample["cloud"], sample["xyz"], args, training=False, use_kdtree=True)
            covb = cov.view(cpu_sample_plots, points, 4)
            for i in range(cpu_sample_plots):
                projection.project_to_2d_rasters(sample["cloud"][i], covb[i].t(), args)
            ctimes.append(time.perf_counter() - t0)
            if it >= 1 and sum(ctimes[1:]) > 8.0:
                break
    ct = sum(ctimes[1:]) / len(ctimes[1:])
    out["cpu_baseline"] = {"value": round(cpu_sample_plots / ct, 3), "unit": "plots/s", "cores": ncores, "kind": "port",
                           "sample": f"{cpu_sample_plots} plots x {points} pts of the same parcel, oracle eval forward + rasters, mean of "
                                     f"{len(ctimes) - 1} passes after 1 warm-up"}
    del batches, model
    torch.cuda.empty_cache()
    return out


def dropin_eager_leg(dev, B, n_points, steps=40, warmup=5):
    """The drop-in as a user of the reference would run it: the loop of /root/reference/learning/train.py:46-66 -- CPU-resident
    batches as the DataLoader collates them, `model(cloud_data)` (which uploads them), `project_to_plotwise_coverages`, the
    three loss terms called one by one as the reference does (`losses.get_*`: on the device each is the fused loss node with the
    other two terms switched off), `loss.backward()`, `torch.optim.Adam.step()`, the three `.item()` reads -- eager, no
    TrainPipeline, no hipGraph, no prefetch, random FPS starts (the reference's are unseeded too)."""
    args = make_args(cuda=dev.index or 0, subsample_size=n_points, ratio1=M1 / n_points, r1=1.0, ratio2=0.25, r2=2.0)
    torch.manual_seed(0)
    model = PointNet2(args).train()
    optimizer = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-3)          # learning/train.py:182
    batches = [make_batch(B, n_points, first_plot=j * B) for j in range(4)]               # host tensors
    times = []
    for it in range(warmup + steps):
        d = batches[it % len(batches)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        cloud_data = {"cloud": d["cloud"], "xyz": d["xyz"]}
        clouds = cloud_data["cloud"]
        gt = d["coverages"].cuda(dev)
        optimizer.zero_grad(set_to_none=True)
        cov, proba = model(cloud_data)
        pred = project_to_plotwise_coverages(cov, clouds, args)
        loss_abs = losses.get_absolute_loss(pred, gt)
        loss_log = losses.get_NLL_loss(proba, d["pdf_all"])               # the reference evaluates its KDE on the CPU; the function uploads
        loss_e = losses.get_entropy_loss(proba)
        loss = loss_abs + args.m * loss_log + args.e * loss_e
        loss.backward()
        optimizer.step()
        _ = (loss_abs.item(), loss_log.item(), loss.item())
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    t = sorted(times[warmup:])[len(times[warmup:]) // 2]
    del model, optimizer
    torch.cuda.empty_cache()
    return {"ms_per_step": round(t * 1e3, 3), "plots_per_s": round(B / t, 1), "statistic": f"median of {steps} steps after {warmup} warm-ups",
            "what": f"learning/train.py:46-66 as written, {B} plots x {n_points} pts per step from HOST tensors (40 MB H2D per step), "
                    "torch.optim.Adam, the three loss terms called one by one (losses.get_*), eager launches, geometry and features back to back"}


def dropin_predict_leg(dev, plots=512, points=10000, batch=20, repeat=3):
    """The drop-in under the reference's INFERENCE loop as written (/root/reference/predict.py:96-126 with
    inference/predict_utils.py:94-102): `model.eval()` under `torch.set_grad_enabled(False)` (predict.py:65,69), CPU-resident batches of
    `args.batch_size` = 20 plots (config.py:85) x `subsample_size` = 10 000 points as its DataLoader collates them,
    `model(cloud_data)`, `get_batch_format`, then PER PLOT `project_to_2d_rasters(clouds[idx], coverages_pointwise[idx], args)`,
    which returns a numpy array (project_to_2d.py:78-113; the drop-in answers a batch's per-plot calls from ONE batched launch
    and ONE device-to-host read, made at the first of them: both arguments are views of the batch's tensors).  The GIS steps behind it (weights
    band, geotransform, GeoTIFF file: GDAL / rasterio) are out of scope and not run.  `config4_parcel_inference` is the same work
    through `inference.predict_parcel` (512 plots per launch, rasters and mosaic on the device)."""
    args = make_args(cuda=dev.index or 0, subsample_size=points)           # reference defaults: ratios .25/.25, r sqrt2/sqrt8
    torch.manual_seed(0)
    model = PointNet2(args).eval()
    batches = []
    for s0 in range(0, plots, batch):
        d = make_batch(min(batch, plots - s0), points, first_plot=s0)
        batches.append({"cloud": d["cloud"], "xyz": d["xyz"]})

    def run():
        n = 0
        for cloud_data in batches:
            clouds = cloud_data["cloud"]
            coverages_pointwise, _ = model(cloud_data)
            if len(coverages_pointwise.shape) < 3:
                coverages_pointwise = model.get_batch_format(coverages_pointwise)
            for idx in range(clouds.shape[0]):
                rasters = project_to_2d_rasters(clouds[idx], coverages_pointwise[idx], args)
                n += 1
        return n, rasters

    times = []
    with torch.no_grad():                 # predict.py:65,69 switch autograd off for the whole script
        run()
        for _ in range(repeat):
            torch.cuda.synchronize()
            t = time.perf_counter()
            n, rasters = run()
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t)
    med = sorted(times)[len(times) // 2]
    del model, batches
    torch.cuda.empty_cache()
    return {"plots_per_s": round(n / med, 1), "ms_per_batch_of_20": round(med / ((n + batch - 1) // batch) * 1e3, 3),
            "runs_s": [round(t, 4) for t in times], "statistic": f"median of {repeat} passes over {n} plots",
            "last_raster_finite_pixels": int(np.isfinite(rasters).sum()),
            "what": f"predict.py:96-126 as written: eval mode, autograd off (predict.py:65,69), {batch} plots x {points} pts per batch from HOST tensors, "
                    "model(cloud_data), get_batch_format, project_to_2d_rasters per plot (numpy out; one launch + one D2H read per BATCH, "
                    "made at the batch's first call); GIS file output not run"}


def secondary_legs(dev):
    """name -> thunk: the other configurations BASELINE.json names, as compact legs of the same process."""
    return {"config2_3sa_arch": lambda: secondary_train_leg(dev, "3sa", 16, 32768, "f32", 100, 10),
            "config5_128k_f32": lambda: secondary_train_leg(dev, "ref", 8, 131072, "f32", 50, 6),
            "config5_128k_bf16": lambda: secondary_train_leg(dev, "ref", 8, 131072, "bf16", 50, 6),
            "config4_parcel_inference": lambda: inference_leg(dev),
            "config4_parcel_inference_bf16": lambda: inference_leg(dev, dtype="bf16"),
            "dropin_eager": lambda: dropin_eager_leg(dev, 16, 32768),
            "dropin_predict": lambda: dropin_predict_leg(dev),
            "exchange_world1": lambda: exchange_leg(dev)}


def main():
    global N_POINTS
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # 200 timed steps = 0.18 s: the timed region ends with a drain of the geometry passes in flight (their counterpart at the
    # start ran before the timer), ~0.5 ms that 50 steps showed as +0.012 ms per step
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--eager", action="store_true", help="do not capture the feature passes into hipGraphs")
    ap.add_argument("--serial", action="store_true",
                    help="headline = the unpipelined step (geometry and features of a batch back to back on one stream)")
    ap.add_argument("--depth", type=int, default=3, help="geometry passes kept in flight ahead of the feature pass")
    ap.add_argument("--no-pair", action="store_true", help="one geometry pass per batch (default: one pass covers PIPE_GROUP = 8 "
                    "consecutive batches -- FPS is one workgroup per plot, so 128 plots take as long as 16)")
    ap.add_argument("--points", type=int, default=N_POINTS, help="points per plot (default = the metric's 32768; other "
                    "values are extra measurements, e.g. 131072 with --plots 8 for BASELINE config 5's plot size)")
    ap.add_argument("--plots", type=int, default=PLOTS_PER_GPU, help="plots per GPU (default = the metric's 16)")
    ap.add_argument("--arch", choices=("ref", "3sa"), default="ref",
                    help="ref = the reference architecture (two ball-query levels + global; parity-checked: the metric); "
                         "3sa = the variant BASELINE config 2 names, three ball-query levels 1024/256/64 with r 1/2/4 m + "
                         "global (not in the reference: throughput only)")
    ap.add_argument("--host-inputs", action="store_true",
                    help="extra measurement, NOT the metric's `value` convention: every step's batch starts in pinned host "
                         "memory and is copied (27 MB) on the side streams in front of its geometry pass")
    ap.add_argument("--dtype", choices=("f32", "bf16"), default="f32",
                    help="f32 = the reference's precision (the metric); bf16 = BASELINE config 5's variant: bfloat16 operands on the "
                         "matrix cores (SA levels and the dense layers over centroids), fp32 accumulate, fp32 everywhere else")
    ap.add_argument("--split-exchange", action="store_true",
                    help="one GPU: use the fallback's multi-GPU launch sequence (backward graph, eager exchange, Adam graph)")
    ap.add_argument("--exchange", choices=("auto", "rccl", "torch", "none"), default="auto",
                    help="the gradient exchange: rccl = ncclAllReduce through the direct binding inside the step's hipGraph (any "
                         "world size, also 1); torch = torch.distributed.all_reduce between two graphs (also at world 1, over a "
                         "one-rank group); auto = none at 1 GPU, rccl at N > 1 when every rank's self-test passes, else torch")
    ap.add_argument("--force-exchange", action="store_true", help="same as --exchange rccl: run the collective even at --gpus 1")
    ap.add_argument("--only-leg", default=None,
                    help="run ONE secondary leg by name and print its JSON (no headline): config2_3sa_arch, config5_128k_f32, "
                         "config5_128k_bf16, config4_parcel_inference, config4_parcel_inference_bf16, dropin_eager -- for profiling and experiments")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary legs (BASELINE configs 2 (3sa-arch), 4, 5 and the eager drop-in loop) that a default "
                         "single-GPU run of the metric's workload attaches under \"secondary\"")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but the launcher set WORLD_SIZE={world}")
    # rehearsal knobs (never set by the driver): all ranks on one device / the gloo backend, to run the N > 1 code path on
    # a one-GPU box
    if os.environ.get("SN2_BENCH_ONE_DEVICE"):
        local_rank = 0
    backend = os.environ.get("SN2_BENCH_BACKEND", "nccl")
    torch.set_num_threads(max(1, min(torch.get_num_threads(), host_cpu_share(), 16)))     # see host_cpu_share
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # HIP deals streams onto its few hardware queues in the order they are created: the loop's own side streams come FIRST, before
    # RCCL / the process group create theirs (a pipeline whose side streams were created after a communicator shared queues
    # with them and ran 15 % slower per step: 0.905 instead of 0.787 ms)
    ops.create_shared_streams(dev)
    # (round 5: the step's own stream at HIGH queue priority -- torch.cuda.Stream(priority=-1), the side streams at 0 -- changed
    # nothing for the pipelined step, 0.7003 against 0.6994 ms, and made the unpipelined one slower, 2.78 against 1.75 ms)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=dev)
        else:
            torch.distributed.init_process_group(backend)

    if a.only_leg:
        legs = secondary_legs(dev)
        if a.only_leg not in legs:
            raise SystemExit(f"--only-leg: one of {sorted(legs)}")
        emit_record({a.only_leg: legs[a.only_leg]()})
        return
    B, N_POINTS = a.plots, a.points
    # depth+1 resident batches (the pipeline's slots)
    pipe_group = 1 if a.no_pair else pipe_group_for(a.steps)
    n_slots = 1 if a.serial else pipe_group * a.depth + pipe_group
    if a.force_exchange:
        a.exchange = "rccl"
    w = build_training(dev, local_rank, rank, world, B, N_POINTS, a.arch, a.dtype, n_slots, exchange=a.exchange)
    args, model, opt, slots, feature_step = w.args, w.model, w.opt, w.slots, w.feature_step
    data = slots[0]

    if hasattr(model, "geometry_fork"):
        model.geometry_fork = False      # eager steps are timed per entry point (HIP events): one stream; the captured
                                         # unpipelined step forks (capture_serial)

    def step():
        """the unpipelined step on slot 0: geometry, features, exchange, Adam back to back on the current stream"""
        loss = feature_step(data)
        opt.step()
        return loss

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # ---- first step instrumented per entry point (also warms every kernel up)
    log("inputs resident; first (instrumented) step")
    with ops.timing() as t0:
        step()
    prof = t0.summary()
    log("first step done: " + ", ".join(f"{k} {v[1]:.3f} ms" for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])[:6]))
    step()
    # which entry point gets the roofline: the longest one ON THE CRITICAL PATH of the timed mode.  Pipelined: the
    # geometry passes (FPS, ball query, 3-NN) run on side streams under the feature pass, which is what bounds the step, so
    # the candidates are the feature-pass entry points; serial: every entry point.  Measured on a second (warm) step.
    # (the shortest of three warm steps per entry point: a single measurement once made the 8 us Adam kernel the "longest"
    # entry point -- an event record that waited on something else)
    prof = {}
    for _ in range(3):
        with ops.timing() as t1:
            step()
        for k, v in t1.summary().items():
            if k not in prof or v[1] < prof[k][1]:
                prof[k] = v
    GEOMETRY = ("sn2_fps", "sn2_ball_query", "sn2_three_nn", "sn2_interp_index", "sn2_sa_order", "sn2_count_sum")
    on_path = {k: v for k, v in prof.items() if a.serial or not k.startswith(GEOMETRY)}
    dominant = max(on_path, key=lambda k: on_path[k][1]) if on_path else None
    longest_geometry = max((k for k in prof if k.startswith(GEOMETRY)), key=lambda k: prof[k][1], default=None)
    log(f"roofline entry point: {dominant}; longest geometry entry point: {longest_geometry}")

    def capture_serial():
        """the whole unpipelined step as ONE hipGraph (single GPU only: no collective inside a graph); the three
        independent chains of the geometry pass become parallel branches of the graph (PointNet2.geometry_fork)"""
        torch.cuda.synchronize()
        fork = getattr(model, "geometry_fork", None)
        if fork is not None:
            model.geometry_fork = True
        try:
            side = ops.shared_stream(dev, "capture")
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                step()                                   # allocator warm-up on the capture stream
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            # (the geometry pass forks into these three and joins them itself; anything else left forked is an error)
            with ops.graph_capture(g, dev, allowed_forks=("fork_b", "fork_c", "pack")):
                l = step()
            g.replay()
            torch.cuda.synchronize()
        finally:
            if fork is not None:
                model.geometry_fork = False
        return g, l

    pipe = None
    graph, loss_static = None, None
    if a.serial:
        launch = "eager"
        for _ in range(max(0, a.warmup - 1)):
            step()
        if not a.eager and world == 1:
            # a failed capture is an error, not a mode switch: an eager run under the same metric string would be a different
            # measurement (--eager asks for it explicitly)
            graph, loss_static = capture_serial()
            launch = "hipGraph"
        mode = f"serial/{launch}"
    else:
        # ---- software pipeline (stratanet2_vegetation_coverage_maps_amd/pipeline.py): geometry of batches i+1..i+depth on
        # side streams while batch i's feature pass (one hipGraph per slot) runs; the all-reduce stays an eager RCCL call
        from stratanet2_vegetation_coverage_maps_amd.pipeline import TrainPipeline
        pipe = TrainPipeline(model, opt, feature_step, slots, depth=a.depth, use_graph=not a.eager,
                             split_exchange=True if a.split_exchange else w.split, group=pipe_group,
                             phase=pipe_phase_for(pipe_group, a.warmup, a.steps))
        pipe.capture()                               # a failed hipGraph capture raises: no silent eager fallback
        launch = "eager" if a.eager else "hipGraph"
        if a.host_inputs:
            pinned = [{k: v.cpu().pin_memory() for k, v in sl.items() if k in ("cloud", "xyz", "gt", "pdf")} for sl in slots]
            pipe.set_feeder(lambda i: pinned[i % len(pinned)])
        pipe.prime()
        for _ in range(a.warmup):
            pipe.step()
        mode = (f"pipelined depth {a.depth}" + (f", one geometry pass per {pipe.group} batches" if pipe.pair else "") + f"/{launch}" +
                (" + H2D of every batch from pinned host memory" if a.host_inputs else ""))
    log(f"mode: {mode}")

    barrier()
    t_start = time.perf_counter()
    tdom = None
    ev_last = ev_end = None
    issued0 = pipe.issued if pipe is not None else 0
    if pipe is not None:
        for _ in range(a.steps):
            loss = pipe.step()
        ev_last, ev_end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev_last.record()                         # behind the last feature pass
        pipe.drain()
        ev_end.record()                          # ... and behind every geometry pass still in flight
    elif graph is not None:
        for _ in range(a.steps):
            graph.replay()
        loss = loss_static
    else:
        with ops.timing({dominant}) as tdom:   # 2 event records per step on the dominant entry point only
            for _ in range(a.steps):
                loss = step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t_start
    barrier()
    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(el, op=torch.distributed.ReduceOp.MAX)
    elapsed = float(el.item())
    drain_ms = None if ev_last is None else float(ev_last.elapsed_time(ev_end))
    log(f"timed region: {elapsed / a.steps * 1e3:.3f} ms/step" + ("" if drain_ms is None else f" (of which the final drain: {drain_ms:.3f} ms in total)"))
    if pipe is not None:
        ops.fps_gave_up(dev)                     # (the region is over: reading the status word costs nothing now)
        ops.global_level_gave_up(dev)
    loss_value = float(loss.item())
    if tdom is None:
        # graph replays / side streams cannot carry the event records: time the dominant entry point over 5 unpipelined
        # eager steps right after the timed region
        with ops.timing({dominant, longest_geometry}) as tdom:
            for _ in range(5):
                step()
    dom = tdom.summary()

    # ---- secondary figure (NOT `value`): the unpipelined step as one hipGraph, the latency of a single batch
    serial_ms = None
    if pipe is not None and world == 1 and not a.eager:
        try:
            g1, _ = capture_serial()
            torch.cuda.synchronize()
            t0s = time.perf_counter()
            for _ in range(a.steps):
                g1.replay()
            torch.cuda.synchronize()
            serial_ms = (time.perf_counter() - t0s) / a.steps * 1e3
            log(f"unpipelined step (one hipGraph): {serial_ms:.3f} ms/step")
        except Exception as exc:                     # noqa: BLE001
            log(f"serial capture failed ({type(exc).__name__}: {exc})")
            torch.cuda.synchronize()

    # ---- one fully instrumented step for the per-entry-point table (after the timed region)
    with ops.timing() as tall:
        step()
    table = tall.summary()
    e1 = e2 = 0
    with torch.no_grad():
        _, _, saved = model._forward_impl(data["xyz"], data["cloud"], data["fps_start"], False)
        e1, e2 = int(saved.tot1.item()), int(saved.tot2.item())
        m1, m2 = saved.M1, saved.M2

    peaks, dom_alone_ms = None, None
    if rank == 0 and world == 1:
        if dominant == "sn2_fp_backward:34+8->34":
            dom_alone_ms = dominant_kernel_alone(step, dev, opt=opt)
        log("measured peaks (stream copy / read, MFMA fp32 / bf16)")
        peaks = measured_peaks(dev)
        log("  " + json.dumps({k: v for k, v in peaks.items() if k.endswith(("GBps", "TFLOPs"))}))
    ranks_seen = world
    if world > 1:
        one = torch.ones(1, device=dev)
        torch.distributed.all_reduce(one)
        ranks_seen = int(one.item())
    if rank == 0:
        ms = elapsed / a.steps * 1e3
        kernels = []
        for k, (c, tms) in sorted(table.items(), key=lambda kv: -kv[1][1]):
            by = algorithmic_bytes(k, B, N_POINTS, m1, m2, e1, e2)
            kernels.append({"entry": k, "ms": round(tms, 4),
                            "alg_GBps": None if by is None else round(by / (tms * 1e-3) / 1e9, 1)})
        CALLS_PER_STEP = {"sn2_interp_index": 3, "sn2_sa_order": 2}     # entry points whose byte model covers all their calls of a step

        def roofline_of(entry, note):
            if not entry or entry not in dom:
                return None
            c, tms = dom[entry]
            avg_ms = tms / (c / CALLS_PER_STEP.get(entry, 1))
            by = algorithmic_bytes(entry, B, N_POINTS, m1, m2, e1, e2)
            ach = None if by is None else by / (avg_ms * 1e-3) / 1e9
            traffic, tsrc = pmc_traffic(entry)
            return {"kernel": entry, "dominant_device_kernel": DOMINANT_KERNEL.get(entry), "bound": "hbm",
                    "achieved": None if ach is None else round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": None if ach is None else round(ach / HBM_PEAK_GBS, 6), "traffic": traffic,
                    "traffic_source": tsrc, "avg_ms": round(avg_ms, 4), "algorithmic_bytes": by,
                    "dominant_device_kernel_avg_ms_rocprof": rocprof_kernel_avg_ms(entry),
                    "timing": ("HIP events inside the timed region" if mode == "serial/eager" else
                               "HIP events over 5 unpipelined eager steps right after the timed region"), "note": note}
        # ---- the whole step against both roofs: compulsory HBM bytes and dense-layer flops of ONE step
        step_bytes, step_flops = step_model_figures(B, N_POINTS, m1, m2, e1, e2)
        MFMA_PEAK = {"f32": 157.3e12, "bf16": 2.5e15}[a.dtype]
        whole_step = {"compulsory_bytes": int(step_bytes), "hbm_frac": round(step_bytes / (ms * 1e-3) / (HBM_PEAK_GBS * 1e9), 4),
                      "dense_flops": int(step_flops), "mfma_frac": round(step_flops / (ms * 1e-3) / MFMA_PEAK, 4),
                      "mfma_peak_TFLOPs": MFMA_PEAK / 1e12,
                      "what": "SURVEY.md 8d: compulsory HBM bytes (2.2 x forward) and dense-layer flops (3 x forward - 352 E1) of "
                              "one step at the measured message counts, over the timed ms_per_step"}
        roof = roofline_of(dominant, "longest entry point of the feature pass, the stream that bounds the pipelined step "
                                     "(the geometry passes run beside it on side streams)" if not a.serial else
                           "longest entry point of the step")
        roof_geo = None if a.serial else roofline_of(
            longest_geometry, "longest entry point overall, off the critical path: fps is a chain of arg-max rounds in one "
                              "workgroup per plot (up to 8 exact samples per round; latency-bound: see us_per_sample and "
                              "workgroups / compute_units); its HBM traffic is 12 B/point once, so the HBM fraction says "
                              "nothing about it")
        out = {"metric": "plots/s fwd+bwd, 32k-pt synthetic plots, batch 16/GPU", "value": round(world * B / (ms * 1e-3), 2),
               "unit": "plots/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 4),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
               "config": {"workload": (("C2 " if (B, N_POINTS) == (16, 32768) else "NOT the metric's size, ") +
                                       ("ref-arch" if a.arch == "ref" else "3sa-arch (not in the reference: throughput only)")) +
                                      f": {B} plots/GPU x {N_POINTS} pts, SA npoint " +
                                      ("1024/256 + global, r 1/2 m, " if a.arch == "ref" else "1024/256/64 + global, r 1/2/4 m, ") +
                                      "train step fwd+P2+loss+bwd+Adam, " +
                                      ("inputs resident in HBM" if not a.host_inputs else
                                       "EVERY batch copied from pinned host memory inside the timed region (not the metric's convention)") +
                                      (", bf16 operands on the matrix cores (SA levels, SA3/FP3/FP2), fp32 accumulate" if a.dtype == "bf16" else ""),
                          "mode": mode + ("" if a.serial else
                                          ": every step runs one feature pass (this batch); the position-only kernels (FPS, "
                                          "ball query, 3-NN) of every batch run exactly once, for later batches on side "
                                          "streams" + (f", {pipe.group} consecutive batches per launch" if (pipe is not None and pipe.pair) else "") +
                                          "; distinct batches in the slots"),
                          "plots_per_gpu": B, "points_per_plot": N_POINTS, "messages_sa1": e1, "messages_sa2": e2,
                          "batches_per_geometry_pass": (pipe.group if pipe is not None else 1),
                          "geometry_passes_launched_in_timed_region": (None if pipe is None else (pipe.issued - issued0) // pipe.group),
                          "drain_ms_inside_timed_region": (None if drain_ms is None else round(drain_ms, 4)),
                          "exchange": w.exchange,
                          "parallelism": f"dp{world} (plots sharded; one 60 KB gradient all-reduce)" if world > 1 else "single GPU"},
               "loss": round(loss_value, 6), "roofline": roof, "kernels": kernels,
               "host_cores_of_this_rank": os.environ.get("SN2_BENCH_PINNED", f"all:{len(os.sched_getaffinity(0))}")}
        if peaks is not None:
            out["measured_peaks"] = peaks
        # what a SCALE record can be audited by: the library, the ranks that took part, the exchange that really ran
        try:
            from stratanet2_vegetation_coverage_maps_amd import rccl as _rccl
            out["config"]["rccl_version"] = _rccl.version()
        except Exception as exc:                         # noqa: BLE001
            out["config"]["rccl_version"] = f"unavailable ({type(exc).__name__})"
        out["config"]["ranks_seen"] = ranks_seen
        out["config"]["exchange_in_graph"] = bool(getattr(opt, "comm", None) is not None)
        if roof is not None:
            # ---- one record, scalar fields, each with ONE denominator:
            #   frac / entry_point_frac : the ENTRY POINT's algorithmic bytes over its live HIP-event time (all its kernels), vs 8 TB/s
            #   dominant_kernel_frac    : the named device kernel ALONE (its own bytes over its own live duration), vs 8 TB/s
            #   *_of_measured_copy      : the same against the stream-copy rate measured in this run
            #   whole_step_*            : SURVEY 8d's compulsory bytes / dense flops of one step over ms_per_step
            roof["entry_point_frac"] = roof["frac"]
            roof["entry_point_kernels"] = ENTRY_KERNELS.get(roof["kernel"], [DOMINANT_KERNEL.get(roof["kernel"])])
            if dom_alone_ms is not None:
                dk_bytes = (288 + 32 + 136) * N_POINTS * B      # h + dy rows, skip columns in, d pre-activation rows out
                roof["dominant_kernel_ms"] = round(dom_alone_ms, 4)
                roof["dominant_kernel_bytes"] = dk_bytes
                roof["dominant_kernel_GBps"] = round(dk_bytes / (dom_alone_ms * 1e-3) / 1e9, 1)
                roof["dominant_kernel_frac"] = round(dk_bytes / (dom_alone_ms * 1e-3) / (HBM_PEAK_GBS * 1e9), 4)
            roof["whole_step_hbm_frac"] = whole_step["hbm_frac"]
            roof["whole_step_mfma_frac"] = whole_step["mfma_frac"]
            roof["whole_step_compulsory_bytes"] = whole_step["compulsory_bytes"]
            roof["whole_step_dense_flops"] = whole_step["dense_flops"]
            if peaks is not None:
                cp = peaks["hbm_copy_GBps"]
                roof["measured_copy_peak_GBps"] = cp
                if roof["achieved"] is not None:
                    roof["frac_of_measured_copy"] = round(roof["achieved"] / cp, 4)
                if dom_alone_ms is not None:
                    roof["dominant_kernel_frac_of_measured_copy"] = round(roof["dominant_kernel_GBps"] / cp, 4)
                roof["whole_step_hbm_frac_of_measured_copy"] = round(whole_step["compulsory_bytes"] / (ms * 1e-3) / (cp * 1e9), 4)
                mp = peaks["mfma_bf16_TFLOPs" if a.dtype == "bf16" else "mfma_f32_TFLOPs"]
                roof["whole_step_mfma_frac_of_measured"] = round(whole_step["dense_flops"] / (ms * 1e-3) / (mp * 1e12), 4)
            roof["whole_step"] = whole_step
        if roof_geo is not None:
            if roof_geo["kernel"].startswith("sn2_fps"):
                # farthest point sampling: M strictly sequential arg-max rounds per plot -- what describes it is the time per
                # sample and how little of the chip it occupies, not a bandwidth
                roof_geo["us_per_sample"] = round(roof_geo["avg_ms"] * 1e3 / max(1, m1 - 1), 4)
                roof_geo["workgroups"] = B * 8           # the timed (unpipelined) form: eight workgroups per plot
                roof_geo["compute_units"] = 256
                roof_geo["bound"] = "latency"
            out["roofline_off_critical_path"] = roof_geo
        if serial_ms is not None:
            out["unpipelined"] = {"ms_per_step": round(serial_ms, 4), "plots_per_s": round(B / (serial_ms * 1e-3), 2),
                                  "what": "the same step with geometry and features of ONE batch back to back on one "
                                          "stream, captured as one hipGraph (single-batch latency; not the headline value)"}
        default_workload = (B, N_POINTS, a.arch, a.dtype) == (PLOTS_PER_GPU, 32768, "ref", "f32") and not (a.serial or a.eager or a.host_inputs)
        if world == 1 and default_workload and not a.no_secondary:
            # ---- the other configurations BASELINE.json names, driver-timed in the same run (compact legs; never `value`)
            pipe = graph = None                         # the headline's pipeline, graphs and resident batches are done with
            slots.clear()
            torch.cuda.empty_cache()
            sec = {}
            for name, fn in secondary_legs(dev).items():
                log(f"secondary leg: {name}")
                try:
                    sec[name] = fn()
                    log(f"  {name}: " + json.dumps({k: v for k, v in sec[name].items() if k in ('ms_per_step', 'plots_per_s', 'seconds_per_parcel')}))
                except Exception as exc:                     # noqa: BLE001  (a failed leg must not lose the headline line)
                    sec[name] = {"error": f"{type(exc).__name__}: {exc}"}
                    log(f"  {name} FAILED: {sec[name]['error']}")
                    torch.cuda.synchronize()
            out["secondary"] = sec
        if world == 1 and not a.no_cpu_baseline:
            log("cpu baseline (oracle on the host cores)")
            out["cpu_baseline"] = cpu_baseline()
            out["speedup_vs_cpu_baseline"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
        emit_record(out)
    if getattr(opt, "comm", None) is not None:
        import gc
        pipe = graph = None                    # the graphs that hold the captured collective go first
        gc.collect()
        torch.cuda.synchronize()
        opt.comm.destroy()                     # before torch's own communicator goes (and never from a destructor at exit)
        opt.comm = None
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
