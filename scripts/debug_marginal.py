"""What does each position-only entry point cost the PIPELINED step?  The slots' inputs never change, so after a warm-up the
tables in the geometry buffers stay valid and an entry point can be turned into a no-op: the step time that disappears with
it is its marginal cost under load (its own duration says little: a 16-workgroup kernel beside a full-chip feature pass is
nearly free, a full-chip kernel costs its whole length)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stratanet2_vegetation_coverage_maps_amd import PointNet2, project_to_plotwise_coverages, losses, hip_ops as ops
from stratanet2_vegetation_coverage_maps_amd.optim import FlatAdam, flatten_parameters
from stratanet2_vegetation_coverage_maps_amd.pipeline import TrainPipeline
from stratanet2_vegetation_coverage_maps_amd.synthetic import make_args, make_batch

B, N = int(os.environ.get("PLOTS", "16")), int(os.environ.get("POINTS", "32768"))
args = make_args(cuda=0, subsample_size=N, ratio1=1024 / N, r1=1.0, ratio2=0.25, r2=2.0)
torch.manual_seed(0)
model = PointNet2(args).train()
model.p2_diam_pix = args.diam_pix          # as bench.py: the geometry passes also compute the projection's pixel ids
flatten_parameters(model)
opt = FlatAdam(model, lr=1e-3, weight_decay=1e-3, fold_gradient_images=True)      # as bench.py
dev = torch.device("cuda:0")
depth = 3


def mk(j):
    h = make_batch(B, N, first_plot=j * B)
    return {"cloud": h["cloud"].to(dev), "xyz": h["xyz"].to(dev), "fps_start": torch.zeros(2, B, dtype=torch.int32, device=dev),
            "gt": h["coverages"].to(dev), "pdf": h["pdf_all"].to(dev)}


def fstep(inp, geo=None):
    opt.zero_grad()
    cd = {"cloud": inp["cloud"], "xyz": inp["xyz"], "fps_start": inp["fps_start"], "geometry": geo}
    cov, proba = model(cd)
    loss, _, _ = losses.projected_total_loss(cov, proba, inp["cloud"], inp["gt"], inp["pdf"], args, geometry=geo, model=model)
    loss.backward()
    return loss


GROUP = int(os.environ.get("GROUP", "8"))          # batches per geometry pass (bench.py: 8)
pipe = TrainPipeline(model, opt, fstep, [mk(j) for j in range(GROUP * depth + GROUP)], depth=depth, group=GROUP)
pipe.capture()
pipe.prime()


def measure(n=200):
    for _ in range(20):
        pipe.step()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        pipe.step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


base = measure()
print(f"all entry points: {base:.4f} ms/step ({pipe.group} batches per geometry pass)", flush=True)
if os.environ.get("ONLY_BASE"):
    pipe.drain(); torch.cuda.synchronize(); sys.exit(0)
real = {}
# (round 5: a pass over several batches builds the per-batch products with the *_group entry points)
CS, SO, II = ("count_sum_group", "sa_order_group", "interp_index_group") if GROUP > 1 else ("count_sum", "sa_order", "interp_index")
groups = [("fps",), ("ball_query",), (CS,), (SO,), ("three_nn",), (II,), ("pack_rows", "plot_pixels"),
          ("ball_query", CS, SO, "three_nn", II),
          ("fps", "ball_query", CS, SO, "three_nn", II),
          ("fps", "ball_query", CS, SO, "three_nn", II, "pack_rows", "plot_pixels")]
for gnames in groups:
    for name in gnames:
        real[name] = getattr(ops, name)
        setattr(ops, name, lambda *a, **k: None)
    t = measure()
    print(f"without {'+'.join(gnames):60s} {t:.4f} ms/step  ({base - t:+.4f})", flush=True)
    for name in gnames:
        setattr(ops, name, real[name])
again = measure()
print(f"all entry points again: {again:.4f} ms/step", flush=True)
if len(sys.argv) > 1 and sys.argv[1] == "three":         # the three 3-NN / inverted-index levels one by one
    for which in (0, 1, 2):
        cnt = {"n": 0}
        for name in ("three_nn", "interp_index"):
            def wrap(fn, name=name):
                state = {"k": 0}
                def f(*a, **k):
                    i = state["k"] % 3
                    state["k"] += 1
                    return None if i == which else fn(*a, **k)
                return f
            real[name] = getattr(ops, name)
            setattr(ops, name, wrap(real[name]))
        t = measure()
        print(f"without 3-NN table + inverted index of level {3 - which}: {t:.4f} ms/step ({base - t:+.4f})", flush=True)
        for name in ("three_nn", "interp_index"):
            setattr(ops, name, real[name])
if len(sys.argv) > 1 and sys.argv[1] == "overhead":      # the loop's own cost: no geometry pass issued at all
    for name in ("fps", "ball_query", "count_sum", "sa_order", "three_nn", "interp_index"):
        real[name] = getattr(ops, name)
        setattr(ops, name, lambda *a, **k: None)
    print(f"entry points as no-ops (copies into the pair buffers, events stay): {measure():.4f} ms/step", flush=True)
    pipe.drain()
    torch.cuda.synchronize()
    issue = pipe.issue_geometry
    def no_issue(i):
        pipe.issued = max(pipe.issued, i + pipe.group)
    pipe.issue_geometry = no_issue
    print(f"no geometry pass issued (main stream: wait on an old event, graph, record): {measure():.4f} ms/step", flush=True)
    k = [0]
    def replay_only(n=200):
        for _ in range(20):
            pipe.graph_fb[k[0] % pipe.slots].replay(); k[0] += 1
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            pipe.graph_fb[k[0] % pipe.slots].replay(); k[0] += 1
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n * 1e3
    print(f"the eight feature graphs replayed back to back: {replay_only():.4f} ms/step", flush=True)
    def replay_one(n=200):
        for _ in range(20):
            pipe.graph_fb[0].replay()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            pipe.graph_fb[0].replay()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n * 1e3
    print(f"one feature graph replayed back to back: {replay_one():.4f} ms/step", flush=True)
    def events_only(i, copies=False):
        i -= i % 2
        k0, k1 = i % pipe.slots, (i + 1) % pipe.slots
        st = pipe.side[(i // 2) % pipe.n_streams]
        for kk in (k0, k1):
            if pipe.slot_done[kk] is not None:
                st.wait_event(pipe.slot_done[kk])
        if copies:
            with torch.cuda.stream(st):
                for h, kk in enumerate((k0, k1)):
                    pipe.xyz2[k0 // 2][h * B:(h + 1) * B].copy_(pipe.inputs[kk]["xyz"], non_blocking=True)
                    pipe.fs2[k0 // 2][:, h * B:(h + 1) * B].copy_(pipe.inputs[kk]["fps_start"], non_blocking=True)
        pipe.geo_ready[k0].record(st)
        pipe.geo_ready[k1].record(st)
        pipe.issued = max(pipe.issued, i + 2)
    pipe.issue_geometry = events_only
    print(f"side streams: wait for the slots, record (no copies, no kernels): {measure():.4f} ms/step", flush=True)
    pipe.issue_geometry = lambda i: events_only(i, True)
    print(f"side streams: wait, the four copies into the pair buffers, record: {measure():.4f} ms/step", flush=True)
    def variant(side_wait, side_record, i):
        i -= i % 2
        k0, k1 = i % pipe.slots, (i + 1) % pipe.slots
        st = pipe.side[(i // 2) % pipe.n_streams]
        if side_wait:
            for kk in (k0, k1):
                if pipe.slot_done[kk] is not None:
                    st.wait_event(pipe.slot_done[kk])
        if side_record:
            pipe.geo_ready[k0].record(st)
            pipe.geo_ready[k1].record(st)
        pipe.issued = max(pipe.issued, i + 2)
    for sw, sr in ((True, False), (False, True)):
        pipe.drain(); torch.cuda.synchronize()
        pipe.issue_geometry = lambda i, sw=sw, sr=sr: variant(sw, sr, i)
        t0 = time.perf_counter()
        for _ in range(200):
            pipe.step()
        th = (time.perf_counter() - t0) / 200 * 1e3
        torch.cuda.synchronize()
        print(f"side streams wait={sw} record={sr}: {measure():.4f} ms/step (host issue {th:.3f} ms/step)", flush=True)
    def variant_host(i):
        i -= i % 2
        k0, k1 = i % pipe.slots, (i + 1) % pipe.slots
        st = pipe.side[(i // 2) % pipe.n_streams]
        ev = pipe.slot_done[k1] if pipe.slot_done[k1] is not None else pipe.slot_done[k0]
        if ev is not None:
            ev.synchronize()                      # the HOST waits for the later of the two feature passes
        pipe.geo_ready[k0].record(st)
        pipe.geo_ready[k1].record(st)
        pipe.issued = max(pipe.issued, i + 2)
    pipe.drain(); torch.cuda.synchronize()
    pipe.issue_geometry = variant_host
    print(f"host waits for the slot (event.synchronize), side streams record: {measure():.4f} ms/step", flush=True)
    pipe.issue_geometry = issue
pipe.drain()
torch.cuda.synchronize()
