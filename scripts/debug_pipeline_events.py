"""Timeline of the pipelined loop from HIP events (no profiler: rocprofv3 makes hipGraphLaunch block for a whole step)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stratanet2_vegetation_coverage_maps_amd import PointNet2, project_to_plotwise_coverages, losses
from stratanet2_vegetation_coverage_maps_amd.optim import FlatAdam, flatten_parameters
from stratanet2_vegetation_coverage_maps_amd.pipeline import TrainPipeline
from stratanet2_vegetation_coverage_maps_amd.synthetic import make_args, make_batch

B, N = 16, 32768
args = make_args(cuda=0, subsample_size=N, ratio1=1024 / N, r1=1.0, ratio2=0.25, r2=2.0)
torch.manual_seed(0)
model = PointNet2(args).train()
flatten_parameters(model)
opt = FlatAdam(model, lr=1e-3, weight_decay=1e-3)
dev = torch.device("cuda:0")


def mk(j):
    h = make_batch(B, N, first_plot=j * B)
    return {"cloud": h["cloud"].to(dev), "xyz": h["xyz"].to(dev), "fps_start": torch.zeros(2, B, dtype=torch.int32, device=dev),
            "gt": h["coverages"].to(dev), "pdf": h["pdf_all"].to(dev)}


def fstep(inp, geo=None):
    opt.zero_grad()
    cd = {"cloud": inp["cloud"], "xyz": inp["xyz"], "fps_start": inp["fps_start"]}
    if geo is not None:
        cd["geometry"] = geo
    cov, proba = model(cd)
    pred = project_to_plotwise_coverages(cov, inp["cloud"], args)
    loss, _ = losses.total_loss(pred, proba, inp["gt"], inp["pdf"], args.m, args.e)
    loss.backward()
    return loss


depth = int(os.environ.get("DEPTH", "2"))
pipe = TrainPipeline(model, opt, fstep, [mk(j) for j in range(depth + 1)], depth=depth)
pipe.capture()
E = lambda: torch.cuda.Event(enable_timing=True)
marks = []          # (label, event)
orig_issue = pipe.issue_geometry


def issue(i=None):
    i = pipe.issued if i is None else i
    st = pipe.side[i % pipe.n_streams]
    a, b = E(), E()
    # the start mark must sit behind the slot_done wait: replicate the wait, then mark
    k = i % pipe.slots
    if pipe.slot_done[k] is not None:
        st.wait_event(pipe.slot_done[k])
    a.record(st)
    orig_issue(i)
    b.record(st)
    marks.append((f"geo{i} s{i % pipe.n_streams}", a, b, time.perf_counter()))


pipe.issue_geometry = issue
pipe.prime()
main = torch.cuda.current_stream()
for _ in range(6):
    pipe.step()
torch.cuda.synchronize()
marks.clear()
ref = E(); ref.record(main)
th0 = time.perf_counter()
for i in range(8):
    a, b = E(), E()
    main.wait_event(pipe.geo_ready[pipe.done % pipe.slots])
    a.record(main)
    th = time.perf_counter()
    pipe.step()
    b.record(main)
    marks.append((f"feat{pipe.done - 1}", a, b, th))
torch.cuda.synchronize()
for lab, a, b, th in sorted(marks, key=lambda m: ref.elapsed_time(m[1])):
    print(f"{lab:12s} gpu start {ref.elapsed_time(a):8.3f} end {ref.elapsed_time(b):8.3f}   host issue at {(th - th0) * 1e3:8.3f} ms")
