"""Where does the HOST spend its time in the reference's eager training loop on the drop-in (bench.py's `dropin_eager` leg)?
Wall-clock stamps between the phases of each step (no extra synchronisation: the last phase, the three .item() reads, absorbs
whatever the device still has queued)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from stratanet2_vegetation_coverage_maps_amd import PointNet2, project_to_plotwise_coverages, losses
from stratanet2_vegetation_coverage_maps_amd.synthetic import make_args, make_batch

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
bench.ops.create_shared_streams(dev)
B, N = 16, 32768
args = make_args(cuda=0, subsample_size=N, ratio1=1024 / N, r1=1.0, ratio2=0.25, r2=2.0)
torch.manual_seed(0)
model = PointNet2(args).train()
opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-3)
batches = [make_batch(B, N, first_plot=j * B) for j in range(4)]
names = ["targets up", "zero_grad", "forward", "projection", "loss ops", "backward", "adam", "item x3"]
acc = [0.0] * len(names)
print("torch threads", torch.get_num_threads(), "host cpu share", bench.host_cpu_share())
if os.environ.get("SN2_THREADS"):
    torch.set_num_threads(int(os.environ["SN2_THREADS"]))
    print("torch threads now", torch.get_num_threads())
import stratanet2_vegetation_coverage_maps_amd.hip_ops as _ops
_up = _ops.PinnedRing.upload
def _timed_upload(self, t, stream=None, dtype=None, out=None, consumer=None):
    a = time.perf_counter()
    r = _up(self, t, stream, dtype, out, consumer)
    print(f"    upload {t.numel() * 4 / 1e6:6.1f} MB: {(time.perf_counter() - a) * 1e3:7.3f} ms", flush=True)
    return r
_ops.PinnedRing.upload = _timed_upload
_ops._UPLOAD_TRACE = []
GPU_EV = os.environ.get("SN2_GPU_EVENTS") == "1"     # also: when did the DEVICE get to the end of each phase (events on the main stream)
def _ev():
    if not GPU_EV:
        return None
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e
for it in range(14):
    d = batches[it % 4]
    torch.cuda.synchronize()
    ev = [_ev()]
    t = [time.perf_counter()]
    gt = d["coverages"].cuda(dev); t.append(time.perf_counter()); ev.append(_ev())
    opt.zero_grad(set_to_none=True); t.append(time.perf_counter()); ev.append(_ev())
    cov, proba = model({"cloud": d["cloud"], "xyz": d["xyz"]}); t.append(time.perf_counter()); ev.append(_ev())
    pred = project_to_plotwise_coverages(cov, d["cloud"], args); t.append(time.perf_counter()); ev.append(_ev())
    la = losses.get_absolute_loss(pred, gt)
    ll = losses.get_NLL_loss(proba, d["pdf_all"])
    le = losses.get_entropy_loss(proba)
    loss = la + args.m * ll + args.e * le; t.append(time.perf_counter()); ev.append(_ev())
    loss.backward(); t.append(time.perf_counter()); ev.append(_ev())
    opt.step(); t.append(time.perf_counter()); ev.append(_ev())
    _ = (la.item(), ll.item(), loss.item()); t.append(time.perf_counter())
    if GPU_EV:
        torch.cuda.synchronize()
        print("   host issued at ", " ".join(f"{(x - t[0]) * 1e3:7.3f}" for x in t[1:]))
        print("   device done at ", " ".join(f"{ev[0].elapsed_time(e):7.3f}" for e in ev[1:]))
    print("    ", [(w, round(v, 3)) for w, v in _ops._UPLOAD_TRACE]); _ops._UPLOAD_TRACE.clear()
    print(it, " ".join(f"{(t[k + 1] - t[k]) * 1e3:7.3f}" for k in range(len(names))), flush=True)
    if it >= 4:
        for k in range(len(names)):
            acc[k] += (t[k + 1] - t[k]) * 1e3 / 10
print("host ms per step by phase:", {n: round(a, 3) for n, a in zip(names, acc)}, "total", round(sum(acc), 3))
if os.environ.get("SN2_CPROFILE"):
    # where inside python: cProfile over ten more steps (forward + projection + backward only), top functions by own time
    import cProfile, pstats
    _ops.PinnedRing.upload = _up
    pr = cProfile.Profile()
    # the backward runs on the autograd engine's thread: profile it there
    from stratanet2_vegetation_coverage_maps_amd import point_net2 as _pn
    pb = cProfile.Profile()
    _bw = _pn.PointNet2._backward_impl
    def _bw_prof(self, s, dcov, dproba):
        pb.enable()
        try:
            return _bw(self, s, dcov, dproba)
        finally:
            pb.disable()
    _pn.PointNet2._backward_impl = _bw_prof
    for it in range(10):
        d = batches[it % 4]
        gt = d["coverages"].cuda(dev)
        opt.zero_grad(set_to_none=True)
        pr.enable()
        cov, proba = model({"cloud": d["cloud"], "xyz": d["xyz"]})
        pred = project_to_plotwise_coverages(cov, d["cloud"], args)
        pr.disable()
        loss = losses.get_absolute_loss(pred, gt) + args.m * losses.get_NLL_loss(proba, d["pdf_all"]) + args.e * losses.get_entropy_loss(proba)
        pr.enable()
        loss.backward()
        pr.disable()
        opt.step()
        torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(25)
    print("---- the network's backward (engine thread)")
    pstats.Stats(pb).sort_stats("tottime").print_stats(25)
    print("---- by cumulative time")
    pstats.Stats(pb).sort_stats("cumtime").print_stats(25)
