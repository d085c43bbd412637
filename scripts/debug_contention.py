"""Which feature-pass entry points slow down while FPS passes (one 1024-lane workgroup per plot, 32 plots = 32 CUs) run on a
side stream?  Eager feature passes timed per entry point with HIP events, alone and under a continuous FPS load."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stratanet2_vegetation_coverage_maps_amd import PointNet2, project_to_plotwise_coverages, losses, hip_ops as ops
from stratanet2_vegetation_coverage_maps_amd.optim import FlatAdam, flatten_parameters
from stratanet2_vegetation_coverage_maps_amd.synthetic import make_args, make_batch

B, N = 16, 32768
args = make_args(cuda=0, subsample_size=N, ratio1=1024 / N, r1=1.0, ratio2=0.25, r2=2.0)
torch.manual_seed(0)
model = PointNet2(args).train()
flatten_parameters(model)
opt = FlatAdam(model, lr=1e-3, weight_decay=1e-3)
dev = torch.device("cuda:0")
h = make_batch(B, N)
inp = {"cloud": h["cloud"].to(dev), "xyz": h["xyz"].to(dev), "fps_start": torch.zeros(2, B, dtype=torch.int32, device=dev),
       "gt": h["coverages"].to(dev), "pdf": h["pdf_all"].to(dev)}
geo = model.alloc_geometry(B, N, dev)
model._geometry(inp["xyz"], inp["fps_start"], out=geo, fork=False)
# the load: FPS passes over LOAD_PLOTS plots (one 8-wave workgroup = one CU each; default 32 = the pipelined loop's pair pass)
LP = int(os.environ.get("LOAD_PLOTS", "32"))
xyz2 = torch.cat([inp["xyz"], inp["xyz"]])[:LP].contiguous()
fs2 = torch.zeros(LP, dtype=torch.int32, device=dev)
gp = model.alloc_geometry(LP, N, dev)
# DUMMY_STREAMS streams created (and used once) before the load's stream: moves it to another hardware queue
_dummies = [torch.cuda.Stream() for _ in range(int(os.environ.get("DUMMY_STREAMS", "0")))]
for _s in _dummies:
    with torch.cuda.stream(_s):
        torch.zeros(1, device=dev)
side = torch.cuda.Stream()
SPIN = int(os.environ.get("SPIN", "0"))


def fstep():
    opt.zero_grad()
    cov, proba = model({"cloud": inp["cloud"], "xyz": inp["xyz"], "fps_start": inp["fps_start"], "geometry": geo})
    pred = project_to_plotwise_coverages(cov, inp["cloud"], args)
    loss, _ = losses.total_loss(pred, proba, inp["gt"], inp["pdf"], args.m, args.e)
    loss.backward()
    opt.step()


def timed(n, load):
    for _ in range(3):
        fstep()
    torch.cuda.synchronize()
    if load:
        with torch.cuda.stream(side):
            if SPIN:
                # SPIN = "blocks": idle one-wave workgroups for the whole timed stretch (60 ms) instead of FPS passes
                from stratanet2_vegetation_coverage_maps_amd import _lib
                _lib.check(_lib.load().sn2_debug_spin(SPIN, int(60e-3 * 2.4e9), None, torch.cuda.current_stream().cuda_stream), "spin")
            else:
                for _ in range(load):
                    ops.fps(xyz2, 1024, fs2, out=(gp.idx1, gp.pos1_soa, gp.pos1_aos, gp.ws1), waves=8)
    with ops.timing() as t:
        for _ in range(n):
            fstep()
    s = t.summary()
    torch.cuda.synchronize()
    return {k: v[1] / n for k, v in s.items()}


n = 10
alone = timed(n, 0)
# an eager pass takes ~2-3 ms of host time: 40 FPS passes (1.26 ms each) cover the 10 timed passes
loaded = timed(n, int(os.environ.get("FPS_PASSES", "40")))
tot_a = sum(alone.values()); tot_l = sum(loaded.values())
print(f"{'entry point':34s} {'alone ms':>9s} {'under FPS':>10s} {'ratio':>6s} {'+us':>7s}")
for k in sorted(alone, key=lambda k: -(loaded.get(k, 0) - alone[k])):
    print(f"{k:34s} {alone[k]:9.4f} {loaded.get(k, 0):10.4f} {loaded.get(k, 0) / alone[k]:6.2f} {(loaded.get(k, 0) - alone[k]) * 1e3:7.1f}")
print(f"{'sum':34s} {tot_a:9.4f} {tot_l:10.4f} {tot_l / tot_a:6.2f} {(tot_l - tot_a) * 1e3:7.1f}")
