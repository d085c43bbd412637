"""Diagnostic: per-wave phase stamps of sa_mfma_bwd_kernel<8, 2, 16, 16, 3> (SA1 backward, layer 1) inside a real training step
(built with -DSN2_SA_STAMPS into gpurun_out/; never shipped).  Prints, over the waves that had work: when each phase ends
relative to the kernel's first stamp (median / 90 % / max, in us at 2.4 GHz) and the items / steps per wave."""
import ctypes, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
csrc = os.path.join(ROOT, "stratanet2_vegetation_coverage_maps_amd/csrc")
so = os.path.join(ROOT, "gpurun_out/libsa_dbg.so")
srcs = [os.path.join(csrc, f) for f in ("geometry.hip", "sa.hip", "sa_mfma.hip", "fp.hip", "project.hip", "loss.hip", "misc.hip", "net.hip")]
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-shared", "-DSN2_SA_STAMPS"] + srcs + ["-o", so])
from stratanet2_vegetation_coverage_maps_amd import _lib
_lib.LIB_PATH = so
import torch
from stratanet2_vegetation_coverage_maps_amd import PointNet2, project_to_plotwise_coverages, losses
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
inp = {k: h[k].to(dev) for k in ("cloud", "xyz")}
gt, pdf = h["coverages"].to(dev), h["pdf_all"].to(dev)
names = ["LDS zeroed / constants", "first item known", "first item done", "all items done", "slab stored", "barrier", "atomics issued"]
for it in range(3):
    opt.zero_grad()
    cov, proba = model({"cloud": inp["cloud"], "xyz": inp["xyz"], "fps_start": torch.zeros(2, B, dtype=torch.int32, device=dev)})
    pred = project_to_plotwise_coverages(cov, inp["cloud"], args)
    loss, _ = losses.total_loss(pred, proba, gt, pdf, args.m, args.e)
    loss.backward()
    opt.step()
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * (4096 * 10))()
    _lib.load().sn2_debug_sa_stamps(out)
    t = np.array(list(out), dtype=np.float64).reshape(4096, 10)
    live = t[:, 7] > 0
    t = t[live]
    t0 = t[:, 0].min()
    print(f"step {it}: {live.sum()} waves; kernel span {(t[:, 7].max() - t0) / 2400:.1f} us; wave start spread {(t[:, 0].max() - t0) / 2400:.1f} us; "
          f"items/wave median {np.median(t[:, 8]):.0f} max {t[:, 8].max():.0f}; steps/wave median {np.median(t[:, 9]):.0f} max {t[:, 9].max():.0f}")
    for i, n in enumerate(names):
        col = t[:, i + 1]
        ok = col > 0
        rel = (col[ok] - t0) / 2400
        dur = (col[ok] - t[ok, 0]) / 2400
        print(f"   {n:26s} at median {np.median(rel):6.1f} p90 {np.percentile(rel, 90):6.1f} max {rel.max():6.1f} us | since wave start: median {np.median(dur):6.1f} max {dur.max():6.1f}")
