"""Diagnostic: phase stamps of one workgroup of fp_bwd_split_kernel<64,32,64> (FP3 backward) inside a real training step
(build with -DSN2_SPLIT_STAMPS into gpurun_out/; never shipped)."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
csrc = os.path.join(ROOT, "stratanet2_vegetation_coverage_maps_amd/csrc")
so = os.path.join(ROOT, "gpurun_out/libsplit_dbg.so")
srcs = [os.path.join(csrc, f) for f in ("geometry.hip", "sa.hip", "sa_mfma.hip", "fp.hip", "project.hip", "loss.hip", "misc.hip", "net.hip")]
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-shared", "-DSN2_SPLIT_STAMPS"] + srcs + ["-o", so])
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
for it in range(3):
    opt.zero_grad()
    cov, proba = model({"cloud": inp["cloud"], "xyz": inp["xyz"], "fps_start": torch.zeros(2, B, dtype=torch.int32, device=dev)})
    pred = project_to_plotwise_coverages(cov, inp["cloud"], args)
    loss, _ = losses.total_loss(pred, proba, gt, pdf, args.m, args.e)
    loss.backward()
    opt.step()
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 16)()
    _lib.load().sn2_debug_split_stamps(out)
    t = list(out)
    names = ["zero LDS + barrier", "build inputs / dp", "barrier", "dW on MFMA", "du on MFMA + slab stores", "barrier", "du / dskip write-out", "dW flush (atomics)"]
    print(f"step {it}: total {t[8] - t[0]} ticks; " + "; ".join(f"{n} {t[i + 1] - t[i]}" for i, n in enumerate(names)), flush=True)
