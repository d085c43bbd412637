"""Diagnostic: per-kernel durations of a real training step's FP1 backward (row pass, source pass, merge) for a diagnostic build
of fp.hip, from a rocprofv3 kernel trace:   SN2_KSTATS_PROG=scripts/time_fp1_bwd.py bash scripts/kstats.sh OUT   (build flags
through SN2_DBG_FLAGS="-DSN2_BR_XCD ...")."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
flags = os.environ.get("SN2_DBG_FLAGS", "").split()
if flags:
    csrc = os.path.join(ROOT, "stratanet2_vegetation_coverage_maps_amd/csrc")
    so = os.path.join(ROOT, "gpurun_out/libfp1b_dbg.so")
    os.makedirs(os.path.dirname(so), exist_ok=True)
    srcs = [os.path.join(csrc, f) for f in ("geometry.hip", "sa.hip", "sa_mfma.hip", "fp.hip", "project.hip", "loss.hip", "misc.hip", "net.hip")]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-shared"] + flags + srcs + ["-o", so])
    from stratanet2_vegetation_coverage_maps_amd import _lib
    _lib.LIB_PATH = so
import torch
from stratanet2_vegetation_coverage_maps_amd import PointNet2, project_to_plotwise_coverages, losses
from stratanet2_vegetation_coverage_maps_amd.synthetic import make_args, make_batch
B, N = 16, 32768
args = make_args(cuda=0, subsample_size=N, ratio1=1024 / N, r1=1.0, ratio2=0.25, r2=2.0)
torch.manual_seed(0)
model = PointNet2(args).train()
dev = torch.device("cuda:0")
h = make_batch(B, N)
inp = {k: h[k].to(dev) for k in ("cloud", "xyz")}
gt, pdf = h["coverages"].to(dev), h["pdf_all"].to(dev)
for it in range(12):
    model.zero_grad(set_to_none=True)
    cov, proba = model({"cloud": inp["cloud"], "xyz": inp["xyz"], "fps_start": torch.zeros(2, B, dtype=torch.int32, device=dev)})
    pred = project_to_plotwise_coverages(cov, inp["cloud"], args)
    loss, _ = losses.total_loss(pred, proba, gt, pdf, args.m, args.e)
    loss.backward()
torch.cuda.synchronize()
print("flags", flags, "loss", float(loss))
