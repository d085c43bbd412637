"""Diagnostic: phase stamps of workgroup 0 of global_level_fwd_kernel inside a real training forward (16 x 32 768), from a
-DSN2_GL_STAMPS build of fp.hip made into gpurun_out/ (never shipped)."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
csrc = os.path.join(ROOT, "stratanet2_vegetation_coverage_maps_amd/csrc")
so = os.path.join(ROOT, "gpurun_out/libgl_dbg.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
srcs = [os.path.join(csrc, f) for f in ("geometry.hip", "sa.hip", "sa_mfma.hip", "fp.hip", "project.hip", "loss.hip", "misc.hip")]
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-shared", "-DSN2_GL_STAMPS"] + srcs + ["-o", so])
from stratanet2_vegetation_coverage_maps_amd import _lib
_lib.LIB_PATH = so
import torch
from stratanet2_vegetation_coverage_maps_amd import PointNet2
from stratanet2_vegetation_coverage_maps_amd.synthetic import make_args, make_batch
B, N = 16, 32768
args = make_args(cuda=0, subsample_size=N, ratio1=1024 / N, r1=1.0, ratio2=0.25, r2=2.0)
torch.manual_seed(0)
model = PointNet2(args).train()
dev = torch.device("cuda:0")
h = make_batch(B, N)
inp = {"cloud": h["cloud"].to(dev), "xyz": h["xyz"].to(dev), "fps_start": torch.zeros(2, B, dtype=torch.int32, device=dev)}
names = ["SA3 tiles", "exchange 1", "finalise", "(sync)", "plot max", "FP3 tiles", "exchange 2", "finalise"]
for it in range(4):
    model(inp)
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 16)()
    _lib.load().sn2_debug_gl_stamps(out)
    t = list(out)
    print(f"forward {it}: total {t[7] - t[0]} ticks (100 MHz); " + "; ".join(f"{n} {t[i + 1] - t[i]}" for i, n in enumerate(names[:7])), flush=True)
