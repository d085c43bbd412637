"""Diagnostic: `sn2_head_forward` alone on the chip (524 288 rows = config 2), HIP events, for several builds of fp.hip:
    python scripts/time_head_fwd.py [-DSN2_HF_OCC=2 ...]     (workgroups per CU the kernel is compiled for: its register budget)
Each switch set is built into gpurun_out/ (never shipped)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
csrc = os.path.join(ROOT, "stratanet2_vegetation_coverage_maps_amd/csrc")
flags = [a for a in sys.argv[1:] if a.startswith("-D")]
if flags:
    so = os.path.join(ROOT, "gpurun_out/libhf_dbg.so")
    os.makedirs(os.path.dirname(so), exist_ok=True)
    srcs = [os.path.join(csrc, f) for f in ("geometry.hip", "sa.hip", "sa_mfma.hip", "fp.hip", "project.hip", "loss.hip", "misc.hip", "net.hip")]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-shared"] + flags + srcs + ["-o", so])
    from stratanet2_vegetation_coverage_maps_amd import _lib
    _lib.LIB_PATH = so
import torch
from stratanet2_vegetation_coverage_maps_amd import hip_ops as ops

dev = torch.device("cuda:0")
R = int(os.environ.get("ROWS", 16 * 32768))
torch.manual_seed(0)
f = torch.randn(R, 36, device=dev)
fa, fc = torch.rand(34, device=dev) + 0.5, torch.randn(34, device=dev) * 0.1
lin1, lin2 = torch.nn.Linear(34, 16).to(dev), torch.nn.Linear(16, 5).to(dev)
cov, proba = torch.empty(R, 4, device=dev), torch.empty(R, 4, device=dev)
d = ops.head_desc(f, fa, fc, lin1, lin2, cov, proba)
for _ in range(5):
    ops.head_forward(d)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(50):
    ops.head_forward(d)
ev[1].record()
torch.cuda.synchronize()
ms = ev[0].elapsed_time(ev[1]) / 50
print(f"{' '.join(flags) or 'shipped build'}: head forward {ms * 1e3:.1f} us for {R} rows = {R * 176 / ms / 1e9:.2f} TB/s of its 176 B/row; "
      f"checksum {float(cov.sum()):.6f} {float(proba.sum()):.3f}", flush=True)
