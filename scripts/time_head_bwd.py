"""Diagnostic: `sn2_head_backward` alone on the chip (524 288 rows = config 2), HIP events, for several builds of fp.hip:
    python scripts/time_head_bwd.py [-DSWITCH ...]        e.g. -DSN2_NO_FLUSH, -DSN2_HB_DIAG=1 (no d-row stores), =2 (no arithmetic),
                                                          =3 (loads only), -DSN2_HB_STAMPS (phase stamps of one wave's second turn)
Each switch set is built into gpurun_out/ (never shipped)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
csrc = os.path.join(ROOT, "stratanet2_vegetation_coverage_maps_amd/csrc")
flags = [a for a in sys.argv[1:] if a.startswith("-D")]
if flags:
    so = os.path.join(ROOT, "gpurun_out/libhb_dbg.so")
    os.makedirs(os.path.dirname(so), exist_ok=True)
    srcs = [os.path.join(csrc, f) for f in ("geometry.hip", "sa.hip", "sa_mfma.hip", "fp.hip", "project.hip", "loss.hip", "misc.hip")]
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
dcov, dproba = torch.randn(R, 4, device=dev), torch.randn(R, 4, device=dev)
dy = torch.empty(R, 36, device=dev)
n_flat = sum(p.numel() for p in list(lin1.parameters()) + list(lin2.parameters()))
stride = (n_flat + 63) // 64 * 64
arena = torch.zeros(ops.GRAD_IMAGES * stride, device=dev)
views, o = [], 0
for p in list(lin1.parameters()) + list(lin2.parameters()):
    views.append(arena[o:o + p.numel()].view(p.shape))
    o += p.numel()
d = ops.head_desc(f, fa, fc, lin1, lin2, dcov=dcov, dproba=dproba, dy=dy, grads=views, grad_images=(ops.GRAD_IMAGES, stride))
for _ in range(5):
    ops.head_backward(d)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(50):
    ops.head_backward(d)
ev[1].record()
torch.cuda.synchronize()
ms = ev[0].elapsed_time(ev[1]) / 50
if any("SN2_HB_STAMPS" in a for a in flags):
    import ctypes
    out = (ctypes.c_ulonglong * 16)()
    _lib.load().sn2_debug_hb_stamps(out)
    tt = list(out)
    names = ["tile in (affine) + prefetch issue", "lin1", "lin2", "softmax bwd", "dW2", "d pre", "dW1", "d rows", "store"]
    print("second turn of one wave, s_memtime ticks: total", tt[9] - tt[0], "; ".join(f"{n} {tt[i + 1] - tt[i]}" for i, n in enumerate(names)), flush=True)
print(f"{' '.join(flags) or 'shipped build'}: head backward {ms * 1e3:.1f} us for {R} rows = {R * 320 / ms / 1e9:.2f} TB/s of its 320 B/row", flush=True)
