"""Diagnostic: FP1's forward (sn2_fp_forward, source-side form: table + row pass + finalisation) and backward alone on the chip at
config 2's shapes (16 plots x 32 768 points, 1024 sources per plot, real 3-NN tables of synthetic plots), HIP events, for several
builds of fp.hip:
    python scripts/time_fp1.py [-DSN2_FR_DIAG=<bits>] ...      bits of the row pass (both forms): 1 no h stores, 2 no table gathers, 4 no skip contraction, 8 (form 1) stores the compiler counts
    SN2_GRID_MULT=2 python scripts/time_fp1.py                 (the row kernels' grids x 2)
Each switch set is built into gpurun_out/ (never shipped).  Per-kernel durations: run it under scripts/kstats.sh."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
csrc = os.path.join(ROOT, "stratanet2_vegetation_coverage_maps_amd/csrc")
flags = [a for a in sys.argv[1:] if a.startswith("-D")]
if flags:
    so = os.path.join(ROOT, "gpurun_out/libfp1_dbg.so")
    os.makedirs(os.path.dirname(so), exist_ok=True)
    srcs = [os.path.join(csrc, f) for f in ("geometry.hip", "sa.hip", "sa_mfma.hip", "fp.hip", "project.hip", "loss.hip", "misc.hip", "net.hip")]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-shared"] + flags + srcs + ["-o", so])
    from stratanet2_vegetation_coverage_maps_amd import _lib
    _lib.LIB_PATH = so
import torch
from stratanet2_vegetation_coverage_maps_amd import hip_ops as ops
from stratanet2_vegetation_coverage_maps_amd.synthetic import make_batch

dev = torch.device("cuda:0")
B, N, M1 = int(os.environ.get("PLOTS", 16)), int(os.environ.get("POINTS", 32768)), 1024
torch.manual_seed(0)
d = make_batch(B, N)
xyz = d["xyz"].to(dev).float().contiguous()
idx1, pos1_soa, pos1_aos = ops.fps(xyz, M1, torch.zeros(B, dtype=torch.int32, device=dev))[:3]
knn = ops.three_nn(pos1_soa, xyz, 3)
lin, bn = torch.nn.Linear(42, 34).to(dev), torch.nn.BatchNorm1d(34).to(dev)
blk = ops.BlockBuffers(lin, bn)
h2 = torch.randn(B * M1, 36, device=dev)
a2, c2 = torch.rand(34, device=dev) + 0.5, torch.randn(34, device=dev) * 0.1
rows0 = torch.randn(B * N, 12, device=dev)
h1 = torch.empty(B * N, 36, device=dev)
fwd = ops.fp_desc(blk, B, N, M1, 34, 8, h2, h1, src_affine=(a2, c2), knn=knn, skip=rows0[:, 0:8])


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(n):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / n * 1e3


from stratanet2_vegetation_coverage_maps_amd import _lib as _l
for form in (1, 0):
    _l.load().sn2_debug_fp_rows_form(form)
    t_f = timed(lambda: ops.fp_forward(fwd, 1))
    print(f"{' '.join(flags) or 'shipped build'} (SN2_GRID_MULT={os.environ.get('SN2_GRID_MULT', '1')}), row pass form {form}: FP1 forward entry "
          f"{t_f:.1f} us (table + row pass + finalisation) for {B} x {N} rows", flush=True)
