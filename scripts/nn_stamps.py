"""Diagnostic: per-wave cost of the grid 3-NN (build with -DSN2_NN_STAMPS; never shipped): cycles, candidates tested,
final ring and query-box size of every wave of 64 Morton-adjacent targets, on bench-like synthetic plots."""
import ctypes, subprocess, sys
import numpy as np
import torch
sys.path.insert(0, ".")
from stratanet2_vegetation_coverage_maps_amd.synthetic import make_batch
csrc = "stratanet2_vegetation_coverage_maps_amd/csrc"
so = "gpurun_out/libnn_dbg.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-shared", "-DSN2_NN_STAMPS",
                       f"{csrc}/geometry.hip", "-o", so])
lib = ctypes.CDLL(so)
B, N, M = 16, 32768, 1024
xyz = make_batch(B, N)["xyz"].cuda().contiguous()
idx = torch.empty(B, M, dtype=torch.int32, device="cuda"); cs = torch.empty(B, 3, M, device="cuda"); ca = torch.empty(B * M, 4, device="cuda")
ws = torch.empty(5 * B * N + 16 * 4200, dtype=torch.int32, device="cuda")
lib.sn2_fps.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 3 + [ctypes.c_void_p] * 6
assert lib.sn2_fps(xyz.data_ptr(), B, N, M, None, idx.data_ptr(), cs.data_ptr(), ca.data_ptr(), ws.data_ptr(), None) == 0
kidx = torch.empty(B * N, 3, dtype=torch.int32, device="cuda"); kw = torch.empty(B * N, 3, device="cuda")
nws = torch.empty(B * (4 * M + 1032) + 64, dtype=torch.int32, device="cuda")
lib.sn2_three_nn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 5
for _ in range(2):
    assert lib.sn2_three_nn(cs.data_ptr(), B, M, xyz.data_ptr(), N, 3, kidx.data_ptr(), kw.data_ptr(), nws.data_ptr(), ws.data_ptr(), None) == 0
torch.cuda.synchronize()
nw = B * N // 64
out = (ctypes.c_ulonglong * (4 * nw))()
lib.sn2_debug_nn_stamps(out, 4 * nw)
a = np.array(list(out), dtype=np.float64).reshape(nw, 4)
cyc, cand, rho, box = a.T
print("waves", nw, " s_memtime ticks per wave: mean %.0f  median %.0f  p90 %.0f  p99 %.0f  max %.0f" % (cyc.mean(), np.median(cyc), np.percentile(cyc, 90), np.percentile(cyc, 99), cyc.max()))
print("candidates per wave: mean %.0f  median %.0f  p90 %.0f  p99 %.0f  max %.0f" % (cand.mean(), np.median(cand), np.percentile(cand, 90), np.percentile(cand, 99), cand.max()))
print("final ring: ", {int(r): int((rho == r).sum()) for r in np.unique(rho)})
print("query box cells: mean %.1f  median %.0f  p90 %.0f  max %.0f" % (box.mean(), np.median(box), np.percentile(box, 90), box.max()))
order = np.argsort(-cyc)[:8]
print("slowest waves (ticks, candidates, ring, box):", [(int(cyc[i]), int(cand[i]), int(rho[i]), int(box[i])) for i in order])
print("ticks per candidate (median): %.1f" % np.median(cyc / np.maximum(cand, 1)))
