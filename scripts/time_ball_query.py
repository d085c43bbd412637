"""Diagnostic: sn2_ball_query (grid path) with GQ_INFLIGHT = 1, 2, 4 candidate blocks per turn, at the sizes that matter:
16 and 32 plots x 32 768 points (one batch / a pair of batches), 8 x 131 072 (config 5)."""
import ctypes, os, subprocess, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stratanet2_vegetation_coverage_maps_amd import hip_ops as ops, _lib
from stratanet2_vegetation_coverage_maps_amd.synthetic import make_batch
csrc = os.path.join(ROOT, "stratanet2_vegetation_coverage_maps_amd/csrc")
cases = [(16, 32768), (32, 32768), (8, 131072)]
data = {}
for B, N in cases:
    xyz = torch.cat([make_batch(min(B, 16), N, first_plot=16 * k)["xyz"] for k in range((B + 15) // 16)])[:B].cuda().contiguous()
    idx, cs, ca, ws = ops.fps(xyz, 1024, None, return_ws=True)
    data[(B, N)] = (xyz, cs, ws)
torch.cuda.synchronize()
ref = {}
for inflight in (1, 2, 4):
    so = os.path.join(ROOT, f"gpurun_out/libgq{inflight}.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-shared",
                           f"-DSN2_GQ_INFLIGHT={inflight}", os.path.join(csrc, "geometry.hip"), "-o", so])
    lib = ctypes.CDLL(so)
    lib.sn2_ball_query.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_float,
                                   ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    for (B, N), (xyz, cs, ws) in data.items():
        M, cap = 1024, 2000
        nbr = torch.empty(B * M, cap, dtype=torch.int32, device="cuda"); cnt = torch.empty(B * M, dtype=torch.int32, device="cuda")
        def run():
            rc = lib.sn2_ball_query(xyz.data_ptr(), B, N, cs.data_ptr(), M, ctypes.c_float(1.0), cap, nbr.data_ptr(), cnt.data_ptr(),
                                    None, ws.data_ptr(), None)
            assert rc == 0, rc
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            run()
        b.record(); torch.cuda.synchronize()
        key = (B, N)
        live = torch.arange(cap, device="cuda")[None, :] < cnt[:, None]
        sig = (cnt.clone(), nbr[live].clone())
        same = True
        if key in ref:
            same = torch.equal(ref[key][0], sig[0]) and torch.equal(ref[key][1], sig[1])
        else:
            ref[key] = sig
        print(f"in flight {inflight}: {B:2d} x {N:6d}: {a.elapsed_time(b) / 20 * 1e3:7.1f} us per launch, lists equal to the first variant: {same}", flush=True)
