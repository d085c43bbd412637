"""Diagnostic: per-phase cycle shares of the bucketed FPS round loop (build with -DSN2_FPS_STAMPS; never shipped)."""
import ctypes, subprocess, sys, os, time
import torch
sys.path.insert(0, ".")
from stratanet2_vegetation_coverage_maps_amd import hip_ops as ops
from stratanet2_vegetation_coverage_maps_amd.synthetic import make_batch
csrc = "stratanet2_vegetation_coverage_maps_amd/csrc"
so = "gpurun_out/libfps_dbg.so"
extra = sys.argv[1:]
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-shared", "-DSN2_FPS_STAMPS"] + extra +
                      [f"{csrc}/geometry.hip", "-o", so])
lib = ctypes.CDLL(so)
B, N, M = 16, 32768, 1024
xyz = make_batch(B, N)["xyz"].cuda()
idx = torch.empty(B, M, dtype=torch.int32, device="cuda"); cs = torch.empty(B, 3, M, device="cuda"); ca = torch.empty(B * M, 4, device="cuda")
ws = torch.empty(ops.fps_ws_words(B, N), dtype=torch.int32, device="cuda")
lib.sn2_fps_waves.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 3 + [ctypes.c_void_p] * 5 + [ctypes.c_int, ctypes.c_void_p]
WAVES = 1 if os.environ.get("SN2_FPS_SPECULATE", "1") == "0" else 0
def run():
    rc = lib.sn2_fps_waves(xyz.data_ptr(), B, N, M, None, idx.data_ptr(), cs.data_ptr(), ca.data_ptr(), ws.data_ptr(), WAVES, None)
    assert rc == 0, rc
run(); torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 8)()
lib.sn2_debug_fps_stamps(out); base = list(out)
t = time.perf_counter(); run(); torch.cuda.synchronize(); el = time.perf_counter() - t
lib.sn2_debug_fps_stamps(out)
d = [out[i] - base[i] for i in range(8)]
rounds = d[6]
if os.environ.get("SN2_FPS_SPECULATE", "1") != "0":
    names = ["(A) tests", "(B) dirty buckets", "(D) select + accept (wave 0)", "barrier waits"]
    tot = sum(d[:4])
    print(f"wall {el*1e3:.3f} ms for {rounds} super-rounds, {d[4]} samples = {d[4]/rounds:.2f} per super-round (stamped build); "
          f"wave 0: dirty buckets/super-round {d[5]/rounds:.2f}, tie searches {d[7]}")
    for n, v in zip(names, d[:4]):
        print(f"  {n:30s} {v/rounds:8.1f} ticks/super-round  {100*v/tot:5.1f}%")
    print(f"  total {tot/rounds:.1f} ticks/super-round = {tot/d[4]:.1f} per sample (s_memtime ticks = shader clocks)")
    o2 = (ctypes.c_ulonglong * 32)()
    lib.sn2_debug_fps_stamps2(o2)
    print(f"  (C) per-wave top-4 (wave 0, both runs): {o2[24] / (2 * rounds):.1f} ticks/super-round")
    print("  accepted per super-round (both runs), histogram 0..16:", list(o2[:17]))
    if o2[25]:
        print(f"  wave 0's queue entries (both runs): {o2[25]}; with a changed point {o2[26] / o2[25]:.2f}; needing the maximum recomputed "
              f"{o2[27] / o2[25]:.2f}; samples per entry {o2[28] / o2[25]:.2f}; changed points per entry {o2[29] / o2[25]:.1f}")
else:
    names = ["(a) tests", "(b) dirty loop", "(c) wave reduce+publish", "barrier wait", "(d) select/next sample"]
    tot = sum(d[:5])
    print(f"wall {el*1e3:.3f} ms for {rounds} rounds (stamped build); wave0: dirty buckets/round {d[5]/rounds:.2f}, tie rounds {d[7]}")
    for n, v in zip(names, d[:5]):
        print(f"  {n:28s} {v/rounds:8.1f} ticks/round  {100*v/tot:5.1f}%")
    print(f"  total {tot/rounds:.1f} ticks/round (s_memtime ticks: 100 MHz)")
