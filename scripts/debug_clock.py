"""What clock does the chip hold while an FPS pass runs on another stream?  A one-wave kernel that waits for a fixed number
of SHADER clocks (sn2_debug_spin) is timed with HIP events: wall time = clocks / clock.  Alone, beside FPS passes over LOAD_PLOTS
plots (8 waves each), and beside an idle kernel of SPIN_BLOCKS one-wave workgroups."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stratanet2_vegetation_coverage_maps_amd import hip_ops as ops, _lib
from stratanet2_vegetation_coverage_maps_amd.synthetic import make_batch

dev = torch.device("cuda:0")
lib = _lib.load()
CLK = int(2.0e6)


def probe(n=20):
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        _lib.check(lib.sn2_debug_spin(1, CLK, None, torch.cuda.current_stream().cuda_stream), "spin")
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


torch.zeros(1, device=dev)
t_alone = probe()
print(f"alone: {t_alone * 1e3:.1f} us for {CLK} clocks -> {CLK / t_alone / 1e3:.0f} MHz")
for LP in (2, 8, 32):
    N = 32768
    xyz = make_batch(LP, N)["xyz"].to(dev)
    fs = torch.zeros(LP, dtype=torch.int32, device=dev)
    side = torch.cuda.Stream()
    ops.fps(xyz, 1024, fs, waves=8)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        for _ in range(40):
            ops.fps(xyz, 1024, fs, waves=8)
    t = probe()
    torch.cuda.synchronize()
    print(f"beside FPS over {LP} plots: {t * 1e3:.1f} us -> {CLK / t / 1e3:.0f} MHz")
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    _lib.check(lib.sn2_debug_spin(256, int(60e-3 * 2.4e9), None, torch.cuda.current_stream().cuda_stream), "spin")
t = probe()
torch.cuda.synchronize()
print(f"beside 256 idle one-wave workgroups: {t * 1e3:.1f} us -> {CLK / t / 1e3:.0f} MHz")
