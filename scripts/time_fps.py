#!/usr/bin/env python3
"""Time sn2_fps alone (HIP events, 20 launches): python scripts/time_fps.py [B] [N] [M]; SN2_FPS_SPECULATE=0 for the
one-sample-per-round kernel (waves = 1 of sn2_fps_waves).  Also checks the indices against the brute-force kernel."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stratanet2_vegetation_coverage_maps_amd import hip_ops as ops  # noqa: E402
from stratanet2_vegetation_coverage_maps_amd.synthetic import make_batch  # noqa: E402

B, N, M = (int(v) for v in (sys.argv[1:4] + ["16", "32768", "1024"][len(sys.argv) - 1:]))
xyz = make_batch(B, N)["xyz"].cuda()
start = torch.zeros(B, dtype=torch.int32, device="cuda")
WAVES = 1 if os.environ.get('SN2_FPS_SPECULATE', '1') == '0' else int(os.environ.get('SN2_FPS_WAVES', '0'))
idx, cs, ca = ops.fps(xyz, M, start, waves=WAVES)
ref, _, _ = ops.fps(xyz, M, start, bucketed=False) if N <= 32768 else (idx, None, None)
print("indices equal to the brute-force kernel:", bool(torch.equal(idx, ref)))
if not torch.equal(idx, ref):
    bad = (idx != ref).nonzero()
    print("first mismatches (plot, sample):", bad[:5].tolist(), idx[bad[0, 0], bad[0, 1] - 2: bad[0, 1] + 3].tolist(),
          ref[bad[0, 0], bad[0, 1] - 2: bad[0, 1] + 3].tolist())
out = (torch.empty_like(idx), torch.empty_like(cs), torch.empty_like(ca),
       torch.empty(ops.fps_ws_words(B, N), dtype=torch.int32, device="cuda"))
for _ in range(3):
    ops.fps(xyz, M, start, out=out, waves=WAVES)
torch.cuda.synchronize()
with ops.timing() as t:
    for _ in range(20):
        ops.fps(xyz, M, start, out=out, waves=WAVES)
for k, (c, ms) in t.summary().items():
    print(f"{k}: {ms / c:.4f} ms per launch ({B} plots x {N} -> {M}; speculate={os.environ.get('SN2_FPS_SPECULATE', '1')})"
          f" = {ms / c / (M - 1) * 1e3:.3f} us per sample")
