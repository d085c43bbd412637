#!/usr/bin/env python3
"""Every FPS kernel variant on one batch: indices against the brute-force kernel, timeouts counted by the multi-workgroup
kernel (control word 1 of the workspace), HIP-event time per launch.  python scripts/time_fps_cluster.py [B] [N] [M] [dup]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stratanet2_vegetation_coverage_maps_amd import hip_ops as ops  # noqa: E402
from stratanet2_vegetation_coverage_maps_amd.synthetic import make_batch  # noqa: E402

B, N, M = (int(v) for v in (sys.argv[1:4] + ["16", "32768", "1024"][len(sys.argv[1:4]):]))
dup = len(sys.argv) > 4 and sys.argv[4] == "dup"
xyz = make_batch(B, N)["xyz"]
if dup:
    q = N // 4
    xyz[:, :, N - q:] = xyz[:, :, :q]
xyz = xyz.cuda()
start = torch.arange(B, dtype=torch.int32, device="cuda") * 977 % N
ref, _, _ = ops.fps(xyz, M, start, bucketed=False) if N <= 32768 else ops.fps(xyz, M, start, waves=16)
VARIANTS = [int(v) for v in os.environ.get("SN2_FPS_VARIANTS", "16,8,34,36,40,66,68,72").split(",")]
for w in VARIANTS:
    out = (torch.empty(B, M, dtype=torch.int32, device="cuda"), torch.empty(B, 3, M, device="cuda"),
           torch.empty(B * M, 4, device="cuda"), torch.zeros(ops.fps_ws_words(B, N), dtype=torch.int32, device="cuda"))
    ops.fps(xyz, M, start, out=out, waves=w)
    torch.cuda.synchronize()
    ok = bool(torch.equal(out[0], ref))
    ctl = ops.fps_ws_ctl(out[3], B, N).tolist()
    msg = f"waves={w:3d}: equal={ok} per-XCD arrivals={ctl[16:24]} overflow={ctl[0]} timeouts={ctl[1]}"
    if not ok:
        bad = (out[0] != ref).nonzero()
        msg += f" first mismatch (plot, sample) {bad[0].tolist()} of {len(bad)}"
    for _ in range(3):
        ops.fps(xyz, M, start, out=out, waves=w)
    torch.cuda.synchronize()
    with ops.timing() as t:
        for _ in range(20):
            ops.fps(xyz, M, start, out=out, waves=w)
    for k, (c, ms) in t.summary().items():
        msg += f"  {ms / c:.4f} ms per launch = {ms / c / (M - 1) * 1e3:.3f} us per sample"
    print(msg, flush=True)
    if ctl[10]:
        names = ["A test", "barrier 1", "B update", "barrier 2", "C select+publish (wave 0)", "wait own records (wave 1)", "sweep (wave 1)", "D rank+tests (wave 1)"]
        r = ctl[10]
        print("    wave 0 of workgroup 0, clocks per super-round: " + ", ".join(f"{n} {ctl[2 + i] * 16 / r:.0f}" for i, n in enumerate(names))
              + f" | rounds {r}, accepted/round {ctl[11] / r:.2f}, fallbacks {ctl[12]}, tie rounds {ctl[13] & 0xFFFF}, empty rounds {ctl[13] >> 16},"
                f" survivors/round {ctl[14] / r:.1f}, sweeps/round {ctl[15] / r:.1f}")
