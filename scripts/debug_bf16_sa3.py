#!/usr/bin/env python3
"""sn2_fp_forward on the global-SA block's shape (32 + 3 -> 64, no interpolation) with bf16 operands against torch on the same
rounded operands; parts of the input zeroed to see which columns go wrong."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stratanet2_vegetation_coverage_maps_amd import hip_ops as ops

torch.manual_seed(0)
dev = "cuda"
B, M = 4, 256
lin = torch.nn.Linear(35, 64).to(dev); bn = torch.nn.BatchNorm1d(64).to(dev)
r = lambda t: t.to(torch.bfloat16).to(torch.float32)
for name, fx, fp in (("all", 1.0, 1.0), ("features only", 1.0, 0.0), ("positions only", 0.0, 1.0)):
    x2 = torch.randn(B * M, 32, device=dev) * fx
    pos = torch.zeros(B * M, 4, device=dev); pos[:, :3] = torch.randn(B * M, 3, device=dev) * 5 * fp
    for bf in (False, True):
        bb = ops.BlockBuffers(lin, bn); bb.mma_bf16 = bf
        h = torch.zeros(B * M, 64, device=dev)
        d = ops.fp_desc(bb, B, M, M, 32, 3, x2, h, skip=pos)
        ops.fp_forward(d, True)
        u = torch.cat([x2, pos[:, :3]], 1)
        W, b = lin.weight.detach(), lin.bias.detach()
        want = torch.relu((r(u).double() @ r(W).double().t()).float() + b) if bf else torch.relu(u @ W.t() + b)
        err = (h - want).abs()
        bad = (err.max(1).values > 1e-2).nonzero().flatten().tolist()
        if bad: print("   bad rows:", bad[:40], "... channels of row", bad[0], ":", (err[bad[0]] > 1e-2).nonzero().flatten().tolist()[:20])
        print(f"{name:15s} bf16={bf!s:5s} max err {float(err.max()):.3e}  worst channel {int(err.max(0).values.argmax())} "
              f"rows with err > 1e-2: {int((err.max(1).values > 1e-2).sum())} of {B * M}; mean |h| {float(want.abs().mean()):.3f}")
