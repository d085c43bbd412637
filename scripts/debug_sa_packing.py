import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stratanet2_vegetation_coverage_maps_amd import hip_ops as ops
from stratanet2_vegetation_coverage_maps_amd.synthetic import make_batch
d = make_batch(16, 32768)
xyz = d["xyz"].cuda()
idx1, cs, ca, ws = ops.fps(xyz, 1024, None, return_ws=True)
nbr, cnt, tot = ops.ball_query(xyz, cs, 1.0, 2000, fps_ws=ws)
n = cnt.cpu().numpy().astype(np.int64)
print("E", n.sum(), "mean", n.mean(), "median", np.median(n), "p90", np.percentile(n, 90), "max", n.max())
print("steps now (64/centroid):", np.ceil(n / 64).sum())
for name, order in (("fps order", n), ("sorted by count", np.sort(n))):
    p2 = order.reshape(-1, 2).max(1); p4 = order.reshape(-1, 4).max(1)
    print(name, " pairs(32):", np.ceil(p2 / 32).sum(), " quads(16):", np.ceil(p4 / 16).sum())
print("ideal E/64:", n.sum() / 64)
