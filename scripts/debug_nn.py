import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stratanet2_vegetation_coverage_maps_amd import hip_ops as ops
from stratanet2_vegetation_coverage_maps_amd.synthetic import make_batch
d = make_batch(16, 32768)
xyz = d["xyz"].cuda()
idx1, cs, ca, ws = ops.fps(xyz, 1024, None, return_ws=True)
i, w = ops.three_nn(cs, xyz, 3, dst_fps_ws=ws)
torch.cuda.synchronize()
c = i[:, 1].cpu().numpy(); r = i[:, 2].cpu().numpy()
print("candidates: mean %.1f median %.0f p90 %.0f p99 %.0f max %d" % (c.mean(), np.median(c), np.percentile(c, 90), np.percentile(c, 99), c.max()))
rings = r // 10000; bw = (r % 10000) // 100; bh = r % 100
print("rings hist", np.bincount(rings)[:20])
print("box w hist", np.bincount(bw)[:20])
print("box h hist", np.bincount(bh)[:20])
