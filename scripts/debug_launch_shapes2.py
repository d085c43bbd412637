import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
import test_gpu_launch_shapes as T
from stratanet2_vegetation_coverage_maps_amd.synthetic import make_args, make_batch
B, N = 512, 10000
args = make_args(subsample_size=N)
m = T._trained_stats_model(args, 0, N)
d = make_batch(B, N, first_plot=0)
cov, proba, rasters, pix = T._eval(m, d["cloud"], d["xyz"], args)
cov2, _, _, _ = T._eval(m, d["cloud"], d["xyz"], args)
torch.cuda.synchronize()
covb = cov.view(B, N, 4)
print("512 twice same bits:", torch.equal(cov, cov2))
for s in (0, 32, 480):
    c32, p32, r32, x32 = T._eval(m, d["cloud"][s:s + 32], d["xyz"][s:s + 32], args)
    c32b, _, _, _ = T._eval(m, d["cloud"][s:s + 32], d["xyz"][s:s + 32], args)
    a, b = c32.view(32, N, 4), covb[s:s + 32]
    ne = (a != b)
    plots = ne.view(32, -1).any(dim=1).nonzero().flatten().tolist()
    print(f"s={s}: 32 twice same {torch.equal(c32, c32b)}; differing elements {int(ne.sum())}, plots {plots}, max |d| {float((a - b).abs().max()):.3e}")
    if plots:
        pl = plots[0]
        pts = ne[pl].any(dim=1).nonzero().flatten()
        print("   plot", pl, "differing points", pts.numel(), pts[:10].tolist())
