import sys
import numpy as np
import torch
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from conftest import golden_args, golden_state_dict, load_golden
from oracle import network
from stratanet2_vegetation_coverage_maps_amd import PointNet2

name = sys.argv[1] if len(sys.argv) > 1 else "c1_ref_defaults"
g, args = load_golden(name), golden_args(name)
sd = golden_state_dict(g)
cloud, xyz = torch.from_numpy(g["in/cloud"]), torch.from_numpy(g["in/xyz"])
fs = torch.from_numpy(g["in/fps_start"])
with torch.no_grad():
    cov_r, proba_r, ex = network.forward(sd, cloud, xyz, args, training=True, fps_start=(fs[0], fs[1]), details=True)
args.cuda = 0
m = PointNet2(args)
m.load_state_dict(sd)
m.train()
cov, proba, s = m._forward_impl(xyz.cuda(), cloud.cuda(), fs.cuda().int(), True)
def cmp(nm, hip, ref):
    hip = hip.cpu()
    print(f"{nm:8s} max err {float((hip - ref).abs().max()):.3e}  sign flips {(int(((hip > 0) != (ref > 0)).sum()))} / {ref.numel()}  exact zeros hip {int((hip == 0).sum())} ref {int((ref == 0).sum())}")
cmp("x1", s.x1, ex["x1"])
cmp("x2", s.x2, ex["x2"])
cmp("x3", s.x3, ex["x3"])
cmp("f3", s.h3 * s.b_fp3.a + s.b_fp3.c, ex["f3"])
cmp("f2", s.h2[:, :34] * s.b_fp2.a + s.b_fp2.c, ex["f2"])
cmp("f1", s.h1[:, :34] * s.b_fp1.a + s.b_fp1.c, ex["f1"])
# pre-BN h of fp1 from the oracle: invert BN
idx1 = ex["idx1"]
print("idx1 equal", bool(torch.equal(s.idx1.cpu().long().view(-1) + (torch.arange(s.B).repeat_interleave(s.M1) * s.N), idx1)))
print("E1 hip", int(s.tot1.item()), "ref", ex["row1"].numel(), " E2 hip", int(s.tot2.item()), "ref", ex["row2"].numel())
print("max cnt1", int(s.cnt1.max()), "max cnt2", int(s.cnt2.max()))
