"""Isolate the FP1 block backward: HIP vs a torch emulation built from the HIP forward's own saved tensors."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from conftest import golden_args, golden_state_dict, load_golden
from stratanet2_vegetation_coverage_maps_amd import PointNet2

name = sys.argv[1] if len(sys.argv) > 1 else "c1_ref_defaults"
g, args = load_golden(name), golden_args(name)
args.cuda = 0
m = PointNet2(args)
m.load_state_dict(golden_state_dict(g))
m.train()
cloud, xyz = torch.from_numpy(g["in/cloud"]).cuda(), torch.from_numpy(g["in/xyz"]).cuda()
fs = torch.from_numpy(g["in/fps_start"]).cuda().int()
cov, proba, s = m._forward_impl(xyz, cloud, fs, True)
R = cov.shape[0]
gen = torch.Generator().manual_seed(9)
dcov, dproba = (torch.randn(R, 4, generator=gen) / R).cuda(), (torch.randn(R, 4, generator=gen) / R).cuda()
grads = m._backward_impl(s, dcov, dproba)
names = [k for k, _ in m.named_parameters()]
G = dict(zip(names, grads))
# ---- torch emulation of fp1 + head from saved tensors
B, N, M1 = s.B, s.N, s.M1
a2, c2 = s.b_fp2.a, s.b_fp2.c
f2 = s.h2[:, :34] * a2 + c2                      # fp2 output rows (B*M1,34)
idx, w = s.knn1
base = (torch.arange(B * N, device="cuda") // N * M1).unsqueeze(1)
gi = idx.long() + base
interp = (f2[gi] * w.unsqueeze(-1)).sum(1) / w.sum(1, keepdim=True)
u = torch.cat([interp, s.rows0[:, :8]], 1)
W = m.fp1_module.nn[0][0].weight.detach().clone().requires_grad_(True)
b = m.fp1_module.nn[0][0].bias.detach().clone().requires_grad_(True)
gam = m.fp1_module.nn[0][2].weight.detach().clone().requires_grad_(True)
bet = m.fp1_module.nn[0][2].bias.detach().clone().requires_grad_(True)
h = torch.relu(u @ W.t() + b)
print("h1 fwd err", float((h - s.h1[:, :34]).abs().max()))
y = torch.nn.functional.batch_norm(h, None, None, gam, bet, True, 0.1, 1e-5)
z = torch.relu(y @ m.lin1.weight.t() + m.lin1.bias)
sc = z @ m.lin2.weight.t() + m.lin2.bias
p = torch.softmax(sc[:, :4], 1)
d = torch.sigmoid(sc[:, 4:5])
((p * d * dcov).sum() + (p * dproba).sum()).backward()
for k, ref in (("fp1_module.nn.0.0.weight", W.grad), ("fp1_module.nn.0.0.bias", b.grad),
               ("fp1_module.nn.0.2.weight", gam.grad), ("fp1_module.nn.0.2.bias", bet.grad)):
    got = G[k]
    err = (got - ref).abs()
    print(k, "max|ref|", float(ref.abs().max()), "max err", float(err.max()), "rel", float(err.max() / ref.abs().max()))
e = (G["fp1_module.nn.0.0.weight"] - W.grad).abs()
print("err by out-channel:", [f"{float(x):.1e}" for x in e.max(1).values])
print("err by in-channel :", [f"{float(x):.1e}" for x in e.max(0).values])
print("ref by in-channel :", [f"{float(x):.1e}" for x in W.grad.abs().max(0).values])
eb = (G["fp1_module.nn.0.0.bias"] - b.grad)
print("bias err:", [f"{float(x):.1e}" for x in eb])
print("bias ref:", [f"{float(x):.1e}" for x in b.grad])
