"""How well conditioned are the golden gradients?  fp32 oracle vs the same oracle with fp64 features/weights."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from conftest import golden_args, golden_state_dict, load_golden
from oracle import network

name = sys.argv[1] if len(sys.argv) > 1 else "c1_ref_defaults"
g, args = load_golden(name), golden_args(name)
cloud, xyz = torch.from_numpy(g["in/cloud"]), torch.from_numpy(g["in/xyz"])
fs = torch.from_numpy(g["in/fps_start"])
gen = torch.Generator().manual_seed(9)
R = cloud.shape[0] * cloud.shape[2]
w1, w2 = torch.randn(R, 4, generator=gen) / R, torch.randn(R, 4, generator=gen) / R
out = {}
for dt in (torch.float32, torch.float64):
    sd = {k: (v.to(dt) if v.is_floating_point() else v) for k, v in golden_state_dict(g).items()}
    keys = network.param_keys(sd)
    for k in keys:
        sd[k].requires_grad_(True)
    cov, proba, _ = network.forward(sd, cloud.to(dt), xyz, args, training=True, fps_start=(fs[0], fs[1]))
    ((cov * w1.to(dt)).sum() + (proba * w2.to(dt)).sum()).backward()
    out[dt] = {k: sd[k].grad.double().numpy() for k in keys}
    out[str(dt) + "cov"] = cov.detach().double()
print("cov diff f32 vs f64", float((out["torch.float32cov"] - out["torch.float64cov"]).abs().max()))
for k in out[torch.float32]:
    a, b = out[torch.float32][k], out[torch.float64][k]
    print(f"{k:45s} rel diff f32 vs f64: {np.abs(a - b).max() / np.abs(b).max():.2e}")
