"""Run-to-run agreement of the parameter gradients: ten backward passes of the same small batch with the same weights
(no optimiser step).  With a bit-reproducible forward pass (scripts/debug_forward_determinism.py) everything agrees to
the order of the float atomics of the weight-gradient flushes (~1e-6 relative on the smallest gradients).  Written when one
pass in four disagreed by 1-2 %: the SA batch statistics were added with LDS float atomics, moved by 1e-7 from run to run,
and a pre-activation next to zero changed sign with them (one ReLU mask flip).
    python scripts/debug_grad_images.py [number of gradient images, default hip_ops.GRAD_IMAGES]"""
import sys
import torch
sys.path.insert(0, ".")
from oracle import network
from stratanet2_vegetation_coverage_maps_amd import PointNet2, losses, project_to_plotwise_coverages
from stratanet2_vegetation_coverage_maps_amd.synthetic import make_args, make_batch

from stratanet2_vegetation_coverage_maps_amd import hip_ops as ops
N, B = 4096, 2
if len(sys.argv) > 1:
    ops.GRAD_IMAGES = int(sys.argv[1])
print("GRAD_IMAGES", ops.GRAD_IMAGES)
args = make_args(cuda=0, subsample_size=N, ratio1=0.125, r1=1.0, ratio2=0.25, r2=2.0)
model = PointNet2(args)
model.load_state_dict(network.init_state_dict(5))
model = model.cuda().train()
d = make_batch(B, N, first_plot=40)
d = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in d.items()}
d["fps_start"] = torch.zeros(2, B, dtype=torch.int32, device="cuda")
gs = []
ITER = 10
for it in range(ITER):
    model.zero_grad()
    cov, proba = model(d)
    pred = project_to_plotwise_coverages(cov, d["cloud"], args)
    loss, _ = losses.total_loss(pred, proba, d["coverages"], d["pdf_all"], args.m, args.e)
    loss.backward()
    gs.append({k: p.grad.detach().clone() for k, p in model.named_parameters()})
torch.cuda.synchronize()
worst = 0.0
for k in gs[0]:
    m = float(gs[0][k].abs().max())
    dev = max(float((gs[i][k] - gs[0][k]).abs().max()) for i in range(1, ITER))
    worst = max(worst, dev / max(m, 1e-12))
    if dev > 1e-5 * max(m, 1e-12):
        print(f"{k:40s} max|g| {m:.3e}  largest deviation from iteration 0: {dev:.3e}")
print("largest relative deviation over all parameters and iterations: %.3e" % worst)
