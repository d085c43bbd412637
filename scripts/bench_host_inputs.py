"""The training step with inputs starting on the HOST, as the reference's DataLoader hands them over
(`model/point_net2.py:119-124`): 16 x 13 x 32768 fp32 = 27 MB H2D per step.  Not `bench.py`'s `value` (that one starts
with the inputs resident in HBM); DESIGN.md section 5 quotes this number."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stratanet2_vegetation_coverage_maps_amd import PointNet2, losses, project_to_plotwise_coverages  # noqa: E402
from stratanet2_vegetation_coverage_maps_amd.optim import FlatAdam, flatten_parameters  # noqa: E402
from stratanet2_vegetation_coverage_maps_amd.synthetic import make_args, make_batch  # noqa: E402

B, N = 16, 32768
args = make_args(cuda=0, subsample_size=N, ratio1=1024 / N, r1=1.0, ratio2=0.25, r2=2.0)
torch.manual_seed(0)
model = PointNet2(args).train()
flatten_parameters(model)
opt = FlatAdam(model, lr=1e-3, weight_decay=1e-3)
host = make_batch(B, N)
gt, pdf = host["coverages"].cuda(), host["pdf_all"].cuda()
out = {}
for label, pin in (("pageable", False), ("pinned", True)):
    cloud, xyz = (host["cloud"].pin_memory(), host["xyz"].pin_memory()) if pin else (host["cloud"], host["xyz"])
    fs = torch.zeros(2, B, dtype=torch.int32, device="cuda")

    def step():
        opt.zero_grad()
        cov, proba = model({"cloud": cloud, "xyz": xyz, "fps_start": fs})
        pred = project_to_plotwise_coverages(cov, cloud, args, model=model)
        loss, _ = losses.total_loss(pred, proba, gt, pdf, args.m, args.e)
        loss.backward()
        opt.step()

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / 20 * 1e3
    out[label] = {"ms_per_step": round(ms, 3), "plots_per_s": round(B / ms * 1e3, 1)}
print(json.dumps({"what": "unpipelined eager step, inputs on the host (27 MB H2D per step)", **out}))
