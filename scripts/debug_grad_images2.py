import sys
import torch
sys.path.insert(0, ".")
from oracle import network
from stratanet2_vegetation_coverage_maps_amd import PointNet2, losses, project_to_plotwise_coverages
from stratanet2_vegetation_coverage_maps_amd import hip_ops as ops
from stratanet2_vegetation_coverage_maps_amd.synthetic import make_args, make_batch

N, B = 4096, 2
args = make_args(cuda=0, subsample_size=N, ratio1=0.125, r1=1.0, ratio2=0.25, r2=2.0)
model = PointNet2(args)
model.load_state_dict(network.init_state_dict(5))
model = model.cuda().train()
d = make_batch(B, N, first_plot=40)
d = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in d.items()}
d["fps_start"] = torch.zeros(2, B, dtype=torch.int32, device="cuda")
cap = {}
orig_fp1 = model._fp1_desc
def fp1_desc(s, **kw):
    if "dy" in kw:
        cap["dy1"] = kw["dy"]
        cap["bn"] = (s.b_fp1.grads[2], s.b_fp1.grads[3])
    return orig_fp1(s, **kw)
model._fp1_desc = fp1_desc
orig_bwd = ops.fp_backward
state = {"n": 0}
def fp_backward(dd):
    if dd.ca == 34 and "dy1" in cap and state["n"] == 0:
        state["n"] = 1
        cap["dy1_copy"] = cap["dy1"].clone()
        cap["bn_copy"] = (cap["bn"][0].clone(), cap["bn"][1].clone())
    return orig_bwd(dd)
ops.fp_backward = fp_backward
rec = []
for it in range(12):
    state["n"] = 0
    model.zero_grad()
    cov, proba = model(d)
    pred = project_to_plotwise_coverages(cov, d["cloud"], args)
    loss, _ = losses.total_loss(pred, proba, d["coverages"], d["pdf_all"], args.m, args.e)
    loss.backward()
    rec.append((cap["dy1_copy"], cap["bn_copy"], model.fp1_module.nn[0][0].weight.grad.detach().clone(),
                model.lin1.weight.grad.detach().clone()))
torch.cuda.synchronize()
for it, (dy, bn, g, gl) in enumerate(rec):
    print(it, "fp1 dW dev %.3e" % float((g - rec[0][2]).abs().max()), " lin1 dW dev %.3e" % float((gl - rec[0][3]).abs().max()),
          " dy1 dev %.3e (max %.3e)" % (float((dy - rec[0][0]).abs().max()), float(dy.abs().max())),
          " bn dgamma dev %.3e dbeta dev %.3e" % (float((bn[0] - rec[0][1][0]).abs().max()), float((bn[1] - rec[0][1][1]).abs().max())))
base = rec[0][0]
for it, (dy, bn, g, gl) in enumerate(rec):
    diff = (dy - base).abs()
    if float(diff.max()) > 1e-6:
        idx = torch.nonzero(diff > 1e-9)
        rows = idx[:, 0]; cols = idx[:, 1]
        print("iteration", it, "differing elements", idx.shape[0], "rows", int(rows.min()), "..", int(rows.max()), "distinct rows", int(rows.unique().numel()),
              "cols", sorted(set(cols.tolist()))[:40])
        r0 = int(rows[0])
        print(" row", r0, "good", base[r0, :8].tolist(), "\n        bad ", dy[r0, :8].tolist())
        break
