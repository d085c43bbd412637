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
        cap["t"] = dict(dy1=kw["dy"], dy2=kw["dsrc"], dg=s.b_fp1.grads[2], db=s.b_fp1.grads[3], dW=s.b_fp1.grads[0], dbias=s.b_fp1.grads[1],
                        h1=s.h1, ok=kw["bn_sums_done"])
    return orig_fp1(s, **kw)
model._fp1_desc = fp1_desc
orig_bwd = ops.fp_backward
def fp_backward(dd):
    r = orig_bwd(dd)
    if dd.ca == 34 and "t" in cap:
        t = cap.pop("t")
        K, stride = dd.blk.grad_replicas, dd.blk.grad_replica_stride
        base = t["dW"].data_ptr()
        imgs = torch.stack([torch.as_strided(t["dW"], t["dW"].shape, t["dW"].stride(), t["dW"].storage_offset() + r_ * stride) for r_ in range(K)])
        cap["snap"] = dict(dy1=t["dy1"].clone(), dy2=t["dy2"].clone(), dg=t["dg"].clone(), db=t["db"].clone(), dWimgs=imgs.clone(),
                           h1=t["h1"].clone(), ok=t["ok"].clone())
    return r
ops.fp_backward = fp_backward
rec = []
for it in range(12):
    model.zero_grad()
    cov, proba = model(d)
    pred = project_to_plotwise_coverages(cov, d["cloud"], args)
    loss, _ = losses.total_loss(pred, proba, d["coverages"], d["pdf_all"], args.m, args.e)
    loss.backward()
    rec.append((cap["snap"], model.fp1_module.nn[0][0].weight.grad.detach().clone()))
torch.cuda.synchronize()
import statistics
# majority value of the final gradient = good
devs = [[float((a[1] - b[1]).abs().max()) for b in rec] for a in rec]
good = min(range(len(rec)), key=lambda i: sum(1 for v in devs[i] if v > 1e-6))
for it, (s, g) in enumerate(rec):
    bad = float((g - rec[good][1]).abs().max()) > 1e-6
    if bad:
        gs = rec[good][0]
        print("iteration", it, "is bad;  differences of its FP1-backward inputs/outputs from good iteration", good)
        for k in ("h1", "dy1", "dg", "db", "ok", "dy2"):
            print("   ", k, "%.3e" % float((s[k].double() - gs[k].double()).abs().max()), "(max %.3e)" % float(gs[k].double().abs().max()))
        hd = (s["h1"] - gs["h1"]).abs()
        idx = torch.nonzero(hd > 1e-7)
        rows = idx[:, 0]
        flips = int(((s["h1"] > 0) != (gs["h1"] > 0)).sum())
        print("    h1: %d differing elements in %d rows (rows %d..%d), blocks of 64 rows touched: %s, relu mask flips: %d" % (
            idx.shape[0], int(rows.unique().numel()), int(rows.min()), int(rows.max()), sorted(set((rows // 64).tolist()))[:20], flips))
        r0 = int(rows[0])
        print("    row", r0, "good", [round(v, 6) for v in gs["h1"][r0, :10].tolist()], "\n            bad ", [round(v, 6) for v in s["h1"][r0, :10].tolist()])
        di = (s["dWimgs"] - gs["dWimgs"]).abs().flatten(1).max(1).values
        print("    dW images: per-image max diff", ["%.1e" % float(v) for v in di], " sum-of-images diff %.3e" % float((s["dWimgs"].sum(0) - gs["dWimgs"].sum(0)).abs().max()))
        break
else:
    print("no bad iteration in this run")
