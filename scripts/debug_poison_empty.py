"""Every torch.empty() the package makes comes back filled with NaN (floats) or 1 (ints): a kernel that reads a word that
nobody wrote shows up as NaN in the step's outputs.  (Fresh allocations are zero pages, so such a read goes unnoticed in a
new process and bites only when the caching allocator hands out a used block.)"""
import sys
import torch
sys.path.insert(0, ".")
_empty = torch.empty
def poisoned_empty(*a, **k):
    t = _empty(*a, **k)
    if t.is_cuda and t.numel():
        if t.dtype in (torch.float32, torch.float64):
            t.fill_(float("nan"))
        elif t.dtype in (torch.int32, torch.int64):
            t.fill_(1)
    return t
torch.empty = poisoned_empty
from oracle import network                                                                             # noqa: E402
from stratanet2_vegetation_coverage_maps_amd import PointNet2, losses, project_to_plotwise_coverages   # noqa: E402
from stratanet2_vegetation_coverage_maps_amd import hip_ops as ops                                      # noqa: E402
from stratanet2_vegetation_coverage_maps_amd.synthetic import make_args, make_batch                    # noqa: E402

for N, B in ((4096, 2), (24001, 3)):
    args = make_args(cuda=0, subsample_size=N, ratio1=min(0.125, 1024 / N), r1=1.0, ratio2=0.25, r2=2.0)
    model = PointNet2(args)
    model.load_state_dict(network.init_state_dict(5))
    model = model.cuda().train()
    d = make_batch(B, N, first_plot=40)
    d = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in d.items()}
    d["fps_start"] = torch.zeros(2, B, dtype=torch.int32, device="cuda")
    for it in range(2):
        model.zero_grad()
        cov, proba = model(d)
        pred = project_to_plotwise_coverages(cov, d["cloud"], args)
        loss, _ = losses.total_loss(pred, proba, d["coverages"], d["pdf_all"], args.m, args.e)
        loss.backward()
        torch.cuda.synchronize()
        bad = [k for k, p in model.named_parameters() if not torch.isfinite(p.grad).all()]
        print(N, B, "iter", it, "loss", float(loss.detach()), "cov nan", bool(torch.isnan(cov).any()),
              "pred nan", bool(torch.isnan(pred).any()), "params with non-finite grad:", bad, flush=True)
    # eval path too
    model.eval()
    with torch.no_grad():
        cov, proba = model(d)
    print(N, B, "eval cov nan", bool(torch.isnan(cov).any()))
