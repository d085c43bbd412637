"""Full-network parity at the METRIC's own size (BASELINE.json configs[1]: 16 plots x 32 768 points, SA 1024/256, r 1/2 m):
forward, plot-wise projection, loss and every parameter gradient against the oracle (kd-tree candidate search + the
canonical fp32 tests), for the reference architecture and for the 3sa variant.  References to match:
/root/reference/model/point_net2.py:106-153, learning/train.py:52-64.

Gradient yardstick: the oracle run again with fp64 features and weights (same fp32 geometry and pixel ids).  The oracle's
own fp32-vs-fp64 discrepancy per tensor is printed next to the HIP error; the bound is max(1e-3, that discrepancy) of the
tensor's magnitude (why a flat 1e-3 cannot hold at default-initialised weights: tests/test_gpu_network.py, scripts/cond_probe.py)."""
import numpy as np
import pytest
import torch

from oracle import losses as olosses, network, projection
from stratanet2_vegetation_coverage_maps_amd import PointNet2, losses, project_to_plotwise_coverages
from stratanet2_vegetation_coverage_maps_amd.point_net2_3sa import PointNet2ThreeSA
from stratanet2_vegetation_coverage_maps_amd.synthetic import make_args, make_batch

pytestmark = pytest.mark.gpu
TOL = 1e-4
B, N = 16, 32768


def _oracle(arch, sd, d, args, fs, dtype):
    s = {k: (v.to(dtype) if v.is_floating_point() else v).clone() for k, v in sd.items()}
    for k in network.param_keys(s):
        s[k].requires_grad_(True)
    cloud = d["cloud"].to(dtype)
    if arch == "3sa":
        cov, proba, _ = network.forward_3sa(s, cloud, d["xyz"], args, training=True, fps_start=fs, use_kdtree=True)
    else:
        cov, proba, _ = network.forward(s, cloud, d["xyz"], args, training=True, fps_start=(fs[0], fs[1]), use_kdtree=True)
    pred = projection.project_to_plotwise_coverages(cov, d["cloud"], args)          # pixel ids from the fp32 cloud
    loss, _ = olosses.total_loss(pred, proba, d["coverages"], d["pdf_all"], args.m, args.e)
    loss.backward()
    return s, cov.detach(), proba.detach(), pred.detach(), float(loss)


@pytest.mark.parametrize("arch", ["ref", "3sa"])
def test_metric_size_forward_loss_backward_vs_oracle(arch):
    torch.set_num_threads(16)
    args = make_args(cuda=0, subsample_size=N, ratio1=1024 / N, r1=1.0, ratio2=0.25, r2=2.0, ratio3=0.25, r3=4.0)
    d = make_batch(B, N)                                       # the plots bench.py's first batch holds
    nf = 3 if arch == "3sa" else 2
    fs = torch.zeros(nf, B, dtype=torch.long)
    d["fps_start"] = fs
    sd = network.init_state_dict_3sa(0) if arch == "3sa" else network.init_state_dict(0)
    m = (PointNet2ThreeSA if arch == "3sa" else PointNet2)(args)
    m.load_state_dict(sd)
    m.train()
    cov, proba = m(d)
    pred = project_to_plotwise_coverages(cov, d["cloud"], args, model=m)
    loss, _ = losses.total_loss(pred, proba, d["coverages"].cuda(), d["pdf_all"].cuda(), args.m, args.e)
    loss.backward()
    torch.cuda.synchronize()

    s32, cov_r, proba_r, pred_r, loss_r = _oracle(arch, sd, d, args, fs, torch.float32)
    np.testing.assert_allclose(cov.detach().cpu().numpy(), cov_r.numpy(), atol=TOL, rtol=0)
    np.testing.assert_allclose(proba.detach().cpu().numpy(), proba_r.numpy(), atol=TOL, rtol=0)
    np.testing.assert_allclose(pred.detach().cpu().numpy(), pred_r.numpy(), atol=TOL, rtol=0)
    assert abs(loss.item() - loss_r) < TOL
    s64, *_ = _oracle(arch, sd, d, args, fs, torch.float64)
    report, worst = [], 0.0
    for k, p in m.named_parameters():
        g64 = s64[k].grad.numpy()
        scale = np.abs(g64).max()
        self_err = np.abs(s32[k].grad.numpy() - g64).max() / scale
        err = np.abs(p.grad.cpu().numpy() - g64).max() / scale
        tol = max(1e-3, self_err)
        report.append(f"{k:42s} err {err:.2e}  tol {tol:.2e}  oracle fp32-vs-fp64 {self_err:.2e}{'  <-- FAIL' if err > tol else ''}")
        worst = max(worst, err / tol)
    print(f"\n[{arch}-arch, {B} x {N}] loss {loss.item():.6f} (oracle {loss_r:.6f}); parameter gradients vs the oracle's fp64 run:\n  "
          + "\n  ".join(report))
    assert worst <= 1.0, "\n".join(report)
