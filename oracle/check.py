"""One training step of the oracle in a chosen precision.  TEST INFRASTRUCTURE ONLY.

`train_step(..., dtype=torch.float64)` is the yardstick of the parity tests: the same network (same fp32 geometry: FPS,
ball query, 3-NN and pixel ids come from the fp32 positions either way) with features, weights and statistics in fp64.
Why not the fp32 run: at default-initialised weights rarely active ReLU channels get BatchNorm outputs of ~100 sigma, and
torch's fp32 CPU batch statistics are good to ~1e-5 relative, so the fp32 oracle itself sits up to 1e-3 (outputs) / 1e-1
(gradients, relative) away from the exact evaluation at the metric's size, while the HIP path (fp64 finalisation of the
statistics) stays within 5e-6 / 2e-5 of it (tests/test_gpu_metric_size.py prints both)."""
import torch

from . import losses, network, projection


def train_step(sd, d, args, fps_start=None, dtype=torch.float64, arch="ref", dropout_mask=None, use_kdtree=False,
               bf16_layers=(), act_bf16=False, training=True):
    """sd: state dict (fp32 tensors); d: make_batch dict (cloud, xyz, coverages, pdf_all); -> dict with the forward
    outputs (detached, `dtype`), the loss terms and `grads` {parameter key: gradient}.  training=False: the same step through
    an eval-mode forward (BatchNorm on its running statistics, under autograd: model/point_net2.py:45-53 after model.eval())."""
    s = {k: (v.to(dtype) if v.is_floating_point() else v).clone() for k, v in sd.items()}
    keys = network.param_keys(s)
    for k in keys:
        s[k].requires_grad_(True)
    cloud = d["cloud"].to(dtype)
    if arch == "3sa":
        cov, proba, ex = network.forward_3sa(s, cloud, d["xyz"], args, training=training, fps_start=fps_start, use_kdtree=use_kdtree,
                                             bf16_layers=bf16_layers, act_bf16=act_bf16)
    else:
        fs = None if fps_start is None else (fps_start[0], fps_start[1])
        cov, proba, ex = network.forward(s, cloud, d["xyz"], args, training=training, fps_start=fs, use_kdtree=use_kdtree,
                                         dropout_mask=dropout_mask, bf16_layers=bf16_layers, act_bf16=act_bf16)
    pred = projection.project_to_plotwise_coverages(cov, d["cloud"], args)          # pixel ids from the fp32 cloud
    loss, parts = losses.total_loss(pred, proba, d["coverages"], d["pdf_all"], args.m, args.e)
    loss.backward()
    return {"cov": cov.detach(), "proba": proba.detach(), "pred": pred.detach(), "loss": float(loss.detach()),
            "parts": [float(p.detach()) for p in parts], "grads": {k: s[k].grad for k in keys}, "new_stats": (ex or {}).get("new_stats")}


def compare(model, cov, proba, loss, ref, tol_out=1e-4, tol_grad=1e-3, pred=None):
    """HIP results (device tensors; gradients in model.named_parameters()) against a `train_step` result.  Returns the list
    of violations (empty = parity) and a printable report with every measured error."""
    import numpy as np
    lines, fails = [], []
    for name, got, want in (("coverages_pointwise", cov, ref["cov"]), ("proba_pointwise", proba, ref["proba"]),
                            ("pred_coverages", pred, ref["pred"])):
        if got is None:
            continue
        e = float(np.abs(got.detach().cpu().double().numpy() - want.double().numpy()).max())
        lines.append(f"{name:42s} max abs err {e:.2e}  (tol {tol_out:.0e})")
        if not e <= tol_out:
            fails.append(lines[-1])
    e = abs(float(loss) - ref["loss"])
    lines.append(f"{'loss':42s} abs err {e:.2e}  (tol {tol_out:.0e})")
    if not e <= tol_out:
        fails.append(lines[-1])
    for k, p in model.named_parameters():
        g = ref["grads"][k].double().numpy()
        scale = np.abs(g).max()
        err = float(np.abs(p.grad.detach().cpu().double().numpy() - g).max() / scale) if scale > 0 else 0.0
        lines.append(f"{k:42s} grad err {err:.2e} of max |grad| = {scale:.2e}  (tol {tol_grad:.0e})")
        if not err <= tol_grad:
            fails.append(lines[-1])
    return fails, "\n  ".join(lines)
