"""Device-side loss block of the timed training step -- mirror of the reference's
`learning/loss_functions.py:9-57` as used by `learning/train.py:58-62`:

    loss = get_absolute_loss(pred, gt) + m * get_NLL_loss(proba, pdf_all) + e * get_entropy_loss(proba)

`total_loss` is ONE autograd node over three HIP kernels (csrc/loss.hip: pointwise partial sums, finalize, backward)
instead of the ~75 elementwise launches torch needs for the same expression and its gradient -- at 2.5 ms per step
those launches were a fifth of the step.  `total_loss_torch` and the three `get_*_torch` functions keep the plain torch-op form
(same as the reference, usable on any device); the tests hold the fused kernels to them.  The three `get_*` functions the
reference's loop calls are that node with two terms switched off on a HIP device, the torch forms elsewhere.

Difference from the reference signature: `get_NLL_loss` takes the KDE-mixture densities `pdf_all (B*N,3)` directly
instead of evaluating `args.kde_mixture` on the CPU each step (`loss_functions.py:30-42`; KDE fitting is out of scope).
"""
import torch

from . import hip_ops as ops
from ._lib import StrataHipError

EPS = 0.0001


def get_absolute_loss_torch(pred_pl, gt):
    # strata [low, med, high] = columns 0, 2, 3; sliced (not list-indexed: a python index list costs a host-to-device
    # copy per step and cannot be captured into a hipGraph)
    d = torch.cat((pred_pl[:, 0:1], pred_pl[:, 2:4]), 1) - torch.cat((gt[:, 0:1], gt[:, 2:4]), 1)
    return (d.pow(2) + EPS).pow(0.5).mean(0).mean()


def get_entropy_loss_torch(pred_pixels):
    p = pred_pixels[:, 2:]
    return -(p * torch.log(p + EPS) + (1 - p) * torch.log(1 - p + EPS)).mean()


def get_NLL_loss_torch(pred_pointwise, pdf_all):
    p_ground = pred_pointwise[:, 0] + pred_pointwise[:, 1]
    lik = p_ground * pdf_all[:, 0] + pred_pointwise[:, 2] * pdf_all[:, 1] + pred_pointwise[:, 3] * pdf_all[:, 2]
    return -torch.log(lik).mean()


def total_loss_torch(pred_coverages, proba_pointwise, gt, pdf_all, m=0.10, e=0.2 / 5):
    l_abs = get_absolute_loss_torch(pred_coverages, gt)
    l_log = get_NLL_loss_torch(proba_pointwise, pdf_all)
    l_e = get_entropy_loss_torch(proba_pointwise)
    return l_abs + m * l_log + e * l_e, (l_abs, l_log, l_e)


# The reference's loop calls the three terms one by one (learning/train.py:58-62).  On a HIP device each of them is ONE autograd
# node over the fused loss kernels with the other two terms SKIPPED (csrc/loss.hip: a switched-off term is not computed and its
# inputs need not exist) -- one or two launches forward and one backward per term instead of ~20 elementwise torch launches
# and their autograd graph (the eager drop-in loop is bound by its host time: DESIGN.md section 5) --, the values in fp64 as
# `total_loss`.  FUSED_TERMS = False (SN2_FUSED_LOSS_TERMS=0), CPU tensors: the plain torch forms above.
import os as _os
FUSED_TERMS = _os.environ.get("SN2_FUSED_LOSS_TERMS", "1") == "1"


def _fused_ok(*tensors):
    return FUSED_TERMS and all(isinstance(t, torch.Tensor) and t.is_cuda for t in tensors)


class _LossTerm(torch.autograd.Function):
    """ONE term of the training loss: kind 1 = absolute (x = plot-wise predictions (B,4), y = ground truth (B,4) fp64),
    2 = NLL (x = pointwise probabilities (R,4), y = densities (R,3) fp64), 3 = entropy (x = (R,4), y = None)."""

    @staticmethod
    def forward(ctx, kind, x, y):
        out = ops.loss_term_forward(kind, x, y)
        ctx.kind = kind
        ctx.save_for_backward(x, y) if y is not None else ctx.save_for_backward(x)
        return out[kind]

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None, None
        saved = ctx.saved_tensors
        x, y = saved[0], (saved[1] if len(saved) > 1 else None)
        return None, ops.loss_term_backward(ctx.kind, x, y, g.to(torch.float64).contiguous()), None


def get_absolute_loss(pred_pl, gt):
    if not _fused_ok(pred_pl, gt):
        return get_absolute_loss_torch(pred_pl, gt)
    with torch.cuda.device(pred_pl.device):
        return _LossTerm.apply(1, pred_pl.float().contiguous(), gt.to(torch.float64).contiguous())


def get_NLL_loss(pred_pointwise, pdf_all):
    if FUSED_TERMS and isinstance(pred_pointwise, torch.Tensor) and pred_pointwise.is_cuda and not pdf_all.is_cuda:
        # the reference evaluates its KDE mixture on the CPU and moves the densities to the device inside this function
        # (loss_functions.py:30-42): through the pinned ring, asynchronously (12.6 MB at C2: a pageable `.cuda()` blocks 0.4 ms),
        # on the upload stream -- the copy runs beside whatever the current stream still has queued (the forward pass: in the
        # reference's loop the device is 0.8 ms behind the host here), not behind it
        dev = pred_pointwise.device
        with torch.cuda.device(dev):
            pdf_all = ops.pinned_ring(dev).upload(pdf_all, stream=ops.shared_stream(dev, "upload"), dtype=torch.float64,
                                                  consumer=torch.cuda.current_stream(dev))
    if not _fused_ok(pred_pointwise, pdf_all):
        return get_NLL_loss_torch(pred_pointwise, pdf_all)
    with torch.cuda.device(pred_pointwise.device):
        return _LossTerm.apply(2, pred_pointwise.float().contiguous(), pdf_all.to(torch.float64).contiguous())


def get_entropy_loss(pred_pixels):
    if not _fused_ok(pred_pixels):
        return get_entropy_loss_torch(pred_pixels)
    with torch.cuda.device(pred_pixels.device):
        return _LossTerm.apply(3, pred_pixels.float().contiguous(), None)


class KdeTables:
    """The three linear-interpolation tables of the reference's `KdeMixture` (`learning/kde_mixture.py:62-70`: X and
    y1, y2, y3 from `evaluate_kdes`, what `interp1d` holds) on the device.  Fitting (KDEpy FFTKDE) stays with the reference:
    `KdeTables.from_mixture(args.kde_mixture, device)` copies the fitted tables."""

    def __init__(self, X, y1, y2, y3, device):
        import numpy as np
        X = np.asarray(X, dtype=np.float64)
        order = np.argsort(X, kind="stable")                      # interp1d(assume_sorted=False) sorts its knots
        Y = np.stack([np.asarray(y, dtype=np.float64)[order] for y in (y1, y2, y3)])
        self.X = torch.from_numpy(np.ascontiguousarray(X[order])).to(device)
        self.Y = torch.from_numpy(np.ascontiguousarray(Y)).to(device)

    @classmethod
    def from_mixture(cls, kde_mixture, device):
        return cls(kde_mixture.f1.x, kde_mixture.f1.y, kde_mixture.f2.y, kde_mixture.f3.y, device)


def kde_densities(clouds_dev, z_max, tables: KdeTables):
    """pdf_all (B*N,3) fp64 of `get_NLL_loss` (`learning/loss_functions.py:30-42`): the three KDE densities at every
    point's height z = cloud[2] * z_max, looked up on the device instead of through scipy on the CPU each step."""
    if not clouds_dev.is_cuda:
        raise StrataHipError("losses.kde_densities runs on the HIP device")
    with torch.cuda.device(clouds_dev.device):
        return ops.kde_lookup(clouds_dev.float().contiguous(), z_max, tables.X, tables.Y)


class _TotalLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, proba, gt, pdf, m, e):
        out = ops.loss_forward(pred, gt, proba, pdf, m, e)
        ctx.save_for_backward(pred, proba, gt, pdf)
        ctx.me = (m, e)
        ctx.set_materialize_grads(False)     # no zero tensors (= three fill launches per step) for the unused component outputs
        total, l_abs, l_log, l_e = out[0], out[1], out[2], out[3]
        ctx.mark_non_differentiable(l_abs, l_log, l_e)
        return total, l_abs, l_log, l_e

    @staticmethod
    def backward(ctx, g, *_):
        if g is None:
            return None, None, None, None, None, None
        pred, proba, gt, pdf = ctx.saved_tensors
        m, e = ctx.me
        dpred, dproba = ops.loss_backward(pred, gt, proba, pdf, m, e, g.to(torch.float64).contiguous())
        return dpred, dproba, None, None, None, None


def total_loss(pred_coverages, proba_pointwise, gt, pdf_all, m=0.10, e=0.2 / 5):
    """-> (total, (absolute, NLL, entropy)), fp64 scalars on the device; differentiable w.r.t. the first two arguments."""
    if not (pred_coverages.is_cuda and proba_pointwise.is_cuda):
        raise StrataHipError("losses.total_loss runs on the HIP device (total_loss_torch is the plain torch form)")
    with torch.cuda.device(pred_coverages.device):
        gt = gt.to(device=pred_coverages.device, dtype=torch.float64).contiguous()
        pdf_all = pdf_all.to(device=pred_coverages.device, dtype=torch.float64).contiguous()
        total, l_abs, l_log, l_e = _TotalLoss.apply(pred_coverages.float().contiguous(), proba_pointwise.float().contiguous(),
                                                    gt, pdf_all, float(m), float(e))
    return total, (l_abs, l_log, l_e)


class _ProjectedLoss(torch.autograd.Function):
    """`project_to_plotwise_coverages` + `total_loss` as ONE autograd node over three launches (csrc/project.hip:
    sn2_projected_loss_forward / _backward) instead of two nodes over seven."""

    @staticmethod
    def forward(ctx, cov, proba, pix, gt, pdf, B, N, D, m, e):
        out, pred, arg, nocc = ops.projected_loss_forward(cov, pix, proba, pdf, gt, B, N, D, m, e)
        ctx.save_for_backward(pred, proba, gt, pdf, arg, nocc, pix)
        ctx.dims = (B, N, D, m, e)
        ctx.set_materialize_grads(False)
        total, l_abs, l_log, l_e = out[0], out[1], out[2], out[3]
        ctx.mark_non_differentiable(l_abs, l_log, l_e, pred)
        return total, l_abs, l_log, l_e, pred

    @staticmethod
    def backward(ctx, g, *_):
        if g is None:
            return (None,) * 10
        pred, proba, gt, pdf, arg, nocc, pix = ctx.saved_tensors
        B, N, D, m, e = ctx.dims
        dcov, dproba = ops.projected_loss_backward(pred, gt, proba, pdf, B, N, D, m, e, g.to(torch.float64).contiguous(), arg, nocc, pix)
        return (dcov, dproba) + (None,) * 8


def projected_total_loss(coverages_pointwise, proba_pointwise, clouds, gt, pdf_all, args, geometry=None, model=None):
    """`pred = project_to_plotwise_coverages(coverages_pointwise, clouds, args)` followed by `total_loss(pred, proba_pointwise, gt,
    pdf_all, args.m, args.e)` (learning/train.py:54-62) -> (total, (absolute, NLL, entropy), pred).  With the pixel ids of a
    geometry pass at hand (`geometry.p2_pix`: `model.p2_diam_pix = args.diam_pix`) the two are ONE autograd node over three
    launches -- the scatter of the coverages beside the pointwise loss sums, the per-plot finalisation whose last workgroup adds
    the loss up, and one backward pass that writes both gradients; same pred, same gradients (bits), the loss to fp64
    re-association.  Without them: the two calls."""
    from .project_to_2d import project_to_plotwise_coverages
    pix = getattr(geometry, "p2_pix", None) if geometry is not None else None
    B = clouds.shape[0]
    N = clouds.shape[2]
    ok = (pix is not None and getattr(geometry, "p2_diam_pix", None) == int(args.diam_pix) and pix.numel() == B * N and
          coverages_pointwise.is_cuda and proba_pointwise.is_cuda)
    if not ok:
        pred = project_to_plotwise_coverages(coverages_pointwise, clouds, args, model=model, geometry=geometry)
        total, parts = total_loss(pred, proba_pointwise, gt, pdf_all, args.m, args.e)
        return total, parts, pred
    dev = coverages_pointwise.device
    with torch.cuda.device(dev):
        gt = gt.to(device=dev, dtype=torch.float64).contiguous()
        pdf_all = pdf_all.to(device=dev, dtype=torch.float64).contiguous()
        total, l_abs, l_log, l_e, pred = _ProjectedLoss.apply(coverages_pointwise.float().contiguous(), proba_pointwise.float().contiguous(),
                                                              pix, gt, pdf_all, B, N, int(args.diam_pix), float(args.m), float(args.e))
    return total, (l_abs, l_log, l_e), pred
