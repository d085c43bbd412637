"""Device-side loss block of the timed training step -- mirror of the reference's
`learning/loss_functions.py:9-57` as used by `learning/train.py:58-62`:

    loss = get_absolute_loss(pred, gt) + m * get_NLL_loss(proba, pdf_all) + e * get_entropy_loss(proba)

Plain torch ops on the device (harness-level code; SURVEY.md 8f ranks a fused HIP loss kernel as the next row).
Difference from the reference signature: `get_NLL_loss` takes the KDE-mixture densities `pdf_all (B*N,3)` directly
instead of evaluating `args.kde_mixture` on the CPU each step (`loss_functions.py:30-42`; KDE fitting is out of scope).
"""
import torch

EPS = 0.0001


def get_absolute_loss(pred_pl, gt):
    # strata [low, med, high] = columns 0, 2, 3; sliced (not list-indexed: a python index list costs a host-to-device
    # copy per step and cannot be captured into a hipGraph)
    d = torch.cat((pred_pl[:, 0:1], pred_pl[:, 2:4]), 1) - torch.cat((gt[:, 0:1], gt[:, 2:4]), 1)
    return (d.pow(2) + EPS).pow(0.5).mean(0).mean()


def get_entropy_loss(pred_pixels):
    p = pred_pixels[:, 2:]
    return -(p * torch.log(p + EPS) + (1 - p) * torch.log(1 - p + EPS)).mean()


def get_NLL_loss(pred_pointwise, pdf_all):
    p_ground = pred_pointwise[:, 0] + pred_pointwise[:, 1]
    lik = p_ground * pdf_all[:, 0] + pred_pointwise[:, 2] * pdf_all[:, 1] + pred_pointwise[:, 3] * pdf_all[:, 2]
    return -torch.log(lik).mean()


def total_loss(pred_coverages, proba_pointwise, gt, pdf_all, m=0.10, e=0.2 / 5):
    l_abs = get_absolute_loss(pred_coverages, gt)
    l_log = get_NLL_loss(proba_pointwise, pdf_all)
    l_e = get_entropy_loss(proba_pointwise)
    return l_abs + m * l_log + e * l_e, (l_abs, l_log, l_e)
