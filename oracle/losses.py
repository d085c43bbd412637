"""The loss of the timed training step, restated from `/root/reference/learning/loss_functions.py:9-57` and
`/root/reference/learning/train.py:58-62`.  TEST INFRASTRUCTURE ONLY.

    loss = loss_abs + m * loss_log + e * loss_entropy          (m = 0.10, e = 0.04: config.py:70-71)

The KDE mixture itself (`learning/kde_mixture.py`, KDEpy) is out of scope: its output `pdf_all (B*N,3)` float64 is
an input here."""
import torch

EPS = 0.0001  # loss_functions.py:6


def absolute_loss(pred_pl: torch.Tensor, gt: torch.Tensor) -> torch.Tensor:
    """:9-16 -- sqrt((pred-gt)^2 + EPS) over strata [low, med, high], mean over plots then strata."""
    return ((pred_pl[:, [0, 2, 3]] - gt[:, [0, 2, 3]]).pow(2) + EPS).pow(0.5).mean(0).mean()


def entropy_loss(proba_pointwise: torch.Tensor) -> torch.Tensor:
    """:19-24 -- binary entropy of the med/high membership probabilities."""
    p = proba_pointwise[:, 2:]
    return -(p * torch.log(p + EPS) + (1 - p) * torch.log(1 - p + EPS)).mean()


def nll_loss(proba_pointwise: torch.Tensor, pdf_all: torch.Tensor) -> torch.Tensor:
    """:27-57 -- p_ground = p_low + p_soil; likelihood = sum_k p_k * pdf_k (float64 because pdf_all is)."""
    p_all = torch.stack([proba_pointwise[:, :2].sum(1), proba_pointwise[:, 2], proba_pointwise[:, 3]], 1)
    return -torch.log((p_all * pdf_all).sum(1)).mean()


def total_loss(pred_coverages, proba_pointwise, gt, pdf_all, m: float = 0.10, e: float = 0.2 / 5):
    l_abs = absolute_loss(pred_coverages, gt)
    l_log = nll_loss(proba_pointwise, pdf_all)
    l_e = entropy_loss(proba_pointwise)
    return l_abs + m * l_log + e * l_e, (l_abs, l_log, l_e)


def kde_predict(X, ys, clouds, z_max):
    """`get_NLL_loss`'s density lookup (`/root/reference/learning/loss_functions.py:30-42`) with `KdeMixture.predict`
    (`learning/kde_mixture.py:65-70`): z = cloud[2] * z_max per plot (a float32 tensor times a python float), concatenated
    into a float64 array, then three scipy `interp1d(kind="linear", assume_sorted=False)` -- scipy itself is the
    checker here.  clouds (B,C,N) float32 tensor -> (B*N,3) float64."""
    import numpy as np
    from scipy.interpolate import interp1d
    z_all = np.empty((0))
    for current_cloud in clouds:
        z = current_cloud[2] * z_max
        z_all = np.append(z_all, z)
    z_all = np.asarray(z_all).reshape(-1)
    fs = [interp1d(X, y, kind="linear", assume_sorted=False) for y in ys]
    return np.concatenate([f(z_all).reshape(-1, 1) for f in fs], 1)
