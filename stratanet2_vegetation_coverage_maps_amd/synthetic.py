"""Seeded synthetic plots in the exact format the reference DataLoader hands to `PointNet2.forward`
(no real LAS data exists offline).

Format contract (reference `data_loader/loader.py:73-87`, `config.py:54-65`):
    cloud (B,10,N) fp32 rows = [x/10, y/10, z/z_max, red, green, blue, nir, intensity, return_num, num_returns]
    xyz   (B,3,N)  fp32      = centred, un-rescaled metres (copied before rescale, `loader.py:79`)
Geometry: 10 m-radius disc (`loader.py:127-132`); z mixture 55 % ground |N(0,0.05)|, 25 % U(0,1.5),
20 % U(1.5,20) (strata limits `learning/kde_mixture.py:54-58`, z_max `config.py:73`).
"""
import math
from types import SimpleNamespace

import torch

BASE_SEED = 20211007
Z_MAX = 24.24


def make_args(**kw):
    """The fields of the reference `config.py` Namespace that the hot path reads
    (`model/point_net2.py:73-85`, `model/project_to_2d.py:21,26,68-78`), with the reference defaults."""
    d = dict(cuda=None, subsample_size=10000, n_class=4, drop=0.0, n_input_feats=10, ratio1=0.25,
             r1=math.sqrt(2.0), ratio2=0.25, r2=math.sqrt(8.0), patience_in_epochs=30, log_embeddings=False,
             diam_pix=20, diam_meters=20, z_max=Z_MAX, m=0.10, e=0.2 / 5, epoch_to_start_early_stop=250,
             current_fold_id=-1, stats_path=".")
    d.update(kw)
    return SimpleNamespace(**d)


def make_plot(n_points: int, seed: int):
    g = torch.Generator().manual_seed(seed)
    u = torch.rand(n_points, 9, generator=g, dtype=torch.float32)
    nrm = torch.randn(n_points, generator=g, dtype=torch.float32)
    rad = 10.0 * torch.sqrt(u[:, 0])
    th = (2.0 * math.pi) * u[:, 1]
    x, y = rad * torch.cos(th), rad * torch.sin(th)
    sel = u[:, 2]
    z = torch.where(sel < 0.55, (0.05 * nrm).abs(),
                    torch.where(sel < 0.80, 1.5 * u[:, 3], 1.5 + 18.5 * u[:, 3]))
    rgbn_i = u[:, 4:9]
    g2 = torch.Generator().manual_seed(seed + 7919)
    ret = torch.randint(0, 7, (n_points, 2), generator=g2).float() / 6.0
    xyz = torch.stack([x, y, z], 0)
    cloud = torch.cat([torch.stack([x / 10.0, y / 10.0, z / Z_MAX], 0), rgbn_i.t(), ret.t()], 0)
    return cloud.contiguous(), xyz.contiguous()


def make_batch(batch_size: int, n_points: int, first_plot: int = 0, base_seed: int = BASE_SEED):
    """Returns the `cloud_data` dict of the reference (CPU tensors) plus the harness-side extras of the
    training step: `coverages` (B,4) float64 ground truth and `pdf_all` (B*N,3) float64 (stand-in for the
    KDE mixture evaluated at the points' z: `learning/loss_functions.py:27-42`)."""
    clouds, xyzs = [], []
    for p in range(first_plot, first_plot + batch_size):
        c, x = make_plot(n_points, base_seed + p)
        clouds.append(c)
        xyzs.append(x)
    g = torch.Generator().manual_seed(base_seed * 31 + first_plot)
    gt = torch.rand(batch_size, 4, generator=g, dtype=torch.float64)
    pdf = 0.05 + 0.95 * torch.rand(batch_size * n_points, 3, generator=g, dtype=torch.float64)
    return {"cloud": torch.stack(clouds), "xyz": torch.stack(xyzs), "coverages": gt, "pdf_all": pdf}
