"""CPU restatement of `/root/reference/model/project_to_2d.py`.  TEST INFRASTRUCTURE ONLY.

P2 `project_to_plotwise_coverages` (:7-55)  and  P1 `project_to_2d_rasters` (:58-113), restated on a DENSE
per-plot pixel grid instead of the reference's `torch.unique` compaction (the set of occupied pixels and the
per-pixel maxima are identical; only the enumeration differs).  Pinned against goldens produced by the
reference functions themselves (oracle/make_golden.py, tests/test_oracle_golden.py).

Index arithmetic is kept operation-for-operation in fp32 (sub, add 1e-4, IEEE divide, multiply, floor):
raster pixel indices must be bit-exact (BASELINE.json north_star).
"""
import numpy as np
import torch

from . import primitives as P


def p2_pixel_ids(clouds: torch.Tensor, diam_pix: int) -> torch.Tensor:
    """Bounding-box-normalised grid of project_to_2d.py:15-22.  clouds (B,>=2,N) -> int32 (B,2,N) in
    [0, diam_pix-1]."""
    xy = clouds[:, :2, :]
    mn = xy.min(dim=2, keepdim=True).values
    mx = xy.max(dim=2, keepdim=True).values
    return torch.floor((xy - mn) / (mx - mn + 0.0001) * diam_pix).int()


def project_to_plotwise_coverages(pred_pointwise: torch.Tensor, clouds: torch.Tensor, args) -> torch.Tensor:
    """pred_pointwise (B*N,4) [low, soil, med, high] -> (B,4): per pixel max of channels 0,2,3, bare soil =
    1 - low-veg pixel max (:41-44), mean over the OCCUPIED pixels of each plot (:46-49)."""
    B, _, N = clouds.shape
    D = args.diam_pix
    pix = p2_pixel_ids(clouds, D).long()
    cell = pix[:, 0, :] * D + pix[:, 1, :] + (torch.arange(B) * D * D).unsqueeze(1)      # (B,N)
    cell = cell.reshape(-1)
    pixel_max, _ = P.scatter_max(pred_pointwise.transpose(1, 0), cell, dim=-1, dim_size=B * D * D)  # (4, B*D*D)
    occupied = torch.zeros(B * D * D).index_add(0, cell, torch.ones(B * N)) > 0
    occ = occupied.view(B, D * D).to(pred_pointwise.dtype)
    n_occ = occ.sum(1)
    pm = pixel_max.view(4, B, D * D)
    low = (pm[0] * occ).sum(1) / n_occ
    soil = ((1 - pm[0]) * occ).sum(1) / n_occ
    med = (pm[2] * occ).sum(1) / n_occ
    high = (pm[3] * occ).sum(1) / n_occ
    return torch.stack([low, soil, med, high]).T


def p1_pixel_ids(cloud: torch.Tensor, diam_pix: int, diam_meters: int) -> torch.Tensor:
    """Fixed grid of project_to_2d.py:68-78. cloud (>=2,N) -> int32 (2,N) clipped to [0, diam_pix-1]."""
    scaling_factor = 10 * (diam_pix / diam_meters)
    xy = cloud[:2, :]
    off = torch.tensor([[float(diam_meters // 2)], [float(diam_meters // 2)]], dtype=torch.float32)
    pix = torch.floor((xy + 0.0001) * scaling_factor + off).int()
    return torch.clip(pix, 0, diam_pix - 1)


def project_to_2d_rasters(cloud: torch.Tensor, coverages_pointwise: torch.Tensor, args) -> np.ndarray:
    """cloud (>=2,N), coverages_pointwise (4,N) -> float64 (3,diam_pix,diam_pix) [low, med, high]; image[y,x];
    NaN where no point falls; rows flipped (:108-110)."""
    D = args.diam_pix
    pix = p1_pixel_ids(cloud.detach().cpu(), D, args.diam_meters).long()
    cell = pix[1] * D + pix[0]                       # image[m = y_pix, k = x_pix]
    vals = coverages_pointwise.detach().cpu().float()
    mx, arg = P.scatter_max(vals, cell, dim=-1, dim_size=D * D)
    img = mx[[0, 2, 3]].double()
    img[:, arg[0] == vals.shape[1]] = float("nan")
    img = img.view(3, D, D).numpy()
    return np.flip(img, axis=1).copy()
