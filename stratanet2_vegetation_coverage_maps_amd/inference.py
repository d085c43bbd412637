"""Parcel inference on the device -- counterpart of the reference loop `predict.py:96-141` +
`inference/predict_utils.py:94-116` + the raster merge of `inference/geotiff_raster.py`, without the GIS file I/O
(GDAL / rasterio / shapefile are out of scope: DESIGN.md section 7).

    per batch of plots:  eval forward -> (B*N,4) coverages -> fixed-grid max rasters (B,3,D,D)   [project_to_2d_rasters]
                         -> radial weight band                                                  [add_weights_band_to_rasters]
                         -> weighted accumulation into the parcel grid                          [rasterio.merge callback]
    at the end:          mosaic = sum(w*v) / sum(w), NaN where no plot has data

Pixel placement follows `get_geotransform` (geotiff_raster.py:46-61): a plot's top-left corner is
(center_x - diam_meters//2, center_y + diam_meters//2) and a pixel is diam_meters/diam_pix metres wide.
"""
import numpy as np
import torch

from . import hip_ops as ops
from .project_to_2d import project_batch_to_2d_rasters


def weights_band(diam_pix: int) -> np.ndarray:
    """`add_weights_band_to_rasters` (geotiff_raster.py:103-118): 1.5 - r on the normalised pixel-centre grid, NaN for
    r > 0.5 (outside the disc inscribed in the raster)."""
    x = (np.arange(-diam_pix // 2, diam_pix // 2, 1) + 0.5) / diam_pix       # loader.py:108-125
    xx, yy = np.meshgrid(x, x, sparse=True)
    r = np.sqrt(xx ** 2 + yy ** 2)
    w = 1.5 - r
    w[r > 0.5] = np.nan
    return w


def add_weights_band_to_rasters(img_to_write: np.ndarray, args) -> np.ndarray:
    """(C,D,D) -> (2C,D,D): one weight band per score band, as the reference writes into each plot GeoTIFF."""
    w = weights_band(args.diam_pix)
    return np.concatenate([img_to_write] + [w[None]] * len(img_to_write), 0)


class ParcelMosaic:
    """Running mosaic of plot rasters on a parcel grid (device resident), merged plot after plot with the rule of the
    reference's rasterio.merge callback (`_weighted_average_of_rasters`, geotiff_raster.py:294-347)."""

    def __init__(self, x_min: float, y_max: float, height_pix: int, width_pix: int, args, device):
        self.args = args
        self.x_min, self.y_max = float(x_min), float(y_max)
        self.pix = args.diam_meters / args.diam_pix
        nan = float("nan")
        self.mean = torch.full((3, height_pix, width_pix), nan, dtype=torch.float32, device=device)
        self.wsum = torch.full((3, height_pix, width_pix), nan, dtype=torch.float32, device=device)
        self.w = torch.from_numpy(weights_band(args.diam_pix).astype(np.float32)).to(device)

    def offsets(self, plot_centers) -> torch.Tensor:
        """(B,2) plot centres in metres -> (B,2) int32 (row, col) of the plots' top-left pixels (`get_geotransform`)."""
        c = torch.as_tensor(plot_centers, dtype=torch.float64).reshape(-1, 2)
        half = self.args.diam_meters // 2
        col = torch.round(((c[:, 0] - half) - self.x_min) / self.pix)
        row = torch.round((self.y_max - (c[:, 1] + half)) / self.pix)
        return torch.stack([row, col], 1).to(torch.int32)

    def add(self, rasters: torch.Tensor, plot_centers):
        off = self.offsets(plot_centers)
        D = self.args.diam_pix
        y0, x0 = int(off[:, 0].min()), int(off[:, 1].min())
        win = (y0, x0, int(off[:, 0].max()) + D - y0, int(off[:, 1].max()) + D - x0)
        ops.mosaic_merge(rasters.contiguous(), self.w, off.to(rasters.device), self.mean, self.wsum, win)

    def result(self) -> torch.Tensor:
        """(4,H,W): [low, med, high] merged scores + ONE weight band (`finalize_merged_raster` :270-275 keeps the
        first of the three identical weight layers)."""
        return torch.cat([self.mean, self.wsum[:1]], 0)

    def finalize(self):
        """`finalize_merged_raster` (geotiff_raster.py:262-285) without its last, GIS step (the admissibility band needs
        rasterio sieve / shapely buffers): (5,H,W) = [Vb, Vm_soft, Vh, Vm_hard, weights] and the hard-medium-vegetation
        threshold that `insert_hard_med_veg_raster_band` (:119-144) searches for."""
        out, thr = ops.mosaic_finalize(self.mean.contiguous(), self.wsum[0].contiguous())
        return out, thr


@torch.no_grad()
def predict_parcel(model, batches, mosaic: ParcelMosaic, args, prefetch: int = 3):
    """`batches`: iterable of dicts with "cloud" (B,10,N), "xyz" (B,3,N), "plot_center" (B,2) (the reference DataLoader's
    collate of `inference/predict_utils.py:74-82`).  Returns the number of plots processed.
    prefetch: how many batches ahead the position-only kernels (FPS, ball query, 3-NN) run, each on its own side stream
    (`PointNet2.prefetch_geometry`), while this batch's feature kernels, rasters and merge run.  FPS is M sequential rounds
    in one workgroup per plot -- 64 plots keep 64 of 256 CUs busy for most of an un-overlapped batch -- so several passes
    in flight is what fills the chip; 0 = no overlap."""
    from collections import deque
    model.eval()
    n = 0
    it = iter(batches)
    window = deque()                       # (batch, geometry handle or None), oldest first
    issued = 0

    def fill():
        nonlocal issued
        while len(window) < max(1, prefetch):
            b = next(it, None)
            if b is None:
                return
            geo = model.prefetch_geometry(b, lane=issued % prefetch) if prefetch > 0 else None
            window.append((b, geo))
            issued += 1

    fill()
    while window:
        cur, geo = window.popleft()
        fill()
        cd = dict(cur)
        if geo is not None:
            cd["geometry"] = geo
        cov, _ = model(cd)
        clouds_dev = model._last_cloud_dev[1]
        model._last_cloud_dev = None
        rasters, _ = project_batch_to_2d_rasters(clouds_dev, cov, args)
        mosaic.add(rasters, cur["plot_center"])
        n += clouds_dev.shape[0]
    # (no check of hip_ops.fps_gave_up here: it reads a device word, i.e. synchronises; callers that synchronise anyway --
    # reading the mosaic back -- may ask for it)
    return n
