"""CPU restatement of the parcel mosaic merge.  TEST INFRASTRUCTURE ONLY.

Follows `/root/reference/inference/geotiff_raster.py`: `add_weights_band_to_rasters` (:103-118), `get_geotransform`
(:46-61), the pairwise `rasterio.merge` callback `_weighted_average_of_rasters` (:294-347) applied plot after plot on the
parcel window, `insert_hard_med_veg_raster_band` (:119-144) and `finalize_merged_raster` (:262-285).

PINNED: `tests/golden/f_mosaic.npz` holds the outputs of those reference functions themselves (run by
`oracle/make_golden_aux.py` in the build container); `tests/test_oracle_golden_aux.py` holds this file to them bit for bit
(merge in float32: the plot GeoTIFFs and hence rasterio.merge's canvas are Float32).  PARITY UNPINNED for one thing only:
the rounding of a geotransform to integer pixel offsets, which rasterio (absent) does."""
import numpy as np


def get_geotransform(plot_center_xy, diam_meters, diam_pix):
    """geotiff_raster.py:46-61: [x_min, pixel width, 0, y_max, 0, -pixel height] of a plot raster."""
    return [plot_center_xy[0] - diam_meters // 2, diam_meters / diam_pix, 0,
            plot_center_xy[1] + diam_meters // 2, 0, -diam_meters / diam_pix]


def weights_band(diam_pix):
    x = (np.arange(-diam_pix // 2, diam_pix // 2, 1) + 0.5) / diam_pix
    xx, yy = np.meshgrid(x, x, sparse=True)
    r = np.sqrt(xx ** 2 + yy ** 2)
    w = 1.5 - r
    w[r > 0.5] = np.nan
    return w


def merge_pair(old, new):
    """One call of the rasterio.merge callback on (2C,h,w) windows: returns the merged (2C,h,w)."""
    old, new = old.copy(), new.copy()
    old_nodata, new_nodata = np.isnan(old), np.isnan(new)
    C = len(old) // 2
    unw = np.zeros_like(old[:C])
    with np.errstate(invalid="ignore", divide="ignore"):
        for k in range(C):
            w = C + k
            old[k] = old[k] * old[w] * (1 - old_nodata[k])
            new[k] = new[k] * new[w] * (1 - new_nodata[k])
            w1 = old[w] * (1 - old_nodata[k])
            w2 = new[w] * (1 - new_nodata[k])
            unw[k] = np.nansum(np.concatenate([[w1], [w2]]), axis=0)
            unw[k][old_nodata[k] & new_nodata[k]] = np.nan
        old[old_nodata] = np.nan
        new[new_nodata] = np.nan
        out = np.nansum([old, new], axis=0)
        out[old_nodata & new_nodata] = np.nan
        out[:C] = out[:C] / unw
    return out


def mosaic(rasters, offsets, H, W, diam_pix, dtype=np.float64):
    """rasters (B,3,D,D) with NaN, offsets (B,2) (row, col) -> (6,H,W) after merging the plots one by one.
    dtype=np.float32 is what the reference does (Float32 GeoTIFFs, geotiff_raster.py:79 => a float32 canvas in
    rasterio.merge; numpy's promotions inside the callback follow from the same expressions); float64 is the yardstick."""
    w = weights_band(diam_pix)
    acc = np.full((6, H, W), np.nan, dtype=dtype)
    D = diam_pix
    for r, (oy, ox) in zip(rasters, offsets):
        img = np.concatenate([r] + [w[None]] * 3, 0).astype(dtype)
        win = acc[:, oy:oy + D, ox:ox + D]
        acc[:, oy:oy + D, ox:ox + D] = merge_pair(win, img)
    return acc


def insert_hard_med_veg_raster_band(mosaic):
    """`/root/reference/inference/geotiff_raster.py:119-144`, operation for operation (numpy.ma replaced by isnan)."""
    image_med_veg = mosaic[1]
    mask = np.isnan(image_med_veg)
    with np.errstate(invalid="ignore"):
        target_coverage = np.nanmean(image_med_veg)
        lin = np.linspace(0, 1, 10001)
        delta = np.ones_like(lin)
        for idx, threshold in enumerate(lin):
            image_med_veg_hard = 1.0 * (image_med_veg > threshold)
            image_med_veg_hard[mask] = np.nan
            delta[idx] = abs(target_coverage - np.nanmean(image_med_veg_hard))
    threshold = lin[np.argmin(delta)]
    image_med_veg_hard = 1.0 * (image_med_veg > threshold)
    image_med_veg_hard[mask] = np.nan
    return np.insert(mosaic, 3, image_med_veg_hard, axis=0), threshold


def finalize_merged_raster(mosaic):
    """`geotiff_raster.py:262-285` without `insert_admissibility_raster` (rasterio / shapely)."""
    mosaic = mosaic[: (3 + 1)]
    mosaic, threshold = insert_hard_med_veg_raster_band(mosaic)
    no_predicted_value = np.nansum(np.isnan(mosaic[:3]), axis=0) == 3
    mosaic = np.nan_to_num(mosaic, nan=0.0, posinf=None, neginf=None)
    mosaic[:, no_predicted_value] = np.nan
    return mosaic, threshold
