"""Golden fixtures for the rows NEXT to the hot path, produced by RUNNING THE REFERENCE'S OWN numpy code.

Runs only in the build container (needs /root/reference; the GPU box never sees it).  Imported from the reference,
unmodified, and CALLED:

* `inference/geotiff_raster.py`: `add_weights_band_to_rasters` (:103-118), `get_geotransform` (:46-61),
  `_weighted_average_of_rasters` (:294-347, the rasterio.merge callback), `insert_hard_med_veg_raster_band` (:119-144),
  `finalize_merged_raster` (:262-285);
* `data_loader/loader.py`: `load_cloud` (:73-87) and everything it calls (`center_cloud`, `add_fake_empty_ground_points`,
  `augment`, `rescale_cloud`, `sample_cloud_data`), under `numpy.random.seed(s)`;
* `utils/load_data.py`: `normalize_z_with_minz_in_a_radius` (:237-249) on top of the REAL
  `sklearn.neighbors.NearestNeighbors(algorithm="kd_tree")` (sklearn 1.7 is installed here).

What is NOT the reference's and why (each a module NAME only, so that `import` statements resolve; none of their code is
needed by the functions above): comet_ml, torch_geometric, torch_scatter, torchnet, seaborn, osgeo (gdal/osr), rasterio,
shapely, shapefile, laspy, numpy_indexed are absent and un-fetchable.  Three adapters, all stated in the fixture:
  1. rasterio.merge itself is absent, so the DRIVER around the callback is emulated: a float32 NaN canvas (the plot
     GeoTIFFs are `gdal.GDT_Float32`, geotiff_raster.py:79, NaN no-data :86,93; rasterio.merge 1.2.6 allocates `dest` in
     the first source's dtype filled with its no-data value) and, plot after plot, `callback(region_view, plot_bands,
     isnan(region_view), isnan(plot_bands))` on the plot's window.  The window placement (rounding of the geotransform to
     pixels) stays a restatement: PARITY UNPINNED for placement only.
  2. `insert_admissibility_raster` (rasterio sieve + shapely buffers) is replaced by the identity while
     `finalize_merged_raster` runs: the fixture holds the five bands before the admissibility band is inserted.
  3. `NearestNeighbors(500, algorithm=...)` (load_data.py:240) passes `n_neighbors` positionally, which sklearn >= 1.0
     rejects; the adapter forwards it as the keyword.  Nothing else of sklearn is touched.

Only data is written (inputs and expected outputs, .npz).  No reference source text is stored.

    python -m oracle.make_golden_aux          # rewrites tests/golden/f_*.npz
"""
import os
import sys
import types

import numpy as np

from oracle.make_golden import OUT, REF, _install_standins


class _Absent:
    def __init__(self, *a, **k):
        raise RuntimeError("absent third-party library: only its name is registered")


def _mod(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def import_reference():
    """-> (geotiff_raster, loader, load_data) modules of the reference."""
    assert os.path.isdir(REF), "the reference is only present in the build container"
    _install_standins()
    _mod("seaborn", set=lambda *a, **k: None)
    _mod("torchnet")
    osgeo = _mod("osgeo")
    osgeo.gdal, osgeo.osr = _mod("osgeo.gdal"), _mod("osgeo.osr")
    r = _mod("rasterio")
    r.features = _mod("rasterio.features")
    r.transform = _mod("rasterio.transform", Affine=_Absent, xy=_Absent, rowcol=_Absent)
    r.merge = _mod("rasterio.merge", merge=_Absent)
    sh = _mod("shapely")
    sh.geometry = _mod("shapely.geometry")
    sh.geometry.point = _mod("shapely.geometry.point", Point=_Absent)
    _mod("shapefile")
    lp = _mod("laspy")
    lp.file = _mod("laspy.file", File=_Absent)
    _mod("numpy_indexed")
    import matplotlib
    matplotlib.use("Agg")
    if REF not in sys.path:
        sys.path.insert(0, REF)
    argv, sys.argv = sys.argv, [sys.argv[0]]          # config.py parses argv at import
    try:
        import inference.geotiff_raster as GR          # noqa: E402  (reference code)
        import data_loader.loader as L                 # noqa: E402
        import utils.load_data as LD                   # noqa: E402
    finally:
        sys.argv = argv
    return GR, L, LD


FEATURE_NAMES = ["x", "y", "z_flat", "red", "green", "blue", "near_infrared", "intensity", "return_num", "num_returns"]


def raw_plot(n, seed, center):
    """A raw (10,n) float32 plot in LAS units (`utils/load_data.py:149-183`): metres, 16-bit colours, returns 1..7."""
    rng = np.random.default_rng(seed)
    rad, th = 10 * np.sqrt(rng.random(n)), 2 * np.pi * rng.random(n)
    return np.stack([center[0] + rad * np.cos(th), center[1] + rad * np.sin(th), 20 * rng.random(n) ** 3,
                     *(65535 * rng.random((4, n))), 32767 * rng.random(n), rng.integers(1, 8, n), rng.integers(1, 8, n)]
                    ).astype(np.float32)


# --------------------------------------------------------------------------------------------------------- mosaic
def mosaic_case(GR, seed, B, D, H, W, band_holes):
    rng = np.random.default_rng(seed)
    args = types.SimpleNamespace(diam_pix=D, diam_meters=D)
    rasters = rng.random((B, 3, D, D)).astype(np.float32)
    rasters[np.broadcast_to(rng.random((B, 1, D, D)) < 0.25, rasters.shape)] = np.nan      # empty pixels (all bands)
    if band_holes:
        rasters[rng.random(rasters.shape) < 0.05] = np.nan                                 # a hole in one band only
    offsets = np.stack([rng.integers(0, H - D + 1, B), rng.integers(0, W - D + 1, B)], 1).astype(np.int32)
    canvas = np.full((6, H, W), np.nan, dtype=np.float32)
    with np.errstate(invalid="ignore", divide="ignore"):
        for rb, (oy, ox) in zip(rasters, offsets):
            new = GR.add_weights_band_to_rasters(rb, args).astype(np.float32)               # what the plot GeoTIFF holds
            region = canvas[:, oy:oy + D, ox:ox + D]
            GR._weighted_average_of_rasters(region, new, np.isnan(region), np.isnan(new))   # updates `region` in place
    return rasters, offsets, canvas


def make_mosaic(GR):
    out = {}
    for D in (20, 32, 5):
        img = np.random.default_rng(D).random((3, D, D))
        out[f"weights/D{D}/in"] = img
        out[f"weights/D{D}/out"] = GR.add_weights_band_to_rasters(img, types.SimpleNamespace(diam_pix=D))
    centers = np.array([[650123.5, 6861234.0], [30.0, 40.0], [12.25, -7.5], [0.0, 0.0]])
    for dm, dp in ((20, 20), (20, 32), (21, 20)):
        a = types.SimpleNamespace(diam_meters=dm, diam_pix=dp)
        out[f"geotransform/m{dm}_p{dp}"] = np.array([GR.get_geotransform(c, a) for c in centers], dtype=np.float64)
    out["geotransform/centers"] = centers
    for name, kw in (("merge_a", dict(seed=1, B=40, D=20, H=70, W=64, band_holes=False)),
                     ("merge_b", dict(seed=2, B=25, D=8, H=20, W=22, band_holes=True))):
        r, off, canvas = mosaic_case(GR, **kw)
        out[f"{name}/rasters"], out[f"{name}/offsets"], out[f"{name}/canvas"] = r, off, canvas
        out[f"{name}/HWD"] = np.array([kw["H"], kw["W"], kw["D"]])

    # finalize: the reference's own function with the GIS step (adapter 2) switched off; the chosen threshold index is
    # read off the `np.argmin(delta)` call the reference makes (geotiff_raster.py:137)
    GR.insert_admissibility_raster = lambda m: m
    for name, (H, W, seed) in (("finalize_a", (60, 70, 0)), ("finalize_b", (33, 129, 1))):
        rng = np.random.default_rng(seed)
        m = rng.random((6, H, W)).astype(np.float32)
        m[1] = (m[1] ** 3).astype(np.float32)
        hole = rng.random((H, W)) < 0.3
        m[:3, hole] = np.nan
        m[3, rng.random((H, W)) < 0.1] = np.nan
        m[3, hole] = np.nan
        m[1, 5, 7] = 0.25                                           # a value exactly on a threshold
        seen = {}
        orig = np.argmin

        def spy(a, *k, **kw):
            r = orig(a, *k, **kw)
            seen["idx"] = int(r)
            return r
        np.argmin = spy
        try:
            with np.errstate(invalid="ignore"):
                fin = GR.finalize_merged_raster(m.copy())
        finally:
            np.argmin = orig
        out[f"{name}/in"], out[f"{name}/out"], out[f"{name}/threshold_index"] = m, fin, np.array(seen["idx"])
    return out


# --------------------------------------------------------------------------------------------------------- load_cloud
def make_load_cloud(L):
    from oracle import prepare
    args = types.SimpleNamespace(diam_meters=20, n_input_feats=10, input_feats=FEATURE_NAMES, z_max=24.24, subsample_size=3000)
    centers = np.array([[650123.5, 6861234.0], [12.25, -7.5], [0.0, 0.0]], dtype=np.float32)
    sizes = (5000, 1200, 2684)                                       # 2684 + 316 fake ground points = 3000 exactly
    raws = [raw_plot(n, i, centers[i]) for i, n in enumerate(sizes)]
    out = {"centers": centers, "seed": np.array(7), "subsample_size": np.array(3000), "numpy_version": np.array(np.__version__)}
    for i, r in enumerate(raws):
        out[f"raw/{i}"] = r
    for train in (False, True):
        tag = "train" if train else "eval"
        dataset = {f"p{i}": {"cloud": r.copy(), "plot_center": centers[i].copy(), "plot_id": f"p{i}", "index": i}
                   for i, r in enumerate(raws)}
        np.random.seed(7)
        rs = np.random.RandomState(7)
        worst = 0.0
        for i in range(len(raws)):
            cd = L.load_cloud(f"p{i}", dataset, args, train=train)
            out[f"{tag}/cloud/{i}"], out[f"{tag}/xyz/{i}"] = cd["cloud"], cd["xyz"]
            assert cd["cloud"].dtype == np.float32 and cd["xyz"].dtype == np.float32
            # the restatement written for numpy 1.21's casting, same random stream: how far is it from the reference
            # running under THIS numpy?  (recorded as data)
            wc, wx = prepare.load_cloud(raws[i], centers[i], args, train, rs)
            worst = max(worst, float(np.abs(wc.astype(np.float64) - cd["cloud"]).max()),
                        float(np.abs(wx.astype(np.float64) - cd["xyz"]).max()))
        out[f"{tag}/next_random"] = np.array(np.random.random())     # both sides must have consumed the same draws
        out[f"{tag}/max_abs_diff_of_numpy121_restatement"] = np.array(worst)
    return out


# --------------------------------------------------------------------------------------------------------- z normalisation
def make_znorm(LD):
    from sklearn.neighbors import NearestNeighbors
    LD.NearestNeighbors = lambda n_neighbors, **kw: NearestNeighbors(n_neighbors=n_neighbors, **kw)      # adapter 3
    out = {}
    rng = np.random.default_rng(11)
    cases = {}
    n = 4000
    rad, th = 10 * np.sqrt(rng.random(n)), 2 * np.pi * rng.random(n)
    cases["disc"] = (np.stack([rad * np.cos(th), rad * np.sin(th), 20 * rng.random(n) ** 2]).astype(np.float32), 1.5)
    # Lambert-93-sized coordinates (float32 spacing 0.0625 m / 0.5 m): many points at exactly the same x,y and pairs at
    # exactly the radius
    cases["lambert"] = (np.stack([650000 + rad * np.cos(th), 6861000 + rad * np.sin(th), 20 * rng.random(n)]).astype(np.float32), 1.5)
    # a 0.75 m lattice: neighbours at exactly r = 1.5 (inclusive test) and r*sqrt(2)/... just outside
    g = np.arange(-6, 6.01, 0.75)
    gx, gy = np.meshgrid(g, g)
    cases["lattice"] = (np.stack([gx.ravel(), gy.ravel(), rng.random(gx.size) * 5]).astype(np.float32), 1.5)
    cases["small_radius"] = (cases["disc"][0][:, :1500].copy(), 0.4)
    for name, (xyz, radius) in cases.items():
        cloud = np.concatenate([xyz, rng.random((2, xyz.shape[1])).astype(np.float32)], 0)
        res = LD.normalize_z_with_minz_in_a_radius(cloud.copy(), radius)
        out[f"{name}/in"], out[f"{name}/radius"], out[f"{name}/out"] = cloud, np.array(radius), res
        assert res.dtype == np.float32
    return out


def main():
    GR, L, LD = import_reference()
    os.makedirs(OUT, exist_ok=True)
    for name, data in (("f_mosaic", make_mosaic(GR)), ("f_load_cloud", make_load_cloud(L)), ("f_znorm", make_znorm(LD))):
        path = os.path.join(OUT, f"{name}.npz")
        np.savez_compressed(path, **data)
        print(f"{name}: {len(data)} arrays -> {path} ({os.path.getsize(path) / 1024:.0f} KiB)")
        for k, v in data.items():
            if "max_abs_diff" in k or "threshold_index" in k:
                print(f"   {k} = {v}")


if __name__ == "__main__":
    main()
