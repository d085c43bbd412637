"""The restatements of the rows next to the hot path against fixtures produced by the REFERENCE'S OWN numpy
code (`oracle/make_golden_aux.py`: geotiff_raster.py merge / weights / finalisation, loader.py `load_cloud`, load_data.py
z-normalisation on the real sklearn kd-tree).  Bit for bit."""
import types

import numpy as np

from conftest import load_golden
from oracle import mosaic as omosaic
from oracle import prepare
from stratanet2_vegetation_coverage_maps_amd import inference, input_pipeline


def test_weights_band_and_geotransform_vs_reference():
    g = load_golden("f_mosaic")
    for D in (20, 32, 5):
        want = g[f"weights/D{D}/out"]
        assert want.shape == (6, D, D)
        np.testing.assert_array_equal(omosaic.weights_band(D), want[3])
        got = inference.add_weights_band_to_rasters(g[f"weights/D{D}/in"], types.SimpleNamespace(diam_pix=D))
        np.testing.assert_array_equal(got, want)
    for key, (dm, dp) in (("m20_p20", (20, 20)), ("m20_p32", (20, 32)), ("m21_p20", (21, 20))):
        want = g[f"geotransform/{key}"]
        got = np.array([omosaic.get_geotransform(c, dm, dp) for c in g["geotransform/centers"]])
        np.testing.assert_array_equal(got, want)


def test_product_pixel_offsets_follow_the_reference_geotransform():
    """`ParcelMosaic.offsets` = the reference geotransform's top-left corner (pinned above) expressed in parcel pixels."""
    g = load_golden("f_mosaic")
    args = types.SimpleNamespace(diam_meters=20, diam_pix=32)
    pm = inference.ParcelMosaic.__new__(inference.ParcelMosaic)
    pm.args, pm.x_min, pm.y_max, pm.pix = args, -100.0, 200.0, 20 / 32
    centers = g["geotransform/centers"][1:]
    gts = g["geotransform/m20_p32"][1:]
    off = pm.offsets(centers).numpy()
    for (row, col), gt in zip(off, gts):
        assert col == round((gt[0] - pm.x_min) / gt[1]) and row == round((pm.y_max - gt[3]) / -gt[5])


def test_merge_restatement_vs_reference_callback():
    g = load_golden("f_mosaic")
    for name in ("merge_a", "merge_b"):
        H, W, D = (int(v) for v in g[f"{name}/HWD"])
        got = omosaic.mosaic(g[f"{name}/rasters"], g[f"{name}/offsets"], H, W, D, dtype=np.float32)
        want = g[f"{name}/canvas"]
        assert got.dtype == want.dtype == np.float32
        np.testing.assert_array_equal(got, want)           # NaN == NaN positionally
        # the fp64 yardstick used by the older tests stays within fp32 rounding of it
        np.testing.assert_allclose(omosaic.mosaic(g[f"{name}/rasters"], g[f"{name}/offsets"], H, W, D), want, rtol=2e-6,
                                   atol=1e-7, equal_nan=True)


def test_finalize_restatement_vs_reference():
    g = load_golden("f_mosaic")
    for name in ("finalize_a", "finalize_b"):
        got, thr = omosaic.finalize_merged_raster(g[f"{name}/in"].copy())
        np.testing.assert_array_equal(got, g[f"{name}/out"])
        assert int(round(thr * 10000)) == int(g[f"{name}/threshold_index"])


def test_load_cloud_restatement_vs_reference_loader():
    g = load_golden("f_load_cloud")
    args = types.SimpleNamespace(diam_meters=20, z_max=24.24, subsample_size=int(g["subsample_size"]))
    for tag, train in (("eval", False), ("train", True)):
        rs = np.random.RandomState(int(g["seed"]))
        for i in range(3):
            wc, wx = prepare.load_cloud(g[f"raw/{i}"], g["centers"][i], args, train, rs)
            np.testing.assert_array_equal(wc, g[f"{tag}/cloud/{i}"])
            np.testing.assert_array_equal(wx, g[f"{tag}/xyz/{i}"])
        assert rs.random() == float(g[f"{tag}/next_random"])        # same number of draws consumed
        assert float(g[f"{tag}/max_abs_diff_of_numpy121_restatement"]) == 0.0
    # the product's host-side draws are the same stream too (the kernel side is a -m gpu test)
    rs_a, rs_b = np.random.RandomState(3), np.random.RandomState(3)
    d = input_pipeline.draw_plot_randoms(5316, 3000, True, rs_a, True)
    prepare.load_cloud(g["raw/0"], g["centers"][0], args, True, rs_b)
    assert rs_a.random() == rs_b.random() and d["idx"].shape == (3000,)
    np.testing.assert_array_equal(input_pipeline.fake_ground_xy(20), g["eval/xyz/1"][:2, 1200:1516].T)


def test_znorm_restatement_vs_reference_on_sklearn():
    g = load_golden("f_znorm")
    for name in ("disc", "lambert", "lattice", "small_radius"):
        got = prepare.normalize_z_with_minz_in_a_radius(g[f"{name}/in"], float(g[f"{name}/radius"]))
        np.testing.assert_array_equal(got, g[f"{name}/out"])
