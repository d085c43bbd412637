"""Parity at the LAUNCH SHAPES bench.py runs its secondary legs in (BASELINE.json configs[3] and configs[4]): 512 plots x 10 000
points per eval launch (parcel inference: the SA kernels take their "plots eight at a time" turns, `count_sum` adds 1 280 000
counts, FP3 runs 320 000 rows on the matrix-core kernel, the sorts run 256 threads per plot, the ball query takes its
bitmap-first path, FP1 and the head run as one kernel) and 8 x 131 072 points in bf16 mode.

The reference's eval forward is a function of ONE plot (`model/point_net2.py:106-153` with BatchNorm on running statistics;
`predict.py:96-126` feeds it whatever batch the DataLoader cut), so a launch of 512 plots must give, plot for plot, the bits
of the same plots in smaller launches -- a size-independent property -- and the oracle pins a sample of them."""
import numpy as np
import pytest
import torch

from oracle import network, projection
from stratanet2_vegetation_coverage_maps_amd import PointNet2, inference, losses, project_to_plotwise_coverages
from stratanet2_vegetation_coverage_maps_amd.project_to_2d import project_batch_to_2d_rasters
from stratanet2_vegetation_coverage_maps_amd.synthetic import make_args, make_batch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _trained_stats_model(args, seed, N, dtype="fp32"):
    """Default-initialised weights + running statistics that are not (0, 1): one training-mode forward on 8 other plots."""
    args.cuda, args.mma_dtype = 0, dtype
    m = PointNet2(args)
    m.load_state_dict(network.init_state_dict(seed))
    m.train()
    w = make_batch(8 if N <= 32768 else 2, N, first_plot=9000)
    w["fps_start"] = torch.zeros(2, w["cloud"].shape[0], dtype=torch.long)
    with torch.no_grad():
        m(w)
    torch.cuda.synchronize()
    return m.eval()


def _eval(m, cloud, xyz, args):
    d = {"cloud": cloud, "xyz": xyz, "fps_start": torch.zeros(2, cloud.shape[0], dtype=torch.long)}
    with torch.no_grad():
        cov, proba = m(d)
        clouds_dev = m._last_cloud_dev[1]
        m._last_cloud_dev = None
        rasters, pix = project_batch_to_2d_rasters(clouds_dev, cov, args)
    return cov, proba, rasters, pix


def test_parcel_launch_of_512_plots_equals_smaller_launches_and_the_oracle():
    B, N = 512, 10000
    args = make_args(subsample_size=N)                       # reference defaults: ratios .25 / .25, r sqrt2 / sqrt8 (config.py:77-80)
    m = _trained_stats_model(args, 0, N)
    d = make_batch(B, N, first_plot=0)
    cov, proba, rasters, pix = _eval(m, d["cloud"], d["xyz"], args)
    torch.cuda.synchronize()
    assert torch.isfinite(cov).all() and torch.isfinite(proba).all()
    covb, probab, pixb = cov.view(B, N, 4), proba.view(B, N, 4), pix.view(B, N)
    # (1) the same plots 32 at a time: the same bits.  (Two layers have two forms each, chosen by the ROW COUNT of a launch:
    # above 65 536 rows FP2 and FP1 hoist their weights through the interpolation -- hip_ops.fp_desc, "source-side form" --,
    # below they run row by row on the matrix cores / per lane.  32 plots = 80 000 FP2 rows and 320 000 FP1 rows take the forms
    # of the 512-plot launch -- in eval FP1 and the head are one kernel at every size --; every other kernel has one form.)  The
    # first and the last two groups + three inside cover every position of the SA kernels' "plots eight at a time" turns.
    for s in (0, 32, 160, 288, 416, 448, 480):
        c32, p32, r32, x32 = _eval(m, d["cloud"][s:s + 32], d["xyz"][s:s + 32], args)
        assert torch.equal(c32.view(32, N, 4), covb[s:s + 32]), f"coverages of plots {s}..{s + 31} depend on the launch size"
        assert torch.equal(p32.view(32, N, 4), probab[s:s + 32]), s
        assert torch.equal(x32.view(32, N), pixb[s:s + 32]), s
        assert torch.equal(torch.nan_to_num(r32, nan=-1.0), torch.nan_to_num(rasters[s:s + 32], nan=-1.0)), s
    # (2) eight / four at a time FP2 (20 000 / 10 000 rows) and FP1 (four: 40 000 rows) take their other form: the same function,
    # another order of fp32 operations -- equal to rounding (measured 9e-8), far inside the 1e-4 of the contract; index
    # structures identical
    e_small = 0.0
    for s, n in ((100, 8), (40, 4), (508, 4)):
        cs, _, rs, xs = _eval(m, d["cloud"][s:s + n], d["xyz"][s:s + n], args)
        assert torch.equal(xs.view(n, N), pixb[s:s + n])
        assert torch.equal(torch.isnan(rs), torch.isnan(rasters[s:s + n]))
        e_small = max(e_small, float((cs.view(n, N, 4) - covb[s:s + n]).abs().max()))
    assert e_small < 2e-6, e_small
    e4 = e_small
    # (3) the oracle on four plots of the launch (first, last, two inside): outputs 1e-4, pixel ids and NaN masks exact
    sel = [0, 97, 300, 511]
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    with torch.no_grad():
        cov_r, proba_r, _ = network.forward(sd, d["cloud"][sel], d["xyz"][sel], args, training=False, use_kdtree=True)
    cov_r, proba_r = cov_r.view(4, N, 4), proba_r.view(4, N, 4)
    worst = 0.0
    for j, b in enumerate(sel):
        ec = float((covb[b].cpu() - cov_r[j]).abs().max())
        ep = float((probab[b].cpu() - proba_r[j]).abs().max())
        worst = max(worst, ec, ep)
        p = projection.p1_pixel_ids(d["cloud"][b], args.diam_pix, args.diam_meters)
        assert torch.equal(pixb[b].cpu(), (p[1] * args.diam_pix + p[0]).int()), f"pixel ids of plot {b}"
        ref = projection.project_to_2d_rasters(d["cloud"][b], cov_r[j].t(), args)
        got = rasters[b].double().cpu().numpy()
        assert np.array_equal(np.isnan(got), np.isnan(ref)), f"NaN mask of plot {b}"
        np.testing.assert_allclose(np.nan_to_num(got), np.nan_to_num(ref), atol=1e-4, rtol=0)
    print(f"\n[C4 launch shape 512 x 10 000] 512-plot launch == 32-plot launches (bits); vs 8- and 4-plot launches {e4:.2e}; "
          f"vs the oracle on plots {sel}: {worst:.2e} (tol 1e-4)")
    assert worst <= 1e-4, worst


def test_predict_parcel_at_512_plots_per_launch_with_three_passes_in_flight():
    """`predict_parcel` as bench.py's config-4 leg calls it (512 plots per launch, prefetch = 3: three geometry passes in flight on
    side streams) against the same parcel fed 32 plots at a time with no overlap at all (32 plots: the layer forms of the big
    launch, see above): the merge is ORDER-DEPENDENT and plot after plot (rasterio.merge callback,
    inference/geotiff_raster.py:294-347), so equal mosaics mean equal rasters in the same order."""
    plots, N, cols, stride = 1536, 10000, 32, 5.0
    args = make_args(subsample_size=N)
    m = _trained_stats_model(args, 0, N)
    rows = plots // cols
    H, W = int(20 + stride * (rows - 1)), int(20 + stride * (cols - 1))

    def batches(per):
        out = []
        for s in range(0, plots, per):
            d = make_batch(per, N, first_plot=s)
            k = torch.arange(s, s + per)
            c = torch.stack([10.0 + stride * (k % cols), 10.0 + stride * (k // cols)], 1).double()
            out.append({"cloud": d["cloud"].to(DEV), "xyz": d["xyz"].to(DEV), "plot_center": c,
                        "fps_start": torch.zeros(2, per, dtype=torch.int64)})
        return out

    big = inference.ParcelMosaic(0.0, float(H), H, W, args, DEV)
    assert inference.predict_parcel(m, batches(512), big, args, prefetch=3) == plots
    small = inference.ParcelMosaic(0.0, float(H), H, W, args, DEV)
    assert inference.predict_parcel(m, batches(32), small, args, prefetch=0) == plots
    torch.cuda.synchronize()
    a, b = big.result(), small.result()
    assert torch.equal(torch.isnan(a), torch.isnan(b))
    assert torch.equal(torch.nan_to_num(a), torch.nan_to_num(b))
    assert float((~torch.isnan(a[0])).float().mean()) > 0.5
    fa, ta = big.finalize()
    fb, tb = small.finalize()
    assert torch.equal(ta, tb) and torch.equal(torch.nan_to_num(fa), torch.nan_to_num(fb))


def test_config5_launch_shape_8_x_131072_bf16():
    """BASELINE configs[4]: 8 plots x 131 072 points, bfloat16 operands on the matrix cores and bfloat16 per-point rows.
    (1) eval: the 8-plot launch == each plot launched alone (bits); (2) one of them against the oracle with the same operand
    and storage roundings (tolerance 3e-3 on pointwise outputs: tests/test_gpu_bf16.py says why a stored activation next to
    a bfloat16 rounding boundary costs a whole bfloat16 ulp on its row; plot-wise coverages 1e-4); (3) a TRAINING step of the
    whole launch in bf16 against the same step in fp32 mode -- a SANITY bound, not parity (the bf16 gradients are held to the
    oracle with the same roundings at 1 x 131 072 and 4 x 32 768 in tests/test_gpu_bf16.py): loss and plot-wise outputs move by
    what bfloat16 must (< 2e-2), every gradient finite, the flat gradient points the fp32 mode's way (cosine >= 0.95; measured
    0.988: bfloat16 storage of dy1 and the d pre-activation rows is a 2^-8 relative perturbation of every per-point gradient)."""
    B, N = 8, 131072
    args = make_args(subsample_size=N, ratio1=1024 / N, r1=1.0, ratio2=0.25, r2=2.0)
    m = _trained_stats_model(args, 1, N, "bf16")
    d = make_batch(B, N, first_plot=77)
    cov, proba, rasters, pix = _eval(m, d["cloud"], d["xyz"], args)
    torch.cuda.synchronize()
    assert m._act_dtype(B * N) == torch.bfloat16 and m._act_dtype(N) == torch.bfloat16
    covb = cov.view(B, N, 4)
    for b in range(B):
        c1, p1, r1, x1 = _eval(m, d["cloud"][b:b + 1], d["xyz"][b:b + 1], args)
        assert torch.equal(c1.view(N, 4), covb[b]), f"plot {b}: the eval result depends on the launch size"
        assert torch.equal(x1.view(N), pix.view(B, N)[b])
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    b = 5
    with torch.no_grad():
        cov_r, proba_r, _ = network.forward({k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()},
                                            d["cloud"][b:b + 1].double(), d["xyz"][b:b + 1], args, training=False,
                                            use_kdtree=True, bf16_layers=PointNet2.BF16_BLOCKS, act_bf16=True)
        pred = project_to_plotwise_coverages(covb[b].contiguous(), d["cloud"][b:b + 1], args)
        pred_r = projection.project_to_plotwise_coverages(cov_r.float(), d["cloud"][b:b + 1], args)
    e_pt = float((covb[b].cpu().double() - cov_r).abs().max())
    e_pr = float((proba.view(B, N, 4)[b].cpu().double() - proba_r).abs().max())
    e_pw = float((pred.cpu() - pred_r).abs().max())
    print(f"\n[C5 launch shape 8 x 131 072, bf16, eval] 8-plot launch == 1-plot launches (bits); plot {b} vs the oracle with the "
          f"same roundings: coverages {e_pt:.2e}, probabilities {e_pr:.2e} (tol 3e-3), plot-wise {e_pw:.2e} (tol 1e-4)")
    assert e_pt <= 3e-3 and e_pr <= 3e-3 and e_pw <= 1e-4
    # (3) the training step of the whole launch, bf16 against fp32 mode
    res = {}
    for dtype in ("bf16", "fp32"):
        args.cuda, args.mma_dtype = 0, dtype
        mt = PointNet2(args)
        mt.load_state_dict(network.init_state_dict(1))
        mt.train()
        dd = {"cloud": d["cloud"], "xyz": d["xyz"], "fps_start": torch.zeros(2, B, dtype=torch.long)}
        c, p = mt(dd)
        pr = project_to_plotwise_coverages(c, d["cloud"], args, model=mt)
        loss, _ = losses.total_loss(pr, p, d["coverages"].cuda(), d["pdf_all"].cuda(), args.m, args.e)
        loss.backward()
        torch.cuda.synchronize()
        res[dtype] = (float(loss), pr.detach().cpu(), {k: q.grad.detach().cpu().double() for k, q in mt.named_parameters()})
        del mt, c, p, pr, loss
        torch.cuda.empty_cache()
    dl = abs(res["bf16"][0] - res["fp32"][0])
    dp = float((res["bf16"][1] - res["fp32"][1]).abs().max())
    dot = sum(float((res["bf16"][2][k] * g).sum()) for k, g in res["fp32"][2].items())
    n16 = sum(float((g ** 2).sum()) for g in res["bf16"][2].values()) ** 0.5
    n32 = sum(float((g ** 2).sum()) for g in res["fp32"][2].values()) ** 0.5
    cos = dot / (n16 * n32)
    print(f"[C5 launch shape, training step] bf16 vs fp32 mode: |d loss| {dl:.2e}, max |d plot-wise| {dp:.2e}, "
          f"cosine of the flat gradients {cos:.4f}, norm ratio {n16 / n32:.3f}")
    assert all(torch.isfinite(g).all() for g in res["bf16"][2].values())
    assert np.isfinite(res["bf16"][0]) and dl < 2e-2 and dp < 2e-2 and cos >= 0.95 and 0.8 < n16 / n32 < 1.25
