"""Drop-in for the reference `model/project_to_2d.py`: same two functions, same arguments, same results, but one
scatter-max kernel sequence on the device instead of per-plot python loops with `torch.unique`, torch_scatter and
device<->CPU bounces (`/root/reference/model/project_to_2d.py:7-55` and `:58-113`)."""
import torch

from . import hip_ops as ops
from ._lib import StrataHipError


class _PlotProject(torch.autograd.Function):
    """P2 as one autograd node: forward = per-pixel max (first point wins ties) + mean over occupied pixels,
    backward = route d pred to the arg-max point of every (pixel, channel) -- torch_scatter's scatter_max/scatter_mean
    gradients."""

    @staticmethod
    def forward(ctx, pred_pointwise, clouds_dev, diam_pix, pix=None):
        B, _, N = clouds_dev.shape
        if pix is not None:          # the ids were computed ahead of time (hip_ops.plot_pixels: they depend on x, y only)
            pred, pix, arg, nocc = ops.plot_project_forward_pix(pred_pointwise.contiguous(), pix, B, N, diam_pix)
        else:
            pred, pix, arg, nocc = ops.plot_project_forward(pred_pointwise.contiguous(), clouds_dev, diam_pix)
        ctx.save_for_backward(arg, nocc, pix)
        ctx.dims = (B, N, int(diam_pix))
        return pred

    @staticmethod
    def backward(ctx, dpred):
        arg, nocc, pix = ctx.saved_tensors
        B, N, D = ctx.dims
        return ops.plot_project_backward(dpred.contiguous().float(), arg, nocc, pix, B, N, D), None, None, None


# The reference calls `project_to_plotwise_coverages(pred_pointwise, clouds, args)` with the very CPU tensor it just gave to
# `PointNet2.forward` (learning/train.py:53-56): the forward remembers its upload here (one entry, weakly referenced host
# tensor), so the unchanged call does not copy 21 MB a second time.
_LAST_UPLOAD = [None, None]          # [weakref to the host tensor, its device copy]


def remember_upload(host, dev_tensor):
    import weakref
    _LAST_UPLOAD[0], _LAST_UPLOAD[1] = weakref.ref(host), dev_tensor


def _clouds_on_device(clouds, device, model_cache=None):
    if isinstance(clouds, torch.Tensor) and clouds.is_cuda:
        return clouds.float().contiguous()
    if model_cache is not None and model_cache[0] is clouds:
        return model_cache[1]
    ref, dev_t = _LAST_UPLOAD
    if ref is not None and ref() is clouds and dev_t is not None and dev_t.device == torch.device(device):
        _LAST_UPLOAD[0] = _LAST_UPLOAD[1] = None      # used once: do not keep the batch alive
        return dev_t
    return clouds.to(device=device, dtype=torch.float32, non_blocking=True).contiguous()


def project_to_plotwise_coverages(pred_pointwise, clouds, args, model=None, geometry=None):
    """pred_pointwise (B*N,4) on the device, clouds (B,10,N) (CPU, as the DataLoader hands them, or device) ->
    (B,4) [low_veg, bare_soil, med_veg, high_veg], differentiable w.r.t. pred_pointwise.
    `model` (optional): the PointNet2 whose forward already uploaded `clouds`, to skip a second H2D copy.
    `geometry` (optional, additive): the handle of a geometry pass that ran with `model.p2_diam_pix = args.diam_pix`
    (PointNet2._geometry): the pixel id of every point is already there (it depends on the plot's x, y only), and the
    projection is two launches instead of three."""
    if not pred_pointwise.is_cuda:
        raise StrataHipError("project_to_plotwise_coverages needs pred_pointwise on a HIP device: no CPU fallback")
    cache = getattr(model, "_last_cloud_dev", None) if model is not None else None
    with torch.cuda.device(pred_pointwise.device):
        clouds_dev = _clouds_on_device(clouds, pred_pointwise.device, cache)
        if cache is not None:
            model._last_cloud_dev = None          # used once: do not keep the batch's host and device clouds alive
        pix = getattr(geometry, "p2_pix", None) if geometry is not None else None
        if pix is not None and (getattr(geometry, "p2_diam_pix", None) != int(args.diam_pix) or
                                pix.numel() != clouds_dev.shape[0] * clouds_dev.shape[2]):
            pix = None                            # ids of another grid or batch: recompute
        return _PlotProject.apply(pred_pointwise, clouds_dev, args.diam_pix, pix)


# The reference's inference loop calls `project_to_2d_rasters` once PER PLOT of a batch it has just sent through the model
# (predict.py:109-126, predict_utils.py:94-102): `coverages_pointwise[idx]` of `model.get_batch_format(...)` with `clouds[idx]`.
# Both arguments are views of the batch's tensors, so the FIRST such call rasterises the whole batch in one launch and brings the
# (B,3,D,D) result to the host in one read; the calls for the other plots of the batch are answered from that array.  Same
# kernel per plot as the single-plot call: the same bits (tests/test_inference.py).  One batch is remembered, by the IDENTITY of
# the batch tensors (weak references: a new batch is a new tensor object even where the allocator hands out the same address)
# and the version counter of the coverages.  BATCH_RASTERS = False: every call rasterises its own plot.
BATCH_RASTERS = True
_RASTER_BATCH = {"cov": None, "clouds": None, "version": None, "grid": None, "rasters": None}


def _batch_view(t, row_len):
    """(base, index) when `t` is plot `index` of a contiguous batch tensor `base` whose plots are `row_len` elements long."""
    base = getattr(t, "_base", None)
    if base is None or not base.is_contiguous() or row_len <= 0:
        return None, None
    off = t.storage_offset() - base.storage_offset()
    if off % row_len:
        return None, None
    return base, off // row_len


def _rasters_of_batch(cloud, coverages_pointwise, args):
    """The plot's rasters out of a batched launch, or None when the arguments are not plots of batch tensors."""
    if coverages_pointwise.dim() != 2 or coverages_pointwise.shape[0] != 4 or cloud.dim() != 2:
        return None
    N = coverages_pointwise.shape[1]
    if coverages_pointwise.stride() != (1, 4) or cloud.shape[1] != N or cloud.stride() != (N, 1):
        return None
    cbase, i = _batch_view(coverages_pointwise, 4 * N)             # (B*N,4) rows as PointNet2.forward returns them
    if cbase is None or cbase.dim() != 2 or cbase.shape[1] != 4 or cbase.dtype != torch.float32 or cbase.shape[0] % N:
        return None
    B = cbase.shape[0] // N
    lbase, j = _batch_view(cloud, cloud.shape[0] * N)              # (B,C,N) as the DataLoader collates them
    if lbase is None or lbase.dim() != 3 or lbase.shape[0] != B or lbase.shape[2] != N or i != j or not (0 <= i < B):
        return None
    grid = (int(args.diam_pix), int(args.diam_meters))
    c = _RASTER_BATCH
    hit = (c["cov"] is not None and c["cov"]() is cbase and c["clouds"]() is lbase and c["version"] == cbase._version and
           c["grid"] == grid)
    if not hit:
        import weakref
        dev = cbase.device
        with torch.cuda.device(dev):
            clouds_dev = _clouds_on_device(lbase, dev)             # the forward's own upload when it is still remembered
            rasters, _ = ops.raster_project(cbase.detach(), clouds_dev, grid[0], grid[1])
            host = rasters.double().cpu().numpy()                  # ONE device-to-host read for the batch
        c.update(cov=weakref.ref(cbase), clouds=weakref.ref(lbase), version=cbase._version, grid=grid, rasters=host)
    return c["rasters"][i].copy()


def project_to_2d_rasters(cloud, coverages_pointwise, args):
    """cloud (>=2,N) normalised coordinates of ONE plot, coverages_pointwise (4,N) -> np.ndarray float64
    (3, diam_pix, diam_pix) [low, med, high], image[y, x], NaN where no point falls, rows flipped."""
    if not coverages_pointwise.is_cuda:
        raise StrataHipError("project_to_2d_rasters needs coverages_pointwise on a HIP device: no CPU fallback")
    if BATCH_RASTERS:
        r = _rasters_of_batch(cloud, coverages_pointwise, args)
        if r is not None:
            return r
    dev = coverages_pointwise.device
    with torch.cuda.device(dev):
        cov = coverages_pointwise.detach().float().t().contiguous()                 # (N,4)
        cl = cloud[:2].to(device=dev, dtype=torch.float32).contiguous().unsqueeze(0)   # (1,2,N)
        rasters, _ = ops.raster_project(cov, cl, args.diam_pix, args.diam_meters)
    return rasters[0].double().cpu().numpy()


def project_batch_to_2d_rasters(clouds_dev, coverages_pointwise, args):
    """Batched form for parcel inference (reference loop `predict.py:115-126`): clouds (B,C,N) on device,
    coverages_pointwise (B*N,4) -> device tensor (B,3,D,D) fp32 with NaN, and the (B*N) pixel ids."""
    return ops.raster_project(coverages_pointwise.detach().contiguous(), clouds_dev, args.diam_pix, args.diam_meters)
