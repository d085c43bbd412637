"""Batch input pipeline on the device -- counterpart of the reference DataLoader's `load_cloud`
(`/root/reference/data_loader/loader.py:73-87`: centre, fake ground points, augmentation, rescale, subsample), which is
numpy code with a `deepcopy` per sample running in the training process.  Here the host only draws the random numbers
(in the reference's order, from the reference's generator: `numpy.random`) and one kernel builds `cloud (B,10,N)` and
`xyz (B,3,N)` for the whole batch from the raw plots resident on the device.

    rs = numpy.random            # or a RandomState: the same draws as the reference under the same seed
    batch = prepare_batch(raw_plots, centers, args, train=True, rs=rs, device=dev)
    cov, proba = model(batch)    # {"cloud": ..., "xyz": ...} device tensors

`noise="numpy"` draws the gaussian noise with `rs.randn` exactly where the reference does (bit-identical batches, used by
the parity tests); `noise="device"` draws it with torch on the device (same distribution, no host work);
`noise=None` leaves it out.
"""
import numpy as np
import torch

from . import hip_ops as ops


def fake_ground_xy(diam_meters: int) -> np.ndarray:
    """Positions of `add_fake_empty_ground_points` (loader.py:90-105): the centres of the 1 m cells of the final raster
    that lie inside the plot disc, in the reference's order.  (316 points for diam_meters = 20.)"""
    x = np.arange(-diam_meters // 2, diam_meters // 2, 1) + 0.5
    xx, yy = np.meshgrid(x, x, sparse=True)
    fx = (xx + 0 * yy).flatten()
    fy = (yy + 0 * xx).flatten()
    r = np.sqrt(fx ** 2 + fy ** 2)
    keep = r < diam_meters // 2
    return np.stack([fx[keep], fy[keep]], 1).astype(np.float32)


def draw_plot_randoms(n_points: int, subsample_size: int, train: bool, rs, noise: bool):
    """The random draws `load_cloud` makes for ONE plot of `n_points` points (fake points included), in its order:
    augment -> get_xyz_augmentation_params (:217-222), xy noise (:186-193), colour noise (:200-208); then sample_cloud
    (:233-247)."""
    out = {}
    if train:
        flip_x = rs.random() > 0.5
        flip_y = rs.random() > 0.5
        angle = np.radians(rs.choice(360, 1)[0])
        out.update(angle=angle, flip_x=bool(flip_x), flip_y=bool(flip_y))
        if noise:
            sigma, clip = 0.01 * 10, 0.03 * 10
            nxy = np.clip(sigma * rs.randn(2, n_points), a_min=-clip, a_max=clip).astype(np.float32)
            clip_c = 0.03 * 65536
            ncol = [np.clip(sigma * rs.randn(n_points), a_min=-clip_c, a_max=clip_c).astype(np.float32)   # sigma of x,y:
                    for _ in range(4)]                                                                      # loader.py:180,202
            out["noise"] = np.concatenate([nxy, np.stack(ncol)], 0)
    if n_points > subsample_size:
        idx = rs.choice(n_points, subsample_size, replace=False)
    else:
        idx = np.concatenate([np.arange(n_points), rs.choice(n_points, subsample_size - n_points, replace=True)])
    out["idx"] = idx.astype(np.int32)
    return out


def prepare_batch(raw_plots, centers, args, train: bool, rs=np.random, device="cuda:0", noise="numpy"):
    """raw_plots: list of (10, n_i) float32 arrays/tensors (host or device); centers: (B,2).  Returns the `cloud_data`
    dict the model takes: {"cloud": (B,10,N), "xyz": (B,3,N)} on the device."""
    dev = torch.device(device)
    B = len(raw_plots)
    N = args.subsample_size
    fake = fake_ground_xy(args.diam_meters)
    n_raw = [int(p.shape[1]) for p in raw_plots]
    draws = [draw_plot_randoms(n + len(fake), N, train, rs, noise == "numpy") for n in n_raw]
    raw = torch.cat([torch.as_tensor(p, dtype=torch.float32).to(dev) for p in raw_plots], 1).contiguous()
    offsets = torch.tensor(np.concatenate([[0], np.cumsum(n_raw)]), dtype=torch.int32, device=dev)
    idx = torch.from_numpy(np.stack([d["idx"] for d in draws])).to(dev)
    rot = flips = nz = noffs = None
    if train:
        rot = torch.tensor([[np.cos(d["angle"]), np.sin(d["angle"])] for d in draws], dtype=torch.float64, device=dev)
        flips = torch.tensor([[int(d["flip_x"]), int(d["flip_y"])] for d in draws], dtype=torch.int32, device=dev)
        tot = [n + len(fake) for n in n_raw]
        noffs = torch.tensor(np.concatenate([[0], np.cumsum(tot)[:-1]]), dtype=torch.int64, device=dev)
        if noise == "numpy":
            nz = torch.from_numpy(np.concatenate([d["noise"] for d in draws], 1)).to(dev).contiguous()
        elif noise == "device":
            with torch.cuda.device(dev):
                g = torch.randn(6, sum(tot), device=dev, dtype=torch.float32) * 0.1
                nz = torch.cat([g[:2].clamp(-0.3, 0.3), g[2:].clamp(-0.03 * 65536, 0.03 * 65536)], 0).contiguous()
        elif noise is not None:
            raise ValueError("noise must be 'numpy', 'device' or None")
    with torch.cuda.device(dev):
        cloud, xyz = ops.prepare_plots(raw, offsets, torch.as_tensor(np.asarray(centers), dtype=torch.float32).to(dev).contiguous(),
                                       torch.from_numpy(fake).to(dev), idx, args.z_max, rot, flips, nz, noffs)
    return {"cloud": cloud, "xyz": xyz}
