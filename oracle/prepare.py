"""CPU restatement of the offline preparation steps next to the hot path.  TEST INFRASTRUCTURE ONLY.

PINNED: `tests/golden/f_load_cloud.npz` and `f_znorm.npz` hold what the reference's own `load_cloud` and
`normalize_z_with_minz_in_a_radius` (on the real sklearn kd-tree) returned in the build container
(`oracle/make_golden_aux.py`); `tests/test_oracle_golden_aux.py` holds this file to them bit for bit.

`normalize_z_with_minz_in_a_radius` follows `/root/reference/utils/load_data.py:237-249`; sklearn's
`NearestNeighbors(algorithm="kd_tree").radius_neighbors` is restated as what it computes: neighbours of point i are the
points j whose float64 reduced distance (dx*dx + dy*dy) is <= radius*radius (inclusive)."""
import numpy as np


def radius_neighbors_min(xy, z, radius):
    """min over {j : |xy_i - xy_j| <= radius} of z_j, brute force in float64 (O(n^2): small clouds only)."""
    xy = np.asarray(xy, dtype=np.float64)
    r2 = float(radius) * float(radius)
    out = np.empty(len(z), dtype=z.dtype)
    for i in range(len(z)):
        d = xy - xy[i]
        rd = d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]
        out[i] = np.min(z[rd <= r2])
    return out


def normalize_z_with_minz_in_a_radius(cloud, znorm_radius_in_meters):
    cloud = cloud.copy()
    xyz = cloud[:3, :].transpose()
    z = xyz[:, 2]
    zmin_neigh = [zz for zz in radius_neighbors_min(xyz[:, :2], z, znorm_radius_in_meters)]
    cloud[2] = cloud[2] - zmin_neigh
    return cloud


# ---------------------------------------------------------------------------------------------------------------------
# load_cloud (`/root/reference/data_loader/loader.py:73-255`), restated with numpy 1.21's casting written out: under the
# reference's pin a float32 array combined with a python / np.float64 SCALAR stays float32 (value-based casting), which
# numpy >= 2 no longer does implicitly (python scalars stay weak under NEP 50, so the reference's own code gives the same
# bits under numpy 2.2: recorded in the fixture).  The random draws come from `rs` in the reference's order.  Each line
# cites the line it follows.
# ---------------------------------------------------------------------------------------------------------------------
def add_fake_empty_ground_points(diam_meters, n_input_feats, cloud):                      # :90-105
    x = np.arange(-diam_meters // 2, diam_meters // 2, 1) + 0.5                           # get_x_y_meshgrid :108-113
    xx, yy = np.meshgrid(x, x, sparse=True)
    fx, fy = (xx + 0 * yy).flatten(), (yy + 0 * xx).flatten()
    r = np.sqrt(fx ** 2 + fy ** 2)
    fake_points = [[a, b, 0.0] + (n_input_feats - 3) * [0.0] for a, b, c in zip(fx, fy, r) if c < diam_meters // 2]
    return np.concatenate([cloud, np.array(fake_points, dtype=np.float32).transpose()], axis=1)


def load_cloud(raw_cloud, plot_center, args, train, rs, with_noise=True):
    f32 = np.float32
    cloud = raw_cloud.astype(np.float32).copy()
    cloud[0] = cloud[0] - f32(plot_center[0])                                             # center_cloud :127-132
    cloud[1] = cloud[1] - f32(plot_center[1])
    cloud = add_fake_empty_ground_points(args.diam_meters, 10, cloud)
    xyz = cloud[:3].copy()                                                                # :80
    if train:                                                                             # augment :161-214
        flip_x = rs.random() > 0.5                                                        # :217-222
        flip_y = rs.random() > 0.5
        angle = np.radians(rs.choice(360, 1)[0])
        c, s = np.cos(angle), np.sin(angle)
        M = np.array(((c, -s), (s, c)))
        cloud[:2] = np.dot(cloud[:2].T, M).T                                              # rotate_around_z :225-230
        xyz[:2] = np.dot(xyz[:2].T, M).T
        if flip_x:
            cloud[0] = -cloud[0]
            xyz[0] = -xyz[0]
        if flip_y:
            cloud[1] = -cloud[1]
            xyz[1] = -xyz[1]
        if with_noise:
            sigma, clip = 0.01 * 10, 0.03 * 10
            cloud[:2] = cloud[:2] + np.clip(sigma * rs.randn(2, cloud.shape[1]), a_min=-clip, a_max=clip).astype(np.float32)
            clip = 0.03 * 65536
            for idx in (3, 4, 5, 6):                                                      # red, green, blue, near_infrared
                cloud[idx] = cloud[idx] + np.clip(sigma * rs.randn(cloud.shape[1]), a_min=-clip, a_max=clip).astype(np.float32)
    cloud[0] = cloud[0] / f32(10)                                                         # rescale_cloud :135-158
    cloud[1] = cloud[1] / f32(10)
    cloud[2] = cloud[2] / f32(args.z_max)
    for idx in (3, 4, 5, 6):
        cloud[idx] = cloud[idx] / f32(65536)
    cloud[7] = cloud[7] / f32(32768)
    for idx in (8, 9):
        cloud[idx] = (cloud[idx] - f32(1)) / f32(7 - 1)
    n_points = cloud.shape[1]                                                             # sample_cloud :233-247
    if n_points > args.subsample_size:
        sampled = rs.choice(n_points, args.subsample_size, replace=False)
    else:
        sampled = np.concatenate([np.arange(n_points), rs.choice(n_points, args.subsample_size - n_points, replace=True)])
    return cloud[:, sampled].copy(), xyz[:, sampled]
