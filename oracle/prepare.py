"""CPU restatement of the offline preparation steps next to the hot path.  TEST INFRASTRUCTURE ONLY.

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
