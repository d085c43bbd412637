"""CPU restatement of the third-party primitives the reference hot path calls.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED for this file: torch-cluster 1.5.9, torch-geometric 1.7.2 and torch-scatter 2.0.7
(`setup_environment/torch_extensions.txt:1-3` of the reference) are not vendored in /root/reference, not
installed here and cannot be fetched; the reference has no tests or golden vectors.  Each function restates
the published algorithm of the named upstream function and is anchored on the reference's call site.

Canonical arithmetic used for every discrete decision (FPS argmax, ball test, kNN):
    d2 = (dx*dx + dy*dy) + dz*dz          each operation individually rounded to fp32, no FMA,
which is bit-identical to torch's `(a - b).pow(2).sum(1)` on an (n,3) fp32 tensor (checked by
tests/test_oracle_primitives.py::test_canonical_d2_matches_torch_sum).

The signatures mirror the upstream ones so that `oracle/make_golden.py` can register this module under
the names `torch_geometric.nn` / `torch_scatter` and run the reference's own glue code on top of it.
"""
import math
from typing import Callable, Optional

import numpy as np
import torch

# ----------------------------------------------------------------------------------------------------------
# canonical squared distance
# ----------------------------------------------------------------------------------------------------------


def canonical_d2(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """(dx*dx + dy*dy) + dz*dz with broadcasting over the leading dims; last dim = xyz."""
    d = a - b
    dx, dy, dz = d[..., 0], d[..., 1], d[..., 2]
    return (dx * dx + dy * dy) + dz * dz


def r2_threshold(r: float) -> torch.Tensor:
    """fp32 value of r*r computed in double: the threshold both `radius` implementations compare with."""
    return torch.tensor(float(r) * float(r), dtype=torch.float32)


# ----------------------------------------------------------------------------------------------------------
# torch_cluster.fps   (reference call: model/point_net2.py:22  `idx = fps(pos, batch, ratio=self.ratio)`)
# ----------------------------------------------------------------------------------------------------------

# The upstream CPU kernel draws its start point with C `rand() % n` (random_start=True is the default and
# the reference never seeds anything: SURVEY.md section 0.4).  For reproducible goldens a provider of start
# indices can be installed: a callable (plot_id, n_points_in_plot, call_counter) -> local start index.
_fps_start_provider: Optional[Callable[[int, int, int], int]] = None
_fps_call_counter = 0


def set_fps_start_provider(fn: Optional[Callable[[int, int, int], int]]) -> None:
    global _fps_start_provider, _fps_call_counter
    _fps_start_provider = fn
    _fps_call_counter = 0


def fps_num_samples(n: int, ratio: float) -> int:
    """Upstream: `deg.toType(kFloat) * ratio` then `.ceil()`: an fp32 product (the double `ratio` is cast to
    the tensor's fp32 opmath type)."""
    return int(math.ceil(float(np.float32(n) * np.float32(ratio))))


def fps(x: torch.Tensor, batch: Optional[torch.Tensor] = None, ratio: float = 0.5,
        random_start: bool = True) -> torch.Tensor:
    """Farthest point sampling, torch-cluster 1.5.9 `fps_cpu` semantics.

    Per plot (plots are contiguous runs of equal `batch` value, ascending): m = ceil(fp32(n) * ratio)
    samples; the first is the start point (random unless random_start=False -> local index 0); then
    repeatedly `dist = min(dist, d2(., last))`, `next = argmax(dist)` (first maximal index).
    Returns GLOBAL indices (int64), plots in order, selection order within a plot.
    """
    global _fps_call_counter
    n_total = x.shape[0]
    if batch is None:
        batch = torch.zeros(n_total, dtype=torch.long)
    counts = torch.bincount(batch).tolist()
    out = []
    start = 0
    for b, n in enumerate(counts):
        y = x[start:start + n].float()
        m = fps_num_samples(n, ratio)
        if _fps_start_provider is not None:
            s = int(_fps_start_provider(b, n, _fps_call_counter))
        elif random_start:
            s = int(torch.randint(0, n, (1,)).item())
        else:
            s = 0
        idx = torch.empty(m, dtype=torch.long)
        idx[0] = s
        dist = canonical_d2(y, y[s])
        for i in range(1, m):
            a = int(torch.argmax(dist))          # first maximal index
            idx[i] = a
            dist = torch.minimum(dist, canonical_d2(y, y[a]))
        out.append(idx + start)
        start += n
    _fps_call_counter += 1
    return torch.cat(out)


def fps_batched(pos: torch.Tensor, m: int, start: torch.Tensor) -> torch.Tensor:
    """Same algorithm on a regular batch `pos (B,N,3)`, all plots advanced together (identical results:
    plots never interact).  Returns LOCAL indices (B,m) int64.  Used by the CPU baseline (the upstream
    kernel also runs plots in parallel, `at::parallel_for` over the batch)."""
    B, N, _ = pos.shape
    idx = torch.empty(B, m, dtype=torch.long)
    idx[:, 0] = start
    ar = torch.arange(B)
    dist = canonical_d2(pos, pos[ar, start].unsqueeze(1))
    for i in range(1, m):
        a = torch.argmax(dist, dim=1)
        idx[:, i] = a
        dist = torch.minimum(dist, canonical_d2(pos, pos[ar, a].unsqueeze(1)))
    return idx


# ----------------------------------------------------------------------------------------------------------
# torch_cluster.radius   (reference call: model/point_net2.py:23-25, max_num_neighbors=2000)
# ----------------------------------------------------------------------------------------------------------


def _plot_ranges(batch: torch.Tensor):
    counts = torch.bincount(batch).tolist()
    starts = np.concatenate([[0], np.cumsum(counts)]).tolist()
    return [(starts[i], starts[i + 1]) for i in range(len(counts))]


def radius(x: torch.Tensor, y: torch.Tensor, r: float, batch_x: Optional[torch.Tensor] = None,
           batch_y: Optional[torch.Tensor] = None, max_num_neighbors: int = 32,
           use_kdtree: bool = False) -> torch.Tensor:
    """For each query `y[i]` all `x[j]` of the same plot with d2(x[j], y[i]) < fp32(r*r) (STRICT, nanoflann
    `RadiusResultSet::addPoint`), at most `max_num_neighbors` per query.  Returns `[2,E]` int64:
    row 0 = query index into y, row 1 = source index into x (the reference unpacks it as `row, col`).

    Order / cap: the upstream CPU kernel keeps the first `max_num_neighbors` hits in kd-tree traversal order
    (implementation-defined).  This restatement defines: ascending source index.  The reference aggregates
    with `max`, so order only matters when a ball exceeds the cap (then parity is build-vs-build only).

    use_kdtree=True finds candidates with scipy cKDTree (double precision, slightly enlarged radius) and then
    applies the canonical fp32 test -- identical output, used at the benchmark sizes.
    """
    if batch_x is None:
        batch_x = torch.zeros(x.shape[0], dtype=torch.long)
    if batch_y is None:
        batch_y = torch.zeros(y.shape[0], dtype=torch.long)
    thr = r2_threshold(r)
    rows, cols = [], []
    rx, ry = _plot_ranges(batch_x), _plot_ranges(batch_y)
    for (xs, xe), (ys, ye) in zip(rx, ry):
        px, py = x[xs:xe].float(), y[ys:ye].float()
        if use_kdtree:
            from scipy.spatial import cKDTree
            tree = cKDTree(px.double().numpy())
            cand = tree.query_ball_point(py.double().numpy(), float(r) * 1.001 + 1e-6, return_sorted=True)
            qi = np.repeat(np.arange(len(cand)), [len(c) for c in cand])
            sj = np.concatenate([np.asarray(c, dtype=np.int64) for c in cand]) if len(qi) else np.zeros(0, np.int64)
            qi_t, sj_t = torch.from_numpy(qi), torch.from_numpy(sj)
            keep = canonical_d2(px[sj_t], py[qi_t]) < thr
            qi_t, sj_t = qi_t[keep], sj_t[keep]
        else:
            qs, ss = [], []
            chunk = max(1, (1 << 24) // max(1, px.shape[0]))
            for c0 in range(0, py.shape[0], chunk):
                d2 = canonical_d2(px.unsqueeze(0), py[c0:c0 + chunk].unsqueeze(1))   # (q, n)
                q, s = torch.nonzero(d2 < thr, as_tuple=True)                         # row-major => ascending s
                qs.append(q + c0)
                ss.append(s)
            qi_t, sj_t = torch.cat(qs), torch.cat(ss)
        # cap: first max_num_neighbors per query in ascending source index
        if qi_t.numel():
            cnt = torch.bincount(qi_t, minlength=py.shape[0])
            if int(cnt.max()) > max_num_neighbors:
                first = torch.cumsum(cnt, 0) - cnt
                rank = torch.arange(qi_t.numel()) - first[qi_t]
                keep = rank < max_num_neighbors
                qi_t, sj_t = qi_t[keep], sj_t[keep]
        rows.append(qi_t + ys)
        cols.append(sj_t + xs)
    return torch.stack([torch.cat(rows), torch.cat(cols)], dim=0)


# ----------------------------------------------------------------------------------------------------------
# torch_cluster.knn + torch_geometric.nn.knn_interpolate   (reference call: model/point_net2.py:63)
# ----------------------------------------------------------------------------------------------------------


def knn(x: torch.Tensor, y: torch.Tensor, k: int, batch_x: Optional[torch.Tensor] = None,
        batch_y: Optional[torch.Tensor] = None, use_kdtree: bool = False):
    """For each `y[i]` the k nearest `x[j]` of the same plot (canonical d2; ties -> lowest source index;
    neighbours listed nearest first).  Returns `[2, E]` int64: row 0 = index into y, row 1 = index into x.
    A plot with fewer than k sources yields that many neighbours (upstream behaviour; only hit by fp3, k=1)."""
    if batch_x is None:
        batch_x = torch.zeros(x.shape[0], dtype=torch.long)
    if batch_y is None:
        batch_y = torch.zeros(y.shape[0], dtype=torch.long)
    rows, cols = [], []
    for (xs, xe), (ys, ye) in zip(_plot_ranges(batch_x), _plot_ranges(batch_y)):
        px, py = x[xs:xe].float(), y[ys:ye].float()
        kk = min(k, px.shape[0])
        if use_kdtree and px.shape[0] > 4 * kk:
            from scipy.spatial import cKDTree
            nc = min(px.shape[0], kk + 5)
            _, cand = cKDTree(px.double().numpy()).query(py.double().numpy(), k=nc)
            cand = torch.from_numpy(np.asarray(cand, dtype=np.int64)).reshape(py.shape[0], nc)
            cand, _ = torch.sort(cand, dim=1)                                    # ascending index
            d2 = canonical_d2(px[cand], py.unsqueeze(1))                         # (q, nc)
            order = torch.argsort(d2, dim=1, stable=True)[:, :kk]                # ties -> lowest index
            nn_idx = torch.gather(cand, 1, order)
        else:
            out = []
            chunk = max(1, (1 << 23) // max(1, px.shape[0]))
            for c0 in range(0, py.shape[0], chunk):
                d2 = canonical_d2(px.unsqueeze(0), py[c0:c0 + chunk].unsqueeze(1))
                out.append(torch.argsort(d2, dim=1, stable=True)[:, :kk])
            nn_idx = torch.cat(out)
        q = torch.arange(py.shape[0]).unsqueeze(1).expand(-1, kk)
        rows.append(q.reshape(-1) + ys)
        cols.append(nn_idx.reshape(-1) + xs)
    return torch.stack([torch.cat(rows), torch.cat(cols)], dim=0)


def scatter_add(src: torch.Tensor, index: torch.Tensor, dim: int = 0, dim_size: Optional[int] = None):
    assert dim == 0
    n = int(index.max()) + 1 if dim_size is None else dim_size
    out = torch.zeros((n,) + tuple(src.shape[1:]), dtype=src.dtype)
    return out.index_add(0, index, src)


def knn_interpolate(x: torch.Tensor, pos_x: torch.Tensor, pos_y: torch.Tensor,
                    batch_x: Optional[torch.Tensor] = None, batch_y: Optional[torch.Tensor] = None,
                    k: int = 3, num_workers: int = 1, use_kdtree: bool = False) -> torch.Tensor:
    """torch-geometric 1.7.2 `knn_interpolate`: inverse-squared-distance interpolation of the features `x`
    living on `pos_x` onto `pos_y`.  Indices and weights under no_grad:
        w = 1 / clamp(d2, min=1e-16),   y = scatter_add(x[j] * w) / scatter_add(w)."""
    with torch.no_grad():
        y_idx, x_idx = knn(pos_x, pos_y, k, batch_x=batch_x, batch_y=batch_y, use_kdtree=use_kdtree)
        diff = pos_x[x_idx] - pos_y[y_idx]
        squared_distance = (diff * diff).sum(dim=-1, keepdim=True)
        weights = 1.0 / torch.clamp(squared_distance, min=1e-16)
    y = scatter_add(x[x_idx] * weights, y_idx, dim=0, dim_size=pos_y.size(0))
    y = y / scatter_add(weights, y_idx, dim=0, dim_size=pos_y.size(0))
    return y


# ----------------------------------------------------------------------------------------------------------
# torch_scatter.scatter_max / scatter_mean   (reference calls: model/project_to_2d.py:39,46-49)
# ----------------------------------------------------------------------------------------------------------


class _ScatterMaxLastDim(torch.autograd.Function):
    """out[..., g] = max over {i : index[i]==g} of src[..., i]; arg = FIRST index attaining it (the upstream CPU
    loop updates on strict `>`); groups that receive nothing -> 0 with arg = src.size(-1).  Backward routes the
    gradient to the arg position only."""

    @staticmethod
    def forward(ctx, src, index, dim_size):
        n = src.shape[-1]
        lead = src.shape[:-1]
        s2 = src.reshape(-1, n)
        idx2 = index.reshape(1, n).expand_as(s2)
        out = torch.full((s2.shape[0], dim_size), float("-inf"), dtype=src.dtype)
        out = out.scatter_reduce(1, idx2, s2, reduce="amax", include_self=True)
        # first position attaining the max
        is_max = s2 == out.gather(1, idx2)
        pos = torch.arange(n).unsqueeze(0).expand_as(s2)
        cand = torch.where(is_max, pos, torch.full_like(pos, n))
        arg = torch.full((s2.shape[0], dim_size), n, dtype=torch.long)
        arg = arg.scatter_reduce(1, idx2, cand, reduce="amin", include_self=True)
        out = torch.where(arg == n, torch.zeros_like(out), out)
        ctx.save_for_backward(arg)
        ctx.n = n
        ctx.src_shape = src.shape
        ctx.mark_non_differentiable(arg)
        return out.reshape(*lead, dim_size), arg.reshape(*lead, dim_size)

    @staticmethod
    def backward(ctx, grad_out, _grad_arg):
        (arg,) = ctx.saved_tensors
        n = ctx.n
        g2 = grad_out.reshape(-1, grad_out.shape[-1])
        grad = torch.zeros((g2.shape[0], n + 1), dtype=grad_out.dtype)
        grad.scatter_(1, arg, g2)
        return grad[:, :n].reshape(ctx.src_shape), None, None


def scatter_max(src: torch.Tensor, index: torch.Tensor, dim: int = -1, out=None,
                dim_size: Optional[int] = None):
    """torch-scatter 2.0.7 `scatter_max(src, index, dim)` -> (out, argmax).  Supports dim=-1 (1-D index
    broadcast over leading dims: the reference's `(4, B*N)` case) and dim=0."""
    if dim_size is None:
        dim_size = int(index.max()) + 1 if index.numel() else 0
    if dim in (-1, src.dim() - 1):
        return _ScatterMaxLastDim.apply(src, index, dim_size)
    if dim == 0:
        o, a = _ScatterMaxLastDim.apply(src.transpose(0, -1), index, dim_size)
        return o.transpose(0, -1), a.transpose(0, -1)
    raise NotImplementedError


def scatter_mean(src: torch.Tensor, index: torch.Tensor, dim: int = -1, out=None,
                 dim_size: Optional[int] = None) -> torch.Tensor:
    """torch-scatter `scatter_mean` for 1-D src: sum / clamp(count, min=1)."""
    assert src.dim() == 1
    if dim_size is None:
        dim_size = int(index.max()) + 1
    s = torch.zeros(dim_size, dtype=src.dtype).index_add(0, index, src)
    c = torch.zeros(dim_size, dtype=src.dtype).index_add(0, index, torch.ones_like(src))
    return s / c.clamp(min=1)


# ----------------------------------------------------------------------------------------------------------
# torch_geometric.nn.PointConv / global_max_pool   (reference: model/point_net2.py:19,27,39)
# ----------------------------------------------------------------------------------------------------------


def global_max_pool(x: torch.Tensor, batch: torch.Tensor) -> torch.Tensor:
    """Per-plot channel-wise max; size = batch.max()+1."""
    size = int(batch.max()) + 1
    return scatter_max(x, batch, dim=0, dim_size=size)[0]


class PointConv(torch.nn.Module):
    """torch-geometric 1.7.2 `PointConv(local_nn, global_nn=None, add_self_loops=...)` with aggr='max':
        out_i = max_{j in N(i)} local_nn(cat[x_j, pos_j - pos_i])          (features FIRST, then relative xyz)
    `edge_index[0]` = source j (indexes x and pos[0]), `edge_index[1]` = target i (indexes pos[1]); targets
    that receive no message get 0.  The reference constructs it with add_self_loops=False."""

    def __init__(self, local_nn=None, global_nn=None, add_self_loops: bool = True, **kwargs):
        super().__init__()
        assert not add_self_loops, "reference uses add_self_loops=False (model/point_net2.py:19)"
        self.local_nn = local_nn
        self.global_nn = global_nn

    def forward(self, x, pos, edge_index):
        if not isinstance(x, tuple):
            x = (x, None)
        if isinstance(pos, torch.Tensor):
            pos = (pos, pos)
        j, i = edge_index[0], edge_index[1]
        msg = pos[0][j] - pos[1][i]
        if x[0] is not None:
            msg = torch.cat([x[0][j], msg], dim=1)
        if self.local_nn is not None:
            msg = self.local_nn(msg)
        out = scatter_max(msg, i, dim=0, dim_size=pos[1].size(0))[0]
        if self.global_nn is not None:
            out = self.global_nn(out)
        return out
