"""Known-answer tests of the restated third-party primitives (oracle/primitives.py).  The reference holds no fixtures for
torch-cluster / torch-scatter / torch-geometric, so these pin the restatement against hand-computed
answers and against independent implementations that ARE available here (scipy's kd-tree, plain torch ops) on the rules
the restatement states: canonical fp32 d2, strict `<` in the ball, first maximum in FPS and scatter_max, lowest index on
kNN ties, the fp32 sample count."""
import numpy as np
import torch

from oracle import primitives as P


def test_canonical_d2_matches_torch_sum():
    """(dx*dx + dy*dy) + dz*dz, each operation rounded to fp32 = what torch computes for (a-b).pow(2).sum(1) on (n,3)."""
    g = torch.Generator().manual_seed(0)
    a = torch.rand(100000, 3, generator=g) * 20 - 10
    b = torch.rand(100000, 3, generator=g) * 20 - 10
    assert torch.equal(P.canonical_d2(a, b), (a - b).pow(2).sum(1))
    assert torch.equal(P.canonical_d2(a, b), ((a - b) * (a - b)).sum(1))
    d = (a - b).double()                                        # and it is NOT the fp64 sum rounded once
    assert not torch.equal(P.canonical_d2(a, b), (d * d).sum(1).float())


def test_fps_num_samples_is_an_fp32_product():
    assert P.fps_num_samples(10000, 0.25) == 2500 and P.fps_num_samples(4096, 0.25) == 1024
    assert P.fps_num_samples(32768, 1024 / 32768) == 1024
    assert P.fps_num_samples(10, 0.33) == 4 and P.fps_num_samples(3, 0.5) == 2
    # fp32(0.1) > 0.1: ten tenths of 1000 round UP in fp32 (ceil(100.0000015)) where exact arithmetic gives 100
    assert P.fps_num_samples(1000, 0.1) == int(np.ceil(np.float32(1000) * np.float32(0.1)))


def test_fps_known_answer_and_first_maximum():
    # points on a line: start 0 -> farthest (index 4), then the middle; a TIE (indices 1 and 3 both at distance 1 from
    # their nearest sample) goes to the lower index
    x = torch.tensor([[0., 0, 0], [1, 0, 0], [2, 0, 0], [3, 0, 0], [4, 0, 0]])
    P.set_fps_start_provider(lambda b, n, call: 0)
    try:
        assert P.fps(x, ratio=0.8).tolist() == [0, 4, 2, 1]
    finally:
        P.set_fps_start_provider(None)
    # two plots: global indices, plots in order; batched form agrees
    pos = torch.rand(2, 50, 3, generator=torch.Generator().manual_seed(1))
    loc = P.fps_batched(pos, 10, torch.tensor([3, 7]))
    P.set_fps_start_provider(lambda b, n, call: [3, 7][b])
    try:
        glob = P.fps(pos.reshape(100, 3), torch.arange(2).repeat_interleave(50), ratio=0.2)
    finally:
        P.set_fps_start_provider(None)
    assert glob.tolist() == (loc + torch.tensor([[0], [50]])).reshape(-1).tolist()


def test_radius_is_strict_capped_and_per_plot():
    x = torch.tensor([[0., 0, 0], [1, 0, 0], [2, 0, 0], [0.5, 0, 0], [10, 0, 0], [10.5, 0, 0]])
    bx = torch.tensor([0, 0, 0, 0, 1, 1])
    y = torch.tensor([[0., 0, 0], [10, 0, 0]])
    by = torch.tensor([0, 1])
    row, col = P.radius(x, y, 1.0, bx, by, max_num_neighbors=32)
    pairs = sorted(zip(row.tolist(), col.tolist()))
    assert pairs == [(0, 0), (0, 3), (1, 4), (1, 5)]            # x[1] at distance exactly 1 is OUT (strict <); plots do not mix
    # against scipy's kd-tree on generic points (no distance within an ulp of r): same sets
    from scipy.spatial import cKDTree
    g = torch.Generator().manual_seed(3)
    pts = torch.rand(400, 3, generator=g) * 4
    q = pts[:40]
    row, col = P.radius(pts, q, 0.7, max_num_neighbors=10 ** 6)
    want = cKDTree(pts.double().numpy()).query_ball_point(q.double().numpy(), 0.7)
    for i in range(40):
        assert sorted(col[row == i].tolist()) == sorted(want[i])
    # the kd-tree candidate path of the oracle gives the same lists as its own full scan
    r2, c2 = P.radius(pts, q, 0.7, max_num_neighbors=10 ** 6, use_kdtree=True)
    assert sorted(zip(row.tolist(), col.tolist())) == sorted(zip(r2.tolist(), c2.tolist()))
    # cap: at most max_num_neighbors per query
    row, col = P.radius(pts, q, 2.0, max_num_neighbors=5)
    assert max(int((row == i).sum()) for i in range(40)) == 5


def test_knn_ties_go_to_the_lowest_index_and_weights():
    x = torch.tensor([[1., 0, 0], [-1, 0, 0], [0, 1, 0], [0, 3, 0]])     # three sources at distance 1 from the origin
    y = torch.zeros(1, 3)
    yi, xi = P.knn(x, y, 2)
    assert xi.tolist() == [0, 1]
    yi, xi = P.knn(x, y, 3, use_kdtree=True)
    assert xi.tolist() == [0, 1, 2]
    # knn_interpolate: inverse squared distance, a coincident source dominates through clamp(d2, 1e-16)
    feats = torch.tensor([[1.], [2.], [4.], [8.]])
    out = P.knn_interpolate(feats, x, torch.tensor([[0., 2, 0]]), k=3)
    w = torch.tensor([1 / 5., 1 / 5., 1 / 1.])                 # nearest three: x[2] (d2 = 1), x[3] (1), x[0] / x[1] (5): lowest index first
    near = P.knn(x, torch.tensor([[0., 2, 0]]), 3)[1].tolist()
    assert near == [2, 3, 0]
    want = (4. * 1 + 8. * 1 + 1. * 0.2) / (1 + 1 + 0.2)
    assert abs(out.item() - want) < 1e-6
    out = P.knn_interpolate(feats, x, x[3:4].clone(), k=3)
    assert abs(out.item() - 8.0) < 1e-6


def test_scatter_max_first_maximum_and_mean():
    src = torch.tensor([[1., 5., 5., 2., 7., 7.]])
    idx = torch.tensor([0, 0, 0, 1, 1, 1])
    out, arg = P.scatter_max(src, idx, dim=-1, dim_size=3)
    assert out.tolist() == [[5., 7., 0.]] and arg[0, :2].tolist() == [1, 4]       # first of the equal maxima; empty group -> 0
    s = src.clone().requires_grad_(True)
    P.scatter_max(s, idx, dim=-1, dim_size=3)[0].sum().backward()
    assert s.grad.tolist() == [[0., 1., 0., 0., 1., 0.]]                            # gradient to the arg-max only
    mean = P.scatter_mean(torch.tensor([2., 4., 6.]), torch.tensor([0, 0, 2]), dim=-1, dim_size=3)
    assert mean.tolist() == [3., 0., 6.]
    assert P.global_max_pool(torch.tensor([[1., 2.], [3., 0.], [5., 5.]]), torch.tensor([0, 0, 1])).tolist() == [[3., 2.], [5., 5.]]
