"""GPU parity, geometry kernels vs the oracle (bit-exact index structures)."""
import numpy as np
import pytest
import torch

from oracle import primitives as P
from stratanet2_vegetation_coverage_maps_amd import hip_ops as ops
from stratanet2_vegetation_coverage_maps_amd.synthetic import make_batch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _pos(B, N, first=0):
    d = make_batch(B, N, first_plot=first)
    return d["xyz"], d["cloud"]


@pytest.mark.parametrize("B,N,M", [(1, 4096, 1024), (2, 2048, 256), (3, 1000, 250), (2, 200, 50), (2, 5000, 313), (40, 2500, 625),
                                   (1, 32768, 256), (2, 16384, 128), (2, 1024, 256), (1, 300, 300)])
def test_fps_matches_oracle_exactly(B, N, M):
    xyz, _ = _pos(B, N, first=7)
    start = torch.tensor([(13 * b + 5) % N for b in range(B)])
    ref = P.fps_batched(xyz.permute(0, 2, 1).contiguous(), M, start)
    idx, cs, ca = ops.fps(xyz.to(DEV), M, start.to(DEV, torch.int32))
    torch.cuda.synchronize()
    assert torch.equal(idx.cpu().long(), ref)
    g = torch.gather(xyz, 2, ref.unsqueeze(1).expand(-1, 3, -1))
    assert torch.equal(cs.cpu(), g)
    assert torch.equal(ca.cpu().view(B, M, 4)[..., :3], g.permute(0, 2, 1))


@pytest.mark.parametrize("B,N,M", [(2, 32768, 1024), (3, 5000, 1250), (2, 20000, 700), (1, 2049, 100)])
def test_bucketed_fps_equals_brute_force_and_oracle(B, N, M):
    """The bucketed kernel prunes work with an exact bound: indices must be identical to the brute-force kernel and to
    the oracle, including on exact distance ties (a quarter of the points are duplicated here)."""
    xyz, _ = _pos(B, N, first=21)
    q = N // 4
    xyz[:, :, N - q:] = xyz[:, :, :q]                                     # duplicates far apart in index
    start = torch.tensor([(977 * b + 13) % N for b in range(B)])
    dev = xyz.to(DEV)
    i_b, cs_b, _ = ops.fps(dev, M, start.to(DEV, torch.int32), bucketed=True)
    i_f, cs_f, _ = ops.fps(dev, M, start.to(DEV, torch.int32), bucketed=False)
    assert torch.equal(i_b, i_f) and torch.equal(cs_b, cs_f)
    ref = P.fps_batched(xyz.permute(0, 2, 1).contiguous(), M, start)
    assert torch.equal(i_b.cpu().long(), ref)
    # eight waves per plot (sn2_fps_waves: the pass that shares its CUs in a pipelined loop): the same samples, and the
    # same workspace for the ball query behind it
    i_8, cs_8, _, ws8 = ops.fps(dev, M, start.to(DEV, torch.int32), waves=8, return_ws=True)
    assert torch.equal(i_8, i_f) and torch.equal(cs_8, cs_f)
    i_4, cs_4, _ = ops.fps(dev, M, start.to(DEV, torch.int32), waves=4)       # four waves per plot (the parcel loop; > 16 384 points: 8)
    assert torch.equal(i_4, i_f) and torch.equal(cs_4, cs_f)
    i_1, cs_1, _ = ops.fps(dev, M, start.to(DEV, torch.int32), waves=1)       # one sample per arg-max round (round 1's kernel)
    assert torch.equal(i_1, i_f) and torch.equal(cs_1, cs_f)
    i_16, cs_16, _ = ops.fps(dev, M, start.to(DEV, torch.int32), waves=16)    # one workgroup of 16 waves per plot
    assert torch.equal(i_16, i_f) and torch.equal(cs_16, cs_f)
    # several workgroups per plot (P = 2, 4, 8 of 16 or 8 waves; what waves = 0 picks where the batch fits the chip): the same
    # samples, and no wait of the exchange ever gave up (control word 1 of the workspace counts the timeouts)
    for w in (34, 36, 40, 66, 68, 72):
        i_c, cs_c, ca_c, ws_c = ops.fps(dev, M, start.to(DEV, torch.int32), waves=w, return_ws=True)
        assert torch.equal(i_c, i_f) and torch.equal(cs_c, cs_f), f"waves={w}"
        assert torch.equal(ca_c.view(B, M, 4)[..., :3], cs_f.permute(0, 2, 1)), f"waves={w}"
        assert ws_c is None or int(ops.fps_ws_ctl(ws_c, B, N)[1]) == 0, f"waves={w}: exchange waits timed out"
    # the workspace is written in EVERY word (round 2 saw the pad word of the last plot's cell table keep the allocation's
    # old contents: no kernel reads it, but a buffer handed to other kernels should not hold undefined words): pre-fill two
    # buffers with different patterns -- whatever the kernel leaves untouched differs between them.  The order of duplicated
    # points inside a cell and the running distances depend on LDS atomics / the wave count, so compare the cell table and the
    # multiset of sorted positions, and require that no word kept its fill pattern.
    def _filled(pattern, waves):
        w = torch.full((ops.fps_ws_words(B, N),), pattern, dtype=torch.int32, device=DEV)
        out = (torch.empty(B, M, dtype=torch.int32, device=DEV), torch.empty(B, 3, M, device=DEV), torch.empty(B * M, 4, device=DEV), w)
        ops.fps(dev, M, start.to(DEV, torch.int32), out=out, waves=waves)
        return w
    if ops.fps_fills_ws(B, N, M):                     # (1 x 2049: no bucketed path, the workspace is not touched)
        wa, wb = _filled(0x01010101, 8), _filled(0x7E7E7E7E, 16)
        assert int((wa == 0x01010101).sum()) == 0 and int((wb == 0x7E7E7E7E).sum()) == 0
        assert torch.equal(wa[5 * B * N:5 * B * N + 4104 * B], wb[5 * B * N:5 * B * N + 4104 * B])    # cell starts, bounding boxes, pad words
        assert torch.equal(wa[:B * N].view(B, N).sort(1).values, wb[:B * N].view(B, N).sort(1).values)    # a permutation of 0..N-1
        order, rank = wa[:B * N].view(B, N).long(), ops.fps_ws_rank(wa, B, N).view(B, N).long()               # and its inverse
        assert torch.equal(torch.gather(rank, 1, order), torch.arange(N, device=DEV).expand(B, N))
    nbr8, cnt8, _ = ops.ball_query(dev, cs_8, 1.0, 64, fps_ws=ws8)
    nbr_f, cnt_f, _ = ops.ball_query(dev, cs_f, 1.0, 64)
    assert torch.equal(cnt8, cnt_f)
    live = torch.arange(64, device=DEV)[None, :] < cnt_f[:, None]
    assert torch.equal(nbr8[live], nbr_f[live])


@pytest.mark.parametrize("B,N,M", [(1, 131072, 1024), (2, 65536, 300), (1, 100000, 257)])
def test_bucketed_fps_on_dense_plots(B, N, M):
    """BASELINE config 5's plot size (128k points): two bucket slots per lane.  No brute-force kernel exists above 32768
    points, so the oracle is the checker; duplicates included."""
    xyz, _ = _pos(B, N, first=5)
    q = N // 8
    xyz[:, :, N - q:] = xyz[:, :, :q]
    start = torch.tensor([(977 * b + 13) % N for b in range(B)])
    i_b, cs_b, ca_b, ws = ops.fps(xyz.to(DEV), M, start.to(DEV, torch.int32), return_ws=True)
    assert ws is not None
    ref = P.fps_batched(xyz.permute(0, 2, 1).contiguous(), M, start)
    assert torch.equal(i_b.cpu().long(), ref)
    for w in (16, 36, 72):                       # one workgroup per plot; 4 x 16 and 8 x 8 waves per plot
        i_w, cs_w, _, ws_w = ops.fps(xyz.to(DEV), M, start.to(DEV, torch.int32), waves=w, return_ws=True)
        assert torch.equal(i_w, i_b) and torch.equal(cs_w, cs_b) and int(ops.fps_ws_ctl(ws_w, B, N)[1]) == 0, f"waves={w}"
    # the ball query and the 3-NN table over the same workspace, against their full scans
    nbr_g, cnt_g, _ = ops.ball_query(xyz.to(DEV), cs_b, 1.0, 2000, fps_ws=ws)
    nbr_f, cnt_f, _ = ops.ball_query(xyz.to(DEV), cs_b, 1.0, 2000)
    mask = torch.arange(nbr_f.shape[1], device=DEV).unsqueeze(0) < cnt_f.unsqueeze(1)
    assert torch.equal(cnt_g, cnt_f) and torch.equal(nbr_g[mask], nbr_f[mask])
    a_i, a_w = ops.three_nn(cs_b, xyz.to(DEV), 3, dst_fps_ws=ws)
    b_i, b_w = ops.three_nn(cs_b, xyz.to(DEV), 3, grid=False)
    assert torch.equal(a_i, b_i) and torch.equal(a_w, b_w)


def test_multi_workgroup_fps_gives_up_safely_and_is_repaired():
    """The workgroups of the multi-workgroup FPS wait for their peers, and nothing guarantees that they are resident together.
    Provoked here with a wait limit of ONE sweep (sn2_debug_fps_spin_limit): every exchange gives up almost at once.  The
    call must return (no hang), count the give-ups (control word 1 of the workspace, and the process-wide status word that
    `ops.fps_gave_up` reads and warns about), write nothing out of range, and still deliver the reference's samples: the
    repair launch behind every multi-workgroup pass (the single-workgroup kernel) samples the plots again."""
    import warnings
    from stratanet2_vegetation_coverage_maps_amd import _lib
    B, N, M = 4, 32768, 512
    xyz, _ = _pos(B, N, first=3)
    start = torch.tensor([(977 * b + 13) % N for b in range(B)])
    dev = xyz.to(DEV)
    ref = P.fps_batched(xyz.permute(0, 2, 1).contiguous(), M, start)
    before = ops.fps_gave_up(DEV, warn=False)
    lib = _lib.load()
    try:
        assert lib.sn2_debug_fps_spin_limit(1) == 0
        for w in (72, 36, 0):
            out = (torch.full((B, M), -7, dtype=torch.int32, device=DEV), torch.full((B, 3, M), float("nan"), device=DEV),
                   torch.full((B * M, 4), float("nan"), device=DEV), torch.empty(ops.fps_ws_words(B, N), dtype=torch.int32, device=DEV))
            idx, cs, ca, ws = ops.fps(dev, M, start.to(DEV, torch.int32), out=out, waves=w, return_ws=True)
            torch.cuda.synchronize()
            assert int(ops.fps_ws_ctl(ws, B, N)[1]) > 0, f"waves={w}: the one-sweep limit did not make a wait give up"
            assert torch.equal(idx.cpu().long(), ref), f"waves={w}: the repair launch did not restore the samples"
            g = torch.gather(xyz, 2, ref.unsqueeze(1).expand(-1, 3, -1))
            assert torch.equal(cs.cpu(), g) and torch.equal(ca.cpu().view(B, M, 4)[..., :3], g.permute(0, 2, 1))
            # the tables the ball query walks are intact as well
            nbr_g, cnt_g, _ = ops.ball_query(dev, cs, 1.0, 64, fps_ws=ws)
            nbr_f, cnt_f, _ = ops.ball_query(dev, cs, 1.0, 64)
            live = torch.arange(64, device=DEV)[None, :] < cnt_f[:, None]
            assert torch.equal(cnt_g, cnt_f) and torch.equal(nbr_g[live], nbr_f[live])
    finally:
        lib.sn2_debug_fps_spin_limit(0)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        after = ops.fps_gave_up(DEV)
    assert after > before and any(issubclass(r.category, ops.StrataHipWarning) for r in rec)
    with warnings.catch_warnings(record=True) as rec:              # reported once per growth
        warnings.simplefilter("always")
        assert ops.fps_gave_up(DEV) == after and not rec
    # and with the default limit nothing gives up
    idx, _, _, ws = ops.fps(dev, M, start.to(DEV, torch.int32), waves=72, return_ws=True)
    assert torch.equal(idx.cpu().long(), ref) and int(ops.fps_ws_ctl(ws, B, N)[1]) == 0 and ops.fps_gave_up(DEV) == after


def test_fps_with_duplicate_points_and_default_start():
    """sample_cloud pads small plots by sampling with replacement (loader.py:238-244) => exact ties."""
    xyz, _ = _pos(1, 500)
    xyz = torch.cat([xyz, xyz[:, :, :300]], dim=2).contiguous()           # 800 points, 300 duplicates
    ref = P.fps_batched(xyz.permute(0, 2, 1).contiguous(), 700, torch.zeros(1, dtype=torch.long))
    idx, _, _ = ops.fps(xyz.to(DEV), 700, None)
    assert torch.equal(idx.cpu().long(), ref)


def _oracle_lists(xyz, cpos, r, cap):
    B, _, N = xyz.shape
    M = cpos.shape[2]
    pos = xyz.permute(0, 2, 1).reshape(B * N, 3)
    cp = cpos.permute(0, 2, 1).reshape(B * M, 3)
    bx = torch.arange(B).repeat_interleave(N)
    by = torch.arange(B).repeat_interleave(M)
    row, col = P.radius(pos, cp, r, bx, by, max_num_neighbors=cap)
    return row, col - bx[col] * N


@pytest.mark.parametrize("B,N,M,r,cap", [(1, 4096, 1024, 2 ** 0.5, 2000), (2, 2048, 256, 1.0, 2000),
                                         (2, 1000, 100, 3.0, 64), (1, 777, 33, 2.0, 2000), (2, 256, 64, 8 ** 0.5, 2000)])
def test_ball_query_matches_oracle_exactly(B, N, M, r, cap):
    xyz, _ = _pos(B, N, first=3)
    fidx = P.fps_batched(xyz.permute(0, 2, 1).contiguous(), M, torch.zeros(B, dtype=torch.long))
    cpos = torch.gather(xyz, 2, fidx.unsqueeze(1).expand(-1, 3, -1)).contiguous()
    row, col = _oracle_lists(xyz, cpos, r, cap)
    nbr, cnt, total = ops.ball_query(xyz.to(DEV), cpos.to(DEV), r, cap)
    torch.cuda.synchronize()
    cnt_ref = torch.bincount(row, minlength=B * M)
    assert torch.equal(cnt.cpu().long(), cnt_ref)
    assert int(total.item()) == int(cnt_ref.sum())
    nbr, cnt = nbr.cpu(), cnt.cpu()
    mask = torch.arange(nbr.shape[1]).unsqueeze(0) < cnt.unsqueeze(1)
    assert torch.equal(nbr[mask].long(), col)          # row-major over (centroid, slot) == oracle's (row, ascending col)
    if cap < 2000:
        assert int(cnt.max()) == cap                   # the cap bites in this case: first `cap` in ascending index


@pytest.mark.parametrize("B,S,T,k", [(1, 1024, 4096, 3), (2, 256, 1024, 3), (2, 64, 1000, 3), (3, 1, 256, 1),
                                     (1, 2, 100, 3), (2, 1500, 700, 3)])
def test_three_nn_matches_oracle(B, S, T, k):
    xyz, _ = _pos(B, max(S, T), first=11)
    src = xyz[:, :, :S].contiguous() if S > 1 else torch.zeros(B, 3, 1)
    dst = xyz[:, :, -T:].contiguous()
    ps = src.permute(0, 2, 1).reshape(B * S, 3)
    pd = dst.permute(0, 2, 1).reshape(B * T, 3)
    bs, bd = torch.arange(B).repeat_interleave(S), torch.arange(B).repeat_interleave(T)
    yi, xi = P.knn(ps, pd, k, bs, bd)
    kk = min(k, S)
    ref_idx = (xi - bs[xi] * S).view(B * T, kk)
    ref_w = 1.0 / torch.clamp(P.canonical_d2(ps[xi], pd[yi]), min=1e-16).view(B * T, kk)
    idx, w = ops.three_nn(src.to(DEV), dst.to(DEV), k)
    idx, w = idx.cpu().long(), w.cpu()
    assert torch.equal(idx[:, :kk], ref_idx)
    assert torch.equal(w[:, :kk], ref_w)               # same canonical d2, IEEE division
    if kk < 3:
        assert torch.all(w[:, kk:] == 0) and torch.equal(idx[:, kk:], idx[:, :1].expand(-1, 3 - kk))


@pytest.mark.parametrize("B,S,T,k", [(2, 1024, 32768, 3), (4, 256, 4096, 3), (2, 2500, 10000, 3), (4, 128, 2051, 2), (40, 625, 2500, 3),
                                     (2, 625, 2500, 1)])
def test_grid_three_nn_equals_full_scan(B, S, T, k):
    """The grid walk returns the full scan's table bit for bit: same indices (lowest index on ties), same weights.
    Sources = an FPS-like subset of the targets plus duplicates; some targets lie outside the sources' bounding box."""
    xyz, _ = _pos(B, T, first=77)
    g = torch.Generator().manual_seed(S)
    pick = torch.stack([torch.randperm(T, generator=g)[:S] for _ in range(B)])
    src = torch.gather(xyz, 2, pick.unsqueeze(1).expand(-1, 3, -1)).contiguous()
    src[:, :, 5] = src[:, :, 9]                       # duplicated sources: exact ties, lowest index must win
    src[:, :, S // 2] = src[:, :, S // 2 + 1]
    dst = xyz.clone()
    dst[:, :2, :50] *= 1.7                            # targets outside the sources' x,y range
    dst[:, 2, 50:100] += 30.0                         # and far above them (many rings)
    _, _, _, fws = ops.fps(dst.to(DEV), 64, None, return_ws=True)      # the targets' Morton order
    a_i, a_w = ops.three_nn(src.to(DEV), dst.to(DEV), k, dst_fps_ws=fws)
    b_i, b_w = ops.three_nn(src.to(DEV), dst.to(DEV), k, grid=False)
    c_i, c_w = ops.three_nn(src.to(DEV), dst.to(DEV), k)          # targets sorted by the source grid's cells in the call
    torch.cuda.synchronize()
    assert torch.equal(a_i, b_i) and torch.equal(a_w, b_w)
    assert torch.equal(c_i, b_i) and torch.equal(c_w, b_w)


@pytest.mark.parametrize("B,N,M,r,cap", [(2, 32768, 1024, 1.0, 2000), (2, 4096, 512, 2 ** 0.5, 2000), (1, 8192, 64, 4.0, 300),
                                         (2, 5000, 100, 2.0, 2000),
                                         # the parcel loop's shape (reference defaults): ~500 candidates and ~150 hits per centroid
                                         # -> the bitmap-first path; with a cap that bites; 40 plots -> the 256-thread sorts
                                         (3, 10000, 2500, 2 ** 0.5, 2000), (2, 10000, 300, 2 ** 0.5, 64), (40, 4100, 128, 2.0, 2000)])
def test_grid_ball_query_equals_full_scan_and_oracle(B, N, M, r, cap):
    """The cell-list ball query (sources sorted by FPS) returns exactly the lists of the full scan: same members, ascending
    source index, same cap behaviour (the r = 4 m case overflows the in-LDS list and takes the fallback)."""
    xyz, _ = _pos(B, N, first=31)
    dev = xyz.to(DEV)
    idx, cs, ca, ws = ops.fps(dev, M, None, return_ws=True)
    assert ws is not None
    nbr_g, cnt_g, tot_g = ops.ball_query(dev, cs, r, cap, fps_ws=ws)
    nbr_f, cnt_f, tot_f = ops.ball_query(dev, cs, r, cap)
    torch.cuda.synchronize()
    assert torch.equal(cnt_g, cnt_f) and int(tot_g) == int(tot_f)
    mask = torch.arange(nbr_f.shape[1], device=DEV).unsqueeze(0) < cnt_f.unsqueeze(1)
    assert torch.equal(nbr_g[mask], nbr_f[mask])
    if N <= 8192 or (B * M <= 7500 and B <= 3):
        row, col = _oracle_lists(xyz, cs.cpu(), r, cap)
        assert torch.equal(nbr_g.cpu()[mask.cpu()].long(), col)


def test_pack_rows():
    xyz, cloud = _pos(2, 777)
    rows = ops.pack_rows(cloud.to(DEV), xyz.to(DEV)).cpu().view(2, 777, 12)
    assert torch.equal(rows[..., :8], cloud[:, 2:].permute(0, 2, 1))
    assert torch.equal(rows[..., 8:11], xyz.permute(0, 2, 1))
    assert torch.all(rows[..., 11] == 0)


@pytest.mark.parametrize("n,r,seed", [(3000, 1.5, 0), (777, 0.7, 1), (5000, 3.0, 2)])
def test_znorm_matches_reference_rule(n, r, seed):
    """Local-minimum z-normalisation of a raw plot vs the restated sklearn rule (fp64 reduced distance, inclusive),
    cross-checked against scipy's cKDTree; points exactly at distance r included (a lattice with spacing r/2)."""
    from oracle import prepare
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(seed)
    xy = rng.uniform(-10, 10, (n, 2)).astype(np.float32) + np.float32(650000.0 if seed == 1 else 0.0)
    k = int(np.sqrt(n // 4))
    lat = (np.stack(np.meshgrid(np.arange(k), np.arange(k)), -1).reshape(-1, 2) * (r / 2)).astype(np.float32)
    xy[: len(lat)] = lat + xy[0]
    z = rng.uniform(0, 20, n).astype(np.float32)
    cloud = np.concatenate([xy.T, z[None], rng.random((7, n)).astype(np.float32)], 0)
    want = prepare.normalize_z_with_minz_in_a_radius(cloud, r)[2]
    zmin_ref = prepare.radius_neighbors_min(xy, z, r)
    tree = cKDTree(xy.astype(np.float64))
    nb = tree.query_ball_point(xy.astype(np.float64), r)
    zmin_kd = np.array([z[j].min() for j in nb], dtype=np.float32)
    assert (zmin_kd == zmin_ref).mean() > 0.999          # the kd-tree computes true distances: boundary ties may differ
    zmin, zout = ops.znorm(torch.from_numpy(cloud[:3].copy()).to(DEV), r)
    assert np.array_equal(zmin.cpu().numpy(), zmin_ref)
    assert np.array_equal(zout.cpu().numpy(), want)


def test_fps_workspace_is_only_handed_out_when_filled():
    """sn2_fps fills its workspace on the bucketed path only (N > 2048, M > 16, B*N % 4 == 0): for other sizes the wrapper
    must return None, or the cell-list walks of ball_query / three_nn would read garbage."""
    for B, N, M, expect in ((1, 4096, 8, False), (1, 4099, 64, False), (2, 4098, 64, True), (1, 2048, 64, False)):
        xyz, _ = _pos(B, N, first=3)
        idx, cs, ca, ws = ops.fps(xyz.to(DEV), M, None, return_ws=True)
        assert (ws is not None) == expect and ops.fps_fills_ws(B, N, M) == expect
        nbr, cnt, _ = ops.ball_query(xyz.to(DEV), cs, 1.5, 200, fps_ws=ws)
        nbr_f, cnt_f, _ = ops.ball_query(xyz.to(DEV), cs, 1.5, 200)
        assert torch.equal(cnt, cnt_f)
        ref = P.fps_batched(xyz.permute(0, 2, 1).contiguous(), M, torch.zeros(B, dtype=torch.long))
        assert torch.equal(idx.cpu().long(), ref)


@pytest.mark.parametrize("B,M", [(3, 1024), (2, 2500), (4, 256), (1, 10000), (2, 77), (40, 700)])      # 40 plots: 256 threads per plot
def test_sa_work_items_order(B, M):
    """sn2_sa_order (the work items of the set-abstraction passes, ranked per plot by neighbour count descending, index
    ascending -- a bitonic sort in LDS) against the rule evaluated on the host (include/strata_hip.h): SOLO centroids (more
    than 64 neighbours) first, one per position in four copies with the flag; QUADs (9..64) four per position; then the
    packed table: OCTs (5..8) eight per 16-int position, every id twice with its flag, HEXes (0..4) sixteen per position;
    -1 wherever nothing is; the trailer holds the largest position counts of any plot."""
    g = torch.Generator().manual_seed(B * 1000 + M)
    cnt = torch.randint(0, 40, (B * M,), generator=g, dtype=torch.int32)         # many ties
    heavy = torch.rand(B * M, generator=g) < 0.1
    cnt[heavy] = torch.randint(40, 2001, (int(heavy.sum()),), generator=g, dtype=torch.int32)
    got = ops.sa_order(cnt.to(DEV), B, M).cpu()
    SOLO_MIN, QUAD_MIN, OCT_MIN, FLAG, OCT = 64, 8, 4, 1 << 30, 1 << 29
    PK = M // 8 + 2
    assert got.numel() == ops.sa_order_len(B, M) == 4 * B * M + 16 * B * PK + 8
    want = torch.full((got.numel(),), -1, dtype=torch.int32)
    offp, offt = 4 * B * M, 4 * B * M + 16 * B * PK
    na = nb = 0
    for b in range(B):
        c = cnt[b * M:(b + 1) * M]
        order = sorted(range(M), key=lambda i: (-int(c[i]), i))
        nsolo = int((c > SOLO_MIN).sum())
        r_oct = nsolo + int(((c > QUAD_MIN) & (c <= SOLO_MIN)).sum())
        r_hex = r_oct + int(((c > OCT_MIN) & (c <= QUAD_MIN)).sum())
        n_oct_items = (r_hex - r_oct + 7) // 8
        for r, i in enumerate(order):
            idv = b * M + i
            if r < nsolo:
                want[4 * (r * B + b):4 * (r * B + b) + 4] = idv | FLAG
            elif r < r_oct:
                rl = r - nsolo
                want[4 * ((nsolo + (rl >> 2)) * B + b) + (rl & 3)] = idv
            elif r < r_hex:
                rl = r - r_oct
                o = offp + 16 * ((rl >> 3) * B + b) + 2 * (rl & 7)
                want[o:o + 2] = idv | OCT
            else:
                rl = r - r_hex
                want[offp + 16 * ((n_oct_items + (rl >> 4)) * B + b) + (rl & 15)] = idv
        na = max(na, nsolo + (r_oct - nsolo + 3) // 4)
        nb = max(nb, n_oct_items + (M - r_hex + 15) // 16)
        assert n_oct_items + (M - r_hex + 15) // 16 <= PK
    want[offt], want[offt + 1] = na, nb
    assert torch.equal(got[:offt + 2], want[:offt + 2])


@pytest.mark.parametrize("B,R,S,with_pos", [(2, 5000, 37, True), (3, 2049, 100, False), (1, 70000, 1024, True), (2, 300, 1, False),
                                             (1, 32768, 8192, True)])       # S = 8192: the limit (64 KB of dynamic LDS)
def test_inverted_index_and_its_chunk_table(B, R, S, with_pos):
    """sn2_interp_index: every (row, slot) of the 3-NN table is on its source's list exactly once with the normalised weight,
    the item table is a permutation of the plot's sources, and the chunk table cuts every list into consecutive pieces of at
    most 63 entries (unused slots: length 0) -- what fp_bwd_src_chunk_kernel walks (csrc/fp.hip).  Lists are made very
    uneven on purpose (a few sources own most rows, some own none)."""
    g = torch.Generator().manual_seed(B * 1000 + S)
    # skewed source choice: squares of uniforms pile up on the low ids; ids >= S - S // 5 are never chosen when S > 4
    top = S - S // 5 if S > 4 else S
    idx = (torch.rand(B * R, 3, generator=g) ** 3 * top).long().clamp_(0, S - 1).int()
    w = torch.rand(B * R, 3, generator=g) + 0.1
    w[::7, 2] = 0.0                                     # a zero weight in slot 1/2 = "no third neighbour": not listed
    pos = torch.rand(B * S, 4, generator=g) if with_pos else None
    ws = ops.interp_index((idx.cuda(), w.cuda()), B, R, S, src_pos=None if pos is None else pos.cuda())
    torch.cuda.synchronize()
    words = ws.cpu().view(torch.int32).numpy()
    fl = ws.cpu().numpy()
    SL = (R + 2047) // 2048
    o_off = B * SL * S
    o_cnt = o_off + B * S
    o_row = o_cnt + B * S
    o_w = o_row + 3 * B * R
    o_items = (o_w + 3 * B * R + 3) // 4 * 4             # 16-byte aligned (the tensor's base is)
    CM = ops.interp_chunks(R, S)
    o_chunks = o_items + 4 * B * S
    assert o_chunks + 4 * B * CM <= ops.interp_ws_words(B, R, S)
    off, cnt = words[o_off:o_off + B * S], words[o_cnt:o_cnt + B * S]
    inv_row, inv_w = words[o_row:o_row + 3 * B * R], fl[o_w:o_w + 3 * B * R]
    items = words[o_items:o_items + 4 * B * S].reshape(B, S, 4)
    chunks = words[o_chunks:o_chunks + 4 * B * CM].reshape(B, CM, 4)
    idx_n, w_n = idx.numpy(), w.numpy()
    wn = w_n / w_n.sum(1, keepdims=True)
    for b in range(B):
        assert sorted(items[b, :, 0].tolist()) == list(range(b * S, (b + 1) * S))
        first = 0
        for k in range(S):
            sid, o, n, f = items[b, k]
            assert o == off[sid] and n == cnt[sid] and f == first
            nch = (n + 62) // 63
            for c in range(nch):
                assert chunks[b, first + c].tolist() == [sid, o + 63 * c, min(63, n - 63 * c), b]
            first += nch
        assert not chunks[b, first:].any()
        rows = slice(b * R, (b + 1) * R)
        listed = (np.arange(3)[None, :] == 0) | (w_n[rows] != 0)
        want = {}
        for r, j in zip(*np.nonzero(listed)):
            want.setdefault(b * S + int(idx_n[rows][r, j]), []).append((int(r), float(wn[rows][r, j])))
        for s in range(b * S, (b + 1) * S):
            got = sorted(zip(inv_row[off[s]:off[s] + cnt[s]].tolist(), inv_w[off[s]:off[s] + cnt[s]].tolist()))
            exp = sorted(want.get(s, []))
            assert [r for r, _ in got] == [r for r, _ in exp]
            np.testing.assert_allclose([x for _, x in got], [x for _, x in exp], rtol=2e-6)
    assert S == 1 or (cnt.max() > 4 * 63 and (cnt == 0).any())
