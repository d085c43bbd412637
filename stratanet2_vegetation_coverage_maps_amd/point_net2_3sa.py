"""The "3sa-arch" variant BASELINE.json's configuration 2 names -- three ball-query set-abstraction levels (npoint
1024 / 256 / 64, radius 1 / 2 / 4 m) before the global one -- on the same kernels ("make the stack generic so a
third ball-query level (npoint 64, r = 4.0, MLP[35,64]) can precede the global pool (then FP4 k=1, FP3/2/1 k=3)").

THE REFERENCE HAS NO SUCH MODEL (`model/point_net2.py:84-96` builds two ball-query levels + the global one): this class is
for throughput measurements next to the reference architecture; its parity partner is the oracle's generalisation
(`oracle/network.py::forward_3sa`), not the reference.

    SA1 [11,16,16] -> SA2 [19,32] -> SA3 [35,64] (ball query, ratio3, r3) -> SA4 global [67,64] -> max
    FP4 k=1 [64+64,64] -> FP3 k=3 [64+32,64] -> FP2 k=3 [64+16,34] -> FP1 k=3 [34+8,34] -> head (as the reference)
"""
from collections import OrderedDict

import torch
from torch import nn

from . import hip_ops as ops
from ._lib import BN_FROZEN_KEEP, STAT_SLOTS
from . import point_net2 as _p2
from .point_net2 import (F32, I32, I64, MLP, FPModule, GlobalSAModule, PointNet2, SAModule, _blocks_of, _Saved)


class PointNet2ThreeSA(PointNet2):
    def __init__(self, args):
        nn.Module.__init__(self)
        self.cuda_device = args.cuda
        self.subsample_size = args.subsample_size
        self.n_class = args.n_class
        self.drop = args.drop
        self.n_input_feats = args.n_input_feats - 2
        self.set_patience_attributes(args)
        self.log_embeddings = args.log_embeddings
        self.last_G_tensor = None
        self._last_flat_grad = None
        self._last_cloud_dev = None
        if self.n_class != 4 or self.n_input_feats != 8:
            raise ValueError("the HIP kernels cover n_class=4, 10 input features")
        self.sa1_module = SAModule(args.ratio1, args.r1, MLP([11, 16, 16]))
        self.sa2_module = SAModule(args.ratio2, args.r2, MLP([19, 32]))
        self.sa3_module = SAModule(getattr(args, "ratio3", 0.25), getattr(args, "r3", 4.0), MLP([35, 64]))
        self.sa4_module = GlobalSAModule(MLP([67, 64]))
        self.fp4_module = FPModule(1, MLP([64 + 64, 64]))
        self.fp3_module = FPModule(3, MLP([64 + 32, 64]))
        self.fp2_module = FPModule(3, MLP([64 + 16, 34]))
        self.fp1_module = FPModule(3, MLP([34 + 8, 34]))
        self.lin1 = nn.Linear(34, 16)
        self.lin2 = nn.Linear(16, self.n_class + 1)
        self.lin2.bias = nn.Parameter(torch.tensor([0.733, 0.266, 0.235, 0.358, 0.500]))
        self.softmax = nn.Softmax(dim=1)
        self.sigmoid = nn.Sigmoid()
        if self.cuda_device is not None:
            self.cuda(self.cuda_device)

    N_FPS = 3
    executor = False                             # (the one-call executor covers the reference architecture)
    geometry_pair_takes_group_cloud = False
    BF16_BLOCKS = ("sa1_module.conv.local_nn", "sa2_module.conv.local_nn", "sa3_module.conv.local_nn", "sa4_module.nn",
                   "fp4_module.nn", "fp3_module.nn", "fp2_module.nn")

    def _sizes3(self, N):
        M1 = ops.fps_num_samples(N, self.sa1_module.ratio)
        M2 = ops.fps_num_samples(M1, self.sa2_module.ratio)
        M3 = ops.fps_num_samples(M2, self.sa3_module.ratio)
        return M1, M2, M3

    def _sizes(self, N):
        return self._sizes3(N)[:2]

    # ------------------------------------------------------------------------------------------ geometry
    def alloc_geometry_pair(self, B, N, device=None, group=2):
        """As `PointNet2.alloc_geometry_pair` with the third ball-query level: buffers of one geometry pass over `group` batches
        and the per-batch views the feature passes read."""
        dev = torch.device(device if device is not None else self.lin1.weight.device)
        M1, M2, M3 = self._sizes3(N)
        gp = self.alloc_geometry(group * B, N, dev)
        e = lambda *shape, dt=F32: torch.empty(*shape, dtype=dt, device=dev)          # noqa: E731
        halves = []
        for h in range(group):
            g = _Saved()
            g.B, g.N, g.M1, g.M2, g.M3 = B, N, M1, M2, M3
            pl = slice(h * B, (h + 1) * B)
            for lvl, M in ((1, M1), (2, M2), (3, M3)):
                rows = slice(h * B * M, (h + 1) * B * M)
                setattr(g, f"idx{lvl}", getattr(gp, f"idx{lvl}")[pl])
                setattr(g, f"pos{lvl}_soa", getattr(gp, f"pos{lvl}_soa")[pl])
                setattr(g, f"pos{lvl}_aos", getattr(gp, f"pos{lvl}_aos")[rows])
                setattr(g, f"nbr{lvl}", getattr(gp, f"nbr{lvl}")[rows])
                setattr(g, f"cnt{lvl}", getattr(gp, f"cnt{lvl}")[rows])
                setattr(g, f"ord{lvl}", e(ops.sa_order_len(B, M), dt=I32))
                setattr(g, f"ws{lvl}", None)
            for name, R in (("knn4", M3), ("knn3", M2), ("knn2", M1), ("knn1", N)):
                rows = slice(h * B * R, (h + 1) * B * R)
                t = getattr(gp, name)
                setattr(g, name, (t[0][rows], t[1][rows]))
            g.totals = torch.zeros(3, dtype=I64, device=dev)
            g.tot1, g.tot2, g.tot3 = g.totals[0:1], g.totals[1:2], g.totals[2:3]
            g.inv4, g.inv3, g.inv2, g.inv1 = (e(ops.interp_ws_words(B, R, S)) for R, S in ((M3, 1), (M2, M3), (M1, M2), (N, M1)))
            g.nn_ws = None
            g.rank1 = gp.rank1[h * B * N:(h + 1) * B * N] if (gp.rank1 is not None and self._fp1_source_side(B * N)) else None
            self._alloc_input_only(g, B, N, dev)
            g.ready = None
            halves.append(g)
        return gp, tuple(halves)

    def _geometry_pair(self, xyz2, fps_start2, gp, halves, clouds=None):
        """`_geometry` for two batches at once: xyz2 (2B,3,N), fps_start2 (3,2B); FPS, ball queries and 3-NN tables on the 2B
        plots in one launch each (into `gp`), the per-batch products (message totals, work items, inverted indices) per half."""
        B2, _, N = xyz2.shape
        B = B2 // len(halves)
        M1, M2, M3 = self._sizes3(N)
        if (gp.B, gp.N) != (B2, N):
            raise ValueError("geometry buffers do not match this batch pair")
        cap = _p2.MAX_NEIGHBORS
        src = xyz2
        for lvl, (M, mod) in enumerate(((M1, self.sa1_module), (M2, self.sa2_module), (M3, self.sa3_module)), 1):
            ws, cs = getattr(gp, f"ws{lvl}"), getattr(gp, f"pos{lvl}_soa")
            ops.fps(src, M, fps_start2[lvl - 1], out=(getattr(gp, f"idx{lvl}"), cs, getattr(gp, f"pos{lvl}_aos"), ws),
                    waves=self.fps_waves_shared if lvl == 1 else 0)
            ops.ball_query(src, cs, mod.r, cap, getattr(gp, f"tot{lvl}"), fps_ws=ws,
                           out=(getattr(gp, f"nbr{lvl}"), getattr(gp, f"cnt{lvl}")))
            src = cs
        ops.three_nn(gp.posg, gp.pos3_soa, 1, out=gp.knn4)
        ops.three_nn(gp.pos3_soa, gp.pos2_soa, 3, out=gp.knn3, ws=gp.nn_ws[0])
        ops.three_nn(gp.pos2_soa, gp.pos1_soa, 3, out=gp.knn2, ws=gp.nn_ws[1])
        ops.three_nn(gp.pos1_soa, xyz2, 3, out=gp.knn1, ws=gp.nn_ws[2])
        for h, g in enumerate(halves):
            g.xyz = xyz2[h * B:(h + 1) * B]
            g.has_rows0 = False
            if clouds is not None:                      # the input-only pieces of the feature pass (PointNet2._input_only)
                self._input_only(g, clouds[h], g.xyz)
            for lvl, M in ((1, M1), (2, M2), (3, M3)):
                ops.count_sum(getattr(g, f"cnt{lvl}"), getattr(g, f"tot{lvl}"))
                ops.sa_order(getattr(g, f"cnt{lvl}"), B, M, out=getattr(g, f"ord{lvl}"))
            ops.interp_index(g.knn4, B, M3, 1, out=g.inv4)
            ops.interp_index(g.knn3, B, M2, M3, out=g.inv3)
            ops.interp_index(g.knn2, B, M1, M2, out=g.inv2)
            ops.interp_index(g.knn1, B, N, M1, out=g.inv1, src_pos=g.pos1_aos, row_perm=g.rank1)
        return halves

    def alloc_geometry(self, B, N, device=None):
        dev = torch.device(device if device is not None else self.lin1.weight.device)
        M1, M2, M3 = self._sizes3(N)
        e = lambda *shape, dt=F32: torch.empty(*shape, dtype=dt, device=dev)          # noqa: E731
        g = _Saved()
        g.B, g.N, g.M1, g.M2, g.M3 = B, N, M1, M2, M3
        cap = _p2.MAX_NEIGHBORS
        for lvl, (S, M) in enumerate(((N, M1), (M1, M2), (M2, M3)), 1):
            setattr(g, f"idx{lvl}", e(B, M, dt=I32))
            setattr(g, f"pos{lvl}_soa", e(B, 3, M))
            setattr(g, f"pos{lvl}_aos", e(B * M, 4))
            setattr(g, f"ws{lvl}", e(ops.fps_ws_words(B, S), dt=I32) if ops.fps_fills_ws(B, S, M) else None)
            setattr(g, f"nbr{lvl}", e(B * M, min(cap, S), dt=I32))
            setattr(g, f"cnt{lvl}", e(B * M, dt=I32))
            setattr(g, f"ord{lvl}", e(ops.sa_order_len(B, M), dt=I32))
        g.totals = torch.zeros(3, dtype=I64, device=dev)
        g.tot1, g.tot2, g.tot3 = g.totals[0:1], g.totals[1:2], g.totals[2:3]
        g.posg = torch.zeros(B, 3, 1, dtype=F32, device=dev)
        g.knn4, g.knn3 = (e(B * M3, 3, dt=I32), e(B * M3, 3)), (e(B * M2, 3, dt=I32), e(B * M2, 3))
        g.knn2, g.knn1 = (e(B * M1, 3, dt=I32), e(B * M1, 3)), (e(B * N, 3, dt=I32), e(B * N, 3))
        g.inv4, g.inv3, g.inv2, g.inv1 = (e(ops.interp_ws_words(B, R, S)) for R, S in ((M3, 1), (M2, M3), (M1, M2), (N, M1)))
        g.nn_ws = tuple(e(ops.three_nn_ws_words(B, S, T), dt=I32) if ops.three_nn_uses_grid(S, T) else None
                        for S, T in ((M3, M2), (M2, M1), (M1, N)))
        g.rank1 = ops.fps_ws_rank(g.ws1, B, N) if (self.fp1_morton_rows and g.ws1 is not None and self._fp1_source_side(B * N)) else None       # as PointNet2.alloc_geometry
        self._alloc_input_only(g, B, N, dev)
        g.ready = None
        return g

    def _geometry(self, xyz, fps_start, out=None, fork=None, shared=False, defer_join=False, inverted=True, cloud=None):
        """As `PointNet2._geometry` with the third ball-query level; one stream (`fork` / `defer_join` are accepted and
        ignored: nothing is forked, so there is nothing to join), `shared` = the level-1 FPS with `fps_waves_shared` waves
        per plot, `inverted=False` = skip the inverted 3-NN tables (an eval-mode forward never gathers through them;
        `g.has_inverted` records it and a later training-mode forward builds them)."""
        dev = xyz.device
        B, _, N = xyz.shape
        M1, M2, M3 = self._sizes3(N)
        g = out if out is not None else self.alloc_geometry(B, N, dev)
        if (g.B, g.N, g.M1, g.M2, g.M3) != (B, N, M1, M2, M3):
            raise ValueError("geometry buffers do not match this batch")
        g.xyz = xyz
        g.has_rows0 = False
        if cloud is not None:
            self._input_only(g, cloud, xyz)
        cap = _p2.MAX_NEIGHBORS
        src = xyz
        for lvl, (M, mod) in enumerate(((M1, self.sa1_module), (M2, self.sa2_module), (M3, self.sa3_module)), 1):
            ws = getattr(g, f"ws{lvl}")
            cs = getattr(g, f"pos{lvl}_soa")
            ops.fps(src, M, fps_start[lvl - 1], out=(getattr(g, f"idx{lvl}"), cs, getattr(g, f"pos{lvl}_aos"), ws),
                    waves=(self.fps_waves_many if B > 32 else self.fps_waves_shared) if (shared and lvl == 1) else 0)
            ops.ball_query(src, cs, mod.r, cap, getattr(g, f"tot{lvl}"), fps_ws=ws,
                           out=(getattr(g, f"nbr{lvl}"), getattr(g, f"cnt{lvl}")))
            ops.sa_order(getattr(g, f"cnt{lvl}"), B, M, out=getattr(g, f"ord{lvl}"))
            src = cs
        ops.three_nn(g.posg, g.pos3_soa, 1, out=g.knn4)
        ops.three_nn(g.pos3_soa, g.pos2_soa, 3, out=g.knn3, ws=g.nn_ws[0])
        ops.three_nn(g.pos2_soa, g.pos1_soa, 3, out=g.knn2, ws=g.nn_ws[1])
        ops.three_nn(g.pos1_soa, xyz, 3, out=g.knn1, ws=g.nn_ws[2])
        if inverted:
            self._inverted_tables(g)
        g.has_inverted = bool(inverted)
        g._join = None
        return g

    @staticmethod
    def _inverted_tables(g):
        B, N, M1, M2, M3 = g.B, g.N, g.M1, g.M2, g.M3
        ops.interp_index(g.knn4, B, M3, 1, out=g.inv4)
        ops.interp_index(g.knn3, B, M2, M3, out=g.inv3)
        ops.interp_index(g.knn2, B, M1, M2, out=g.inv2)
        ops.interp_index(g.knn1, B, N, M1, out=g.inv1, src_pos=g.pos1_aos, row_perm=getattr(g, "rank1", None))

    def _stage_positions(self, cloud_data, dev, ring=None):
        xyz = cloud_data["xyz"]
        if ring is not None and not xyz.is_cuda:
            xyz_d = ring.upload(xyz, dtype=F32)           # (hip_ops.PinnedRing: PointNet2._forward_from_host)
        else:
            xyz_d = xyz.to(device=dev, dtype=F32, non_blocking=True).contiguous()
        B, _, N = xyz_d.shape
        fs = cloud_data.get("fps_start", None) if isinstance(cloud_data, dict) else None
        if fs is None:
            M1, M2, _ = self._sizes3(N)
            fs = torch.stack([torch.randint(0, N, (B,)), torch.randint(0, M1, (B,)), torch.randint(0, M2, (B,))])
        fs = torch.as_tensor(fs).to(device=dev, dtype=I32, non_blocking=True).contiguous()
        if fs.shape != (3, B):
            raise ValueError(f"fps_start must have shape (3,{B})")
        return xyz_d, fs

    # ------------------------------------------------------------------------------------------ forward
    def _forward_impl(self, xyz, cloud, fps_start, training, geo=None, drop_keep=None, need_grad=True):
        dev = xyz.device
        B, _, N = xyz.shape
        M1, M2, M3 = self._sizes3(N)
        # eval mode with gradients wanted: what a training forward keeps, on the running statistics (PointNet2._forward_impl)
        frozen = (not training) and bool(need_grad)
        keep = bool(training) or frozen
        mode = 1 if training else (BN_FROZEN_KEEP if frozen else 0)
        if geo is None:
            geo = self._geometry(xyz, fps_start, inverted=keep)
        elif (geo.B, geo.N, geo.M1, geo.M2, geo.M3) != (B, N, M1, M2, M3):
            raise ValueError("prefetched geometry does not match this batch")
        if keep and not getattr(geo, "has_inverted", True):
            # tables prefetched in eval mode, forward in training mode: the backward pass needs the inverted indices
            self._inverted_tables(geo)
            geo.has_inverted = True
        s = _Saved()
        s.__dict__.update({k: v for k, v in geo.__dict__.items()
                           if k not in ("ready", "stream", "totals", "nn_ws", "fps_start", "_join", "has_inverted", "has_rows0")})
        s.xyz = xyz
        widths = [16, 16, 32, 64, 64, 64, 64, 34, 34]
        aux = torch.empty(4 * sum(widths), dtype=F32, device=dev)
        stats = torch.empty(STAT_SLOTS * 2 * sum(widths), dtype=F32, device=dev)
        cur = [0, 0]
        bf = lambda prefix: self.mma_dtype == "bf16" and prefix in self.BF16_BLOCKS      # noqa: E731
        s.b_sa1 = _blocks_of(self.sa1_module.conv.local_nn, aux, stats, cur, bf("sa1_module.conv.local_nn"))
        s.b_sa2 = _blocks_of(self.sa2_module.conv.local_nn, aux, stats, cur, bf("sa2_module.conv.local_nn"))
        s.b_sa3 = _blocks_of(self.sa3_module.conv.local_nn, aux, stats, cur, bf("sa3_module.conv.local_nn"))
        s.b_sa4 = _blocks_of(self.sa4_module.nn, aux, stats, cur, bf("sa4_module.nn"))[0]
        s.b_fp4 = _blocks_of(self.fp4_module.nn, aux, stats, cur, bf("fp4_module.nn"))[0]
        s.b_fp3 = _blocks_of(self.fp3_module.nn, aux, stats, cur, bf("fp3_module.nn"))[0]
        s.b_fp2 = _blocks_of(self.fp2_module.nn, aux, stats, cur, bf("fp2_module.nn"))[0]
        s.b_fp1 = _blocks_of(self.fp1_module.nn, aux, stats, cur)[0]
        s.aux, s.stats = aux, stats
        for bb in s.b_sa1 + s.b_sa2 + s.b_sa3 + [s.b_sa4, s.b_fp4, s.b_fp3, s.b_fp2, s.b_fp1]:
            bb.frozen = frozen
        e = lambda *shape, dt=F32: torch.empty(*shape, dtype=dt, device=dev)          # noqa: E731
        if getattr(geo, "has_rows0", False) and geo.rows0.shape[0] == B * N:
            s.rows0 = geo.rows0                   # packed by the geometry pass (PointNet2._input_only)
        else:
            s.rows0 = ops.pack_rows(cloud, xyz)
        s.ext1, s.arg1, s.x1 = e(B * M1, 16), e(B * M1, 16, dt=I32), e(B * M1, 16)
        ops.sa_forward(self._sa1_desc(s), mode)
        s.ext2, s.arg2, s.x2 = e(B * M2, 32), e(B * M2, 32, dt=I32), e(B * M2, 32)
        ops.sa_forward(self._sa2_desc(s), mode)
        s.ext3, s.arg3l, s.x3 = e(B * M3, 64), e(B * M3, 64, dt=I32), e(B * M3, 64)
        ops.sa_forward(self._sa3l_desc(s), mode)
        s.h_sa4 = e(B * M3, 64)
        ops.fp_forward(self._sa4_desc(s), mode)
        s.xg, s.argg = ops.plot_max_forward(s.h_sa4, s.b_sa4.a, s.b_sa4.c, B, M3, 64)
        if self.log_embeddings:
            self.last_G_tensor = s.xg
        s.h4 = e(B * M3, 64)
        ops.fp_forward(self._fp4_desc(s), mode)
        s.h3 = e(B * M2, 64)
        ops.fp_forward(self._fp3_desc(s), mode)
        s.h2 = e(B * M1, 36)
        ops.fp_forward(self._fp2_desc(s), mode)
        s.h1 = torch.empty(B * N, 36, dtype=self._act_dtype(B * N), device=dev)
        ops.fp_forward(self._fp1_desc(s), mode)
        cov, proba = e(B * N, 4), e(B * N, 4)
        s.drop_keep = drop_keep
        ops.head_forward(ops.head_desc(s.h1, s.b_fp1.a, s.b_fp1.c, self.lin1, self.lin2, cov, proba, drop_mask=drop_keep,
                                       drop_p=self.drop))
        return cov, proba, s

    def _sa3l_desc(self, s, dout=None, dfeat=None, g=False):
        return ops.sa_desc(s.b_sa3, s.x2, 32, s.pos2_aos, s.pos3_aos, s.nbr3, s.cnt3, s.tot3, s.B, s.M2, s.M3, s.ext3,
                           s.arg3l, s.x3, dout=dout, dfeat=dfeat, with_grads=g, order=s.ord3)

    def _sa4_desc(self, s, **kw):
        return ops.fp_desc(s.b_sa4, s.B, s.M3, s.M3, 64, 3, s.x3, s.h_sa4, skip=s.pos3_aos, **kw)

    def _fp4_desc(self, s, **kw):
        return ops.fp_desc(s.b_fp4, s.B, s.M3, 1, 64, 64, s.xg, s.h4, knn=s.knn4, skip=s.x3, **kw)

    def _fp3_desc(self, s, **kw):
        return ops.fp_desc(s.b_fp3, s.B, s.M2, s.M3, 64, 32, s.h4, s.h3, src_affine=(s.b_fp4.a, s.b_fp4.c), knn=s.knn3,
                           skip=s.x2, **kw)

    # ------------------------------------------------------------------------------------------ backward
    def _backward_impl(self, s, dcov, dproba):
        dev = s.xyz.device
        B, N, M1, M2, M3 = s.B, s.N, s.M1, s.M2, s.M3
        params = list(self.parameters())
        sizes = OrderedDict(dy2=B * M1 * 36, dy3=B * M2 * 64, dy4=B * M3 * 64, dx1=B * M1 * 16, dx2=B * M2 * 32,
                            dx3=B * M3 * 64, dxg=B * 64, dy_sa4=B * M3 * 64)
        flat, buf, views, images, arena = self._grad_arena(params, sizes, dev)
        for bb in s.b_sa1 + s.b_sa2 + s.b_sa3 + [s.b_sa4, s.b_fp4, s.b_fp3, s.b_fp2, s.b_fp1]:
            bb.grads = (views[id(bb.lin.weight)], views[id(bb.lin.bias)], views[id(bb.bn.weight)], views[id(bb.bn.bias)])
            bb.grad_images = images
        dcov = None if dcov is None else dcov.contiguous()
        dproba = None if dproba is None else dproba.contiguous()
        e = lambda *shape: torch.empty(*shape, dtype=F32, device=dev)                  # noqa: E731
        dy1 = torch.empty(B * N, 36, dtype=s.h1.dtype, device=dev)
        hg = (views[id(self.lin1.weight)], views[id(self.lin1.bias)], views[id(self.lin2.weight)], views[id(self.lin2.bias)])
        hd = ops.head_desc(s.h1, s.b_fp1.a, s.b_fp1.c, self.lin1, self.lin2, dcov=dcov, dproba=dproba, dy=dy1, grads=hg,
                           grad_images=images, drop_mask=getattr(s, "drop_keep", None), drop_p=self.drop)
        ops.head_backward(hd)
        bn_ok = torch.empty(4, dtype=I32, device=dev)

        def bn_of(mod):
            return mod.nn[0][2]

        def sums(fn, d, bn, blk, k):
            fn(d, bn.weight.detach(), bn.bias.detach(), blk.aux[2], blk.aux[3], views[id(bn.weight)], views[id(bn.bias)],
               bn_ok[k:k + 1])

        sums(ops.head_bn_sums, hd, bn_of(self.fp1_module), s.b_fp1, 0)
        dy2, dy3, dy4 = buf["dy2"].view(B * M1, 36), buf["dy3"].view(B * M2, 64), buf["dy4"].view(B * M3, 64)
        dx1, dx2, dx3 = buf["dx1"].view(B * M1, 16), buf["dx2"].view(B * M2, 32), buf["dx3"].view(B * M3, 64)
        dxg, dy_sa4 = buf["dxg"].view(B, 64), buf["dy_sa4"].view(B * M3, 64)
        d1 = self._fp1_desc(s, dy=dy1, dsrc=dy2, du_scratch=torch.empty(B * N, 36, dtype=s.h1.dtype, device=dev), with_grads=True, interp_index=s.inv1,
                            bn_sums_done=bn_ok[0:1])
        ops.fp_backward(d1)
        sums(ops.fp_bn_sums, d1, bn_of(self.fp2_module), s.b_fp2, 1)
        d2 = self._fp2_desc(s, dy=dy2, dsrc=dy3, dskip=dx1, du_scratch=e(B * M1, 64), with_grads=True, interp_index=s.inv2,
                            bn_sums_done=bn_ok[1:2])
        ops.fp_backward(d2)
        sums(ops.fp_bn_sums, d2, bn_of(self.fp3_module), s.b_fp3, 2)
        d3 = self._fp3_desc(s, dy=dy3, dsrc=dy4, dskip=dx2, du_scratch=e(B * M2, 64), with_grads=True, interp_index=s.inv3,
                            bn_sums_done=bn_ok[2:3])
        ops.fp_backward(d3)
        sums(ops.fp_bn_sums, d3, bn_of(self.fp4_module), s.b_fp4, 3)
        ops.fp_backward(self._fp4_desc(s, dy=dy4, dsrc=dxg, dskip=dx3, du_scratch=e(B * M3, 64), with_grads=True,
                                       interp_index=s.inv4, bn_sums_done=bn_ok[3:4]))
        ops.plot_max_backward(dxg, s.argg, B, M3, 64, dy_sa4)
        ops.fp_backward(self._sa4_desc(s, dy=dy_sa4, dsrc=dx3, with_grads=True))
        ops.sa_backward(self._sa3l_desc(s, dout=dx3, dfeat=dx2, g=True))
        ops.sa_backward(self._sa2_desc(s, dout=dx2, dfeat=dx1, g=True))
        ops.sa_backward(self._sa1_desc(s, dout=dx1, g=True))
        ops.grad_reduce(arena, flat.numel(), images)
        s.flat_grad = flat
        self._last_flat_grad = flat
        return [views[id(p)] for p in params]
