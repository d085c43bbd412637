"""PointNet2 -- drop-in for the reference `model/point_net2.py` (IGNF/StrataNet2), MI355X-native underneath.

Same public surface as the reference class (`/root/reference/model/point_net2.py:70-220`): constructor fields read
from `args`, `forward(cloud_data) -> (coverages_pointwise, proba_pointwise)`, `get_long_form`, `get_batch_format`,
early-stopping / checkpoint helpers, and a `state_dict()` with the reference's exact keys and shapes
(`sa1_module.conv.local_nn.0.0.weight`, ..., `lin2.bias`), so reference checkpoints load and the reference training
and inference drivers (`learning/train.py:53-56`, `predict.py:103-114`) can call it unchanged.

Underneath, `forward` is ONE autograd node whose forward and backward are sequences of hand-written HIP kernels
(libstrata_hip.so via ctypes, raw device pointers, torch's current stream): FPS -> ball query -> fused
gather+MLP+BN+max (SA1, SA2) -> global SA -> 3-NN interpolation + MLP (FP3..FP1) -> head.  No torch_cluster /
torch_scatter / torch_geometric, no per-edge tensors, no host synchronisation anywhere in a step.

Additive extension: `cloud_data["fps_start"]` -- int tensor (2,B) of LOCAL start indices for the two FPS calls
(the reference's `fps` starts at a C `rand()` point and is unseeded).  Absent: random starts
drawn with torch's generator in training mode... the reference draws them in eval mode too, so does this class.
"""
import os
from collections import OrderedDict

import torch
import torch.nn as nn
from torch.nn import BatchNorm1d as BN
from torch.nn import Linear as Lin
from torch.nn import ReLU
from torch.nn import Sequential as Seq

from . import executor as X
from . import hip_ops as ops
from ._lib import BN_FROZEN_KEEP, MAX_NEIGHBORS as _MAXN, STAT_SLOTS, StrataHipError

MAX_NEIGHBORS = _MAXN      # radius(..., max_num_neighbors=2000), model/point_net2.py:24 (tests lower it by monkeypatching)

F32, I32, I64, F64 = torch.float32, torch.int32, torch.int64, torch.float64


def MLP(channels):
    """(Linear -> ReLU -> BatchNorm1d) blocks with the reference's module nesting, hence its state-dict keys
    `<i>.0.*` (Linear) and `<i>.2.*` (BatchNorm)  -- model/point_net2.py:45-53."""
    return Seq(*[Seq(Lin(channels[i - 1], channels[i]), ReLU(), BN(channels[i])) for i in range(1, len(channels))])


class PointConv(nn.Module):
    """Parameter holder named like torch_geometric's PointConv (`.local_nn`); the computation is sn2_sa_forward."""

    def __init__(self, local_nn):
        super().__init__()
        self.local_nn = local_nn


class SAModule(nn.Module):
    def __init__(self, ratio, r, nn_):
        super().__init__()
        self.ratio, self.r = ratio, r
        self.conv = PointConv(nn_)


class GlobalSAModule(nn.Module):
    def __init__(self, nn_):
        super().__init__()
        self.nn = nn_


class FPModule(nn.Module):
    def __init__(self, k, nn_):
        super().__init__()
        self.k = k
        self.nn = nn_


class _Saved:
    """Everything the backward pass needs from one forward (device tensors + sizes)."""
    pass


def _blocks_of(seq, aux_arena, stats_arena, cursor, mma_bf16=False):
    out = []
    for blk in seq:
        lin, bn = blk[0], blk[2]
        c = lin.out_features
        aux = aux_arena[cursor[0]:cursor[0] + 4 * c].view(4, c)
        ns = STAT_SLOTS * 2 * c
        st = stats_arena[cursor[1]:cursor[1] + ns]
        cursor[0] += 4 * c
        cursor[1] += ns
        out.append(ops.BlockBuffers(lin, bn, aux, st))
        out[-1].mma_bf16 = mma_bf16
    return out


class _PointNet2Fn(torch.autograd.Function):
    """forward/backward of the whole network as one autograd node; `params` are passed so autograd routes their
    gradients, the kernels read them through the modules (same storage)."""

    @staticmethod
    def forward(ctx, model, xyz, cloud, fps_start, geo, drop_keep, *params):
        training = model.training
        # will a backward pass follow?  Grad mode is off inside Function.forward, and ctx.needs_input_grad reports the inputs'
        # requires_grad flags WHATEVER the caller's grad mode (round 5: under torch.no_grad() it said yes, and an eval forward
        # took the everything-kept path of the eval-mode backward): the callers record torch.is_grad_enabled() in front of apply
        need_grad = bool(getattr(model, "_grad_mode_at_call", True)) and any(ctx.needs_input_grad[6:])
        cov, proba, saved = model._forward_impl(xyz, cloud, fps_start, training, geo, drop_keep, need_grad=need_grad)
        ctx.model = model
        if not isinstance(saved, X.NetSaved):
            saved.training = training
        ctx.saved = saved if need_grad else None
        ctx.n_params = len(params)
        return cov, proba

    @staticmethod
    def backward(ctx, dcov, dproba):
        if ctx.saved is None:
            raise RuntimeError("PointNet2: backward through a forward that recorded no graph")
        # (after an eval-mode forward -- BatchNorm on its running statistics, model/point_net2.py:45-53 -- the gradient is the
        # running-statistics one, gamma * invstd * dy: the forward kept what a training forward keeps and marked its blocks
        # `frozen`, sn2_block.frozen_stats; round 5)
        grads = ctx.model._backward_impl(ctx.saved, dcov, dproba)
        ctx.saved = None
        return (None, None, None, None, None, None) + tuple(grads)


class PointNet2(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.cuda_device = args.cuda
        self.subsample_size = args.subsample_size
        self.n_class = args.n_class
        self.drop = args.drop
        self.n_input_feats = args.n_input_feats - 2  # x and y are not fed to the network (point_net2.py:77)
        self.set_patience_attributes(args)
        self.log_embeddings = args.log_embeddings
        self.last_G_tensor = None
        self._last_flat_grad = None
        self._last_cloud_dev = None
        # additive extension (not a reference flag): "bf16" = bfloat16 operands on the matrix cores (BASELINE.json
        # configs[4]); default "fp32" = the reference's precision
        self.set_mma_dtype(getattr(args, "mma_dtype", "fp32"))
        if self.n_class != 4 or self.n_input_feats != 8:
            raise ValueError("the HIP kernels cover the reference architecture: n_class=4, 10 input features")
        ndim = 3
        mlp1 = [self.n_input_feats + ndim, 16, 16]
        mlp2 = [mlp1[-1] + ndim, 32]
        mlp3 = [mlp2[-1] + ndim, 64]
        # construction order = the reference's (point_net2.py:84-96): same RNG stream => same default weights
        self.sa1_module = SAModule(args.ratio1, args.r1, MLP(mlp1))
        self.sa2_module = SAModule(args.ratio2, args.r2, MLP(mlp2))
        self.sa3_module = GlobalSAModule(MLP(mlp3))
        mlp3_fp = [mlp3[-1] + mlp2[-1], 64]
        mlp2_fp = [mlp3_fp[-1] + mlp1[-1], 34]
        mlp1_fp = [mlp2_fp[-1] + self.n_input_feats, 34]
        self.fp3_module = FPModule(1, MLP(mlp3_fp))
        self.fp2_module = FPModule(3, MLP(mlp2_fp))
        self.fp1_module = FPModule(3, MLP(mlp1_fp))
        self.lin1 = nn.Linear(mlp1_fp[-1], 16)
        self.lin2 = nn.Linear(16, self.n_class + 1)
        self.lin2.bias = nn.Parameter(torch.tensor([0.733, 0.266, 0.235, 0.358, 0.500]))  # point_net2.py:97-99
        self.softmax = nn.Softmax(dim=1)
        self.sigmoid = nn.Sigmoid()
        if self.cuda_device is not None:
            self.cuda(self.cuda_device)

    # the blocks `mma_dtype = "bf16"` applies to: everything that runs on the matrix cores -- the set-abstraction levels and
    # the dense layers over centroids (SA3, FP3, FP2).  The two per-point layers (FP1 in its source-side form, the head)
    # are VALU streaming kernels bound by HBM, not by arithmetic: they stay fp32.
    BF16_BLOCKS = ("sa1_module.conv.local_nn", "sa2_module.conv.local_nn", "sa3_module.nn", "fp3_module.nn", "fp2_module.nn")
    mma_dtype = "fp32"
    # the level-1 FPS kernel when its pass shares the chip with feature kernels (sn2_fps_waves: 8 = one workgroup of 8 waves per plot)
    fps_waves_shared = int(os.environ.get("SN2_FPS_WAVES_SHARED", "8"))
    # ... and with MANY plots in the pass (more than 32: the parcel loop's 512 per launch = two FPS workgroups per CU) 4 waves per
    # plot where the kernel has that form (plots of at most 16 384 points; larger ones take 8): under that much concurrency the pass
    # itself is shorter with fewer waves (4.1 against 5.1 ms per 512 plots of 10 000 points) and the loop 2.6 % faster
    fps_waves_many = int(os.environ.get("SN2_FPS_WAVES_MANY", "4"))
    geometry_fork = True       # `_geometry`: the three independent chains behind the level-1 FPS on three streams
    # One C-ABI call per pass (executor.py; include/strata_hip.h: sn2_net_geometry / sn2_net_forward / sn2_net_backward) instead of
    # ~25 + ~10 calls issued from Python: same entry points, same descriptors, same order, same bits -- the host time between the
    # launches is what bounded the reference's loop as written (learning/train.py:44-71).  False: the per-call path below (also
    # taken while hip_ops.timing measures single entry points).
    executor = os.environ.get("SN2_EXECUTOR", "1") == "1"
    # True (set by optim.FlatAdam(fold_gradient_images=True) where no exchange sits between backward and update): the backward
    # pass leaves the 32 images of the flat gradient unfolded and the optimiser's kernel folds them (sn2_adam_step_images: one
    # launch less per step); until that step `p.grad` holds image 0 only.
    defer_grad_reduce = False
    _grad_images_pending = None
    geometry_pair_takes_group_cloud = True      # `_geometry_pair(..., cloud2=)`: the input-only pieces once per group of batches
    fuse_eval_head = os.environ.get("SN2_FUSE_EVAL_HEAD", "1") == "1"     # eval: FP1 + head in one kernel (sn2_fp_head_eval)
    # training: SA3, its BatchNorm, the plot max, FP3 and its BatchNorm in one launch (sn2_global_level_forward) instead of five;
    # its workgroups exchange the batch statistics among themselves -- False where other processes share the device
    fuse_global_level = os.environ.get("SN2_FUSE_GLOBAL_LEVEL", "1") == "1"
    # additive: a geometry pass that is handed the batch's `cloud` also does the two INPUT-only pieces of the feature pass --
    # the level-0 rows (`sn2_pack_rows`: 12 us of the step's critical path at C2) and, when `p2_diam_pix` is set (to
    # args.diam_pix), the pixel ids of `project_to_plotwise_coverages` (project_to_2d.py:16-22: a function of x, y only; 10 us) --
    # so that a loop which runs its geometry passes ahead (pipeline.TrainPipeline, prefetch_geometry) takes them off the
    # feature pass.  Same kernels, same results.
    p2_diam_pix = None

    def set_mma_dtype(self, dtype: str):
        """"fp32" (default: exact fp32 products, the reference's precision) or "bf16": the dense contractions of
        `BF16_BLOCKS` -- forward, input gradient, weight gradient -- take bfloat16 operands on v_mfma_f32_16x16x32_bf16 /
        16x16x16 with fp32 accumulation; ReLU, BatchNorm, statistics, every arg-max and all position-only kernels stay
        fp32, so the index structures are the same bits in both modes.  A dense block with more than 64 * SN2_STAT_SLOTS
        (65 536) rows -- FP2 at the reference's default ratio1 = 0.5 on 32 768-point plots -- has no bfloat16 kernel and
        runs in fp32 (`hip_ops.fp_desc`)."""
        if dtype not in ("fp32", "bf16"):
            raise ValueError("mma_dtype must be 'fp32' or 'bf16'")
        self.mma_dtype = dtype
        return self

    # ------------------------------------------------------------------------------------------ forward
    def forward(self, cloud_data):
        cloud, xyz = cloud_data["cloud"], cloud_data["xyz"]
        dev = self.lin1.weight.device
        if dev.type != "cuda":
            raise StrataHipError("PointNet2.forward needs a HIP device (args.cuda): the product path has no CPU "
                                 "fallback; the CPU restatement lives in oracle/ and is test infrastructure")
        if cloud.dim() != 3 or cloud.shape[1] != self.n_input_feats + 2 or xyz.shape != (cloud.shape[0], 3, cloud.shape[2]):
            raise ValueError(f"expected cloud (B,{self.n_input_feats + 2},N) and xyz (B,3,N), got "
                             f"{tuple(cloud.shape)} and {tuple(xyz.shape)}")
        with torch.cuda.device(dev):
            geo = cloud_data.get("geometry", None) if isinstance(cloud_data, dict) else None
            if geo is None and not cloud.is_cuda and not xyz.is_cuda and self.host_upload_overlap:
                return self._forward_from_host(cloud_data, cloud, dev)
            cloud_d = cloud.to(device=dev, dtype=F32, non_blocking=True).contiguous()
            if geo is not None:
                if getattr(geo, "ready", None) is not None:
                    # position-only kernels already ran (or are running) on the side stream: wait for them here
                    cs = torch.cuda.current_stream()
                    cs.wait_event(geo.ready)
                    for v in geo.__dict__.values():          # allocated on the side stream, consumed on this one
                        for t in (v if isinstance(v, tuple) else (v,)):
                            if isinstance(t, torch.Tensor):
                                t.record_stream(cs)
                # ready is None: persistent buffers (alloc_geometry); the caller orders the streams itself
                xyz_d, fs = geo.xyz, None
            else:
                xyz_d, fs = self._stage_positions(cloud_data, dev)
            self._last_cloud_dev = (cloud, cloud_d)  # lets project_to_plotwise_coverages skip a second H2D copy
            params = self._params()
            self._grad_mode_at_call = torch.is_grad_enabled()
            cov, proba = _PointNet2Fn.apply(self, xyz_d, cloud_d, fs, geo, self._dropout_keep(cloud_data, cloud_d), *params)
        return cov, proba

    # CPU inputs (the reference's calling convention, learning/train.py:46-56): upload `xyz` (6 MB at C2) first through a pinned
    # ring, launch the position-only kernels on it, and let `cloud` (21 MB) follow on a copy stream while they run
    host_upload_overlap = os.environ.get("SN2_HOST_UPLOAD_OVERLAP", "1") == "1"

    def _forward_from_host(self, cloud_data, cloud, dev):
        """`forward` for CPU-resident inputs: same kernels, same results; only the order of uploads and launches differs."""
        ring = ops.pinned_ring(dev)
        cur = torch.cuda.current_stream(dev)
        # the device copy of `cloud` is allocated HERE, on the stream that consumes it, before anything of this forward is
        # launched: whatever used the block before lies in front of `start` on this stream
        cloud_d = torch.empty(cloud.shape, dtype=F32, device=dev)
        start = torch.cuda.Event()
        start.record(cur)
        xyz_d, fs = self._stage_positions(cloud_data, dev, ring=ring)
        # (the inverted tables: whenever a backward pass may follow -- training, or eval mode under autograd)
        g = self._geometry(xyz_d, fs, defer_join=True, inverted=self.training or torch.is_grad_enabled())     # launched: the device is busy from here on
        up = ops.shared_stream(dev, "upload")
        up.wait_event(start)                                                          # not for the geometry pass: only for the block's past
        ring.upload(cloud, stream=up, dtype=F32, out=cloud_d, consumer=cur)            # host memcpy + DMA beside the geometry pass
        self._last_cloud_dev = (cloud, cloud_d)
        from .project_to_2d import remember_upload
        remember_upload(cloud, cloud_d)              # `project_to_plotwise_coverages(pred, clouds, args)` as the reference calls it
        params = self._params()
        self._grad_mode_at_call = torch.is_grad_enabled()
        return _PointNet2Fn.apply(self, xyz_d, cloud_d, None, g, self._dropout_keep(cloud_data, cloud_d), *params)

    def _params(self):
        """The module's parameters in `parameters()` order, walked once (the walk was 0.1 ms of host time per forward and per
        backward) and VALIDATED per use: the direct children and every cached leaf's Parameter / buffer objects must still be the
        ones the walk saw -- a swapped `lin2`, `to_empty()`, `load_state_dict(assign=True)` or a Parameter assigned by hand
        rebuild the list (and with it the executor's model struct) instead of routing gradients to stale objects.  (A module
        replaced deeper in the tree, e.g. `fp1_module.nn[0] = ...`, is not seen by the cheap check: call `invalidate_caches()`.)"""
        c = self.__dict__.get("_leafs")
        if c is not None:
            mods = self._modules
            ok = all(mods.get(k) is v for k, v in c[0]) and all(m._parameters.get(n) is p for m, n, p in c[1]) and \
                all(m._buffers.get(n) is b for m, n, b in c[2])
            if ok:
                return self.__dict__["_param_list"]
        self.invalidate_caches()
        ps = list(self.parameters())
        leaf_p, leaf_b = [], []
        for mod in self.modules():
            leaf_p += [(mod, n, p) for n, p in mod._parameters.items() if p is not None]
            leaf_b += [(mod, n, b) for n, b in mod._buffers.items() if b is not None]
        self.__dict__["_param_list"] = ps
        self.__dict__["_leafs"] = (tuple(self._modules.items()), leaf_p, leaf_b)
        return ps

    def invalidate_caches(self):
        """Forget everything derived from the module tree (parameter list, the executor's model struct and plans)."""
        for k in ("_param_list", "_leafs", "_net_ms"):
            self.__dict__.pop(k, None)

    def _apply(self, fn, *a, **kw):          # .to() / .cuda() / .float() / to_empty(): tensors may move or be replaced
        self.invalidate_caches()
        return super()._apply(fn, *a, **kw)

    def load_state_dict(self, *a, **kw):
        self.invalidate_caches()
        return super().load_state_dict(*a, **kw)

    def _use_executor(self):
        return bool(self.executor) and ops._timing is None

    def _net_model(self):
        """The executor's view of this model (executor.ModelStruct), rebuilt when a parameter / buffer object or address or one
        of the settings it carries changed."""
        params = self._params()                       # (validates the module tree, drops `_net_ms` when it changed)
        ms = self.__dict__.get("_net_ms")
        if ms is None or not ms.current(self, MAX_NEIGHBORS):
            ms = self.__dict__["_net_ms"] = X.ModelStruct(self, params, MAX_NEIGHBORS)
        return ms

    def _net_ctx(self):
        c = self.__dict__.get("_net_ctx_obj")
        if c is None:
            c = self.__dict__["_net_ctx_obj"] = X.NetCtx()
        return c

    def _dropout_keep(self, cloud_data, cloud_d):
        """F.dropout(x, p=self.drop, training=self.training) between lin1 and lin2 (model/point_net2.py:142): the (B*N) words
        of kept hidden channels for the head kernels, or None (eval mode, p = 0).  The mask is drawn on the device from
        torch's generator (Bernoulli(1-p) per element, as F.dropout draws it; torch's own CUDA dropout stream is not
        reproducible across devices either); additive extension for parity tests: `cloud_data["dropout_mask"]`, a
        (B*N,16) tensor, non-zero = keep."""
        if not (self.training and self.drop > 0):
            return None
        R = cloud_d.shape[0] * cloud_d.shape[2]
        keep = cloud_data.get("dropout_mask", None) if isinstance(cloud_data, dict) else None
        if keep is None:
            keep = torch.empty(R, 16, dtype=F32, device=cloud_d.device).bernoulli_(max(0.0, 1.0 - float(self.drop)))
        else:
            keep = torch.as_tensor(keep).to(device=cloud_d.device)
            if tuple(keep.shape) != (R, 16):
                raise ValueError(f"dropout_mask must have shape ({R},16)")
        return ops.dropout_mask_words(keep)

    # Keep FP1's d pre-activation rows in the plots' Morton order (sn2_fp.row_perm).  OFF: measured at 16 x 32 768, the source
    # pass's gather did not get faster (45.2 against 46.8 us: neighbouring sources already share an XCD's L2, and the pass is
    # bound by the number of cache lines its gather instructions touch, not by where they come from), while the row pass's
    # permuted stores cost 7.8 us (57.9 against 50.1).  The path stays tested (tests/test_gpu_network.py).
    fp1_morton_rows = os.environ.get("SN2_FP1_MORTON_ROWS", "0") == "1"

    @staticmethod
    def _fp1_source_side(rows):
        """Whether the per-point layer FP1 runs in its source-side form (hip_ops.fp_desc hands out `src_ws`): only that form
        keeps bfloat16 rows (`_act_dtype`) and a permuted d pre-activation buffer (`rank1`)."""
        return bool(ops.SOURCE_SIDE) and rows > 64 * STAT_SLOTS

    def _act_dtype(self, rows):
        """Storage type of the three per-point activation buffers (FP1's output h1, the head's gradient dy1, FP1's
        d pre-activation): bfloat16 under `mma_dtype = "bf16"` where the per-point layer takes its source-side form (more
        than 64 * SN2_STAT_SLOTS rows), else fp32.  These 75 MB buffers are what the per-point kernels stream."""
        return torch.bfloat16 if (self.mma_dtype == "bf16" and self._fp1_source_side(rows)) else F32

    def _sizes(self, N):
        M1 = ops.fps_num_samples(N, self.sa1_module.ratio)
        M2 = ops.fps_num_samples(M1, self.sa2_module.ratio)
        return M1, M2

    # ------------------------------------------------------------------------------------------ geometry
    def alloc_geometry(self, B, N, device=None):
        """Persistent result buffers for `_geometry(..., out=)`: what a software-pipelined training loop hands to the
        position-only kernels of the batches in flight (pipeline.TrainPipeline)."""
        dev = torch.device(device if device is not None else self.lin1.weight.device)
        if self._use_executor():
            with torch.cuda.device(dev):
                return X.ArenaGeometry(self._net_model().plan(self, B, N), dev, self)      # one allocation, views on demand
        M1, M2 = self._sizes(N)
        e = lambda *shape, dt=F32: torch.empty(*shape, dtype=dt, device=dev)          # noqa: E731
        g = _Saved()
        g.B, g.N, g.M1, g.M2 = B, N, M1, M2
        g.idx1, g.pos1_soa, g.pos1_aos = e(B, M1, dt=I32), e(B, 3, M1), e(B * M1, 4)
        g.ws1 = e(ops.fps_ws_words(B, N), dt=I32) if ops.fps_fills_ws(B, N, M1) else None
        # every point's position along the plot's Morton curve (the level-1 FPS leaves it in its workspace): the order FP1's
        # backward keeps its d pre-activation rows in (hip_ops.fp_desc: row_perm)
        g.rank1 = ops.fps_ws_rank(g.ws1, B, N) if (self.fp1_morton_rows and g.ws1 is not None and self._fp1_source_side(B * N)) else None
        g.nbr1, g.cnt1 = e(B * M1, min(MAX_NEIGHBORS, N), dt=I32), e(B * M1, dt=I32)
        g.idx2, g.pos2_soa, g.pos2_aos = e(B, M2, dt=I32), e(B, 3, M2), e(B * M2, 4)
        g.ws2 = e(ops.fps_ws_words(B, M1), dt=I32) if ops.fps_fills_ws(B, M1, M2) else None
        g.nbr2, g.cnt2 = e(B * M2, min(MAX_NEIGHBORS, M1), dt=I32), e(B * M2, dt=I32)
        g.totals = torch.zeros(2, dtype=I64, device=dev)
        g.pos3 = torch.zeros(B, 3, 1, dtype=F32, device=dev)
        g.knn3 = (e(B * M2, 3, dt=I32), e(B * M2, 3))
        g.knn2 = (e(B * M1, 3, dt=I32), e(B * M1, 3))
        g.knn1 = (e(B * N, 3, dt=I32), e(B * N, 3))
        g.tot1, g.tot2 = g.totals[0:1], g.totals[1:2]
        g.ord1, g.ord2 = e(ops.sa_order_len(B, M1), dt=I32), e(ops.sa_order_len(B, M2), dt=I32)
        g.inv3, g.inv2, g.inv1 = (e(ops.interp_ws_words(B, R, S)) for R, S in ((M2, 1), (M1, M2), (N, M1)))
        g.nn_ws = tuple(e(ops.three_nn_ws_words(B, S, T), dt=I32) if ops.three_nn_uses_grid(S, T) else None
                        for S, T in ((M2, M1), (M1, N)))
        self._alloc_input_only(g, B, N, dev)
        g.ready = None
        return g

    def _alloc_input_only(self, g, B, N, dev):
        """Buffers of the input-only pieces a geometry pass may produce (see `p2_diam_pix`)."""
        g.rows0 = torch.empty(B * N, 12, dtype=F32, device=dev)
        g.has_rows0 = False
        g.p2_pix, g.p2_mm, g.p2_diam_pix = None, None, None
        if self.p2_diam_pix is not None:
            g.p2_pix = torch.empty(B * N, dtype=I32, device=dev)
            g.p2_mm = torch.empty(B, 4, dtype=F32, device=dev)

    def _input_only(self, g, cloud, xyz):
        """The input-only pieces of the feature pass, run with the geometry pass when it is handed the batch's `cloud` (device)."""
        ops.pack_rows(cloud, xyz, out=g.rows0)
        g.has_rows0 = True
        if g.p2_pix is not None and self.p2_diam_pix is not None:
            ops.plot_pixels(cloud, self.p2_diam_pix, out=(g.p2_mm, g.p2_pix))
            g.p2_diam_pix = int(self.p2_diam_pix)

    def _geometry(self, xyz, fps_start, out=None, fork=None, shared=False, defer_join=False, inverted=True, cloud=None):
        """Everything that depends on the point POSITIONS only (no weights, no features): both FPS levels, both ball
        queries, the three 3-NN tables.  In the reference these are the torch_cluster calls inside SAModule / FPModule
        (point_net2.py:22-25, 63).  Because they need no parameters they can run ahead of the feature kernels: see
        `prefetch_geometry`.  `out`: buffers from `alloc_geometry` to write into (no allocation, same addresses every
        time: what a hipGraph-replayed feature pass needs).
        `fork` (default `self.geometry_fork`): after the level-1 FPS the three independent chains -- (a) ball query 1 +
        its work items, (b) level-2 FPS, ball query 2, the two small 3-NN tables, (c) the per-point 3-NN table + its
        inverted index -- run on three streams and join before returning (captured into a hipGraph they become parallel
        branches): the level-2 FPS is 16 workgroups for 0.15 ms, chains (a) and (c) fill the chip beside it.
        `defer_join` (with `fork`): return without joining; `g._join = (stream of chain b, stream of chain c)` for the caller
        to wait on where it first needs them (`_forward_impl`: the first set-abstraction level starts beside chain b).
        `inverted=False`: skip the inverted 3-NN tables (only the backward pass gathers through them: an eval-mode forward
        does not need them -- a tenth of the geometry pass of the parcel loop); `g.has_inverted` records it.
        `shared`: the pass runs beside other batches' feature kernels (a pipelined loop, `prefetch_geometry`): the level-1
        FPS takes `fps_waves_shared` waves per plot (include/strata_hip.h: sn2_fps_waves).
        `cloud` (B,10,N) on the device: also run the input-only pieces of the feature pass here (`_input_only`)."""
        if self._use_executor():
            return X.geometry(self, self._net_model(), xyz, fps_start, out=out, fork=fork, shared=shared, defer_join=defer_join,
                              inverted=inverted, cloud=cloud)
        dev = xyz.device
        B, _, N = xyz.shape
        M1, M2 = self._sizes(N)
        g = out if out is not None else self.alloc_geometry(B, N, dev)
        if (g.B, g.N, g.M1, g.M2) != (B, N, M1, M2):
            raise ValueError("geometry buffers do not match this batch")
        g.xyz = xyz
        fork = self.geometry_fork if fork is None else fork
        cur = torch.cuda.current_stream(dev)
        ops.fps(xyz, M1, fps_start[0], out=(g.idx1, g.pos1_soa, g.pos1_aos, g.ws1),
                waves=(self.fps_waves_many if B > 32 else self.fps_waves_shared) if shared else 0)
        if fork:
            sb, sc = ops.shared_stream(dev, "fork_b"), ops.shared_stream(dev, "fork_c")
            sb.wait_stream(cur)
            sc.wait_stream(cur)
        else:
            sb = sc = cur
        with torch.cuda.stream(sb):                                        # (b) the level-2 chain
            ops.fps(g.pos1_soa, M2, fps_start[1], out=(g.idx2, g.pos2_soa, g.pos2_aos, g.ws2))
            # (the message totals: only where a backward may follow -- `inverted`; sn2_net_geometry)
            ops.ball_query(g.pos1_soa, g.pos2_soa, self.sa2_module.r, MAX_NEIGHBORS, g.tot2 if inverted else False, fps_ws=g.ws2,
                           out=(g.nbr2, g.cnt2))
            ops.sa_order(g.cnt2, B, M2, out=g.ord2)
            ops.three_nn(g.pos3, g.pos2_soa, 1, out=g.knn3)
            ops.three_nn(g.pos2_soa, g.pos1_soa, 3, out=g.knn2, ws=g.nn_ws[0])
            # the inverted 3-NN tables the backward pass gathers through: positions only, so they belong here
            if inverted:
                ops.interp_index(g.knn3, B, M2, 1, out=g.inv3)
                ops.interp_index(g.knn2, B, M1, M2, out=g.inv2)
        with torch.cuda.stream(sc):                                        # (c) the per-point table
            ops.three_nn(g.pos1_soa, xyz, 3, out=g.knn1, ws=g.nn_ws[1])
            if inverted:
                ops.interp_index(g.knn1, B, N, M1, out=g.inv1, src_pos=g.pos1_aos, row_perm=g.rank1)
        g.has_inverted = bool(inverted)
        g.has_rows0 = False
        if cloud is not None:
            self._input_only(g, cloud, xyz)
        # (a)
        ops.ball_query(xyz, g.pos1_soa, self.sa1_module.r, MAX_NEIGHBORS, g.tot1 if inverted else False, fps_ws=g.ws1, out=(g.nbr1, g.cnt1))
        ops.sa_order(g.cnt1, B, M1, out=g.ord1)
        g._join = None
        if fork and defer_join:
            g._join = (sb, sc)
        elif fork:
            cur.wait_stream(sb)
            cur.wait_stream(sc)
        return g

    def alloc_geometry_pair(self, B, N, device=None, group=2):
        """Buffers for `_geometry_pair`: the position-only kernels of `group` (2, or more) batches of B plots launched together
        (FPS is one workgroup per plot and M sequential rounds: several batches take as long as one), plus the per-batch views
        the feature passes read.  -> (combined buffers, (geometry of the first batch, of the second, ...))."""
        dev = torch.device(device if device is not None else self.lin1.weight.device)
        M1, M2 = self._sizes(N)
        gp = self.alloc_geometry(group * B, N, dev)
        e = lambda *shape, dt=F32: torch.empty(*shape, dtype=dt, device=dev)          # noqa: E731
        # the per-batch products -- message totals, SA work items, inverted 3-NN indices -- of all `group` batches live in ONE
        # strided buffer each, so that the pass builds them with one (set of) launch(es) per kind instead of one per batch
        # (hip_ops.*_group; round 5: 18 small launches per pass of eight batches instead of 144); a batch's slice is an ordinary
        # per-batch table
        up4 = lambda n: (n + 3) // 4 * 4          # noqa: E731
        so1, so2 = ops.sa_order_len(B, M1), ops.sa_order_len(B, M2)
        si = [up4(ops.interp_ws_words(B, R, S)) for R, S in ((M2, 1), (M1, M2), (N, M1))]
        grp = _Saved()
        grp.G, grp.so1, grp.so2, grp.si = group, so1, so2, si
        grp.ord1, grp.ord2 = e(group * so1, dt=I32), e(group * so2, dt=I32)
        grp.inv = [e(group * w) for w in si]
        grp.totals = torch.zeros(group, 2, dtype=I64, device=dev)
        gp._grp = grp
        halves = []
        for h in range(group):
            g = _Saved()
            g.B, g.N, g.M1, g.M2 = B, N, M1, M2
            pl, r1, r2, rn = slice(h * B, (h + 1) * B), slice(h * B * M1, (h + 1) * B * M1), \
                slice(h * B * M2, (h + 1) * B * M2), slice(h * B * N, (h + 1) * B * N)
            g.idx1, g.pos1_soa, g.pos1_aos, g.nbr1, g.cnt1 = gp.idx1[pl], gp.pos1_soa[pl], gp.pos1_aos[r1], gp.nbr1[r1], gp.cnt1[r1]
            g.idx2, g.pos2_soa, g.pos2_aos, g.nbr2, g.cnt2 = gp.idx2[pl], gp.pos2_soa[pl], gp.pos2_aos[r2], gp.nbr2[r2], gp.cnt2[r2]
            g.knn3 = (gp.knn3[0][r2], gp.knn3[1][r2])
            g.knn2 = (gp.knn2[0][r1], gp.knn2[1][r1])
            g.knn1 = (gp.knn1[0][rn], gp.knn1[1][rn])
            g.totals = grp.totals[h]
            g.tot1, g.tot2 = grp.totals[h, 0:1], grp.totals[h, 1:2]
            g.ord1, g.ord2 = grp.ord1[h * so1:(h + 1) * so1], grp.ord2[h * so2:(h + 1) * so2]
            g.inv3, g.inv2, g.inv1 = (grp.inv[k][h * si[k]:h * si[k] + ops.interp_ws_words(B, R, S)]
                                      for k, (R, S) in enumerate(((M2, 1), (M1, M2), (N, M1))))
            g.ws1 = g.ws2 = g.nn_ws = None
            g.rank1 = gp.rank1[rn] if (gp.rank1 is not None and self._fp1_source_side(B * N)) else None
            # the input-only pieces of the feature pass: the batch's slices of the GROUP's buffers (one launch each for the whole
            # group when the pass is handed the group's clouds in one tensor: `_geometry_pair(..., cloud2=)`)
            g.rows0, g.has_rows0 = gp.rows0[rn], False
            g.p2_pix = gp.p2_pix[rn] if gp.p2_pix is not None else None
            g.p2_mm = gp.p2_mm[pl] if gp.p2_mm is not None else None
            g.p2_diam_pix = None
            g.ready = None
            halves.append(g)
        return gp, tuple(halves)

    def _geometry_pair(self, xyz2, fps_start2, gp, halves, clouds=None, cloud2=None):
        """`_geometry` for len(halves) batches at once: xyz2 (G B,3,N), fps_start2 (2,G B); FPS, ball queries and 3-NN tables
        run on all plots in one launch each (into `gp`), the per-batch products (message totals, SA work items, inverted 3-NN
        indices) per batch.  Same tables as G `_geometry` calls.  clouds: the G batches' (B,10,N) device tensors -> also the
        input-only pieces of their feature passes (`_input_only`), batch by batch; cloud2 (G B,10,N): the same for the whole group
        in one launch each (the batches' clouds live in one tensor: what TrainPipeline arranges)."""
        B2, _, N = xyz2.shape
        B = B2 // len(halves)
        M1, M2 = self._sizes(N)
        if (gp.B, gp.N) != (B2, N):
            raise ValueError("geometry buffers do not match this batch pair")
        # 8 waves per plot: this pass runs beside other batches' feature kernels (sn2_fps_waves)
        ops.fps(xyz2, M1, fps_start2[0], out=(gp.idx1, gp.pos1_soa, gp.pos1_aos, gp.ws1), waves=self.fps_waves_shared)
        # (no message total of the GROUP: the batches' totals come from count_sum_group below)
        ops.ball_query(xyz2, gp.pos1_soa, self.sa1_module.r, MAX_NEIGHBORS, False, fps_ws=gp.ws1, out=(gp.nbr1, gp.cnt1))
        ops.fps(gp.pos1_soa, M2, fps_start2[1], out=(gp.idx2, gp.pos2_soa, gp.pos2_aos, gp.ws2))
        ops.ball_query(gp.pos1_soa, gp.pos2_soa, self.sa2_module.r, MAX_NEIGHBORS, False, fps_ws=gp.ws2,
                       out=(gp.nbr2, gp.cnt2))
        ops.three_nn(gp.pos3, gp.pos2_soa, 1, out=gp.knn3)
        ops.three_nn(gp.pos2_soa, gp.pos1_soa, 3, out=gp.knn2, ws=gp.nn_ws[0])
        # (the targets in the FPS pass's existing Morton order -- `dst_fps_ws=gp.ws1`, no target sort -- made the pipelined step
        # SLOWER, 0.739 against 0.718 ms: the search's query boxes grow more than the sort costs; round 5)
        ops.three_nn(gp.pos1_soa, xyz2, 3, out=gp.knn1, ws=gp.nn_ws[1])
        grp, G = gp._grp, len(halves)
        tot_flat = grp.totals.view(-1)                       # (G,2): [h][0] = level-1 messages of batch h, [h][1] = level-2
        ops.count_sum_group(gp.cnt1, G, B * M1, tot_flat, 2)
        ops.count_sum_group(gp.cnt2, G, B * M2, tot_flat[1:], 2)
        ops.sa_order_group(gp.cnt1, G, B, M1, grp.ord1, grp.so1)
        ops.sa_order_group(gp.cnt2, G, B, M2, grp.ord2, grp.so2)
        ops.interp_index_group(gp.knn3, G, B, M2, 1, grp.inv[0], grp.si[0])
        ops.interp_index_group(gp.knn2, G, B, M1, M2, grp.inv[1], grp.si[1])
        rank = gp.rank1 if halves[0].rank1 is not None else None
        ops.interp_index_group(gp.knn1, G, B, N, M1, grp.inv[2], grp.si[2], src_pos=gp.pos1_aos, row_perm=rank)
        if cloud2 is not None:
            self._input_only(gp, cloud2, xyz2)                    # one launch each over the whole group
        for h, g in enumerate(halves):
            g.xyz = xyz2[h * B:(h + 1) * B]
            g.has_rows0 = cloud2 is not None
            g.p2_diam_pix = gp.p2_diam_pix if cloud2 is not None else None
            g.has_inverted = True
            if clouds is not None and cloud2 is None:
                self._input_only(g, clouds[h], g.xyz)
        return halves

    def prefetch_geometry(self, cloud_data, lane: int = 0):
        """Run the position-only kernels of a batch on a side stream, ahead of time (typically for batch k+1 while
        batch k is in its backward pass: the FPS rounds are sequential and occupy one CU per plot, the feature kernels
        fill the rest of the chip).  Returns a handle to put into `cloud_data["geometry"]` for the forward call.
        `lane` selects one of several side streams, so that the passes of several batches can be in flight at once."""
        dev = self.lin1.weight.device
        if dev.type != "cuda":
            raise StrataHipError("prefetch_geometry needs a HIP device")
        with torch.cuda.device(dev):
            xyz_d, fs = self._stage_positions(cloud_data, dev)
            side = ops.shared_stream(dev, f"side{lane}")
            side.wait_stream(torch.cuda.current_stream())
            # staged on the current stream, read by kernels of the side stream long after this function has returned:
            # without this the allocator may hand the start indices' memory to the next forward while FPS level 2 still
            # has to read them
            xyz_d.record_stream(side)
            fs.record_stream(side)
            with torch.cuda.stream(side):
                # one stream per pass: several passes are in flight on their own lanes already, and a fork inside each
                # (three more streams + their events) cost the parcel loop 18 % (33 300 -> 27 100 plots/s)
                cl = cloud_data.get("cloud", None) if isinstance(cloud_data, dict) else None
                cl = cl if (isinstance(cl, torch.Tensor) and cl.is_cuda and cl.dtype == F32 and cl.is_contiguous()) else None
                g = self._geometry(xyz_d, fs, shared=True, fork=False, inverted=self.training, cloud=cl)
                g.fps_start = fs
                g.ready = torch.cuda.Event()
                g.ready.record(side)
            g.stream = side
        return g

    def _stage_positions(self, cloud_data, dev, ring=None):
        xyz = cloud_data["xyz"]
        if ring is not None and not xyz.is_cuda:
            xyz_d = ring.upload(xyz, dtype=F32)
        else:
            xyz_d = xyz.to(device=dev, dtype=F32, non_blocking=True).contiguous()
        B, _, N = xyz_d.shape
        fs = cloud_data.get("fps_start", None) if isinstance(cloud_data, dict) else None
        if fs is None:
            # reference behaviour: an independent random start per plot and per FPS call
            m1 = ops.fps_num_samples(N, self.sa1_module.ratio)
            fs = torch.stack([torch.randint(0, N, (B,)), torch.randint(0, m1, (B,))])
        fs = torch.as_tensor(fs).to(device=dev, dtype=I32, non_blocking=True).contiguous()
        if fs.shape != (2, B):
            raise ValueError(f"fps_start must have shape (2,{B})")
        return xyz_d, fs

    def _forward_impl(self, xyz, cloud, fps_start, training, geo=None, drop_keep=None, need_grad=True):
        if self._use_executor():
            cov, proba, s = X.forward(self, self._net_model(), xyz, cloud, fps_start, training, geo, drop_keep, need_grad=need_grad)
            if self.log_embeddings:
                self.last_G_tensor = s.x3
            return cov, proba, s
        dev = xyz.device
        B, _, N = xyz.shape
        M1, M2 = self._sizes(N)
        cur_stream = torch.cuda.current_stream(dev)
        # eval mode with gradients wanted: everything a training forward keeps, on the running statistics (SN2_BN_FROZEN_KEEP)
        frozen = (not training) and bool(need_grad)
        keep = bool(training) or frozen
        mode = 1 if training else (BN_FROZEN_KEEP if frozen else 0)
        rows0, packed = None, None
        if geo is not None and getattr(geo, "has_rows0", False) and geo.rows0.shape[0] == B * N:
            rows0 = geo.rows0                    # packed by the geometry pass (`_input_only`)
        if geo is None:
            if self.geometry_fork:
                # the row packing needs the inputs only: beside the level-1 FPS (16 workgroups) instead of behind it
                rows0 = torch.empty(B * N, 12, dtype=F32, device=dev)
                pack_stream = ops.shared_stream(dev, "pack")
                pack_stream.wait_stream(cur_stream)
                with torch.cuda.stream(pack_stream):
                    ops.pack_rows(cloud, xyz, out=rows0)
                    packed = torch.cuda.Event()
                    packed.record(pack_stream)
            geo = self._geometry(xyz, fps_start, defer_join=True, inverted=keep)
        elif (geo.B, geo.N, geo.M1, geo.M2) != (B, N, M1, M2):
            raise ValueError("prefetched geometry does not match this batch")
        join = getattr(geo, "_join", None)
        geo._join = None
        if join == "ctx":
            raise StrataHipError("a geometry pass launched by the executor with a deferred join must be consumed by the executor")
        if keep and not getattr(geo, "has_inverted", True):
            # tables prefetched in eval mode, forward in training mode: the backward pass needs the inverted indices -- of 3-NN
            # tables that a pass with a deferred join is still writing on its side streams (round 5: joined here first; at the
            # metric's size the indices were built from tables half written)
            if join is not None:
                cur_stream.wait_stream(join[0])
                cur_stream.wait_stream(join[1])
                join = None
            ops.count_sum(geo.cnt1, geo.tot1)          # (the message totals an eval-mode geometry pass did not make)
            ops.count_sum(geo.cnt2, geo.tot2)
            ops.interp_index(geo.knn3, B, M2, 1, out=geo.inv3)
            ops.interp_index(geo.knn2, B, M1, M2, out=geo.inv2)
            ops.interp_index(geo.knn1, B, N, M1, out=geo.inv1, src_pos=geo.pos1_aos, row_perm=getattr(geo, "rank1", None))
            geo.has_inverted = True
        s = _Saved()
        s.__dict__.update({k: v for k, v in geo.__dict__.items()
                           if k not in ("ready", "stream", "ws1", "ws2", "totals", "nn_ws", "fps_start", "_join", "has_rows0")})
        s.xyz = xyz
        # per-forward arenas for the BN side buffers of the 7 blocks: a,c,mean,invstd and the per-workgroup statistics
        # slots (written before they are read: no zero fill)
        widths = [16, 16, 32, 64, 64, 34, 34]
        aux = torch.empty(4 * sum(widths), dtype=F32, device=dev)
        stats = torch.empty(STAT_SLOTS * 2 * sum(widths), dtype=F32, device=dev)
        cur = [0, 0]
        bf = lambda prefix: self.mma_dtype == "bf16" and prefix in self.BF16_BLOCKS      # noqa: E731
        s.b_sa1 = _blocks_of(self.sa1_module.conv.local_nn, aux, stats, cur, bf("sa1_module.conv.local_nn"))
        s.b_sa2 = _blocks_of(self.sa2_module.conv.local_nn, aux, stats, cur, bf("sa2_module.conv.local_nn"))
        s.b_sa3 = _blocks_of(self.sa3_module.nn, aux, stats, cur, bf("sa3_module.nn"))[0]
        s.b_fp3 = _blocks_of(self.fp3_module.nn, aux, stats, cur, bf("fp3_module.nn"))[0]
        s.b_fp2 = _blocks_of(self.fp2_module.nn, aux, stats, cur, bf("fp2_module.nn"))[0]
        s.b_fp1 = _blocks_of(self.fp1_module.nn, aux, stats, cur)[0]
        s.aux, s.stats = aux, stats
        for bb in s.b_sa1 + s.b_sa2 + [s.b_sa3, s.b_fp3, s.b_fp2, s.b_fp1]:
            bb.frozen = frozen

        # ---- level 0 rows: [8 features | x y z 0]
        if packed is not None:
            cur_stream.wait_event(packed)
            s.rows0 = rows0
        elif rows0 is not None:
            s.rows0 = rows0
        else:
            s.rows0 = ops.pack_rows(cloud, xyz)
        # ---- SA1: gather + MLP[11,16,16] + BN + max over the ball-query lists     (point_net2.py:131, 21-29)
        s.ext1 = torch.empty(B * M1, 16, dtype=F32, device=dev)
        s.arg1 = torch.empty(B * M1, 16, dtype=I32, device=dev)
        s.x1 = torch.empty(B * M1, 16, dtype=F32, device=dev)
        ops.sa_forward(self._sa1_desc(s), mode)
        if join is not None:
            cur_stream.wait_stream(join[0])      # chain b of the forked geometry pass: level-2 tables, small 3-NN tables
        # ---- SA2: MLP[19,32]                                                     (:132)
        s.ext2 = torch.empty(B * M2, 32, dtype=F32, device=dev)
        s.arg2 = torch.empty(B * M2, 32, dtype=I32, device=dev)
        s.x2 = torch.empty(B * M2, 32, dtype=F32, device=dev)
        ops.sa_forward(self._sa2_desc(s), mode)
        # ---- SA3: MLP[35,64] on cat[x2, pos2] -> per-plot max                    (:133, 37-42)
        s.h_sa3 = torch.empty(B * M2, 64, dtype=F32, device=dev)
        s.h3 = torch.empty(B * M2, 64, dtype=F32, device=dev)
        if training and self.fuse_global_level and B <= 28 and not (s.b_sa3.mma_bf16 or s.b_fp3.mma_bf16):
            # ... and its max, FP3 (k=1 from the plot's global feature) and both BatchNorms: one launch  (:133-137)
            s.x3 = torch.empty(B, 64, dtype=F32, device=dev)
            s.arg3 = torch.empty(B, 64, dtype=I32, device=dev)
            ops.global_level_forward(self._sa3_desc(s), self._fp3_desc(s), s.x3, s.arg3, owner=self)
        else:
            ops.fp_forward(self._sa3_desc(s), mode)
            s.x3, s.arg3 = ops.plot_max_forward(s.h_sa3, s.b_sa3.a, s.b_sa3.c, B, M2, 64)
            # ---- FP3 (k=1 from the plot's global feature at the origin), FP2, FP1 (k=3)   (:137-139, 62-67)
            ops.fp_forward(self._fp3_desc(s), mode)
        if self.log_embeddings:
            self.last_G_tensor = s.x3
        s.h2 = torch.empty(B * M1, 36, dtype=F32, device=dev)
        ops.fp_forward(self._fp2_desc(s), mode)
        if join is not None:
            cur_stream.wait_stream(join[1])      # chain c: the per-point 3-NN table and its inverted index
        cov = torch.empty(B * N, 4, dtype=F32, device=dev)
        proba = torch.empty(B * N, 4, dtype=F32, device=dev)
        s.drop_keep = drop_keep
        if not keep and self.fuse_eval_head and self._act_dtype(B * N) == F32 and ops.SOURCE_SIDE:
            # EVAL: FP1 and the head (:139-151) in one pass, the (B*N,36) rows of h1 never reach memory (nothing is kept for a backward)
            s.h1 = None
            d1 = ops.fp_desc(s.b_fp1, B, N, M1, 34, 8, s.h2, None, src_affine=(s.b_fp2.a, s.b_fp2.c), knn=s.knn1,
                             skip=s.rows0[:, 0:8], force_src_ws=True)
            ops.fp_head_eval(d1, ops.head_desc(None, s.b_fp1.a, s.b_fp1.c, self.lin1, self.lin2, cov, proba, rows=B * N))
            return cov, proba, s
        s.h1 = torch.empty(B * N, 36, dtype=self._act_dtype(B * N), device=dev)
        ops.fp_forward(self._fp1_desc(s), mode)
        # ---- head                                                                  (:141-151)
        ops.head_forward(ops.head_desc(s.h1, s.b_fp1.a, s.b_fp1.c, self.lin1, self.lin2, cov, proba, drop_mask=drop_keep,
                                       drop_p=self.drop))
        return cov, proba, s

    # ---- descriptors (shared by forward and backward; gradient views are attached for the backward call)
    def _sa1_desc(self, s, dout=None, g=False):
        return ops.sa_desc(s.b_sa1, s.rows0[:, 0:8], 8, s.rows0[:, 8:12], s.pos1_aos, s.nbr1, s.cnt1, s.tot1, s.B, s.N,
                           s.M1, s.ext1, s.arg1, s.x1, dout=dout, dfeat=None, with_grads=g, order=getattr(s, "ord1", None))

    def _sa2_desc(self, s, dout=None, dfeat=None, g=False):
        return ops.sa_desc(s.b_sa2, s.x1, 16, s.pos1_aos, s.pos2_aos, s.nbr2, s.cnt2, s.tot2, s.B, s.M1, s.M2, s.ext2,
                           s.arg2, s.x2, dout=dout, dfeat=dfeat, with_grads=g, order=getattr(s, "ord2", None))

    def _sa3_desc(self, s, **kw):
        return ops.fp_desc(s.b_sa3, s.B, s.M2, s.M2, 32, 3, s.x2, s.h_sa3, skip=s.pos2_aos, **kw)

    def _fp3_desc(self, s, **kw):
        return ops.fp_desc(s.b_fp3, s.B, s.M2, 1, 64, 32, s.x3, s.h3, knn=s.knn3, skip=s.x2, **kw)

    def _fp2_desc(self, s, **kw):
        return ops.fp_desc(s.b_fp2, s.B, s.M1, s.M2, 64, 16, s.h3, s.h2, src_affine=(s.b_fp3.a, s.b_fp3.c), knn=s.knn2,
                           skip=s.x1, **kw)

    def _fp1_desc(self, s, **kw):
        return ops.fp_desc(s.b_fp1, s.B, s.N, s.M1, 34, 8, s.h2, s.h1, src_affine=(s.b_fp2.a, s.b_fp2.c), knn=s.knn1,
                           skip=s.rows0[:, 0:8], row_perm=getattr(s, "rank1", None), **kw)

    # ------------------------------------------------------------------------------------------ backward
    def _backward_impl(self, s, dcov, dproba):
        if isinstance(s, X.NetSaved):
            return X.backward(self, s, dcov, dproba)
        dev = s.xyz.device
        B, N, M1, M2 = s.B, s.N, s.M1, s.M2
        params = self._params()
        # one zero-filled arena: flat parameter gradient + every accumulate-into buffer of the backward chain
        sizes = OrderedDict(dy2=B * M1 * 36, dy3=B * M2 * 64, dx1=B * M1 * 16, dx2=B * M2 * 32, dx3=B * 64,
                            dy_sa3=B * M2 * 64)
        flat, buf, views, images, arena = self._grad_arena(params, sizes, dev)

        def attach(bb):
            bb.grads = (views[id(bb.lin.weight)], views[id(bb.lin.bias)], views[id(bb.bn.weight)], views[id(bb.bn.bias)])
            bb.grad_images = images

        for bb in s.b_sa1 + s.b_sa2 + [s.b_sa3, s.b_fp3, s.b_fp2, s.b_fp1]:
            attach(bb)
        dcov = None if dcov is None else dcov.contiguous()
        dproba = None if dproba is None else dproba.contiguous()
        # head
        dy1 = torch.empty(B * N, 36, dtype=s.h1.dtype, device=dev)
        hg = (views[id(self.lin1.weight)], views[id(self.lin1.bias)], views[id(self.lin2.weight)], views[id(self.lin2.bias)])
        hd = ops.head_desc(s.h1, s.b_fp1.a, s.b_fp1.c, self.lin1, self.lin2, dcov=dcov, dproba=dproba, dy=dy1, grads=hg,
                           grad_images=images, drop_mask=getattr(s, "drop_keep", None), drop_p=self.drop)
        ops.head_backward(hd)
        # FP1's BatchNorm gradients fall out of lin1's (hip_ops.head_bn_sums): no extra pass over the B*N rows
        bn1 = self.fp1_module.nn[0][2]
        bn_ok = torch.empty(4, dtype=I32, device=dev)      # per BatchNorm: did the shortcut apply (else the same kernel's row pass)
        ops.head_bn_sums(hd, bn1.weight.detach(), bn1.bias.detach(), s.b_fp1.aux[2], s.b_fp1.aux[3], views[id(bn1.weight)],
                         views[id(bn1.bias)], bn_ok[0:1])
        # FP1 -> d(fp2 output)
        dy2 = buf["dy2"].view(B * M1, 36)
        d1 = self._fp1_desc(s, dy=dy1, dsrc=dy2, du_scratch=torch.empty(B * N, 36, dtype=s.h1.dtype, device=dev), with_grads=True,
                            interp_index=s.inv1, bn_sums_done=bn_ok[0:1])
        ops.fp_backward(d1)
        bn2 = self.fp2_module.nn[0][2]      # FP2's BatchNorm feeds FP1's interpolation: its gradients from FP1's dW, db
        ops.fp_bn_sums(d1, bn2.weight.detach(), bn2.bias.detach(), s.b_fp2.aux[2], s.b_fp2.aux[3], views[id(bn2.weight)],
                       views[id(bn2.bias)], bn_ok[1:2])
        # FP2 -> d(fp3 output), d x1
        dy3, dx1 = buf["dy3"].view(B * M2, 64), buf["dx1"].view(B * M1, 16)
        d2 = self._fp2_desc(s, dy=dy2, dsrc=dy3, dskip=dx1, du_scratch=torch.empty(B * M1, 64, dtype=F32, device=dev),
                            with_grads=True, interp_index=s.inv2, bn_sums_done=bn_ok[1:2])
        ops.fp_backward(d2)
        bn3 = self.fp3_module.nn[0][2]      # and FP3's from FP2's
        ops.fp_bn_sums(d2, bn3.weight.detach(), bn3.bias.detach(), s.b_fp3.aux[2], s.b_fp3.aux[3], views[id(bn3.weight)],
                       views[id(bn3.bias)], bn_ok[2:3])
        # FP3 -> d x2 and the per-row gradients of its interpolated part (left in du3: scatter_ready = -1); then the pool between
        # FP3 and SA3 in one launch (hip_ops.global_pool_backward): d x3, its routing to the SA3 rows that attained the maximum, and
        # SA3's BatchNorm sums over those B x 64 entries
        dx3, dx2 = buf["dx3"].view(B, 64), buf["dx2"].view(B * M2, 32)
        du3 = torch.empty(B * M2, 64, dtype=F32, device=dev)
        ops.fp_backward(self._fp3_desc(s, dy=dy3, dsrc=dx3, dskip=dx2, du_scratch=du3, with_grads=True,
                                       interp_index=s.inv3, bn_sums_done=bn_ok[2:3], gather=False))
        dy_sa3 = buf["dy_sa3"].view(B * M2, 64)
        bn_sa3 = self.sa3_module.nn[0][2]
        ops.global_pool_backward(du3, s.arg3, s.h_sa3, s.b_sa3.aux[2], s.b_sa3.aux[3], B, M2, dx3, dy_sa3,
                                 views[id(bn_sa3.weight)], views[id(bn_sa3.bias)])
        ops.fp_backward(self._sa3_desc(s, dy=dy_sa3, dsrc=dx2, with_grads=True, bn_sums_done=bn_ok[3:4]))
        # SA2 -> d x1 ; SA1
        ops.sa_backward(self._sa2_desc(s, dout=dx2, dfeat=dx1, g=True))
        ops.sa_backward(self._sa1_desc(s, dout=dx1, g=True))
        if getattr(self, "defer_grad_reduce", False):
            self._grad_images_pending = (arena,) + tuple(images)       # FlatAdam folds the images inside its own kernel
        else:
            ops.grad_reduce(arena, flat.numel(), images)      # the images of (dW, db) -> image 0 = `flat`
            self._grad_images_pending = None
        s.flat_grad = flat
        self._last_flat_grad = flat
        return [views[id(p)] for p in params]

    @staticmethod
    def _grad_arena(params, sizes, dev):
        """One zero-filled arena per backward: the images of the flat parameter gradient (hip_ops.grad_images_alloc) and
        every accumulate-into buffer of the backward chain.  -> (flat = image 0, buffers, per-parameter views of image
        0, (replicas, stride), arena)."""
        poffs, n_flat = ops.flat_layout(params)
        offs, tot = {}, 0
        for k, n in sizes.items():
            offs[k] = tot
            tot += (n + 3) // 4 * 4
        arena, flat, images, extra = ops.grad_images_alloc(n_flat, dev, tot)
        buf = {k: extra[offs[k]:offs[k] + n] for k, n in sizes.items()}
        views = {id(p): flat[o:o + p.numel()].view(p.shape) for p, o in zip(params, poffs)}
        return flat, buf, views, images, arena

    # ------------------------------------------------------------------------------------------ layout helpers
    @staticmethod
    def get_long_form(data):
        """(B,f,N) -> (B*N,f), plot-major (point_net2.py:155-158)."""
        B, f, N = data.shape
        return data.permute(1, 0, 2).reshape(f, B * N).transpose(1, 0)

    def get_batch_format(self, data):
        """(B*N,f) -> (B,f,N) (point_net2.py:160-163)."""
        n = self.subsample_size
        return data.view(-1, n, data.shape[1]).transpose(1, 2)

    # ------------------------------------------------------------------------------------------ early stopping / ckpt
    def set_patience_attributes(self, args):
        self.stopped_early = False
        self.best_metric_value = 10 ** 6
        self.best_metric_epoch = 1
        self.patience_in_epochs = args.patience_in_epochs

    def stop_early(self, val_metric, epoch, args):
        """Keep the best state so far (by a metric to minimise); True once patience is exhausted
        (point_net2.py:172-184)."""
        if val_metric < self.best_metric_value:
            self.best_metric_value, self.best_metric_epoch = val_metric, epoch
            self.save_state(args)
            return False
        if epoch < args.epoch_to_start_early_stop:
            return False
        if epoch >= self.best_metric_epoch + self.patience_in_epochs:
            self.stopped_early = True
            return True
        return False

    @staticmethod
    def _checkpoint_path(args):
        tag = f"fold_n={args.current_fold_id}" if args.current_fold_id > 0 else "full"
        return os.path.join(args.stats_path, f"PCC_model_{tag}.pt")

    def save_state(self, args):
        torch.save({"best_metric_epoch": self.best_metric_epoch, "state_dict": self.state_dict(),
                    "best_metric_value": self.best_metric_value}, self._checkpoint_path(args))

    def load_state(self, save_path):
        checkpoint = torch.load(save_path, map_location=None if self.cuda_device is not None else torch.device("cpu"))
        self.load_state_dict(checkpoint["state_dict"])
        self.best_metric_epoch = checkpoint["best_metric_epoch"]
        self.best_metric_value = checkpoint["best_metric_value"]
        return self

    def load_best_state(self, args):
        return self.load_state(self._checkpoint_path(args))
