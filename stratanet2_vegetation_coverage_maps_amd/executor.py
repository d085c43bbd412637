"""PointNet2's passes behind ONE C-ABI call each (include/strata_hip.h: sn2_net_geometry / sn2_net_forward / sn2_net_backward).

`PointNet2.forward` of the reference (`/root/reference/model/point_net2.py:106-153`) is ~25 kernel-launching entry points in a
fixed order and `loss.backward()` through it (`/root/reference/learning/train.py:64`) ~10 more.  Issued one by one from Python
(descriptor by descriptor, ~40 `torch.empty` per pass) the reference's loop as written spent more host time between the launches
than the device needs for the kernels.  Here the host hands the library

    * the model once (`model_struct`: parameter / buffer pointers and the gradient offsets, rebuilt only when a pointer moved),
    * ONE arena per pass (geometry tables / activations / backward buffers; `sn2_net_*_carve` lays them out),

and makes one call.  The entry points behind it, their descriptors and their order are those of the per-call path
(`PointNet2._forward_impl` etc. with `PointNet2.executor = False`), so the results are the same bits
(tests/test_gpu_executor.py).  Views of single buffers (`saved.h1`, `geo.nbr1`, ...) are made on demand.
"""
import ctypes
import weakref
from ctypes import byref

import torch

from . import _lib
from . import hip_ops as ops
from ._lib import NetAct, NetBwd, NetDims, NetGeo, NetIO, NetModel, StrataHipError

F32, I32, I64, BF16, U8 = torch.float32, torch.int32, torch.int64, torch.bfloat16, torch.uint8
_FAKE_BASE = 1 << 40                 # carve against a fake base: a field's offset is its value minus this (0 stays distinguishable from NULL)
_ITEM = {F32: 4, I32: 4, I64: 8, BF16: 2, U8: 1}


def _geo_shapes(B, N, M1, M2, cap1, cap2):
    return {
        "idx1": (I32, (B, M1)), "pos1_soa": (F32, (B, 3, M1)), "pos1_aos": (F32, (B * M1, 4)), "ws1": (I32, (ops.fps_ws_words(B, N),)),
        "nbr1": (I32, (B * M1, cap1)), "cnt1": (I32, (B * M1,)), "tot1": (I64, (1,)), "ord1": (I32, (ops.sa_order_len(B, M1),)),
        "idx2": (I32, (B, M2)), "pos2_soa": (F32, (B, 3, M2)), "pos2_aos": (F32, (B * M2, 4)), "ws2": (I32, (ops.fps_ws_words(B, M1),)),
        "nbr2": (I32, (B * M2, cap2)), "cnt2": (I32, (B * M2,)), "tot2": (I64, (1,)), "ord2": (I32, (ops.sa_order_len(B, M2),)),
        "knn3_idx": (I32, (B * M2, 3)), "knn3_w": (F32, (B * M2, 3)), "knn2_idx": (I32, (B * M1, 3)), "knn2_w": (F32, (B * M1, 3)),
        "knn1_idx": (I32, (B * N, 3)), "knn1_w": (F32, (B * N, 3)),
        "inv3": (F32, (ops.interp_ws_words(B, M2, 1),)), "inv2": (F32, (ops.interp_ws_words(B, M1, M2),)),
        "inv1": (F32, (ops.interp_ws_words(B, N, M1),)),
        "nn_ws2": (I32, (ops.three_nn_ws_words(B, M2, M1),)), "nn_ws1": (I32, (ops.three_nn_ws_words(B, M1, N),)),
        "rows0": (F32, (B * N, 12)), "p2_pix": (I32, (B * N,)), "p2_mm": (F32, (B, 4)),
    }


def _act_shapes(B, N, M1, M2, act_bf16):
    W = 260                                            # 16 + 16 + 32 + 64 + 64 + 34 + 34: the seven blocks' widths
    return {
        "aux": (F32, (4 * W,)), "stats": (F32, (_lib.STAT_SLOTS * 2 * W,)),
        "ext1": (F32, (B * M1, 16)), "arg1": (I32, (B * M1, 16)), "x1": (F32, (B * M1, 16)),
        "ext2": (F32, (B * M2, 32)), "arg2": (I32, (B * M2, 32)), "x2": (F32, (B * M2, 32)),
        "h_sa3": (F32, (B * M2, 64)), "h3": (F32, (B * M2, 64)), "x3": (F32, (B, 64)), "arg3": (I32, (B, 64)),
        "h2": (F32, (B * M1, 36)), "h1": (BF16 if act_bf16 else F32, (B * N, 36)),
        "src_ws1": (F32, (B * ops.interp_chunks(N, M1) * 36,)), "src_ws2": (F32, (B * ops.interp_chunks(M1, M2) * 36,)),
    }


def _offsets(struct, shapes):
    """name -> (byte offset, dtype, shape) of the fields a carve call filled (NULL fields are left out)."""
    out = {}
    for name, (dt, shape) in shapes.items():
        v = getattr(struct, name)
        if v is not None:
            out[name] = (int(v) - _FAKE_BASE, dt, shape)
    return out


def _view(arena, off, dt, shape):
    n = _ITEM[dt]
    for s in shape:
        n *= s
    return arena[off:off + n].view(dt).view(shape)


class Plan:
    """Everything that depends on (model settings, B, N) only: the dims struct, arena sizes and field offsets."""

    def __init__(self, model, ms, B, N, max_neighbors):
        lib = _lib.load()
        M1, M2 = model._sizes(N)
        self.B, self.N, self.M1, self.M2 = B, N, M1, M2
        d = NetDims()
        d.B, d.N, d.M1, d.M2 = B, N, M1, M2
        d.cap1, d.cap2 = min(max_neighbors, N), min(max_neighbors, M1)
        d.act_bf16 = int(model._act_dtype(B * N) == BF16)
        d.p2_diam_pix = int(model.p2_diam_pix) if model.p2_diam_pix is not None else 0
        self.dims = d
        sz, sz2 = ctypes.c_size_t(), ctypes.c_size_t()
        g = NetGeo()
        _lib.check(lib.sn2_net_geo_carve(byref(ms), byref(d), _FAKE_BASE, byref(g), byref(sz)), "sn2_net_geo_carve")
        self.geo_bytes = int(sz.value)
        self.geo_offsets = _offsets(g, _geo_shapes(B, N, M1, M2, d.cap1, d.cap2))
        self.act_bytes, self.act_offsets = {}, {}
        for training in (0, 1):
            a = NetAct()
            _lib.check(lib.sn2_net_act_carve(byref(ms), byref(d), training, _FAKE_BASE, byref(a), byref(sz)), "sn2_net_act_carve")
            self.act_bytes[training] = int(sz.value)
            self.act_offsets[training] = _offsets(a, _act_shapes(B, N, M1, M2, d.act_bf16))
        b = NetBwd()
        _lib.check(lib.sn2_net_bwd_carve(byref(ms), byref(d), _FAKE_BASE, _FAKE_BASE, byref(b), byref(sz), byref(sz2)), "sn2_net_bwd_carve")
        self.bwd_arena_words, self.bwd_scratch_bytes = int(sz.value) // 4, int(sz2.value)


_ZEROS = {}


def _pos3(dev, B):
    """(B,3,1) zeros: the position of every plot's global feature (model/point_net2.py:41), a constant nobody writes."""
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), B)
    z = _ZEROS.get(key)
    if z is None:
        if len(_ZEROS) > 16:
            _ZEROS.clear()
        z = _ZEROS[key] = torch.zeros(B, 3, 1, dtype=F32, device=dev)
    return z


class ArenaGeometry:
    """The position-only tables of one batch in ONE device allocation (what `PointNet2._geometry` fills and the feature passes
    read).  Same attribute names as the per-tensor handle of the per-call path (idx1, nbr1, cnt1, knn1 = (idx, w), inv1, rows0,
    ...): each is a view made when first asked for."""

    def __init__(self, plan, dev, model):
        self.plan = plan
        self.B, self.N, self.M1, self.M2 = plan.B, plan.N, plan.M1, plan.M2
        self.arena = torch.empty(plan.geo_bytes, dtype=U8, device=dev)
        self.pos3 = _pos3(dev, plan.B)
        self.xyz = None
        self.has_rows0, self.has_inverted = False, False
        self.ready, self._join = None, None
        self.p2_diam_pix = None
        g = NetGeo()
        base = self.arena.data_ptr()
        for name, (off, _, _) in plan.geo_offsets.items():
            setattr(g, name, base + off)
        g.pos3 = self.pos3.data_ptr()
        # every point's position along the plot's Morton curve (left by the level-1 FPS in its workspace): FP1's row order
        self._want_rank1 = bool(model.fp1_morton_rows and "ws1" in plan.geo_offsets and model._fp1_source_side(plan.B * plan.N))
        if self._want_rank1:
            g.rank1 = self.rank1.data_ptr()
        self.cgeo = g

    _PAIRS = {"knn3": ("knn3_idx", "knn3_w"), "knn2": ("knn2_idx", "knn2_w"), "knn1": ("knn1_idx", "knn1_w")}

    def __getattr__(self, name):                      # only reached for names not yet in __dict__
        d = self.__dict__
        plan = d.get("plan")
        if plan is None:
            raise AttributeError(name)
        spec = plan.geo_offsets.get(name)
        if spec is not None:
            v = _view(d["arena"], *spec)
        elif name in self._PAIRS:
            v = tuple(getattr(self, n) for n in self._PAIRS[name])
        elif name == "nn_ws":
            v = tuple(getattr(self, n) if n in plan.geo_offsets else None for n in ("nn_ws2", "nn_ws1"))
        elif name == "rank1":
            v = ops.fps_ws_rank(self.ws1, self.B, self.N) if d.get("_want_rank1") else None
        elif name in ("ws1", "ws2", "nn_ws1", "nn_ws2", "p2_pix", "p2_mm"):
            v = None                                   # a table this shape does not have
        else:
            raise AttributeError(name)
        d[name] = v
        return v


def geo_struct(g):
    """The sn2_net_geo of a geometry handle: an ArenaGeometry carries it; a handle of separate tensors (the per-batch views of a
    grouped pass, `alloc_geometry_pair`) gets one built from its tensors' addresses, once -- its buffers are persistent."""
    cg = getattr(g, "cgeo", None)
    if cg is not None:
        return cg
    cg = NetGeo()
    p = lambda t: None if t is None else t.data_ptr()          # noqa: E731
    for n in ("idx1", "pos1_soa", "pos1_aos", "nbr1", "cnt1", "tot1", "ord1", "idx2", "pos2_soa", "pos2_aos", "nbr2", "cnt2", "tot2",
              "ord2", "inv3", "inv2", "inv1", "rows0"):
        setattr(cg, n, p(getattr(g, n)))
    for n in ("ws1", "ws2", "rank1", "p2_pix", "p2_mm"):
        setattr(cg, n, p(getattr(g, n, None)))
    nn_ws = getattr(g, "nn_ws", None) or (None, None)
    cg.nn_ws2, cg.nn_ws1 = p(nn_ws[0]), p(nn_ws[1])
    for k in ("knn3", "knn2", "knn1"):
        idx, w = getattr(g, k)
        setattr(cg, k + "_idx", idx.data_ptr())
        setattr(cg, k + "_w", w.data_ptr())
    pos3 = getattr(g, "pos3", None)
    if pos3 is None:
        pos3 = g.pos3 = _pos3(g.idx1.device, g.B)
    cg.pos3 = pos3.data_ptr()
    g.cgeo = cg
    return cg


def _check_handle(g, plan, what):
    """A handle's buffers must have been laid out for this batch shape AND this neighbour cap (the lists' row stride)."""
    d = plan.dims
    ok = (g.B, g.N, g.M1, g.M2) == (d.B, d.N, d.M1, d.M2)
    if ok:
        gp = getattr(g, "plan", None)
        if gp is not None:
            ok = gp is plan or (gp.dims.cap1, gp.dims.cap2, gp.geo_bytes) == (d.cap1, d.cap2, plan.geo_bytes) or \
                (gp.dims.cap1, gp.dims.cap2) == (d.cap1, d.cap2) and gp.dims.p2_diam_pix >= d.p2_diam_pix
        else:
            ok = g.nbr1.shape[1] == d.cap1 and g.nbr2.shape[1] == d.cap2
    if not ok:
        raise ValueError(f"{what} do not match this batch")


class NetSaved:
    """What a forward pass keeps for its backward pass: the geometry handle, ONE activation arena and the two C structs over
    them.  Buffers read as attributes (h1, x3, arg1, ... and, through the handle, knn1, tot1, ...) are views made on demand."""

    def __init__(self, plan, geo, arena, cact, training):
        self.plan, self.geo, self.arena, self.cact, self.training = plan, geo, arena, cact, training
        self.B, self.N, self.M1, self.M2 = plan.B, plan.N, plan.M1, plan.M2

    def __getattr__(self, name):
        d = self.__dict__
        plan = d.get("plan")
        if plan is None:
            raise AttributeError(name)
        spec = plan.act_offsets[1 if d["training"] else 0].get(name)
        if spec is not None:
            v = d[name] = _view(d["arena"], *spec)
            return v
        return getattr(d["geo"], name)


# ---------------------------------------------------------------------------------------------------------- the model
_BLOCKS = ("sa1_module.conv.local_nn.0", "sa1_module.conv.local_nn.1", "sa2_module.conv.local_nn.0", "sa3_module.nn.0",
           "fp3_module.nn.0", "fp2_module.nn.0", "fp1_module.nn.0")


class ModelStruct:
    """sn2_net_model of a PointNet2 + what the host needs beside it (the flat gradient's layout).  Valid as long as `key` (the
    addresses of every parameter and buffer, and the settings the struct carries) is unchanged."""

    def __init__(self, model, params, max_neighbors):
        offs, n_flat = ops.flat_layout(params)
        off_of = {id(p): o for p, o in zip(params, offs)}
        self.param_offsets, self.n_flat = offs, n_flat
        self.param_shapes = [tuple(p.shape) for p in params]
        self.param_numels = [p.numel() for p in params]
        m = NetModel()
        mods = dict(model.named_modules())
        layers = [m.sa1[0], m.sa1[1], m.sa2, m.sa3, m.fp3, m.fp2, m.fp1]
        self.tensors = []
        for L, name in zip(layers, _BLOCKS):
            blk = mods[name]
            lin, bn = blk[0], blk[2]
            prefix = name.rsplit(".", 1)[0]
            for t in (lin.weight, lin.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var):
                ops._chk(t, F32, None, name)
            L.cin, L.cout = lin.in_features, lin.out_features
            L.W, L.b, L.gamma, L.beta = lin.weight.data_ptr(), lin.bias.data_ptr(), bn.weight.data_ptr(), bn.bias.data_ptr()
            L.running_mean, L.running_var = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
            nbt = bn.num_batches_tracked
            if nbt is not None:
                ops._chk(nbt, I64, None, "num_batches_tracked")
            L.num_batches_tracked = None if nbt is None else nbt.data_ptr()
            L.gW, L.gb, L.ggamma, L.gbeta = (off_of[id(t)] for t in (lin.weight, lin.bias, bn.weight, bn.bias))
            L.mma_bf16 = int(model.mma_dtype == "bf16" and prefix in model.BF16_BLOCKS)
            self.tensors += [lin.weight, lin.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var] + ([nbt] if nbt is not None else [])
        for t, shape in ((model.lin1.weight, (16, 34)), (model.lin1.bias, (16,)), (model.lin2.weight, (5, 16)), (model.lin2.bias, (5,))):
            ops._chk(t, F32, shape, "head parameter")
        m.lin1_W, m.lin1_b, m.lin2_W, m.lin2_b = (t.data_ptr() for t in (model.lin1.weight, model.lin1.bias, model.lin2.weight, model.lin2.bias))
        m.g_lin1_W, m.g_lin1_b, m.g_lin2_W, m.g_lin2_b = (off_of[id(t)] for t in (model.lin1.weight, model.lin1.bias, model.lin2.weight,
                                                                                model.lin2.bias))
        self.tensors += [model.lin1.weight, model.lin1.bias, model.lin2.weight, model.lin2.bias]
        m.n_flat = n_flat
        m.r1_sq, m.r2_sq = ops.r2_threshold(model.sa1_module.r), ops.r2_threshold(model.sa2_module.r)
        m.max_neighbors = int(max_neighbors)
        m.drop_p = float(model.drop)
        m.fuse_global_level, m.fuse_eval_head, m.source_side = int(bool(model.fuse_global_level)), int(bool(model.fuse_eval_head)), int(bool(ops.SOURCE_SIDE))
        m.fps_waves_shared, m.fps_waves_many = int(model.fps_waves_shared), int(model.fps_waves_many)
        self.c = m
        self.settings = settings_of(model, max_neighbors)
        self.key = tuple(t.data_ptr() for t in self.tensors)
        self.plans = {}
        dev = model.lin1.weight.device
        if any(t.device != dev for t in self.tensors):
            raise StrataHipError("PointNet2: parameters and buffers must live on one HIP device")

    def current(self, model, max_neighbors):
        return self.settings == settings_of(model, max_neighbors) and self.key == tuple(t.data_ptr() for t in self.tensors)

    def plan(self, model, B, N):
        k = (B, N)
        p = self.plans.get(k)
        if p is None:
            if len(self.plans) > 32:
                self.plans.clear()
            p = self.plans[k] = Plan(model, self.c, B, N, self.c.max_neighbors)
        return p


def settings_of(model, max_neighbors):
    return (model.mma_dtype, float(model.drop), bool(model.fuse_global_level), bool(model.fuse_eval_head), bool(ops.SOURCE_SIDE),
            int(model.fps_waves_shared), int(model.fps_waves_many), float(model.sa1_module.r), float(model.sa2_module.r),
            float(model.sa1_module.ratio), float(model.sa2_module.ratio), int(max_neighbors), model.p2_diam_pix,
            bool(model.fp1_morton_rows))


class NetCtx:
    """The events of a forked geometry pass (sn2_net_ctx_create): one per model and device, destroyed with it."""

    def __init__(self):
        h = ctypes.c_void_p()
        _lib.check(_lib.load().sn2_net_ctx_create(byref(h)), "sn2_net_ctx_create")
        self.handle = h.value
        self._fin = weakref.finalize(self, _destroy_ctx, self.handle)


def _destroy_ctx(handle):
    try:
        _lib.load().sn2_net_ctx_destroy(handle)
    except Exception:           # noqa: BLE001  (interpreter shutdown)
        pass


def _io(model, dev, training, flags, cloud=None, fps_start=None, fork=False):
    io = NetIO()
    io.cloud = None if cloud is None else cloud.data_ptr()
    io.fps_start = None if fps_start is None else fps_start.data_ptr()
    io.fps_status = ops.fps_status_word(dev).data_ptr()
    io.training = int(training) if training in (0, 1, 2) else int(bool(training))      # (2 = SN2_BN_FROZEN_KEEP)
    if fork:
        io.stream_b = ops.shared_stream(dev, "fork_b").cuda_stream
        io.stream_c = ops.shared_stream(dev, "fork_c").cuda_stream
        io.stream_pack = ops.shared_stream(dev, "pack").cuda_stream
        io.ctx = model._net_ctx().handle
    io.flags = flags
    return io


def geometry(model, ms, xyz, fps_start, out=None, fork=None, shared=False, defer_join=False, inverted=True, cloud=None):
    """`PointNet2._geometry` as one call (sn2_net_geometry)."""
    dev = xyz.device
    B, _, N = xyz.shape
    ops._chk(xyz, F32, (B, 3, N), "xyz")
    ops._chk(fps_start, I32, (2, B), "fps_start")
    plan = ms.plan(model, B, N)
    g = out if out is not None else ArenaGeometry(plan, dev, model)
    _check_handle(g, plan, "geometry buffers")
    cg = geo_struct(g)
    g.xyz = xyz
    cg.xyz = xyz.data_ptr()
    fork = model.geometry_fork if fork is None else fork
    flags = (_lib.NET_FORK if fork else 0) | (_lib.NET_SHARED if shared else 0) | (_lib.NET_INVERTED if inverted else 0)
    if fork and defer_join:
        flags |= _lib.NET_DEFER_JOIN
    if cloud is not None:
        ops._chk(cloud, F32, (B, 10, N), "cloud")
        flags |= _lib.NET_INPUT_ONLY
    io = _io(model, dev, model.training, flags, cloud=cloud, fps_start=fps_start, fork=fork)
    _lib.check(_lib.load().sn2_net_geometry(byref(ms.c), byref(plan.dims), byref(cg), byref(io), ops._stream()), "sn2_net_geometry")
    g.has_inverted = bool(inverted)
    g.has_rows0 = cloud is not None
    if cloud is not None and plan.dims.p2_diam_pix > 0 and getattr(g, "p2_pix", None) is not None:
        g.p2_diam_pix = int(plan.dims.p2_diam_pix)
    g._join = "ctx" if (fork and defer_join) else None
    return g


def forward(model, ms, xyz, cloud, fps_start, training, geo=None, drop_keep=None, need_grad=True):
    """`PointNet2._forward_impl` as one call (sn2_net_forward) -> (coverages_pointwise, proba_pointwise, NetSaved)."""
    dev = xyz.device
    B, _, N = xyz.shape
    ops._chk(xyz, F32, (B, 3, N), "xyz")
    ops._chk(cloud, F32, (B, 10, N), "cloud")
    plan = ms.plan(model, B, N)
    lib = _lib.load()
    fork = False
    # eval mode with gradients wanted (torch's BatchNorm in eval under autograd, model/point_net2.py:45-53): the forward keeps what
    # a training forward keeps, on the running statistics (SN2_BN_FROZEN_KEEP)
    frozen = (not training) and bool(need_grad)
    keep = bool(training) or frozen
    mode = 1 if training else (_lib.BN_FROZEN_KEEP if frozen else 0)
    if geo is None:
        ops._chk(fps_start, I32, (2, B), "fps_start")
        geo = ArenaGeometry(plan, dev, model)
        fork = bool(model.geometry_fork)
        flags = _lib.NET_WITH_GEOMETRY | (_lib.NET_FORK if fork else 0)
        geo.has_inverted = keep
        fs = fps_start
    else:
        _check_handle(geo, plan, "prefetched geometry tables")
        flags, fs = 0, None
        if getattr(geo, "has_rows0", False):
            flags |= _lib.NET_HAS_ROWS0
        if getattr(geo, "_join", None) is not None:
            flags |= _lib.NET_JOIN_PENDING
            geo._join = None
            fork = True                               # (the context whose events the deferred pass recorded)
        if getattr(geo, "has_inverted", True):
            flags |= _lib.NET_HAS_INVERTED
    cg = geo_struct(geo)
    geo.xyz = xyz
    cg.xyz = xyz.data_ptr()
    t = 1 if keep else 0
    arena = torch.empty(plan.act_bytes[t], dtype=U8, device=dev)
    ca = NetAct()
    base = arena.data_ptr()
    for name, (off, _, _) in plan.act_offsets[t].items():
        setattr(ca, name, base + off)
    cov = torch.empty(B * N, 4, dtype=F32, device=dev)
    proba = torch.empty(B * N, 4, dtype=F32, device=dev)
    ca.cov, ca.proba = cov.data_ptr(), proba.data_ptr()
    if drop_keep is not None:
        ops._chk(drop_keep, I32, (B * N,), "drop_mask")
        ca.drop_mask = drop_keep.data_ptr()
    bwd_arena = None
    if keep and need_grad:
        # the zero-filled arena of the backward pass that will follow: cleared by the forward's last kernel instead of a launch
        # of its own in front of that pass
        bwd_arena = torch.empty(plan.bwd_arena_words, dtype=F32, device=dev)
        ca.bwd_arena, ca.bwd_arena_words = bwd_arena.data_ptr(), plan.bwd_arena_words
    io = _io(model, dev, mode, flags, cloud=cloud, fps_start=fs, fork=fork)
    if training and model.fuse_global_level:
        ws = ops.global_level_ws(dev, owner=model)
        io.gl_xchg, io.gl_ctl = ws[0].data_ptr(), ws[1].data_ptr()
    _lib.check(lib.sn2_net_forward(byref(ms.c), byref(plan.dims), byref(cg), byref(ca), byref(io), ops._stream()), "sn2_net_forward")
    geo.has_rows0 = True
    if keep:
        geo.has_inverted = True
    s = NetSaved(plan, geo, arena, ca, keep)              # (`training` of the saved set = which activation layout it has)
    s.frozen = frozen
    s.drop_keep = drop_keep
    s.xyz = xyz
    s.ms = ms
    s.bwd_arena = bwd_arena
    return cov, proba, s


def backward(model, s, dcov, dproba):
    """`PointNet2._backward_impl` as one call (sn2_net_backward) -> the parameter gradients (views of ONE flat buffer)."""
    plan, ms = s.plan, s.ms
    dev = s.xyz.device
    R = plan.B * plan.N
    arena = s.__dict__.get("bwd_arena")                                       # cleared by the forward pass's last kernel ...
    s.bwd_arena = None                                                        # (once: a second backward over this forward clears its own)
    pre_zeroed = arena is not None
    if arena is None:
        arena = torch.empty(plan.bwd_arena_words, dtype=F32, device=dev)    # ... or zero-filled inside the call
    scratch = torch.empty(plan.bwd_scratch_bytes, dtype=U8, device=dev)
    cb = NetBwd()
    sz, sz2 = ctypes.c_size_t(), ctypes.c_size_t()
    lib = _lib.load()
    _lib.check(lib.sn2_net_bwd_carve(byref(ms.c), byref(plan.dims), arena.data_ptr(), scratch.data_ptr(), byref(cb), byref(sz), byref(sz2)),
               "sn2_net_bwd_carve")
    if dcov is not None:
        dcov = ops._chk(dcov.contiguous(), F32, (R, 4), "dcoverages")
        cb.dcov = dcov.data_ptr()
    if dproba is not None:
        dproba = ops._chk(dproba.contiguous(), F32, (R, 4), "dproba")
        cb.dproba = dproba.data_ptr()
    defer = bool(getattr(model, "defer_grad_reduce", False))
    cb.defer_grad_reduce = int(defer)
    cb.arena_is_zero = int(pre_zeroed)
    cb.frozen_stats = int(bool(s.__dict__.get("frozen", False)))
    _lib.check(lib.sn2_net_backward(byref(ms.c), byref(plan.dims), byref(geo_struct(s.geo)), byref(s.cact), byref(cb), ops._stream()),
               "sn2_net_backward")
    flat = arena[:ms.n_flat]
    s.flat_grad = flat
    model._last_flat_grad = flat
    model._grad_images_pending = (arena, cb.images, cb.image_stride) if defer else None
    # (one split call instead of 32 slices: the views are made in C++)
    return [p if len(shape) == 1 else p.view(shape) for p, shape in zip(flat.split(ms.param_numels), ms.param_shapes)]
