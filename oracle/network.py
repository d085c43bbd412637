"""Functional CPU restatement of the reference `PointNet2.forward`.  TEST INFRASTRUCTURE ONLY.

Follows `/root/reference/model/point_net2.py`:
    SAModule.forward        :21-29     fps -> radius(cap 2000) -> PointConv(max) -> (x, pos[idx], batch[idx])
    GlobalSAModule.forward  :37-42     MLP(cat[x, pos]) -> global max pool
    MLP                     :45-53     (Linear -> ReLU -> BatchNorm1d) blocks, BN AFTER ReLU
    FPModule.forward        :62-67     knn_interpolate -> cat[interp, skip] -> MLP
    PointNet2.__init__      :71-104    layer sizes, lin2 bias constant
    PointNet2.forward       :106-153   long form, drop x/y, SA1-3, FP3-1, head
Pinned against goldens produced by running that very code (oracle/make_golden.py).

Unlike the reference (module objects) this is a pure function of a `state_dict`-shaped dict of tensors, so that
autograd yields the gradient of every parameter and the BatchNorm running statistics come back explicitly.
"""
from collections import OrderedDict
from typing import Dict, Optional, Sequence

import torch
import torch.nn.functional as F

from . import primitives as P

MAX_NUM_NEIGHBORS = 2000  # model/point_net2.py:24

# (prefix, [channel sizes]) in the construction order of model/point_net2.py:81-96
LAYERS = OrderedDict([
    ("sa1_module.conv.local_nn", [11, 16, 16]),
    ("sa2_module.conv.local_nn", [19, 32]),
    ("sa3_module.nn", [35, 64]),
    ("fp3_module.nn", [96, 64]),
    ("fp2_module.nn", [80, 34]),
    ("fp1_module.nn", [42, 34]),
])
LIN2_BIAS = [0.733, 0.266, 0.235, 0.358, 0.500]  # model/point_net2.py:97-99


def init_state_dict(seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    """Default-initialised weights with the key names/shapes of the reference `state_dict()`.
    Construction order equals the reference's so `torch.manual_seed(seed)` yields the same numbers."""
    torch.manual_seed(seed)
    sd = OrderedDict()
    for prefix, ch in LAYERS.items():
        for i in range(1, len(ch)):
            lin = torch.nn.Linear(ch[i - 1], ch[i])
            bn = torch.nn.BatchNorm1d(ch[i])
            for k, v in lin.state_dict().items():
                sd[f"{prefix}.{i - 1}.0.{k}"] = v.detach().clone()
            for k, v in bn.state_dict().items():
                sd[f"{prefix}.{i - 1}.2.{k}"] = v.detach().clone()
    lin1 = torch.nn.Linear(34, 16)
    lin2 = torch.nn.Linear(16, 5)
    sd["lin1.weight"], sd["lin1.bias"] = lin1.weight.detach().clone(), lin1.bias.detach().clone()
    sd["lin2.weight"] = lin2.weight.detach().clone()
    sd["lin2.bias"] = torch.tensor(LIN2_BIAS)
    return sd


def param_keys(sd) -> Sequence[str]:
    return [k for k in sd if not (k.endswith("running_mean") or k.endswith("running_var")
                                  or k.endswith("num_batches_tracked"))]


def _bf16(t):
    return t.to(torch.bfloat16).to(t.dtype)          # round to nearest even, as v_cvt_pk_bf16_f32


class _LinearBF16(torch.autograd.Function):
    """A Linear layer whose three contractions take bfloat16 operands and accumulate wider -- the checker of the build's
    bf16 variant (sn2_block.mma_bf16; BASELINE.json configs[4]; NOT in the reference): y = r(x) r(W)^T + b,
    dx = r(dy) r(W), dW = r(dy)^T r(x), db = sum(dy), r = rounding to bfloat16."""

    @staticmethod
    def forward(ctx, x, W, b):
        xr, Wr = _bf16(x), _bf16(W)
        ctx.save_for_backward(xr, Wr)
        return xr @ Wr.t() + b

    @staticmethod
    def backward(ctx, dy):
        xr, Wr = ctx.saved_tensors
        dyr = _bf16(dy)
        return dyr @ Wr, dyr.t() @ xr, dy.sum(0)


class _RoundSTE(torch.autograd.Function):
    """A tensor that is STORED in bfloat16 on its way forward (rounded to nearest even; the gradient passes unchanged)."""

    @staticmethod
    def forward(ctx, x):
        return _bf16(x)

    @staticmethod
    def backward(ctx, g):
        return g


class _GradRound(torch.autograd.Function):
    """A tensor whose GRADIENT is stored in bfloat16 on its way back (the forward value passes unchanged)."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return _bf16(g)


def _fp1_block(f, x0, sd, training, new_stats, act_bf16):
    """FP1's (Linear, ReLU, BatchNorm) block on cat[interpolated, skip] (`model/point_net2.py:62-67,90,93`).  act_bf16: the
    checker of the build's bfloat16 ACTIVATION STORAGE on the per-point layer (sn2_fp.act_bf16; BASELINE.json configs[4];
    not in the reference): the block's output rows (after the ReLU, before the BatchNorm, whose batch statistics describe
    the stored rows), the gradient of the BatchNorm's output, and the gradient of the pre-activation on its way to the
    interpolation's sources are each rounded to bfloat16 once; sums, statistics and weight gradients stay wide."""
    if not act_bf16:
        return _mlp(torch.cat([f, x0], dim=1), sd, "fp1_module.nn", 1, training, new_stats)
    prefix = "fp1_module.nn"
    W, b = sd[f"{prefix}.0.0.weight"], sd[f"{prefix}.0.0.bias"]
    ca = f.shape[1]
    tA = _GradRound.apply(F.linear(f, W[:, :ca]))                 # d pre-activation -> sources through bfloat16 rows
    a = _RoundSTE.apply(F.relu(tA + F.linear(x0, W[:, ca:]) + b))  # h1 as stored
    rm = sd[f"{prefix}.0.2.running_mean"].clone()
    rv = sd[f"{prefix}.0.2.running_var"].clone()
    if BN_STATS_F64 and training:
        y = _batch_norm_f64_stats(a, rm, rv, sd[f"{prefix}.0.2.weight"], sd[f"{prefix}.0.2.bias"])
    else:
        y = F.batch_norm(a, rm, rv, sd[f"{prefix}.0.2.weight"], sd[f"{prefix}.0.2.bias"], training=training, momentum=0.1, eps=1e-5)
    if new_stats is not None:
        new_stats[f"{prefix}.0.2.running_mean"] = rm
        new_stats[f"{prefix}.0.2.running_var"] = rv
    return _GradRound.apply(y)                                     # dy1 as stored


BF16_LAYERS = ()      # prefixes of the blocks evaluated with bf16 operands (set through `forward(..., bf16_layers=)`)
BN_STATS_F64 = False  # training-mode BatchNorm: batch mean / variance summed in fp64, everything else in the input's dtype
                      # (set through `forward(..., bn_stats_f64=True)`; isolates torch's fp32 batch-statistics error)


def _batch_norm_f64_stats(x, rm, rv, gamma, beta, momentum=0.1, eps=1e-5):
    """torch BatchNorm1d in training mode (biased variance to normalise, unbiased into running_var) with ONLY the two
    batch statistics evaluated in fp64: mean and variance are summed in double and rounded to x's dtype once, the
    normalisation, the affine and every gradient stay in x's dtype."""
    xd = x.double()
    mean = xd.mean(0)
    var = (xd - mean).pow(2).mean(0)
    n = x.shape[0]
    with torch.no_grad():
        rm.mul_(1 - momentum).add_(momentum * mean.to(rm.dtype))
        rv.mul_(1 - momentum).add_(momentum * (var * (n / max(n - 1, 1))).to(rv.dtype))
    invstd = (var + eps).rsqrt().to(x.dtype)
    return (x - mean.to(x.dtype)) * invstd * gamma + beta


def _mlp(x, sd, prefix, n_blocks, training, new_stats):
    """(Linear -> ReLU -> BN) x n_blocks, `model/point_net2.py:45-53`; torch BatchNorm1d defaults."""
    for i in range(n_blocks):
        if prefix in BF16_LAYERS:
            x = _LinearBF16.apply(x, sd[f"{prefix}.{i}.0.weight"], sd[f"{prefix}.{i}.0.bias"])
        else:
            x = F.linear(x, sd[f"{prefix}.{i}.0.weight"], sd[f"{prefix}.{i}.0.bias"])
        x = F.relu(x)
        rm = sd[f"{prefix}.{i}.2.running_mean"].clone()
        rv = sd[f"{prefix}.{i}.2.running_var"].clone()
        if BN_STATS_F64 and training:
            x = _batch_norm_f64_stats(x, rm, rv, sd[f"{prefix}.{i}.2.weight"], sd[f"{prefix}.{i}.2.bias"])
        else:
            x = F.batch_norm(x, rm, rv, sd[f"{prefix}.{i}.2.weight"], sd[f"{prefix}.{i}.2.bias"],
                             training=training, momentum=0.1, eps=1e-5)
        if new_stats is not None:
            new_stats[f"{prefix}.{i}.2.running_mean"] = rm
            new_stats[f"{prefix}.{i}.2.running_var"] = rv
    return x


def _fps_regular(pos_long, B, n, ratio, start):
    m = P.fps_num_samples(n, ratio)
    loc = P.fps_batched(pos_long.view(B, n, 3), m, start)
    return (loc + (torch.arange(B) * n).unsqueeze(1)).reshape(-1), m


def forward(sd: Dict[str, torch.Tensor], cloud: torch.Tensor, xyz: torch.Tensor, args, training: bool,
            fps_start: Optional[Sequence[torch.Tensor]] = None, use_kdtree: bool = False, details: bool = False,
            dropout_mask: Optional[torch.Tensor] = None, bf16_layers: Sequence[str] = (), bn_stats_f64: bool = False,
            act_bf16: bool = False):
    """cloud (B,10,N), xyz (B,3,N) fp32 CPU tensors (the DataLoader collate of `loader.py:73-87`).
    fps_start = (start1 (B,), start2 (B,)) LOCAL start indices of the two FPS calls (None -> 0).
    dropout_mask (B*N,16), non-zero = keep: the mask F.dropout (point_net2.py:142) would have drawn (None: torch draws).
    bf16_layers: prefixes of the (Linear, ReLU, BN) stacks whose Linear layers take bfloat16 operands (`_LinearBF16`).
    bn_stats_f64: training-mode BatchNorm statistics summed in fp64, everything else in `cloud`'s dtype (`BN_STATS_F64`).
    act_bf16: the per-point activation buffers pass through bfloat16 storage (`_fp1_block`).
    Returns (coverages_pointwise (B*N,4), proba_pointwise (B*N,4), extras) where extras holds the new BN
    running statistics (training) and, with details=True, the intermediate tensors."""
    global BF16_LAYERS, BN_STATS_F64
    BF16_LAYERS, BN_STATS_F64 = tuple(bf16_layers), bool(bn_stats_f64)
    B, _, N = cloud.shape
    # long form (B*N, f), plot-major (point_net2.py:155-158), drop normalised x,y (:118)
    pos0 = xyz.permute(0, 2, 1).reshape(B * N, 3).contiguous()
    x0 = cloud.permute(0, 2, 1).reshape(B * N, -1)[:, 2:].contiguous()
    batch0 = torch.arange(B).repeat_interleave(N)
    s1 = torch.zeros(B, dtype=torch.long) if fps_start is None else fps_start[0].long()
    s2 = torch.zeros(B, dtype=torch.long) if fps_start is None else fps_start[1].long()
    new_stats = {} if training else None
    ex = {}

    # ---- SA1 (point_net2.py:131, 21-29)
    idx1, M1 = _fps_regular(pos0, B, N, args.ratio1, s1)
    pos1, batch1 = pos0[idx1], batch0[idx1]
    row, col = P.radius(pos0, pos1, args.r1, batch0, batch1, max_num_neighbors=MAX_NUM_NEIGHBORS,
                        use_kdtree=use_kdtree)
    msg = torch.cat([x0[col], pos0[col] - pos1[row]], dim=1)
    msg = _mlp(msg, sd, "sa1_module.conv.local_nn", 2, training, new_stats)
    x1 = P.scatter_max(msg, row, dim=0, dim_size=pos1.shape[0])[0]
    ex.update(idx1=idx1, row1=row, col1=col)

    # ---- SA2 (:132)
    idx2, M2 = _fps_regular(pos1, B, M1, args.ratio2, s2)
    pos2, batch2 = pos1[idx2], batch1[idx2]
    row, col = P.radius(pos1, pos2, args.r2, batch1, batch2, max_num_neighbors=MAX_NUM_NEIGHBORS,
                        use_kdtree=use_kdtree)
    msg = torch.cat([x1[col], pos1[col] - pos2[row]], dim=1)
    msg = _mlp(msg, sd, "sa2_module.conv.local_nn", 1, training, new_stats)
    x2 = P.scatter_max(msg, row, dim=0, dim_size=pos2.shape[0])[0]
    ex.update(idx2=idx2, row2=row, col2=col)

    # ---- SA3: global (:133, 37-42)
    h = _mlp(torch.cat([x2, pos2], dim=1), sd, "sa3_module.nn", 1, training, new_stats)
    x3 = P.global_max_pool(h, batch2)
    pos3 = pos2.new_zeros((B, 3))
    batch3 = torch.arange(B)

    # ---- FP3 (k=1), FP2 (k=3), FP1 (k=3)  (:137-139, 62-67)
    f = P.knn_interpolate(x3, pos3, pos2, batch3, batch2, k=1)
    f3 = _mlp(torch.cat([f, x2], dim=1), sd, "fp3_module.nn", 1, training, new_stats)
    f = P.knn_interpolate(f3, pos2, pos1, batch2, batch1, k=3, use_kdtree=use_kdtree)
    f2 = _mlp(torch.cat([f, x1], dim=1), sd, "fp2_module.nn", 1, training, new_stats)
    f = P.knn_interpolate(f2, pos1, pos0, batch1, batch0, k=3, use_kdtree=use_kdtree)
    f1 = _fp1_block(f, x0, sd, training, new_stats, act_bf16)

    # ---- head (:141-151)
    h = F.relu(F.linear(f1, sd["lin1.weight"], sd["lin1.bias"]))
    if dropout_mask is not None and training and args.drop > 0:
        # F.dropout with the Bernoulli(1-p) keep-mask handed in (so that the checker and the kernels see the same mask):
        # kept elements are scaled by 1/(1-p), the others are zero (torch.nn.functional.dropout; p = 1 gives zeros)
        h = h * (dropout_mask != 0).to(h.dtype) * (1.0 / (1.0 - args.drop) if args.drop < 1 else 0.0)
    else:
        h = F.dropout(h, p=args.drop, training=training)
    scores = F.linear(h, sd["lin2.weight"], sd["lin2.bias"])
    proba = torch.softmax(scores[:, :4], dim=1)
    density = torch.sigmoid(scores[:, 4:5])
    coverages = proba * density

    ex["new_stats"] = new_stats
    if details:
        ex.update(pos1=pos1, pos2=pos2, x1=x1, x2=x2, x3=x3, f3=f3, f2=f2, f1=f1, scores=scores)
    return coverages, proba, ex


# ---------------------------------------------------------------------------------------------------------------------
# The "3sa-arch" variant (NOT in the reference: BASELINE.json's configuration 2 names three ball-query levels; SURVEY.md 8d
# asks for it next to the reference architecture).  Same primitives, one more SAModule before the global one and one more
# FPModule; the checker of stratanet2_vegetation_coverage_maps_amd/point_net2_3sa.py.
# ---------------------------------------------------------------------------------------------------------------------
LAYERS_3SA = OrderedDict([
    ("sa1_module.conv.local_nn", [11, 16, 16]),
    ("sa2_module.conv.local_nn", [19, 32]),
    ("sa3_module.conv.local_nn", [35, 64]),
    ("sa4_module.nn", [67, 64]),
    ("fp4_module.nn", [128, 64]),
    ("fp3_module.nn", [96, 64]),
    ("fp2_module.nn", [80, 34]),
    ("fp1_module.nn", [42, 34]),
])


def init_state_dict_3sa(seed: int = 0):
    torch.manual_seed(seed)
    sd = OrderedDict()
    for prefix, ch in LAYERS_3SA.items():
        for i in range(1, len(ch)):
            lin, bn = torch.nn.Linear(ch[i - 1], ch[i]), torch.nn.BatchNorm1d(ch[i])
            for k, v in lin.state_dict().items():
                sd[f"{prefix}.{i - 1}.0.{k}"] = v.detach().clone()
            for k, v in bn.state_dict().items():
                sd[f"{prefix}.{i - 1}.2.{k}"] = v.detach().clone()
    lin1, lin2 = torch.nn.Linear(34, 16), torch.nn.Linear(16, 5)
    sd["lin1.weight"], sd["lin1.bias"] = lin1.weight.detach().clone(), lin1.bias.detach().clone()
    sd["lin2.weight"] = lin2.weight.detach().clone()
    sd["lin2.bias"] = torch.tensor(LIN2_BIAS)
    return sd


def forward_3sa(sd, cloud, xyz, args, training, fps_start=None, use_kdtree=False, bf16_layers=(), bn_stats_f64=False,
                act_bf16=False):
    global BF16_LAYERS, BN_STATS_F64
    BF16_LAYERS, BN_STATS_F64 = tuple(bf16_layers), bool(bn_stats_f64)
    B, _, N = cloud.shape
    pos0 = xyz.permute(0, 2, 1).reshape(B * N, 3).contiguous()
    x0 = cloud.permute(0, 2, 1).reshape(B * N, -1)[:, 2:].contiguous()
    batch0 = torch.arange(B).repeat_interleave(N)
    starts = [torch.zeros(B, dtype=torch.long) if fps_start is None else fps_start[i].long() for i in range(3)]
    new_stats = {} if training else None
    levels = [(pos0, batch0, x0, N)]
    for lvl, (ratio, r, prefix, nb) in enumerate(((args.ratio1, args.r1, "sa1_module.conv.local_nn", 2),
                                                  (args.ratio2, args.r2, "sa2_module.conv.local_nn", 1),
                                                  (args.ratio3, args.r3, "sa3_module.conv.local_nn", 1))):
        ps, bs, xs, n = levels[-1]
        idx, M = _fps_regular(ps, B, n, ratio, starts[lvl])
        pc, bc = ps[idx], bs[idx]
        row, col = P.radius(ps, pc, r, bs, bc, max_num_neighbors=MAX_NUM_NEIGHBORS, use_kdtree=use_kdtree)
        msg = _mlp(torch.cat([xs[col], ps[col] - pc[row]], dim=1), sd, prefix, nb, training, new_stats)
        levels.append((pc, bc, P.scatter_max(msg, row, dim=0, dim_size=pc.shape[0])[0], M))
    (pos1, batch1, x1, _), (pos2, batch2, x2, _), (pos3, batch3, x3, _) = levels[1:]
    h = _mlp(torch.cat([x3, pos3], dim=1), sd, "sa4_module.nn", 1, training, new_stats)
    xg = P.global_max_pool(h, batch3)
    posg, batchg = pos3.new_zeros((B, 3)), torch.arange(B)
    f = P.knn_interpolate(xg, posg, pos3, batchg, batch3, k=1)
    f4 = _mlp(torch.cat([f, x3], dim=1), sd, "fp4_module.nn", 1, training, new_stats)
    f = P.knn_interpolate(f4, pos3, pos2, batch3, batch2, k=3, use_kdtree=use_kdtree)
    f3 = _mlp(torch.cat([f, x2], dim=1), sd, "fp3_module.nn", 1, training, new_stats)
    f = P.knn_interpolate(f3, pos2, pos1, batch2, batch1, k=3, use_kdtree=use_kdtree)
    f2 = _mlp(torch.cat([f, x1], dim=1), sd, "fp2_module.nn", 1, training, new_stats)
    f = P.knn_interpolate(f2, pos1, pos0, batch1, batch0, k=3, use_kdtree=use_kdtree)
    f1 = _fp1_block(f, x0, sd, training, new_stats, act_bf16)
    h = F.relu(F.linear(f1, sd["lin1.weight"], sd["lin1.bias"]))
    scores = F.linear(h, sd["lin2.weight"], sd["lin2.bias"])
    proba = torch.softmax(scores[:, :4], dim=1)
    return proba * torch.sigmoid(scores[:, 4:5]), proba, {"new_stats": new_stats}
