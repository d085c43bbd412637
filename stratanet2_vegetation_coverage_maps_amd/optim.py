"""Flat-buffer optimiser and data-parallel gradient exchange for the timed training step.

The reference's harness builds `optim.Adam(model.parameters(), lr=args.lr, weight_decay=args.wd)`
(`/root/reference/learning/train.py:180-185`).  `FlatAdam` is the same update rule (amsgrad off, L2 decay folded into
the gradient, bias correction) as ONE kernel over the model's 14 997 parameters, which `flatten_parameters` re-homes
into a single contiguous buffer (every nn.Parameter becomes a view of it, so `state_dict()` / `load_state_dict()` and
any torch optimiser keep working).  The backward pass already writes all gradients into one flat buffer
(`PointNet2._last_flat_grad`), so data parallelism is exactly one `all_reduce` of 60 KB per step.
"""
import torch

from . import hip_ops as ops


def shard_of_rank(rank: int, plots_per_rank: int):
    """(first plot, number of plots) of a rank: the global batch of `world * plots_per_rank` plots is cut into contiguous
    shards; plots are independent units (per-GPU BatchNorm statistics, as torch DDP without SyncBN)."""
    return rank * plots_per_rank, plots_per_rank


def allreduce_flat_grad(flat_grad: torch.Tensor, world_size: int, group=None, comm=None, force: bool = False) -> float:
    """The only exchange step of the data-parallel path: one SUM all-reduce of the flat gradient buffer (14 997 fp32 =
    60 KB, latency-bound on xGMI).  Returns the 1/world scale the optimiser kernel applies afterwards.
    comm: an `rccl.RcclComm` -- `ncclAllReduce` on torch's CURRENT stream ("on the compute stream"; inside a
    graph capture it becomes a node of the step's hipGraph), run at ANY world size including 1; None: torch's process group
    (its own stream and event hand-offs; skipped at world 1 unless `force`: a one-rank process group then runs the call, which
    is how bench.py times this path on a one-GPU box)."""
    if comm is not None:
        comm.all_reduce_sum_(flat_grad)
        return 1.0 / comm.world
    if world_size > 1 or force:
        torch.distributed.all_reduce(flat_grad, op=torch.distributed.ReduceOp.SUM, group=group)
    return 1.0 / world_size


def broadcast_bn_buffers(model, world_size: int, src: int = 0, group=None) -> int:
    """Optional second exchange of the data-parallel path (torch DDP's `broadcast_buffers`): every rank takes
    rank `src`'s BatchNorm running statistics -- 520 floats + 7 counters, one small broadcast.  The training step does not
    need it (batch statistics are per GPU, as DDP without SyncBN; the weights stay identical through the all-reduced
    gradient), but without it the replicas' EVAL-mode outputs drift apart, because each rank's running statistics follow
    its own shard.  Call it every K steps or before evaluating / checkpointing from a rank other than `src`.
    Returns the number of floats exchanged."""
    bufs = [b for n, b in model.named_buffers() if n.endswith("running_mean") or n.endswith("running_var")]
    cnts = [b for n, b in model.named_buffers() if n.endswith("num_batches_tracked")]
    if world_size <= 1 or not bufs:
        return 0
    flat = torch.cat([b.reshape(-1).float() for b in bufs])
    torch.distributed.broadcast(flat, src=src, group=group)
    o = 0
    for b in bufs:
        b.copy_(flat[o:o + b.numel()].view_as(b))
        o += b.numel()
    n = int(flat.numel())
    if cnts:
        # the int64 `num_batches_tracked` counters travel as int64 (a float32 carries integers only up to 2^24)
        ic = torch.stack([c.reshape(()).to(torch.int64) for c in cnts])
        torch.distributed.broadcast(ic, src=src, group=group)
        for c, v in zip(cnts, ic):
            c.copy_(v.to(c.dtype))
        n += int(ic.numel())
    return n


def flatten_parameters(model) -> torch.Tensor:
    params = list(model.parameters())
    offs, n = ops.flat_layout(params)
    flat = torch.zeros(n, dtype=params[0].dtype, device=params[0].device)
    for p, o in zip(params, offs):
        flat[o:o + p.numel()].copy_(p.detach().reshape(-1))
        p.data = flat[o:o + p.numel()].view(p.shape)
    model._flat_params = flat
    return flat


class FlatAdam:
    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, process_group=None,
                 world_size=1, comm=None, fold_gradient_images=False):
        """fold_gradient_images: with no exchange between backward and update (one rank, no communicator) let THIS step's kernel
        fold the 32 images of the flat gradient (sn2_adam_step_images) instead of a launch of its own at the end of the backward
        pass: `model.defer_grad_reduce = True`; the parameters' `.grad` views hold the whole gradient after `step()` (before
        it: image 0 only).  Ignored when an exchange needs the folded gradient first."""
        self.model = model
        self.flat = getattr(model, "_flat_params", None)
        if self.flat is None:
            self.flat = flatten_parameters(model)
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        # on the device (graph-replay safe): {steps taken so far, the kernel's arrival ticket}
        self.step_words = torch.zeros(2, dtype=torch.int32, device=self.flat.device)
        self.step_dev = self.step_words[:1]
        self.world_size = world_size
        self.process_group = process_group
        self.comm = comm                      # rccl.RcclComm: the exchange as ncclAllReduce on the step's own stream (graph-capturable)
        self.force_exchange = False           # torch's all_reduce even at world 1 (needs an initialised one-rank process group)
        if comm is not None and comm.world != world_size:
            raise ValueError("FlatAdam: the RCCL communicator and world_size disagree")
        self.fold_gradient_images = bool(fold_gradient_images) and world_size == 1 and comm is None
        if self.fold_gradient_images:
            model.defer_grad_reduce = True

    def reset(self):
        """Forget the optimiser state: moments, step count AND the Adam kernel's arrival ticket (`step_words[1]`: the last
        workgroup of a launch advances `step_words[0]` and zeroes the ticket; a ticket left non-zero -- state restored by
        hand, a launch that was aborted -- would keep the count from ever advancing again)."""
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        self.step_words.zero_()

    def state_dict(self):
        """What to save: the moments and the step count (`step_words[0]`; the ticket word is not state)."""
        return {"exp_avg": self.exp_avg.clone(), "exp_avg_sq": self.exp_avg_sq.clone(), "step": int(self.step_words[0].item())}

    def load_state_dict(self, sd):
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.step_words.copy_(torch.tensor([int(sd["step"]), 0], dtype=torch.int32))   # ticket zeroed whatever it was

    def zero_grad(self, set_to_none=True):
        for p in self.model.parameters():
            p.grad = None
        self.model._last_flat_grad = None
        self.model._grad_images_pending = None

    def step(self):
        g = self.model._last_flat_grad
        if g is None:
            raise RuntimeError("FlatAdam.step: no gradient (run backward through PointNet2 first)")
        pending = getattr(self.model, "_grad_images_pending", None)
        if pending is not None and not (self.world_size > 1 or self.force_exchange or self.comm is not None):
            arena, replicas, stride = pending
            self.model._grad_images_pending = None
            ops.adam_step_images(self.flat, arena, replicas, stride, self.exp_avg, self.exp_avg_sq, self.lr, self.betas[0],
                                 self.betas[1], self.eps, self.weight_decay, self.step_words, 1.0)
            return
        if pending is not None:                # an exchange was switched on after the backward pass: fold first
            arena, replicas, stride = pending
            self.model._grad_images_pending = None
            ops.grad_reduce(arena, g.numel(), (replicas, stride))
        scale = allreduce_flat_grad(g, self.world_size, self.process_group, self.comm, self.force_exchange)   # RCCL over xGMI when world > 1
        ops.adam_step(self.flat, g, self.exp_avg, self.exp_avg_sq, self.lr, self.betas[0], self.betas[1], self.eps,
                      self.weight_decay, self.step_words, scale)
