"""The N > 1 path on CPU (gloo, world_size 2): plots are sharded by rank with no data-path collective, the only exchange
is ONE all-reduce of the flat gradient buffer, and every rank ends up with the mean of the shard gradients (SURVEY.md
8e).  The HIP backward cannot run here, so the oracle produces each shard's gradient (test infrastructure); what is
under test is the product's sharding + exchange code in `optim.py`."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _shard_grad(rank, per_rank, N, args):
    from oracle import losses, network, projection
    from stratanet2_vegetation_coverage_maps_amd.synthetic import make_batch
    d = make_batch(per_rank, N, first_plot=rank * per_rank)       # the same sharding rule as bench.py
    sd = network.init_state_dict(0)
    keys = network.param_keys(sd)
    for k in keys:
        sd[k].requires_grad_(True)
    cov, proba, _ = network.forward(sd, d["cloud"], d["xyz"], args, training=True)
    pred = projection.project_to_plotwise_coverages(cov, d["cloud"], args)
    loss, _ = losses.total_loss(pred, proba, d["coverages"], d["pdf_all"], args.m, args.e)
    loss.backward()
    return torch.cat([sd[k].grad.reshape(-1) for k in keys])


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from stratanet2_vegetation_coverage_maps_amd.optim import allreduce_flat_grad, shard_of_rank
        from stratanet2_vegetation_coverage_maps_amd.synthetic import make_args
        torch.set_num_threads(2 if world <= 2 else 1)
        N, per_rank = (512, 2) if world <= 2 else (256, 1)
        args = make_args(subsample_size=N, ratio1=0.125, r1=1.5, ratio2=0.25, r2=3.0)
        assert shard_of_rank(rank, per_rank) == (rank * per_rank, per_rank)
        g = _shard_grad(rank, per_rank, N, args)
        assert g.numel() == 14997
        scale = allreduce_flat_grad(g, world)                       # ONE collective: sum; the 1/world scale goes to Adam
        mean = torch.stack([_shard_grad(r, per_rank, N, args) for r in range(world)]).mean(0)
        np.testing.assert_allclose((g * scale).numpy(), mean.numpy(), rtol=1e-5, atol=1e-7)
        # the optional second exchange: rank 0's BatchNorm running statistics to everyone (520 floats + 7 counters)
        from stratanet2_vegetation_coverage_maps_amd.optim import broadcast_bn_buffers
        bnm = torch.nn.Sequential(torch.nn.Linear(4, 8), torch.nn.BatchNorm1d(8), torch.nn.BatchNorm1d(8))
        big = 2 ** 24 + 1                                           # not representable in float32: the counters travel as int64
        with torch.no_grad():
            for name, b in bnm.named_buffers():
                b.fill_(big + rank if name.endswith("num_batches_tracked") else float(rank + 1))
        n = broadcast_bn_buffers(bnm, world)
        assert n == 2 * 16 + 2
        for name, b in bnm.named_buffers():
            if name.endswith("num_batches_tracked"):
                assert b.dtype == torch.int64 and int(b) == big
            else:
                assert float(b.mean()) == 1.0
        out[rank] = 1
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_ranks_gloo_allreduce_of_flat_gradient(world):
    """world 2, and the node's size 8 (tiny shards: the exchange and sharding code is what runs, 8 processes of 1 thread)."""
    port = 29500 + (os.getpid() % 2000)
    out = mp.Manager().dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    assert dict(out) == {r: 1 for r in range(world)}
