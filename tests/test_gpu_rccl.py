"""The direct RCCL binding (stratanet2_vegetation_coverage_maps_amd/rccl.py): `ncclAllReduce` on the step's own stream, captured
into the slot's hipGraph.  A one-GPU box can only form a ONE-rank communicator -- the library still initialises, enqueues
and (inside a capture) records the collective, which is what these tests hold; the N-rank arithmetic of the exchange is
covered over gloo (tests/test_distributed_cpu.py, tests/test_gpu_distributed.py)."""
import numpy as np
import pytest
import torch

from stratanet2_vegetation_coverage_maps_amd import rccl


def test_binding_loads_and_draws_unique_ids():
    """CPU: the library loads through ctypes, reports its version and hands out 128-byte ids (no device call)."""
    assert rccl.version() >= 21000
    a, b = rccl.unique_id(), rccl.unique_id()
    assert len(a) == len(b) == rccl.NCCL_UNIQUE_ID_BYTES and a != b
    with pytest.raises(rccl.RcclError):
        rccl.RcclComm(0, 1, b"short", "cuda:0")


@pytest.mark.gpu
def test_one_rank_all_reduce_eager_and_captured():
    comm = rccl.comm_from_torch_group("cuda:0")
    assert (comm.rank, comm.world) == (0, 1)
    assert rccl.self_test(comm, graph=True)
    x = torch.arange(14997, dtype=torch.float32, device="cuda:0")      # the flat gradient's size
    want = x.clone()
    comm.all_reduce_sum_(x)
    torch.cuda.synchronize()
    assert torch.equal(x, want)                                         # SUM over one rank
    with pytest.raises(rccl.RcclError):
        comm.all_reduce_sum_(torch.zeros(4, dtype=torch.float64, device="cuda:0"))
    with pytest.raises(rccl.RcclError):
        comm.all_reduce_sum_(torch.zeros(4))
    comm.destroy()
    with pytest.raises(rccl.RcclError):
        comm.all_reduce_sum_(x)


@pytest.mark.gpu
def test_pipeline_with_the_exchange_inside_the_graph_matches_no_exchange():
    """TrainPipeline with an RCCL communicator on the optimiser: ONE graph per slot (no split), the collective captured between
    backward and Adam; at world 1 the sum is the identity, so losses and weights equal the run without any exchange, bit for
    bit in the forward (learning rate 0: every loss depends on its batch only) and to 1e-6 with the optimiser running."""
    from test_gpu_pipeline import _setup
    from oracle import network
    from stratanet2_vegetation_coverage_maps_amd.pipeline import TrainPipeline
    N, B, depth, G = 4096, 2, 2, 2
    n_slots, steps = G * depth + G, 9
    runs = {}
    for with_comm in (False, True):
        model, opt, slots, fstep = _setup(N, B, depth, n_slots, lr=1e-3)
        comm = rccl.comm_from_torch_group("cuda:0") if with_comm else None
        opt.comm = comm
        pipe = TrainPipeline(model, opt, fstep, slots, depth=depth, use_graph=True, group=G)
        assert pipe.split_exchange is False and all(g is None for g in pipe.graph_opt)
        pipe.capture()
        model.load_state_dict(network.init_state_dict(5))
        opt.reset()
        pipe.prime()
        losses = [float(pipe.step().detach()) for _ in range(steps)]
        pipe.drain(check=True)
        torch.cuda.synchronize()
        runs[with_comm] = (losses, model._flat_params.clone().cpu().numpy(), int(opt.step_dev.item()))
        if comm is not None:
            comm.destroy()
    np.testing.assert_allclose(runs[True][0][:3], runs[False][0][:3], rtol=0, atol=1e-6)
    np.testing.assert_allclose(runs[True][0], runs[False][0], rtol=0, atol=1e-3)
    assert runs[True][2] == runs[False][2] == steps
    assert np.abs(runs[True][1] - runs[False][1]).max() < 4e-3
