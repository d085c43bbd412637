"""TrainPipeline (geometry of later batches on side streams, feature pass per slot as a hipGraph) == the plain loop."""
import numpy as np
import pytest
import torch

from oracle import network
from stratanet2_vegetation_coverage_maps_amd import PointNet2, losses, project_to_plotwise_coverages
from stratanet2_vegetation_coverage_maps_amd import hip_ops as ops
from stratanet2_vegetation_coverage_maps_amd.optim import FlatAdam, flatten_parameters
from stratanet2_vegetation_coverage_maps_amd.pipeline import TrainPipeline
from stratanet2_vegetation_coverage_maps_amd.synthetic import make_args, make_batch

pytestmark = pytest.mark.gpu


def _setup(N, B, depth, n_slots=None, lr=1e-3, input_only=False):
    """input_only: the geometry passes also compute the projection's pixel ids (`model.p2_diam_pix`; the row packing moves there
    whenever a pass is handed the batch's cloud) and the feature step projects from them -- bench.py's configuration."""
    args = make_args(cuda=0, subsample_size=N, ratio1=0.125, r1=1.0, ratio2=0.25, r2=2.0)
    model = PointNet2(args)
    model.load_state_dict(network.init_state_dict(5))
    model = model.cuda().train()
    if input_only:
        model.p2_diam_pix = args.diam_pix
    flatten_parameters(model)
    # eps = 1e-3, not Adam's 1e-8: the two loops compared here differ in the order of a few fp32 atomic adds, and with the
    # default eps the FIRST update of a weight is lr * g / (|g| + eps) -- rounding noise on a gradient of ~1e-8 moves that
    # weight by a fraction of lr and the next losses by ~1e-3 (seen as a bimodal 1.488e-3 on one parameter).  The tests
    # are about launch order and data movement, which a damped optimiser shows just as well.
    opt = FlatAdam(model, lr=lr, eps=1e-3, weight_decay=1e-3)
    slots = []
    for j in range(n_slots or depth + 1):
        h = make_batch(B, N, first_plot=40 + j * B)
        slots.append({"cloud": h["cloud"].cuda(), "xyz": h["xyz"].cuda(),
                      "fps_start": torch.full((2, B), j, dtype=torch.int32, device="cuda"),
                      "gt": h["coverages"].cuda(), "pdf": h["pdf_all"].cuda()})

    def feature_step(inp, geo=None):
        opt.zero_grad()
        cd = {"cloud": inp["cloud"], "xyz": inp["xyz"], "fps_start": inp["fps_start"]}
        if geo is not None:
            cd["geometry"] = geo
        cov, proba = model(cd)
        pred = project_to_plotwise_coverages(cov, inp["cloud"], args, geometry=geo if input_only else None)
        loss, _ = losses.total_loss(pred, proba, inp["gt"], inp["pdf"], args.m, args.e)
        loss.backward()
        return loss
    return model, opt, slots, feature_step


def _assert_same_losses(got, ref):
    """The forward pass is bit-reproducible, the backward pass adds its weight gradients with float atomics, so the two
    loops carry weights that differ by ~1e-10 after the first update.  That is far below 1e-6 in the loss -- until a
    pre-activation that sits next to zero lands on different sides in the two loops: one ReLU mask flip moves the next
    losses by a few 1e-6 (seen: 4.8e-6 from step 4 on).  So: the first three steps (every slot's first use: geometry hand-
    over, input copies, graph replays) to 1e-6, the later ones (slot reuse; a stale or late buffer there means another
    batch's data and shows as 1e-2 to 1e-1: the batches' losses differ by that much) to 1e-3 -- after a flip the two
    trajectories drift apart by ~1.5e-4 within nine steps."""
    np.testing.assert_allclose(got[:3], ref[:3], rtol=0, atol=1e-6)
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-3)


@pytest.mark.parametrize("use_graph,split,pair", [(False, False, False), (True, False, False), (True, True, False),
                                                  (False, False, True), (True, True, True), (True, False, 4), (False, True, 3)])
def test_pipeline_matches_plain_loop(use_graph, split, pair):
    """pair: one geometry pass per TWO batches (2*depth+2 slots); an integer G > 2: per G batches (G*depth+G slots: the
    mode bench.py runs, G = 8)."""
    G = int(pair) if pair not in (False, True) else (2 if pair else 1)
    pair = G > 1
    N, B, depth, steps = 4096, 2, 2, (2 * G + 5) if pair else 7
    n_slots = G * depth + G
    model, opt, slots, fstep = _setup(N, B, depth, n_slots)
    ref_losses = []
    for i in range(steps):
        l = fstep(slots[i % n_slots])
        opt.step()
        ref_losses.append(float(l.detach()))
    ref_params = model._flat_params.clone()
    ref_rm = model.fp1_module.nn[0][2].running_mean.clone()

    model2, opt2, slots2, fstep2 = _setup(N, B, depth, n_slots)
    pipe = TrainPipeline(model2, opt2, fstep2, slots2, depth=depth, use_graph=use_graph, split_exchange=split, group=G)
    assert pipe.pair == pair and pipe.group == G
    pipe.capture()
    # capture() warms each slot with feature passes that update the BN running statistics but not the weights; reset
    # the model/optimiser state so both loops start equal
    model2.load_state_dict(network.init_state_dict(5))
    opt2.reset()                                  # moments, step count and the Adam kernel's arrival ticket
    pipe.prime()
    got = []
    for i in range(steps):
        got.append(float(pipe.step().detach()))
    pipe.drain()
    torch.cuda.synchronize()
    # same kernels on the same data; only the order of a few fp32 atomic adds (dW flushes) may differ.  Losses: see
    # _assert_same_losses; parameters 1e-4 (with Adam's default eps a near-zero gradient's rounding noise moved 66 of 14 997 weights by 3.4e-5
    # in 7 steps; see _setup).
    _assert_same_losses(got, ref_losses)
    # parameters: a ReLU mask flip (see _assert_same_losses) lets the two trajectories drift: up to ~5e-4 on half of the
    # weights within nine steps; a missed or doubled update would show in the optimiser's step counter
    dp = np.abs(model2._flat_params.cpu().numpy() - ref_params.cpu().numpy())
    drm = np.abs(model2.fp1_module.nn[0][2].running_mean.cpu().numpy() - ref_rm.cpu().numpy()).max()
    # stated bounds = 2.5 x the largest value ever measured (parameters 1.6e-3 after the 11 steps of the G = 3 case, 5e-4 after
    # 9; running mean 6.6e-4), below what a real defect shows (a missed, doubled or misplaced update: 1e-2 and more); the
    # measured values of this run are printed
    print(f"\n[pipeline graph={use_graph} split={split} pair={pair}] max |d parameters| {dp.max():.2e} (bound 4e-3), "
          f"max |d running mean| {drm:.2e} (bound 2e-3), max |d loss| {np.abs(np.array(got) - np.array(ref_losses)).max():.2e}")
    assert dp.max() < 4e-3, dp.max()
    assert int(opt2.step_dev.item()) == steps == int(opt.step_dev.item())
    np.testing.assert_allclose(model2.fp1_module.nn[0][2].running_mean.cpu().numpy(), ref_rm.cpu().numpy(), atol=2e-3)   # the same drift (seen: 5.3e-4); one update more or less: 1e-2


def test_pipeline_in_the_headline_mode_matches_plain_loop():
    """The mode bench.py's `value` is measured in: EIGHT consecutive batches per geometry pass, depth 3 => 32 slots, one
    hipGraph per slot, Adam in the same graph -- on tiny plots, for more steps than there are slots (every slot is reused: its
    tables are overwritten by a later pass while earlier feature passes are still queued).  Learning rate 0 (the optimiser
    kernel still runs and counts): every loss is a function of that step's batch and tables only, so the pipelined loop must
    reproduce the plain loop to 1e-6 at EVERY step; a stale, late or misplaced table shows as 1e-2 and more.
    The pipelined side runs as bench.py does: the geometry passes also pack the level-0 rows and compute the projection's pixel
    ids (input-only pieces of the feature pass), and the passes are issued with a phase; the plain loop does neither."""
    N, B, depth, G = 4096, 2, 3, 8
    n_slots = G * depth + G
    steps = n_slots + 2 * G + 3
    model, opt, slots, fstep = _setup(N, B, depth, n_slots, lr=0.0)
    ref = []
    for i in range(steps):
        l = fstep(slots[i % n_slots])
        opt.step()
        ref.append(float(l.detach()))
    assert np.ptp(ref[:n_slots]) > 1e-3               # the slots hold different batches: their losses tell them apart
    model2, opt2, slots2, fstep2 = _setup(N, B, depth, n_slots, lr=0.0, input_only=True)
    pipe = TrainPipeline(model2, opt2, fstep2, slots2, depth=depth, use_graph=True, group=G, phase=3)
    assert pipe.group == G and pipe.slots == 32 and pipe.ahead == G * depth - 3 and pipe.input_only
    assert all(g.p2_pix is not None for g in pipe.geo)
    pipe.capture()
    model2.load_state_dict(network.init_state_dict(5))
    opt2.reset()
    pipe.prime()
    out = torch.zeros(steps, dtype=torch.float64, device="cuda")
    for i in range(steps):                              # no host synchronisation inside the loop
        out[i] = pipe.step().detach()
    pipe.drain(check=True)
    torch.cuda.synchronize()
    got = out.cpu().tolist()
    print(f"\n[pipeline G = 8, 32 slots, {steps} steps] max |d loss| {np.abs(np.array(got) - np.array(ref)).max():.2e}")
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-6)
    assert int(opt2.step_dev.item()) == steps
    # BatchNorm running statistics went through the same `steps` updates in both loops (bit-reproducible forward)
    np.testing.assert_allclose(model2.fp1_module.nn[0][2].running_mean.cpu().numpy(),
                               model.fp1_module.nn[0][2].running_mean.cpu().numpy(), rtol=0, atol=1e-6)


@pytest.mark.parametrize("pair", [False, True])
def test_pipeline_with_host_feeder_matches_plain_loop(pair):
    """Batches fed from pinned host memory on the side streams (the way a DataLoader user of
    /root/reference/learning/train.py:33-44 would feed the pipeline) give the losses of the plain loop over the same
    batches.  Built to hit what round 1's `bench.py --host-inputs` NaN needed: hipGraph replays, pair mode (2*depth+2 slots,
    one geometry pass per two batches), host sources created AFTER the capture with `.cpu()` (a device-to-host copy between
    capture and replay), more DISTINCT batches than slots (a slot gets new data on every reuse; the slots start zeroed), and
    a spin kernel in front of every geometry pass so that the feature passes really wait on `geo_ready`.
    The learning rate is 0 (the optimiser kernel still runs and counts its steps): the weights stay put, so every loss is a
    function of that step's data and tables only and must agree to 1e-6 at EVERY step -- no trajectory drift to allow for."""
    N, B, depth = 4096, 2, 2
    n_slots = 2 * depth + 2 if pair else depth + 1
    n_host, steps = n_slots + 3, 2 * n_slots + 3
    batches = [make_batch(B, N, first_plot=40 + j * B) for j in range(n_host)]

    model, opt, slots, fstep = _setup(N, B, depth, n_slots, lr=0.0)
    ref = []
    for i in range(steps):
        h = batches[i % n_host]
        inp = {"cloud": h["cloud"].cuda(), "xyz": h["xyz"].cuda(), "gt": h["coverages"].cuda(), "pdf": h["pdf_all"].cuda(),
               "fps_start": torch.full((2, B), i % n_slots, dtype=torch.int32, device="cuda")}
        l = fstep(inp)
        opt.step()
        ref.append(float(l.detach()))

    model2, opt2, slots2, fstep2 = _setup(N, B, depth, n_slots, lr=0.0)
    pipe = TrainPipeline(model2, opt2, fstep2, slots2, depth=depth, use_graph=True)
    assert pipe.pair == pair
    pipe.capture()
    # AFTER the capture: device-to-host copies + pinning (the order bench.py --host-inputs uses)
    dev_batches = [{"cloud": h["cloud"].cuda(), "xyz": h["xyz"].cuda(), "gt": h["coverages"].cuda(), "pdf": h["pdf_all"].cuda()}
                   for h in batches]
    host = [{k: v.cpu().pin_memory() for k, v in b.items()} for b in dev_batches]
    for sl in slots2:                                   # wipe the resident copies: the feeder must bring the data
        for k in ("cloud", "xyz", "gt", "pdf"):
            sl[k].zero_()
    model2.load_state_dict(network.init_state_dict(5))
    opt2.reset()                                  # moments, step count and the Adam kernel's arrival ticket
    pipe.issued = pipe.done = 0
    pipe.set_feeder(lambda i: host[i % n_host])
    issue = pipe.issue_geometry

    def delayed(i=None):
        for st in pipe.side:
            with torch.cuda.stream(st):
                torch.cuda._sleep(2_000_000)            # ~1 ms in front of whatever that side stream does next
        return issue(i)
    pipe.issue_geometry = delayed
    pipe.prime()
    out = torch.zeros(steps, dtype=torch.float64, device="cuda")
    for i in range(steps):                              # no host synchronisation inside the loop
        out[i] = pipe.step().detach()
    pipe.drain()
    torch.cuda.synchronize()
    got = out.cpu().tolist()
    assert all(np.isfinite(got)), got
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-6)
    assert int(opt2.step_dev.item()) == steps


def test_capture_guard_joins_and_refuses_a_stream_left_forked():
    """`hip_ops.graph_capture` (what TrainPipeline.capture and bench.py's serial capture go through; DESIGN.md section 4, "the
    capture_end crash of round 4"): a shared stream that the captured code forked into and did not join back is JOINED before
    the capture ends (hipStreamEndCapture never sees an unjoined fork: on ROCm 7.2 that crashed the process instead of returning
    an error) and the capture is refused with StrataHipError -- unless the stream is one the caller says it forks into on
    purpose, in which case the graph is whole and replays.  A capture that starts while a shared stream is inside another
    capture is refused before anything is recorded."""
    from stratanet2_vegetation_coverage_maps_amd._lib import StrataHipError
    dev = torch.device("cuda:0")
    side = ops.shared_stream(dev, "side0")
    x = torch.zeros(64, device=dev)
    torch.cuda.synchronize()

    def fork_and_forget(cap):
        side.wait_stream(cap)
        with torch.cuda.stream(side):
            x.add_(1.0)                         # work on a forked stream that nothing joins back

    g = torch.cuda.CUDAGraph()
    with pytest.raises(StrataHipError, match="left work on shared stream"):
        with ops.graph_capture(g, dev) as cap:
            fork_and_forget(cap)
    torch.cuda.synchronize()
    assert float(x.sum()) == 0.0                # captured, never run; the graph was dropped
    g2 = torch.cuda.CUDAGraph()
    with ops.graph_capture(g2, dev, allowed_forks=("side0",)) as cap:
        fork_and_forget(cap)                    # the guard's join makes the graph whole
    for _ in range(2):
        g2.replay()
    torch.cuda.synchronize()
    assert float(x.min()) == 2.0 and float(x.max()) == 2.0
    assert ops.forked_streams(dev) == []
    # a capture while a shared stream is already capturing: refused up front
    outer = torch.cuda.CUDAGraph()
    cap_stream = ops.shared_stream(dev, "capture")
    with torch.cuda.stream(cap_stream):
        with torch.cuda.graph(outer, stream=cap_stream):
            with pytest.raises(StrataHipError, match="already part of a stream capture"):
                with ops.graph_capture(torch.cuda.CUDAGraph(), dev):
                    pass
            x.add_(0.0)
    torch.cuda.synchronize()
