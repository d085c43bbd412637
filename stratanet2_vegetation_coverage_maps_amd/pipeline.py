"""Software-pipelined training step: position-only kernels of later batches overlap the feature kernels of this one.

The reference runs, per batch and in this order (`model/point_net2.py:21-29,62-67` under `learning/train.py:52-66`):
fps -> radius -> PointConv ... knn_interpolate ... -> loss -> backward -> Adam.  `fps`, `radius` and `knn` read point
POSITIONS only -- no weights, no features -- so for batch i+1, i+2 they can run while batch i is still in its
forward/backward.  That matters on MI355X because FPS is M strictly sequential rounds in ONE workgroup per plot: with
16 plots per GPU it keeps 16 of 256 CUs busy for ~2 ms while the feature kernels (which fill the chip) need ~2 ms too.

The step is bound by (latency of a geometry pass under load) / (batches a pass covers x passes in flight), and the passes
in flight by the hardware queues (three side streams + the main one; more streams share queues and serialise).  FPS is
one workgroup per plot, so a pass over SEVERAL batches has the latency of a pass over one: with `group = G` (G*depth+G
slots; bench.py runs G = 8, `pair=True` means G = 2) every side-stream pass delivers the tables of G consecutive batches.
What the FPS workgroups cost the feature pass beside them is the time they are resident, not the CUs they hold (two plots'
worth costs as much as thirty-two): G batches per pass divide that time by G -- 0.96 / 0.83 / 0.79 / 0.78 ms per step at
G = 1 / 2 / 4 / 8 (C2), until the pass itself takes a third of the chip (G = 16: 0.81).

Layout: `slots` = depth+1 sets of persistent buffers (inputs + `PointNet2.alloc_geometry`), `depth` side streams.

    side[j]  : wait slot_done[slot of batch i-1] ; geometry(batch i+depth) -> that slot     launched eagerly (10 kernels)
    main     : wait geo_ready[slot i] ; features(batch i) = zero_grad, forward, projection, loss, backward
               [; all-reduce of the flat gradient when world > 1] ; Adam ; record slot_done[slot i]

The feature pass of each slot is captured once into a hipGraph (static addresses: the slot's buffers) and replayed; the
all-reduce stays an ordinary eager RCCL call between the backward graph and the Adam graph.  Every step executes exactly
one geometry pass and one feature pass; results are identical to the unpipelined step (tests/test_gpu_pipeline.py).
"""
import torch

from . import hip_ops as ops
from .optim import allreduce_flat_grad


class TrainPipeline:
    def __init__(self, model, opt, feature_step, slot_inputs, depth=2, use_graph=True, n_streams=None,
                 split_exchange=None, pair=None, group=None, phase=0):
        """model: PointNet2 (train mode); opt: FlatAdam; slot_inputs: list of depth+1 dicts with device tensors "cloud"
        (B,10,N), "xyz" (B,3,N), "fps_start" (2,B) int32 + whatever `feature_step` needs;
        feature_step(inputs, geometry) -> loss: zero_grad, forward (with cloud_data["geometry"] = geometry),
        projection, loss, backward -- everything of the step except the gradient exchange and the optimiser."""
        # pair mode: one geometry pass covers TWO consecutive batches (FPS is one workgroup per plot and M sequential
        # rounds: 32 plots take as long as 16), so a side stream delivers two batches per pass.  The step is bound by
        # (latency of a geometry pass under load) / (batches it covers x passes in flight), and the passes in flight are
        # limited by the hardware queues (three side streams + the main one).  Needs 2*depth+2 slots.
        # group = G > 2: G consecutive batches per pass (G*depth+G slots).  The FPS workgroups cost the feature pass about as
        # much with 2 plots as with 32: what counts is how long they are resident per step.
        can = getattr(model, "alloc_geometry_pair", None) is not None
        if group is None:
            if pair is None:
                pair = len(slot_inputs) >= 2 * depth + 2 and len(slot_inputs) % 2 == 0 and can
            group = 2 if pair else 1
        self.group = G = int(group)
        self.pair = G > 1
        if G > 1 and not can:
            raise ValueError("this model has no grouped geometry pass")
        if len(slot_inputs) < G * depth + G:
            raise ValueError("need at least depth+1 input slots (G*depth+G with G batches per geometry pass)")
        if len(slot_inputs) % G:
            raise ValueError("the number of slots must be a multiple of the batches per geometry pass")
        self.model, self.opt, self.feature_step = model, opt, feature_step
        # the passes already run beside each other on the side streams: no further fork inside a pass
        self._geo_kw = {"fork": False, "shared": True} if hasattr(model, "geometry_fork") else {}
        self.inputs = slot_inputs
        self.depth, self.slots = depth, len(slot_inputs)
        dev = slot_inputs[0]["xyz"].device
        self.dev = dev
        B, _, N = slot_inputs[0]["xyz"].shape
        if self.pair:
            self.geo, self.geo_pairs, self.xyz2, self.fs2 = [None] * self.slots, [], [], []
            for pb in range(self.slots // G):
                gp, parts = model.alloc_geometry_pair(B, N, dev, group=G) if G != 2 else model.alloc_geometry_pair(B, N, dev)
                self.geo_pairs.append(gp)
                for h in range(G):
                    self.geo[G * pb + h] = parts[h]
                self.xyz2.append(torch.empty(G * B, 3, N, dtype=slot_inputs[0]["xyz"].dtype, device=dev))
                self.fs2.append(torch.zeros(slot_inputs[0]["fps_start"].shape[0], G * B, dtype=torch.int32, device=dev))
            # The positions (and, where the slots carry them, the clouds) of the G batches of a pass live in ONE tensor per pass
            # group, and the slots' entries are VIEWS of it (round 5): the pass reads the group's tensor directly -- no G + G
            # device-to-device copies in front of every pass -- and its input-only pieces run once over the whole group.
            self.cloud2 = []
            self._fs_synced = [False] * (self.slots // G)
            has_cloud = all("cloud" in d for d in slot_inputs)
            for pb in range(self.slots // G):
                c2 = None
                if has_cloud:
                    c0 = slot_inputs[G * pb]["cloud"]
                    c2 = torch.empty((G * B,) + tuple(c0.shape[1:]), dtype=c0.dtype, device=dev)
                for h in range(G):
                    d = slot_inputs[G * pb + h]
                    self.xyz2[pb][h * B:(h + 1) * B].copy_(d["xyz"])
                    d["xyz"] = self.xyz2[pb][h * B:(h + 1) * B]
                    if c2 is not None:
                        c2[h * B:(h + 1) * B].copy_(d["cloud"])
                        d["cloud"] = c2[h * B:(h + 1) * B]
                self.cloud2.append(c2)
        else:
            self.geo = [model.alloc_geometry(B, N, dev) for _ in range(self.slots)]
        self.B = B
        # batches the geometry may run ahead of the feature passes (never into a slot whose feature pass is not launched yet)
        self.ahead = min(G * depth, self.slots - G) if self.pair else depth
        # phase (0 <= phase < G, grouped passes only): the passes are issued at the end of the steps that complete batch numbers
        # = phase (mod G) -- phase 0 issues a pass as late as the slots allow, phase s > 0 issues it G - s steps earlier (the
        # tables then wait longer in their slots: depth - 1 whole passes + s batches ahead instead of depth passes).  A timed
        # region that must END with all streams drained wants its last pass issued early in a group, not at its last step
        # (bench.py picks the phase from its step counts; steady-state throughput does not depend on it).
        self.phase = int(phase) % G if self.pair else 0
        if self.phase:
            self.ahead -= self.phase
        self.n_streams = n_streams or depth
        self.side = [ops.shared_stream(dev, f"side{j}") for j in range(self.n_streams)]     # one set per process: hip_ops.shared_stream
        self.geo_ready = [torch.cuda.Event() for _ in range(self.slots)]
        self.slot_done = [None] * self.slots
        self.graph_fb = [None] * self.slots       # zero_grad .. backward
        self.graph_opt = [None] * self.slots      # Adam on that slot's flat gradient
        self.flat_grad = [None] * self.slots
        self.loss = [None] * self.slots
        self.issued = 0                           # geometry passes launched so far
        self.done = 0                             # feature passes launched so far
        self.use_graph = use_graph
        self.slot_wait = "host"                   # how a geometry pass waits for its slots to be free: _wait_slots
        self.feeder_blocking = False              # diagnostic: make the feeder's host-to-device copies synchronous
        self.feeder = None                        # optional: feeder(i) -> dict of HOST tensors for batch number i
        # the geometry passes also run the input-only pieces of the feature pass (PointNet2._input_only: row packing, P2 pixel
        # ids) when the model has them: they are off the feature pass's critical path then
        self.input_only = hasattr(model, "_input_only") and all("cloud" in d for d in slot_inputs)
        # exchange between the backward graph and the Adam graph (always when world > 1; can be forced on one GPU to
        # exercise exactly the launch sequence the multi-GPU run uses)
        # With an RCCL communicator on the optimiser (opt.comm: ncclAllReduce on the step's own stream) the exchange is a node
        # of the slot's graph like any kernel: ONE graph per step at any world size, nothing split.
        in_graph = getattr(opt, "comm", None) is not None
        self.split_exchange = ((getattr(opt, "world_size", 1) > 1) and not in_graph) if split_exchange is None else bool(split_exchange)

    def _wait_slots(self, st, ks):
        """Before a geometry pass overwrites the tables of slots `ks`, the feature passes that last read them must have
        finished.  `slot_wait == "host"` (default): the HOST waits on the later of their events; the passes are issued
        `ahead` batches in front, so the device still has one or more whole steps queued while the host waits.
        `"device"`: `hipStreamWaitEvent` on the side stream.  Measured (scripts/debug_marginal.py): a side stream waiting on
        events of the main stream costs the MAIN stream 0.07 ms per step even with nothing else on the side streams
        (0.694 -> 0.766 ms; every recorded event then has to signal another queue), the host wait costs nothing."""
        evs = [self.slot_done[k] for k in ks if self.slot_done[k] is not None]
        if len(evs) < len(ks):
            st.wait_stream(torch.cuda.current_stream(self.dev))      # first use of a slot: after whatever filled it
        if not evs:
            return
        if self.slot_wait == "host":
            evs[-1].synchronize()                 # recorded in main-stream order: the last one implies the others
        else:
            for ev in evs:
                st.wait_event(ev)

    # ---- geometry of batch number i (its inputs must already be in slot i % slots)
    def issue_geometry(self, i=None):
        i = self.issued if i is None else i
        if self.pair:
            return self._issue_pair(i)
        k = i % self.slots
        st = self.side[i % self.n_streams]
        self._wait_slots(st, (k,))                # the feature pass that last read this slot's tables has finished
        with torch.cuda.stream(st):
            d = self.inputs[k]
            if self.feeder is not None:
                # the next batch arrives from the host (pinned buffers): its copy into the slot rides on the side stream,
                # in front of the geometry pass that reads it and behind the feature pass that last read the slot
                for name, t in self.feeder(i).items():
                    d[name].copy_(t, non_blocking=not self.feeder_blocking)
            self.model._geometry(d["xyz"], d["fps_start"], out=self.geo[k], **self._geo_kw, **self._cloud_kw(d))
            self.geo_ready[k].record(st)
        self.issued = max(self.issued, i + 1)

    def _issue_pair(self, i):
        """Geometry of the G consecutive batches i .. i+G-1 (i a multiple of G) in one pass on one side stream."""
        G = self.group
        i -= i % G
        B = self.B
        ks = [(i + h) % self.slots for h in range(G)]
        pb = ks[0] // G
        st = self.side[(i // G) % self.n_streams]
        self._wait_slots(st, tuple(ks))
        with torch.cuda.stream(st):
            for h, k in enumerate(ks):
                d = self.inputs[k]
                if self.feeder is not None:
                    for name, t in self.feeder(i + h).items():
                        d[name].copy_(t, non_blocking=not self.feeder_blocking)
                # (d["xyz"] / d["cloud"] ARE slices of the group's tensors: nothing to copy; the start indices -- 2 x B ints in a
                # (2, G B) table -- are copied when a feeder may have changed them, and once otherwise)
                if self.feeder is not None or not self._fs_synced[pb]:
                    self.fs2[pb][:, h * B:(h + 1) * B].copy_(d["fps_start"], non_blocking=True)
            self._fs_synced[pb] = True
            kw = {}
            if self.input_only:
                group_cloud = self.cloud2[pb] is not None and getattr(self.model, "geometry_pair_takes_group_cloud", False)
                kw = {"cloud2": self.cloud2[pb]} if group_cloud else {"clouds": [self.inputs[k]["cloud"] for k in ks]}
            self.model._geometry_pair(self.xyz2[pb], self.fs2[pb], self.geo_pairs[pb], tuple(self.geo[k] for k in ks), **kw)
            for k in ks:
                self.geo_ready[k].record(st)
        self.issued = max(self.issued, i + G)

    def _cloud_kw(self, d):
        return {"cloud": d["cloud"]} if self.input_only else {}

    def _exchange_and_update(self, k):
        g = self.flat_grad[k]
        self.model._last_flat_grad = g
        self.opt.step()                           # all-reduce (world > 1) + Adam kernel

    def capture(self):
        """Warm every slot eagerly (allocator, lazy loads), then capture each slot's feature pass.  Geometry of all
        slots must be valid while warming: computed here, and left valid for steps 0..slots-1."""
        main = torch.cuda.current_stream(self.dev)
        for i in range(0, self.slots, self.group):
            self.issue_geometry(i)
        self.issued = 0
        for st in self.side:
            main.wait_stream(st)
        torch.cuda.synchronize(self.dev)
        if not self.use_graph:
            return
        cap = ops.shared_stream(self.dev, "capture")
        cap.wait_stream(main)
        with torch.cuda.stream(cap):
            for k in range(self.slots):           # warm-up on the capture stream
                self.feature_step(self.inputs[k], self.geo[k])
            comm = getattr(self.opt, "comm", None)
            if comm is not None:
                # RCCL connects lazily at a communicator's FIRST collective: that one must not be the node being captured
                #  -- one eager all-reduce of a scratch buffer on the capture stream, on every rank
                comm.all_reduce_sum_(torch.zeros(256, dtype=torch.float32, device=self.dev))
        main.wait_stream(cap)
        torch.cuda.synchronize(self.dev)
        pool = torch.cuda.graph_pool_handle()
        world = getattr(self.opt, "world_size", 1)
        # (ops.graph_capture: torch.cuda.graph + the guards of DESIGN.md section 4 -- no shared stream inside another capture
        # at the start, none left forked at the end: a feature pass forks nowhere)
        for k in range(self.slots):
            g = torch.cuda.CUDAGraph()
            with ops.graph_capture(g, self.dev, pool=pool):
                self.loss[k] = self.feature_step(self.inputs[k], self.geo[k])
                self.flat_grad[k] = self.model._last_flat_grad
                if not self.split_exchange:
                    self._exchange_and_update(k)  # no exchange: Adam rides in the same graph
            self.graph_fb[k] = g
            if self.split_exchange:
                g2 = torch.cuda.CUDAGraph()
                with ops.graph_capture(g2, self.dev, pool=pool):
                    ops.adam_step(self.opt.flat, self.flat_grad[k], self.opt.exp_avg, self.opt.exp_avg_sq, self.opt.lr,
                                  self.opt.betas[0], self.opt.betas[1], self.opt.eps, self.opt.weight_decay,
                                  self.opt.step_words, 1.0 / world)
                self.graph_opt[k] = g2
        torch.cuda.synchronize(self.dev)

    def prime(self):
        """Launch the geometry of the first `depth` batches (pairs of batches in pair mode) before the first step."""
        self._run_ahead()

    def _run_ahead(self):
        step = self.group
        while self.issued + step <= self.done + self.ahead:
            self.issue_geometry(self.issued)

    def step(self):
        """One training step on batch number `done` (slot done % slots); keeps `depth` geometry passes in flight."""
        i = self.done
        k = i % self.slots
        main = torch.cuda.current_stream(self.dev)
        if self.issued <= i:
            self.issue_geometry(i)
        main.wait_event(self.geo_ready[k])
        if self.graph_fb[k] is not None:
            self.graph_fb[k].replay()
            if self.graph_opt[k] is not None:
                allreduce_flat_grad(self.flat_grad[k], self.opt.world_size, self.opt.process_group, getattr(self.opt, "comm", None),
                                    getattr(self.opt, "force_exchange", False))
                self.graph_opt[k].replay()
            loss = self.loss[k]
        else:
            loss = self.feature_step(self.inputs[k], self.geo[k])
            self.flat_grad[k] = self.model._last_flat_grad
            self._exchange_and_update(k)
        ev = torch.cuda.Event()
        ev.record(main)
        self.slot_done[k] = ev
        self.done = i + 1
        # geometry of batch i+depth goes into the slot whose tables batch i+depth-slots (= i-1 when slots = depth+1) read
        self._run_ahead()
        return loss

    def set_feeder(self, feeder):
        """feeder(i) -> {"cloud": ..., "xyz": ..., ...} HOST tensors (ideally pinned) for batch number i, same shapes and dtypes
        as the slot tensors of those names; they are copied into slot i % slots on the side stream before that batch's
        geometry pass.  None = the slots already hold the data (the resident-input mode of bench.py)."""
        self.feeder = feeder

    def drain(self, check: bool = False):
        """Make the main stream wait for every geometry pass in flight.  check=True (where the caller synchronises with the
        device anyway): also read the FPS status word and the fused global level's give-up count and warn when passes had to be repeated
        (hip_ops.fps_gave_up, hip_ops.global_level_gave_up: the results are right either way)."""
        for st in self.side:
            torch.cuda.current_stream(self.dev).wait_stream(st)
        if check:
            ops.fps_gave_up(self.dev)
            ops.global_level_gave_up(self.dev)      # (warns once per growth: launches of the fused global level were repaired)
