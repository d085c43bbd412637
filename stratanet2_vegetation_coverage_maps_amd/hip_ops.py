"""Host-side wrappers of the C ABI (include/strata_hip.h): PyTorch-ROCm tensors in, raw device pointers across
`ctypes`, torch's current stream.  Every wrapper validates dtype / device / contiguity / shape on the host before a
kernel is launched (a kernel that faults can take the whole node down) and raises on a non-zero return code.

Names follow the reference's third-party call sites (model/point_net2.py:9): fps, radius -> ball_query,
knn_interpolate -> three_nn + the interpolation inside `fp_forward`, PointConv -> `sa_forward`.
"""
import math
from typing import Optional

import numpy as np
import torch

from . import _lib
from ._lib import FP, SA, Block, Head, StrataHipError, check

I32, F32, F64, I64, BF16 = torch.int32, torch.float32, torch.float64, torch.int64, torch.bfloat16


# ---- optional per-entry-point timing with HIP events on torch's current stream (the stream every kernel is launched
# on).  bench.py switches it on for the entry points whose roofline it reports; None = off (no overhead).
_timing = None


class timing:
    """with timing({"sn2_fps", ...}) as t: ... ; t.summary() -> {name: (calls, total_ms)} after a device sync."""

    def __init__(self, names=None):
        self.names = None if names is None else set(names)
        self.events = []

    def __enter__(self):
        global _timing
        self._prev, _timing = _timing, self
        return self

    def __exit__(self, *exc):
        global _timing
        _timing = self._prev

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, a, b in self.events:
            c, t = out.get(name, (0, 0.0))
            out[name] = (c + 1, t + a.elapsed_time(b))
        return out


def _call(name, *args, tag=None, key=None):
    fn = getattr(_lib.load(), name)
    t = _timing
    key = key or name                       # the name the entry point is reported under
    if tag is not None:
        key = f"{key}:{tag}"
    if t is not None and (t.names is None or key in t.names):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        rc = fn(*args)
        b.record()
        t.events.append((key, a, b))
    else:
        rc = fn(*args)
    check(rc, name)


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream() -> int:
    """The raw handle of torch's current stream on the current device (every entry point launches there).  Through torch's
    own raw accessor where it exists: `torch.cuda.current_stream()` builds a Stream object per call, ~2 us x 60 calls per
    eager step."""
    if _RAW_STREAM is not None:
        return _RAW_STREAM(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


_SHARED_STREAMS = {}      # (device index, name) -> stream


def shared_stream(dev, name: str) -> torch.cuda.Stream:
    """The process-wide side stream `name` of a device ("side0".."sideK": geometry passes of pipelined loops and prefetch
    lanes; "fork_b", "fork_c", "pack": the branches of an unpipelined geometry pass; "capture").  Created ONCE: HIP deals
    streams onto a few hardware queues round-robin as they are created, so a second pipeline (or model) that created its own
    streams could find two of them -- or one and the main stream -- on one queue, where the passes it meant to overlap run
    back to back (seen: +25 % per step for a pipeline built after another one in the same process).
    SINGLE OWNER AT A TIME: the streams are shared by everything in the process that asks for them by name -- a TrainPipeline,
    an eval model's prefetch lanes, bench.py's serial capture.  Two such users that are LIVE at once serialise on them (a pass
    meant to overlap another runs behind it), and a hipGraph capture on "capture" would record foreign work enqueued on it
    meanwhile: run one pipelined loop / one capture at a time per device (what bench.py and the tests do: legs run one after
    the other), or give a second concurrent user its own names."""
    dev = torch.device(dev)
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), name)
    if key not in _SHARED_STREAMS:
        _SHARED_STREAMS[key] = torch.cuda.Stream(device=dev)
    return _SHARED_STREAMS[key]


def forked_streams(dev, exclude=None):
    """Names of the process's shared streams of `dev` that are part of a stream capture RIGHT NOW (hipStreamIsCapturing):
    streams a capture forked into (`side.wait_stream(capture stream)` + work on `side`)."""
    dev = torch.device(dev)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    out = []
    for (di, name), s in list(_SHARED_STREAMS.items()):
        if di != idx or (exclude is not None and s == exclude):
            continue
        with torch.cuda.stream(s):
            if torch.cuda.is_current_stream_capturing():
                out.append(name)
    return out


import contextlib as _contextlib


@_contextlib.contextmanager
def graph_capture(graph, dev, pool=None, allowed_forks=()):
    """`torch.cuda.graph(graph, pool=pool)` with the two rules this package's captures live by ENFORCED (round 5; DESIGN.md
    section 4, "the capture_end crash of round 4"):
      * a capture starts with none of the process's shared streams inside another capture (two live users of `shared_stream`
        names) -- refused with StrataHipError before anything is captured;
      * a capture ENDS with every stream it forked into joined back.  ROCm 7.2's hipStreamEndCapture does not return
        hipErrorStreamCaptureUnjoined for a stream that is still forked: it crashed the process (round 4: a loss-value branch
        on a fourth stream whose autograd node the engine ran -- and left work -- on that stream after the step's join).
        Here every shared stream found capturing at the end of the block is joined first (so the runtime never sees the
        state), and unless its name is in `allowed_forks` (streams the captured code forks into and joins itself: the
        unpipelined step's geometry branches) the graph is dropped and StrataHipError raised."""
    dev = torch.device(dev)
    busy = forked_streams(dev)
    if busy:
        raise StrataHipError(f"graph_capture: shared stream(s) {busy} are already part of a stream capture: one capture at a time "
                             "per device (hip_ops.shared_stream: single owner at a time)")
    stray = []
    kw = {} if pool is None else {"pool": pool}
    with torch.cuda.graph(graph, **kw):
        cap = torch.cuda.current_stream(dev)
        try:
            yield cap
        finally:
            for name in forked_streams(dev, exclude=cap):
                cap.wait_stream(shared_stream(dev, name))      # joined: hipStreamEndCapture sees a legal graph whatever happened
                if name not in allowed_forks:
                    stray.append(name)
    if stray:
        graph.reset()
        raise StrataHipError(f"graph_capture: the captured code left work on shared stream(s) {stray} that nothing joined back "
                             "into the capture stream (an autograd node created on a side stream runs its backward there); "
                             "the graph was dropped")


_UPLOAD_TRACE = None       # scripts/profile_dropin_host.py sets a list: (what, host ms) of every wait / copy inside PinnedRing.upload


class PinnedRing:
    """Host-to-device uploads of pageable tensors through a small ring of PINNED staging buffers.

    The reference hands `PointNet2.forward` CPU tensors from a DataLoader without `pin_memory` (`learning/train.py:33-44`,
    `model/point_net2.py:119-124` moves them with `.cuda()`): a copy from pageable memory is staged by the runtime chunk by chunk
    and blocks the host for its whole length (40 MB per step at C2: 5 of the eager loop's 5.8 ms).  Here the host copies into a
    pinned buffer itself (a multi-threaded memcpy) and the DMA runs asynchronously on the stream it is given, so the kernels
    that need only the first tensor (the geometry pass needs `xyz` only) start while the rest is still on its way.
    A slot is reused only after the DMA that read it has finished (an event per slot, waited for on the host)."""

    def __init__(self, dev, slots: int = 4):
        self.dev = torch.device(dev)
        self.bufs = [None] * slots
        self.events = [None] * slots
        self.k = 0

    def upload(self, t: torch.Tensor, stream=None, dtype=None, out=None, consumer=None) -> torch.Tensor:
        """CPU tensor -> device tensor of `dtype` (default: its own) with the same shape, copied asynchronously on `stream`
        (default: torch's current stream).  The result is safe to use on that stream; events for others are the caller's
        business.  out: the device tensor to fill -- allocate it on the stream that will CONSUME it (a 21 MB block allocated on
        a side stream and handed to the main stream with record_stream kept the caching allocator from reusing it: a fresh
        hipMalloc per step, 80 ms instead of 5), and make `stream` wait for whatever last used the block.
        consumer: the stream that will read `out` when that is not `stream`: it is made to wait for the copy here, and the
        slot's reuse event is recorded on IT.  (An event that is the last thing ever submitted to an otherwise idle side stream
        was only seen as complete by a later `event.synchronize()` after ~86 ms on ROCm 7.2 -- every third step of the eager
        loop; an event on the busy main stream completes when the stream gets there.)"""
        dtype = dtype or t.dtype
        src = t.detach()
        if src.dtype != dtype:
            src = src.to(dtype)
        src = src.contiguous()
        n = src.numel() * src.element_size()
        i = self.k
        self.k = (self.k + 1) % len(self.bufs)
        if self.events[i] is not None:
            if _UPLOAD_TRACE is not None:
                import time as _t
                a = _t.perf_counter()
                self.events[i].synchronize()
                _UPLOAD_TRACE.append(("event", (_t.perf_counter() - a) * 1e3))
            else:
                self.events[i].synchronize()                  # the DMA that last read this slot
        if self.bufs[i] is None or self.bufs[i].numel() < n:
            self.bufs[i] = torch.empty(max(n, 1 << 20), dtype=torch.uint8).pin_memory()
        stage = self.bufs[i][:n].view(dtype).view(src.shape)
        if _UPLOAD_TRACE is not None:
            import time as _t
            a = _t.perf_counter()
            stage.copy_(src)
            _UPLOAD_TRACE.append(("memcpy", (_t.perf_counter() - a) * 1e3))
        else:
            stage.copy_(src)                                  # host memcpy into pinned memory
        st = torch.cuda.current_stream(self.dev) if stream is None else stream
        allocated_here = out is None
        with torch.cuda.stream(st):
            if out is None:
                out = torch.empty(src.shape, dtype=dtype, device=self.dev)
            elif out.shape != src.shape or out.dtype != dtype or not out.is_contiguous():
                raise ValueError("PinnedRing.upload: `out` does not match the source")
            out.copy_(stage, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(st)
        if consumer is not None and consumer != st:
            consumer.wait_event(ev)
            ev = torch.cuda.Event()
            ev.record(consumer)
            if allocated_here:
                # allocated on the copy stream, read on the consumer's: the caching allocator must not hand the block to a later
                # allocation of the copy stream before the consumer is done with it.  (Two blocks then alternate from step to
                # step -- scripts/debug_side_alloc.py: no hipMalloc after the second iteration.)
                out.record_stream(consumer)
        self.events[i] = ev
        return out


_RINGS = {}


def pinned_ring(dev) -> PinnedRing:
    """The process-wide upload ring of a device."""
    dev = torch.device(dev)
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    if key not in _RINGS:
        _RINGS[key] = PinnedRing(dev)
    return _RINGS[key]


def create_shared_streams(dev, lanes: int = 4):
    """Create the process's side streams NOW ("side0".."side<lanes-1>", the fork / pack / capture streams), in a fixed order.
    Call it before anything else creates streams on the device (RCCL communicators, torch.distributed process groups): the
    hardware queues go to the first streams created, and a loop whose side streams come late shares queues with the
    communicator's streams (measured: 0.905 instead of 0.787 ms per step)."""
    for name in [f"side{j}" for j in range(lanes)] + ["fork_b", "fork_c", "pack", "capture", "upload"]:
        shared_stream(dev, name)


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _chk(t: torch.Tensor, dtype, shape=None, name="tensor"):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise ValueError(f"{name}: expected a tensor on the HIP device")
    if t.dtype != dtype:
        raise ValueError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: must be contiguous")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
    return t


def _chk_rows(t: torch.Tensor, dtype, rows: int, cols: int, name="rows", align: int = 4) -> int:
    """A 2-D row view (rows, >=cols) with unit inner stride; returns the row stride in elements.  Row starts must be
    16-byte aligned when align == 4 (the kernels read rows with 16-byte loads)."""
    if not isinstance(t, torch.Tensor) or not t.is_cuda or t.dtype != dtype or t.dim() != 2:
        raise ValueError(f"{name}: expected a 2-D {dtype} tensor on the HIP device")
    if t.shape[0] != rows or t.shape[1] < cols or t.stride(1) != 1:
        raise ValueError(f"{name}: expected ({rows}, >={cols}) with unit inner stride, got {tuple(t.shape)} {t.stride()}")
    st = t.stride(0) if rows > 1 else max(t.stride(0), t.shape[1])
    if align > 1 and (st % align or (t.data_ptr() % (4 * align))):
        raise ValueError(f"{name}: rows must be {4 * align}-byte aligned")
    return st


def fps_num_samples(n: int, ratio: float) -> int:
    """ceil(fp32(n) * fp32(ratio)) -- torch-cluster 1.5.9 `fps` sample count (see oracle/primitives.py)."""
    return int(math.ceil(float(np.float32(n) * np.float32(ratio))))


def r2_threshold(r: float) -> float:
    """fp32(r*r) with r*r evaluated in double: what `radius` compares squared distances with."""
    return float(np.float32(float(r) * float(r)))


# ---------------------------------------------------------------------------------------------- geometry
def pack_rows(cloud: torch.Tensor, xyz: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    B, C, N = cloud.shape
    _chk(cloud, F32, (B, C, N), "cloud")
    _chk(xyz, F32, (B, 3, N), "xyz")
    rows0 = torch.empty(B * N, 12, dtype=F32, device=cloud.device) if out is None else _chk(out, F32, (B * N, 12), "out")
    _call("sn2_pack_rows", _ptr(cloud), _ptr(xyz), B, C, N, _ptr(rows0), _stream())
    return rows0


def fps_ws_words(B: int, N: int) -> int:
    """SN2_FPS_WS_WORDS of include/strata_hip.h."""
    return 6 * B * N + (4104 + 4096) * B + 32


def fps_ws_ctl(ws: torch.Tensor, B: int, N: int) -> torch.Tensor:
    """The 32 control words of a filled FPS workspace ([1] = exchange waits of the multi-workgroup kernel that gave up in the LAST
    pass over this workspace; the process-wide running total is `fps_gave_up`)."""
    o = 5 * B * N + (4104 + 4096) * B
    return ws[o:o + 32]


def fps_ws_rank(ws: torch.Tensor, B: int, N: int) -> torch.Tensor:
    """The (B*N) int32 view of a filled FPS workspace that holds every point's position in the plot's spatial (Morton) order."""
    return ws[5 * B * N + (4104 + 4096) * B + 32:]


def fps_fills_ws(B: int, N: int, m: int) -> bool:
    """Whether sn2_fps takes its bucketed path for these sizes, i.e. FILLS the workspace (Morton order, sorted table, cell
    starts) that ball_query / three_nn may then walk.  Must mirror the condition in csrc/geometry.hip (sn2_fps): handing
    those kernels a workspace nobody filled would send them through garbage cell lists."""
    many_small = N <= 4096 and B > 32          # sn2_fps_status: many small plots take the brute-force kernel (no tables)
    return N > 2048 and not many_small and m > 16 and (B * N) % 4 == 0 and N <= 131072


_FPS_STATUS = {}          # device index -> [status word (1,) int32 on the device, count already reported]


class StrataHipWarning(RuntimeWarning):
    pass


def fps_status_word(dev) -> torch.Tensor:
    """The device word every `fps` call of this process hands to sn2_fps_status (include/strata_hip.h): it accumulates the
    waits of the multi-workgroup FPS kernel that gave up (each such pass is repeated by the single-workgroup kernel inside
    the same call: the samples are right either way)."""
    dev = torch.device(dev)
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    if key not in _FPS_STATUS:
        _FPS_STATUS[key] = [torch.zeros(1, dtype=I32, device=dev), 0]
    return _FPS_STATUS[key][0]


def fps_gave_up(dev, warn: bool = True) -> int:
    """Waits of the multi-workgroup FPS that gave up on this device since the process started.  Reads one word from the
    device: call it where the host synchronises anyway (TrainPipeline.drain, the end of predict_parcel, after a test).
    A count that grew since the last call means FPS passes are being run twice (the workgroups of the multi-workgroup
    kernel were not resident together -- another stream or process held the CUs): results are unaffected, the pass takes
    its bounded wait + the single-workgroup kernel longer; `warn` reports that once per growth as a StrataHipWarning."""
    dev = torch.device(dev)
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    if key not in _FPS_STATUS:
        return 0
    word, seen = _FPS_STATUS[key]
    n = int(word.item())
    if n > seen:
        _FPS_STATUS[key][1] = n
        if warn:
            import warnings
            warnings.warn(f"multi-workgroup FPS: {n - seen} exchange wait(s) gave up on {dev} and the passes were repeated by "
                          "the single-workgroup kernel (results unaffected, the passes took longer); ask for one workgroup "
                          "per plot (fps(..., waves=8 or 16)) where other kernels or processes share the device",
                          StrataHipWarning, stacklevel=2)
    return n


def fps(pos_soa: torch.Tensor, m: int, start: Optional[torch.Tensor] = None, bucketed: bool = True,
        return_ws: bool = False, out=None, waves: int = 0):
    """pos_soa (B,3,N) -> idx (B,m) int32 local indices, cpos_soa (B,3,m), cpos_aos (B*m,4).
    waves: which bucketed kernel (include/strata_hip.h: sn2_fps_waves): 0 = the shortest pass (several workgroups per plot
    where the batch fits the chip), 16 / 8 / 4 = one workgroup of 16 / 8 / 4 waves per plot (8 = the pass that shares its CUs with
    concurrent kernels, what the pipelined training loop asks for; 4 = the same for passes over hundreds of small plots, the
    parcel loop; plots of more than 16 384 points take 8), 32 + P / 64 + P = P workgroups of 16 / 8 waves per plot,
    1 = the one-sample-per-round kernel; same indices whichever runs.
    bucketed=False forces the brute-force kernel (same result; kept for cross-checks).  return_ws=True also returns
    the spatial-order workspace (or None), which `ball_query` over the same points can reuse.
    out = (idx, cpos_soa, cpos_aos, workspace-or-None): caller-owned result buffers (persistent pipelines)."""
    B, three, N = pos_soa.shape
    _chk(pos_soa, F32, (B, 3, N), "pos_soa")
    if not (1 <= m <= N):
        raise ValueError(f"fps: need 1 <= m <= N, got m={m}, N={N}")
    if start is not None:
        _chk(start, I32, (B,), "start")
    dev = pos_soa.device
    use_ws = bucketed and fps_fills_ws(B, N, m)
    if out is not None:
        idx, cs, ca, order = out
        _chk(idx, I32, (B, m), "out idx")
        _chk(cs, F32, (B, 3, m), "out cpos_soa")
        _chk(ca, F32, (B * m, 4), "out cpos_aos")
        if order is not None:
            _chk(order, I32, (fps_ws_words(B, N),), "out workspace")
        if not use_ws:
            order = None                      # the kernel would not fill it: nobody may read it
    else:
        idx = torch.empty(B, m, dtype=I32, device=dev)
        cs = torch.empty(B, 3, m, dtype=F32, device=dev)
        ca = torch.empty(B * m, 4, dtype=F32, device=dev)
        order = torch.empty(fps_ws_words(B, N), dtype=I32, device=dev) if use_ws else None
    _call("sn2_fps_status", _ptr(pos_soa), B, N, m, _ptr(start), _ptr(idx), _ptr(cs), _ptr(ca), _ptr(order), int(waves),
          _ptr(fps_status_word(dev)), _stream(), tag=f"N={N}", key="sn2_fps")
    if return_ws:
        return idx, cs, ca, order
    return idx, cs, ca


def ball_query(src_soa: torch.Tensor, cpos_soa: torch.Tensor, r: float, cap: int = _lib.MAX_NEIGHBORS,
               total: Optional[torch.Tensor] = None, fps_ws: Optional[torch.Tensor] = None, out=None):
    """-> nbr (B*M,cap) int32 (first cnt entries valid, ascending source index), cnt (B*M) int32, total (1) int64.
    out = (nbr, cnt): caller-owned result buffers."""
    B, _, N = src_soa.shape
    M = cpos_soa.shape[2]
    _chk(src_soa, F32, (B, 3, N), "src_soa")
    _chk(cpos_soa, F32, (B, 3, M), "cpos_soa")
    cap = min(cap, N)
    dev = src_soa.device
    if out is not None:
        nbr, cnt = out
        _chk(nbr, I32, (B * M, cap), "out nbr")
        _chk(cnt, I32, (B * M,), "out cnt")
    else:
        nbr = torch.empty(B * M, cap, dtype=I32, device=dev)
        cnt = torch.empty(B * M, dtype=I32, device=dev)
    if total is False:
        total = None                          # no message total wanted (a grouped pass sums the counts per batch: count_sum_group)
    elif total is None:
        total = torch.zeros(1, dtype=I64, device=dev)
    else:
        _chk(total, I64, (1,), "total")
    if fps_ws is not None:
        _chk(fps_ws, I32, (fps_ws_words(B, N),), "fps_ws")
    _call("sn2_ball_query", _ptr(src_soa), B, N, _ptr(cpos_soa), M, r2_threshold(r), cap, _ptr(nbr), _ptr(cnt), _ptr(total),
          _ptr(fps_ws), _stream(), tag=f"N={N}")
    return nbr, cnt, total


def count_sum(cnt: torch.Tensor, total: torch.Tensor):
    """include/strata_hip.h: sn2_count_sum -- total (1,) int64 = sum of the neighbour counts cnt (n,) int32."""
    _chk(cnt, I32, None, "cnt")
    _chk(total, I64, (1,), "total")
    _call("sn2_count_sum", _ptr(cnt), cnt.numel(), _ptr(total), _stream())


def count_sum_group(cnt: torch.Tensor, G: int, n: int, totals: torch.Tensor, stride: int):
    """include/strata_hip.h: sn2_count_sum_group -- totals[h * stride] = sum of cnt[h*n : (h+1)*n], h < G, in one launch."""
    _chk(cnt, I32, (G * n,), "cnt")
    _chk(totals, I64, None, "totals")
    if totals.numel() < (G - 1) * stride + 1:
        raise ValueError("count_sum_group: totals too small")
    _call("sn2_count_sum_group", _ptr(cnt), G, n, _ptr(totals), stride, _stream(), key="sn2_count_sum")


def three_nn_ws_words(B: int, S: int, T: int = 0) -> int:
    """SN2_THREE_NN_XY_WS_WORDS (T > 0) / SN2_THREE_NN_WS_WORDS (T = 0) of include/strata_hip.h."""
    return B * (4 * S + 5 * T + 1032)


def three_nn_uses_grid(S: int, T: int) -> bool:
    """The grid search of sn2_three_nn_xy applies (otherwise the full scan runs)."""
    return 128 <= S <= 8192 and T > 2048


def three_nn(src_soa: torch.Tensor, dst_soa: torch.Tensor, k: int, out=None, grid: bool = True, ws=None,
             dst_fps_ws: Optional[torch.Tensor] = None):
    """-> idx (B*T,3) int32 local source indices, w (B*T,3) = 1/max(d2,1e-16) (0 on unused slots).
    out = (idx, w): caller-owned result buffers.
    grid=True: for T > 2048 targets and 128..8192 sources the search walks a per-plot x,y grid of the sources, a wave taking
    64 targets of adjacent grid cells (sn2_three_nn_xy sorts the targets by cell itself; same result as the full scan, bit
    for bit); ws: caller-owned workspace of three_nn_ws_words(B, S, T) int32 for it (allocated here when None).
    dst_fps_ws (with grid=True): use the order `fps(..., return_ws=True)` left for the TARGET points instead of the
    cell sort (the first form of the grid search, kept for cross-checks); grid=False forces the full scan."""
    B, _, S = src_soa.shape
    T = dst_soa.shape[2]
    _chk(src_soa, F32, (B, 3, S), "src_soa")
    _chk(dst_soa, F32, (B, 3, T), "dst_soa")
    dev = src_soa.device
    if out is not None:
        idx, w = out
        _chk(idx, I32, (B * T, 3), "out idx")
        _chk(w, F32, (B * T, 3), "out w")
    else:
        idx = torch.empty(B * T, 3, dtype=I32, device=dev)
        w = torch.empty(B * T, 3, dtype=F32, device=dev)
    if grid and three_nn_uses_grid(S, T) and dst_fps_ws is None:
        n = three_nn_ws_words(B, S, T)
        if ws is None:
            ws = torch.empty(n, dtype=I32, device=dev)
        else:
            _chk(ws, I32, (n,), "ws")
        _call("sn2_three_nn_xy", _ptr(src_soa), B, S, _ptr(dst_soa), T, k, _ptr(idx), _ptr(w), _ptr(ws), _stream(), tag=f"T={T}")
        return idx, w
    if grid and dst_fps_ws is not None and three_nn_uses_grid(S, T):
        if ws is None:
            ws = torch.empty(three_nn_ws_words(B, S), dtype=I32, device=dev)
        elif ws.numel() < three_nn_ws_words(B, S):
            raise ValueError("three_nn: workspace too small")
        _chk(dst_fps_ws, I32, (fps_ws_words(B, T),), "dst_fps_ws")
    else:
        ws = dst_fps_ws = None
    _call("sn2_three_nn", _ptr(src_soa), B, S, _ptr(dst_soa), T, k, _ptr(idx), _ptr(w), _ptr(ws), _ptr(dst_fps_ws), _stream(),
          tag=f"T={T}")
    return idx, w


# ---------------------------------------------------------------------------------------------- blocks
class BlockBuffers:
    """Device-side companions of one (Linear -> ReLU -> BatchNorm1d) block: affine (a, c), saved batch statistics,
    fp64 accumulators.  `params` = (W, b, gamma, beta, running_mean, running_var) tensors of the nn modules."""

    def __init__(self, lin: torch.nn.Linear, bn: torch.nn.BatchNorm1d, aux: Optional[torch.Tensor] = None,
                 stats: Optional[torch.Tensor] = None):
        self.lin, self.bn = lin, bn
        self.cin, self.cout = lin.in_features, lin.out_features
        dev = lin.weight.device
        # aux rows: a, c, mean, invstd (fp32);  stats: per-workgroup batch-statistics slots (no initialisation needed)
        self.aux = torch.empty(4, self.cout, dtype=F32, device=dev) if aux is None else aux
        self.stats = torch.empty(_lib.STAT_SLOTS * 2 * self.cout, dtype=F32, device=dev) if stats is None else stats
        _chk(self.aux, F32, (4, self.cout), "aux")
        _chk(self.stats, F32, (_lib.STAT_SLOTS * 2 * self.cout,), "stat_slots")
        self.grads = None                                                 # (dW, db, dgamma, dbeta) views, set per backward
        self.grad_images = (1, 0)                                         # (replicas, stride in floats) of dW / db
        self.mma_bf16 = False                                             # bfloat16 operands on the matrix cores (sn2_block.mma_bf16)
        self.frozen = False                                               # the forward ran on the running statistics (sn2_block.frozen_stats)

    def fill(self, blk: Block, with_grads: bool = False):
        for t, n in ((self.lin.weight, "weight"), (self.lin.bias, "bias"), (self.bn.weight, "bn.weight"),
                     (self.bn.bias, "bn.bias"), (self.bn.running_mean, "running_mean"),
                     (self.bn.running_var, "running_var")):
            _chk(t, F32, None, n)
        blk.cin, blk.cout = self.cin, self.cout
        blk.mma_bf16 = int(bool(self.mma_bf16))
        blk.frozen_stats = int(bool(self.frozen))
        blk.W, blk.b = _ptr(self.lin.weight), _ptr(self.lin.bias)
        blk.gamma, blk.beta = _ptr(self.bn.weight), _ptr(self.bn.bias)
        blk.running_mean, blk.running_var = _ptr(self.bn.running_mean), _ptr(self.bn.running_var)
        blk.a, blk.c, blk.mean, blk.invstd = (_ptr(self.aux[i]) for i in range(4))
        blk.stat_slots = _ptr(self.stats)
        nbt = self.bn.num_batches_tracked
        if nbt is not None:
            _chk(nbt, I64, None, "num_batches_tracked")
        blk.num_batches_tracked = _ptr(nbt) if nbt is not None else None
        if with_grads:
            dW, db, dg, dbeta = self.grads
            for t, ref in ((dW, self.lin.weight), (db, self.lin.bias), (dg, self.bn.weight), (dbeta, self.bn.bias)):
                _chk(t, F32, ref.shape, "grad view")
            blk.dW, blk.db, blk.dgamma, blk.dbeta = _ptr(dW), _ptr(db), _ptr(dg), _ptr(dbeta)
            blk.grad_replicas, blk.grad_replica_stride = self.grad_images
        else:
            blk.dW = blk.db = blk.dgamma = blk.dbeta = None
            blk.grad_replicas, blk.grad_replica_stride = 1, 0

    @property
    def a(self):
        return self.aux[0]

    @property
    def c(self):
        return self.aux[1]


def prepare_plots(raw, offsets, centers, fake_xy, idx, z_max: float, rot=None, flips=None, noise=None, noise_offsets=None):
    """include/strata_hip.h: sn2_prepare_plots.  raw (10,T) f32, offsets (B+1) i32, centers (B,2) f32, fake_xy (F,2) f32,
    idx (B,N) i32; train iff rot (B,2) f64 and flips (B,2) i32 are given; noise (6,Tn) f32 + noise_offsets (B) i64 optional.
    -> cloud (B,10,N), xyz (B,3,N)."""
    C, T = raw.shape
    B, N = idx.shape
    _chk(raw, F32, (10, T), "raw")
    _chk(offsets, I32, (B + 1,), "offsets")
    _chk(centers, F32, (B, 2), "centers")
    F = fake_xy.shape[0]
    _chk(fake_xy, F32, (F, 2), "fake_xy")
    _chk(idx, I32, (B, N), "idx")
    train = rot is not None
    if train:
        _chk(rot, F64, (B, 2), "rot")
        _chk(flips, I32, (B, 2), "flips")
    Tn = 0
    if noise is not None:
        Tn = noise.shape[1]
        _chk(noise, F32, (6, Tn), "noise")
        _chk(noise_offsets, I64, (B,), "noise_offsets")
    dev = raw.device
    cloud = torch.empty(B, 10, N, dtype=F32, device=dev)
    xyz = torch.empty(B, 3, N, dtype=F32, device=dev)
    _call("sn2_prepare_plots", _ptr(raw), T, _ptr(offsets), _ptr(centers), _ptr(fake_xy), F, _ptr(idx), B, N, int(train),
          _ptr(rot), _ptr(flips), _ptr(noise), _ptr(noise_offsets), Tn, float(z_max), _ptr(cloud), _ptr(xyz), _stream())
    return cloud, xyz


def znorm(xyz: torch.Tensor, radius: float):
    """xyz (3,n) fp32 of ONE raw plot on the device -> (zmin (n), z - zmin (n)): the local-minimum z-normalisation of
    `normalize_z_with_minz_in_a_radius` (utils/load_data.py:237-249)."""
    _, n = xyz.shape
    _chk(xyz, F32, (3, n), "xyz")
    if not radius > 0:
        raise ValueError("znorm: radius must be positive")
    lo = xyz[:2].min(1).values.tolist()          # the caller usually has the bounding box already; here one small sync
    hi = xyz[:2].max(1).values.tolist()
    cells = (int((hi[0] - lo[0]) / radius) + 2) * (int((hi[1] - lo[1]) / radius) + 2)
    ws = torch.empty(5 * n + 3 * cells + 8, dtype=I32, device=xyz.device)
    zmin = torch.empty(n, dtype=F32, device=xyz.device)
    zout = torch.empty(n, dtype=F32, device=xyz.device)
    _call("sn2_znorm", _ptr(xyz[0]), _ptr(xyz[1]), _ptr(xyz[2]), n, float(radius), lo[0], lo[1], hi[0], hi[1], _ptr(ws),
          _ptr(zmin), _ptr(zout), _stream())
    return zmin, zout


def sa_order_len(B: int, M: int) -> int:
    """SN2_SA_ORDER_WORDS of include/strata_hip.h."""
    return 4 * B * M + 16 * B * (M // 8 + 2) + 8


def sa_order(cnt: torch.Tensor, B: int, M: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """cnt (B*M) from ball_query -> the work items of the SA passes (include/strata_hip.h: sn2_sa_order): solo centroids
    with long lists, quads of centroids with medium ones, and eight / sixteen centroids with at most 8 / 4 neighbours packed into
    one step, heaviest first."""
    _chk(cnt, I32, (B * M,), "cnt")
    n = sa_order_len(B, M)
    if out is None:
        out = torch.empty(n, dtype=I32, device=cnt.device)
    else:
        _chk(out, I32, (n,), "out order")
    _call("sn2_sa_order", _ptr(cnt), B, M, _ptr(out), _stream())
    return out


def sa_order_group(cnt: torch.Tensor, G: int, B: int, M: int, out_all: torch.Tensor, stride: int):
    """include/strata_hip.h: sn2_sa_order_group -- the work items of G consecutive batches of B plots in one launch pair; batch h's
    table = out_all[h * stride : h * stride + sa_order_len(B, M)]."""
    _chk(cnt, I32, (G * B * M,), "cnt")
    if stride < sa_order_len(B, M):
        raise ValueError("sa_order_group: stride smaller than a table")
    _chk(out_all, I32, (G * stride,), "out order tables")
    _call("sn2_sa_order_group", _ptr(cnt), G, B, M, _ptr(out_all), stride, _stream(), key="sn2_sa_order")


def sa_desc(blocks, feat, cf, spos, cpos_aos, nbr, cnt, total, B, Nsrc, M, ext, arg, out, dout=None, dfeat=None,
            with_grads=False, order=None) -> SA:
    """feat: (B*Nsrc, >=cf) row view; spos: (B*Nsrc, >=4) row view holding x,y,z,."""
    cap = nbr.shape[1]
    cl = blocks[-1].cout
    feat_stride = _chk_rows(feat, F32, B * Nsrc, cf, "feat")
    spos_stride = _chk_rows(spos, F32, B * Nsrc, 4, "spos")
    _chk(cpos_aos, F32, (B * M, 4), "cpos_aos")
    _chk(nbr, I32, (B * M, cap), "nbr")
    _chk(cnt, I32, (B * M,), "cnt")
    _chk(total, I64, (1,), "total")
    _chk(ext, F32, (B * M, cl), "ext")
    _chk(arg, I32, (B * M, cl), "arg")
    _chk(out, F32, (B * M, cl), "out")
    d = SA()
    d.B, d.Nsrc, d.M, d.cap, d.cf, d.nl = B, Nsrc, M, cap, cf, len(blocks)
    d.feat, d.feat_stride, d.spos, d.spos_stride = _ptr(feat), feat_stride, _ptr(spos), spos_stride
    d.cpos, d.nbr, d.cnt, d.total = _ptr(cpos_aos), _ptr(nbr), _ptr(cnt), _ptr(total)
    if order is not None:
        _chk(order, I32, (sa_order_len(B, M),), "order")
    d.order = _ptr(order)
    for i, bb in enumerate(blocks):
        bb.fill(d.blk[i], with_grads)
    d.ext, d.arg, d.out = _ptr(ext), _ptr(arg), _ptr(out)
    if dout is not None:
        _chk(dout, F32, (B * M, cl), "dout")
    if dfeat is not None:
        _chk(dfeat, F32, (B * Nsrc, cf), "dfeat")
    d.dout, d.dfeat = _ptr(dout), _ptr(dfeat)
    return d


def sa_forward(d: SA, training: bool):
    _call("sn2_sa_forward", d, int(training), _stream(), tag=f"cf={d.cf}")


def sa_backward(d: SA):
    _call("sn2_sa_backward", d, _stream(), tag=f"cf={d.cf}")


def interp_ws_words(B: int, R_per_plot: int, S_per_plot: int) -> int:
    """SN2_INTERP_WS_WORDS of include/strata_hip.h."""
    return (B * S_per_plot * ((R_per_plot + 2047) // 2048 + 6) + 6 * B * R_per_plot + 64
            + 4 * B * interp_chunks(R_per_plot, S_per_plot))


def interp_chunks(R_per_plot: int, S_per_plot: int) -> int:
    """SN2_INTERP_CHUNKS of include/strata_hip.h: slots of the per-plot chunk table of an inverted index."""
    return (3 * R_per_plot + 62) // 63 + S_per_plot


def interp_index(knn, B: int, R_per_plot: int, S_per_plot: int, out: Optional[torch.Tensor] = None,
                 src_pos: Optional[torch.Tensor] = None, row_perm: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Inverted index of a 3-NN table (source -> list of (target row, normalised weight)): what the backward of the
    interpolation gathers through.  Position-only, so it can be built in the geometry pass.  src_pos (B*S,4): the
    source positions; given, the index also orders every plot's sources along a Morton curve (L2 locality of the
    source-side backward).  row_perm (B*R_per_plot) int32, a permutation of 0..R-1 per plot: the lists name row_perm[row]
    instead of row -- where `fp_desc(..., row_perm=)` keeps the row's d pre-activation (include/strata_hip.h: sn2_fp.row_perm)."""
    if src_pos is not None:
        _chk(src_pos, F32, (B * S_per_plot, 4), "src_pos")
    if row_perm is not None:
        _chk(row_perm, I32, (B * R_per_plot,), "row_perm")
    R = B * R_per_plot
    _chk(knn[0], I32, (R, 3), "knn_idx")
    _chk(knn[1], F32, (R, 3), "knn_w")
    n = interp_ws_words(B, R_per_plot, S_per_plot)
    if out is None:
        out = torch.empty(n, dtype=F32, device=knn[0].device)
    else:
        _chk(out, F32, (n,), "out index")
    _call("sn2_interp_index_perm", _ptr(knn[0]), _ptr(knn[1]), _ptr(src_pos), _ptr(row_perm), B, R_per_plot, S_per_plot, _ptr(out),
          _stream(), key="sn2_interp_index")
    return out


def interp_index_group(knn, G: int, B: int, R_per_plot: int, S_per_plot: int, out_all: torch.Tensor, stride: int,
                       src_pos: Optional[torch.Tensor] = None, row_perm: Optional[torch.Tensor] = None):
    """include/strata_hip.h: sn2_interp_index_group -- the inverted indices of G consecutive batches of B plots in ONE set of four
    launches; knn / src_pos / row_perm: the group's arrays; batch h's index = out_all[h * stride : h * stride + interp_ws_words(...)]."""
    R = G * B * R_per_plot
    _chk(knn[0], I32, (R, 3), "knn_idx")
    _chk(knn[1], F32, (R, 3), "knn_w")
    if src_pos is not None:
        _chk(src_pos, F32, (G * B * S_per_plot, 4), "src_pos")
    if row_perm is not None:
        _chk(row_perm, I32, (R,), "row_perm")
    if stride % 4 or stride < interp_ws_words(B, R_per_plot, S_per_plot):
        raise ValueError("interp_index_group: stride must be a multiple of 4 words and hold one index")
    _chk(out_all, F32, (G * stride,), "out indices")
    _call("sn2_interp_index_group", _ptr(knn[0]), _ptr(knn[1]), _ptr(src_pos), _ptr(row_perm), G, B, R_per_plot, S_per_plot, _ptr(out_all),
          stride, _stream(), key="sn2_interp_index")


SOURCE_SIDE = True     # False: never hand out src_ws, i.e. every row rebuilds its interpolated input (tests compare both)


def fp_desc(block: BlockBuffers, B, R_per_plot, S_per_plot, ca, cb, src, h, src_affine=None, knn=None, skip=None,
            dy=None, dsrc=None, dskip=None, du_scratch=None, with_grads=False, interp_index=None,
            bn_sums_done=None, row_perm=None, force_src_ws=False, gather=True) -> FP:
    """src: (B*S_per_plot, >=ca) rows when knn is given, else (B*R_per_plot, >=ca); skip: (B*R_per_plot, >=cb) row view.
    h of dtype bfloat16 (then dy and du_scratch too): the per-point layer stores its three activation buffers in bfloat16
    (include/strata_hip.h: sn2_fp.act_bf16; BASELINE config 5) -- only the source-side form of a layer of more than
    64 * SN2_STAT_SLOTS rows has those kernels."""
    R = B * R_per_plot
    hs = (block.cout + 3) // 4 * 4
    AT = BF16 if (h is not None and h.dtype == BF16) else F32         # the storage type of h / dy / du_scratch
    n_src_rows = R if knn is None else B * S_per_plot
    src_stride = _chk_rows(src, F32, n_src_rows, ca, "src")
    if src.shape[1] < (ca + 3) // 4 * 4 and src_stride < (ca + 3) // 4 * 4:
        raise ValueError("fp: src rows must be padded to a multiple of 4 floats")
    if h is not None or not force_src_ws:
        _chk(h, AT, (R, hs), "h")
    d = FP()
    d.B, d.R_per_plot, d.S_per_plot, d.ca, d.cb = B, R_per_plot, S_per_plot, ca, cb
    d.src, d.src_stride = _ptr(src), src_stride
    if src_affine is not None:
        _chk(src_affine[0], F32, (ca,), "src_a")
        _chk(src_affine[1], F32, (ca,), "src_c")
        d.src_a, d.src_c = _ptr(src_affine[0]), _ptr(src_affine[1])
    else:
        d.src_a = d.src_c = None
    if knn is not None:
        _chk(knn[0], I32, (R, 3), "knn_idx")
        _chk(knn[1], F32, (R, 3), "knn_w")
        d.knn_idx, d.knn_w = _ptr(knn[0]), _ptr(knn[1])
    else:
        d.knn_idx = d.knn_w = None
    if cb > 0:
        d.skip_stride = _chk_rows(skip, F32, R, cb, "skip", align=4 if cb % 4 == 0 else 1)
        d.skip = _ptr(skip)
    else:
        d.skip, d.skip_stride = None, 0
    block.fill(d.blk, with_grads)
    if R > 64 * _lib.STAT_SLOTS:
        # bfloat16 operands exist on the matrix-core kernels of the layers over centroids (<= 64 * SN2_STAT_SLOTS rows);
        # a block with more rows (FP2 at the reference's default ratio1 = 0.5: 262 144 rows) runs its fp32 row kernels
        d.blk.mma_bf16 = 0
    d.h, d.h_stride = _ptr(h), hs
    if dy is not None:
        _chk(dy, AT, (R, hs), "dy")
    d.dsrc_stride = d.dskip_stride = 0
    if dsrc is not None:
        d.dsrc_stride = _chk_rows(dsrc, F32, n_src_rows, ca, "dsrc", align=1)
    if dskip is not None:
        d.dskip_stride = _chk_rows(dskip, F32, R, cb, "dskip", align=1)
    d.scatter_ws, d.scatter_ready = None, 0
    if bn_sums_done is not None:
        _chk(bn_sums_done, I32, (1,), "bn_sums_done")
    d.bn_sums_done = _ptr(bn_sums_done)
    # source-side workspace of the per-point layer (include/strata_hip.h: src_ws); scratch, so one per descriptor
    d._src_ws = None
    if SOURCE_SIDE and knn is not None and 0 < cb <= 16 and cb % 4 == 0 and (R > 64 * _lib.STAT_SLOTS or force_src_ws):
        d._src_ws = torch.empty(B * interp_chunks(R_per_plot, S_per_plot) * hs, dtype=F32, device=src.device)
    d.src_ws = _ptr(d._src_ws)
    d.row_perm = None
    if row_perm is not None and d._src_ws is not None and du_scratch is not None:
        # the backward pass keeps its d pre-activation rows in this order; the interp_index handed in must have been built
        # with the same permutation
        _chk(row_perm, I32, (R,), "row_perm")
        d.row_perm = _ptr(row_perm)
    d.act_bf16 = int(AT == BF16)
    if d.act_bf16 and d._src_ws is None:
        raise ValueError("fp: bfloat16 activation rows need the source-side form (a k-NN layer of more than 65 536 rows)")
    if du_scratch is not None:
        _chk(du_scratch, AT, (R, max(ca, hs)), "du_scratch")
        if knn is not None and dsrc is not None:
            # inverted index of the 3-NN table: prebuilt by interp_index (geometry pass) or built by the backward call
            words = interp_ws_words(B, R_per_plot, S_per_plot)
            if interp_index is not None:
                _chk(interp_index, F32, (words,), "interp_index")
                d._scatter_ws, d.scatter_ready = interp_index, 1
            else:
                d._scatter_ws = torch.empty(words, dtype=F32, device=src.device)
            d.scatter_ws = _ptr(d._scatter_ws)
            if not gather:
                d.scatter_ready = -1          # the per-row input gradients stay in du_scratch: the caller transposes (global_pool_backward)
    d.dy, d.dsrc, d.dskip, d.du_scratch = _ptr(dy), _ptr(dsrc), _ptr(dskip), _ptr(du_scratch)
    return d


def fp_bn_sums(d: FP, gamma, beta, mean, invstd, dgamma, dbeta, ok):
    """After fp_backward(d): dgamma/dbeta (ACCUMULATED, complete) of the BatchNorm whose output block d interpolates, from d's
    own weight and bias gradients -- or, where a |gamma| is too small for that, summed over the rows by the same kernel
    (include/strata_hip.h: sn2_fp_bn_sums).  mean, invstd: the BatchNorm's saved batch statistics."""
    for t, n in ((gamma, "gamma"), (beta, "beta"), (mean, "mean"), (invstd, "invstd"), (dgamma, "dgamma"), (dbeta, "dbeta")):
        _chk(t, F32, (d.ca,), n)
    _chk(ok, I32, (1,), "ok")
    _call("sn2_fp_bn_sums", d, _ptr(gamma), _ptr(beta), _ptr(mean), _ptr(invstd), _ptr(dgamma), _ptr(dbeta), _ptr(ok), _stream())


def fp_forward(d: FP, training: bool):
    _call("sn2_fp_forward", d, int(training), _stream(), tag=f"{d.ca}+{d.cb}->{d.blk.cout}")


def fp_backward(d: FP):
    _call("sn2_fp_backward", d, _stream(), tag=f"{d.ca}+{d.cb}->{d.blk.cout}")


def plot_max_forward(h, a, c, B, R_per_plot, C):
    hs = (C + 3) // 4 * 4
    _chk(h, F32, (B * R_per_plot, hs), "h")
    _chk(a, F32, (C,), "a")
    _chk(c, F32, (C,), "c")
    out = torch.empty(B, C, dtype=F32, device=h.device)
    arg = torch.empty(B, C, dtype=I32, device=h.device)
    _call("sn2_plot_max_forward", _ptr(h), _ptr(a), _ptr(c), B, R_per_plot, C, _ptr(out), _ptr(arg), _stream())
    return out, arg


GL_MAX_PLOTS = 28        # sn2_global_level_forward's limit (csrc/fp.hip: GL_MAX_PLOTS)
_GLOBAL_WS = {}          # device index -> workspace of callers that name no owner
_GLOBAL_WS_ALL = []      # weak references to every live workspace (global_level_gave_up looks at all of them)


class _GlobalWs(list):
    """[exchange granules (int64), control words (int32), give-ups already reported]"""
    __slots__ = ("__weakref__",)


def global_level_ws(dev, B: int = GL_MAX_PLOTS, owner=None):
    """The exchange area and control words of `global_level_forward` (zero-filled once, then the library's).
    owner: the object whose training forwards use it (a PointNet2): the workspace lives on it, so two models that train on two
    streams of one process never share an exchange area; without an owner there is one per device.  Launches that share a
    workspace must be on one stream at a time.  Allocated ONCE at the largest size the kernel takes (28 plots: 229 KB) and
    never again: hipGraphs captured earlier hold its raw address and its launch epoch (round 4 reallocated it when a later call
    had more plots, under the feet of the graphs captured before)."""
    dev = torch.device(dev)
    if B > GL_MAX_PLOTS:
        raise ValueError(f"global_level_forward takes at most {GL_MAX_PLOTS} plots")
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    store = _GLOBAL_WS if owner is None else owner.__dict__.setdefault("_gl_ws", {})
    ws = store.get(key)
    if ws is None:
        if torch.cuda.is_current_stream_capturing():
            # zero fills captured into a graph would run at every replay and reset the launch epoch under the other graphs
            raise StrataHipError("global_level_forward: its exchange area must exist before a stream capture starts -- run one "
                                 "eager training forward first (TrainPipeline.capture does) or call hip_ops.global_level_ws(dev, owner=model)")
        ws = _GlobalWs([torch.zeros(2 * GL_MAX_PLOTS * 4 * 128, dtype=I64, device=dev), torch.zeros(8, dtype=I32, device=dev), 0])
        store[key] = ws
        import weakref
        _GLOBAL_WS_ALL[:] = [r for r in _GLOBAL_WS_ALL if r() is not None]
        _GLOBAL_WS_ALL.append(weakref.ref(ws))
    return ws


def global_level_forward(d_sa3: FP, d_fp3: FP, x3: torch.Tensor, arg3: torch.Tensor, owner=None):
    """SA3 -> BatchNorm -> plot max -> FP3 -> BatchNorm in one launch (training mode; include/strata_hip.h).  d_fp3 must have
    been built with src = x3.  owner: see `global_level_ws`."""
    B = d_sa3.B
    _chk(x3, F32, (B, 64), "x3")
    _chk(arg3, I32, (B, 64), "arg3")
    ws = global_level_ws(x3.device, B, owner=owner)
    _call("sn2_global_level_forward", d_sa3, d_fp3, _ptr(x3), _ptr(arg3), _ptr(ws[0]), _ptr(ws[1]), _stream())


def global_level_gave_up(dev, warn: bool = True) -> int:
    """Exchange waits of `global_level_forward` that gave up on this device since the process started (a workgroup of a launch
    was not resident within the spin limit: another stream or process held the CUs), over every live workspace.  Each such
    launch was REPAIRED by the gated launch behind it (sn2_global_level_forward: the level is recomputed by one workgroup),
    so the results are unaffected; a count that grew since the last call is reported once as a StrataHipWarning (the passes took
    longer; `PointNet2.fuse_global_level = False` runs the level as separate launches).  Reads device words: call it where the
    host synchronises anyway (TrainPipeline.drain, after a test)."""
    dev = torch.device(dev)
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    total = 0
    for r in list(_GLOBAL_WS_ALL):
        ws = r()
        if ws is None or ws[1].device.index != key:
            continue
        n = int(ws[1][1].item())
        total += n
        if n > ws[2]:
            seen, ws[2] = ws[2], n
            if warn:
                import warnings
                warnings.warn(f"global_level_forward: {n - seen} exchange wait(s) gave up on {dev}; those launches were repeated "
                              "by the single-workgroup repair launch (results unaffected, the passes took longer); set "
                              "PointNet2.fuse_global_level = False where other kernels or processes share the device",
                              StrataHipWarning, stacklevel=2)
    return total


def plot_max_backward(dout, arg, B, R_per_plot, C, dy):
    hs = (C + 3) // 4 * 4
    _chk(dout, F32, (B, C), "dout")
    _chk(arg, I32, (B, C), "arg")
    _chk(dy, F32, (B * R_per_plot, hs), "dy")
    _call("sn2_plot_max_backward", _ptr(dout), _ptr(arg), B, R_per_plot, C, _ptr(dy), _stream())


def global_pool_backward(du, arg, h, mean, invstd, B, R_per_plot, dx, dy, dgamma, dbeta):
    """include/strata_hip.h: sn2_global_pool_backward -- the backward of the pool between FP3 and SA3 in one launch."""
    _chk(du, F32, (B * R_per_plot, 64), "du")
    _chk(arg, I32, (B, 64), "arg")
    _chk(h, F32, (B * R_per_plot, 64), "h")
    for t, n in ((mean, "mean"), (invstd, "invstd"), (dgamma, "dgamma"), (dbeta, "dbeta")):
        _chk(t, F32, (64,), n)
    _chk(dx, F32, (B, 64), "dx")
    _chk(dy, F32, (B * R_per_plot, 64), "dy")
    _call("sn2_global_pool_backward", _ptr(du), 64, _ptr(arg), _ptr(h), _ptr(mean), _ptr(invstd), B, R_per_plot, 64, _ptr(dx), _ptr(dy),
          _ptr(dgamma), _ptr(dbeta), _stream())


def dropout_mask_words(keep: torch.Tensor) -> torch.Tensor:
    """keep (R,16) bool / 0-1 values (True = the hidden channel survives F.dropout) -> (R) int32 words, bit j = channel j."""
    if keep.dim() != 2 or keep.shape[1] != 16:
        raise ValueError("dropout mask must be (rows, 16)")
    pow2 = (2 ** torch.arange(16, device=keep.device, dtype=torch.int32))
    return ((keep != 0).to(torch.int32) * pow2).sum(1, dtype=torch.int32).contiguous()


def head_desc(f, fa, fc, lin1, lin2, coverages=None, proba=None, dcov=None, dproba=None, dy=None, grads=None,
              grad_images=(1, 0), drop_mask=None, drop_p: float = 0.0, rows: Optional[int] = None) -> Head:
    """f (R,36): the rows the head reads -- or None with `rows` = R for `fp_head_eval`, whose rows never reach memory."""
    if f is None:
        if rows is None or coverages is None:
            raise ValueError("head_desc: without f the row count and the outputs must be given")
        R, AT = int(rows), F32
    else:
        R = f.shape[0]
        AT = BF16 if f.dtype == BF16 else F32         # bfloat16 rows of f (and dy): sn2_head.act_bf16
        _chk(f, AT, (R, 36), "f")
    _chk(fa, F32, (34,), "fa")
    _chk(fc, F32, (34,), "fc")
    _chk(lin1.weight, F32, (16, 34), "lin1.weight")
    _chk(lin1.bias, F32, (16,), "lin1.bias")
    _chk(lin2.weight, F32, (5, 16), "lin2.weight")
    _chk(lin2.bias, F32, (5,), "lin2.bias")
    d = Head()
    d.R, d.cin, d.f_stride = R, 34, 36
    d.f, d.fa, d.fc = _ptr(f), _ptr(fa), _ptr(fc)
    d.W1, d.b1, d.W2, d.b2 = _ptr(lin1.weight), _ptr(lin1.bias), _ptr(lin2.weight), _ptr(lin2.bias)
    for t, n in ((coverages, "coverages"), (proba, "proba"), (dcov, "dcoverages"), (dproba, "dproba")):
        if t is not None:
            _chk(t, F32, (R, 4), n)
    d.coverages, d.proba, d.dcoverages, d.dproba = _ptr(coverages), _ptr(proba), _ptr(dcov), _ptr(dproba)
    if dy is not None:
        _chk(dy, AT, (R, 36), "dy")
    d.dy = _ptr(dy)
    d.act_bf16 = int(AT == BF16)
    if grads is not None:
        for t, ref in zip(grads, (lin1.weight, lin1.bias, lin2.weight, lin2.bias)):
            _chk(t, F32, ref.shape, "head grad view")
        d.dW1, d.db1, d.dW2, d.db2 = (_ptr(t) for t in grads)
        d.grad_replicas, d.grad_replica_stride = grad_images
    else:
        d.dW1 = d.db1 = d.dW2 = d.db2 = None
        d.grad_replicas, d.grad_replica_stride = 1, 0
    if drop_mask is not None:
        _chk(drop_mask, I32, (R,), "drop_mask")
        d.drop_mask, d.drop_scale = _ptr(drop_mask), (1.0 / (1.0 - drop_p) if drop_p < 1.0 else 0.0)
    else:
        d.drop_mask, d.drop_scale = None, 1.0
    return d


GRAD_IMAGES = 32       # images of the flat parameter gradient the backward kernels spread their atomics over


def flat_layout(params):
    """Offsets of the parameters inside the flat parameter / gradient vectors (back to back, in model.parameters() order)
    and the vectors' length."""
    offs, o = [], 0
    for p in params:
        offs.append(o)
        o += p.numel()
    return offs, o


def grad_images_alloc(n_flat: int, device, extra_words: int = 0):
    """One zero-filled arena: GRAD_IMAGES images of the flat gradient (image stride = n_flat rounded up to 64 floats)
    followed by `extra_words` floats.  Returns (arena, image 0 view (n_flat), (replicas, stride), extra view)."""
    stride = (n_flat + 63) // 64 * 64
    arena = torch.zeros(GRAD_IMAGES * stride + extra_words, dtype=F32, device=device)
    return arena, arena[:n_flat], (GRAD_IMAGES, stride), arena[GRAD_IMAGES * stride:]


def grad_reduce(arena, n_flat: int, images):
    """include/strata_hip.h: sn2_grad_reduce -- fold the images into image 0 (arena[:n_flat])."""
    replicas, stride = images
    _chk(arena, F32, None, "arena")
    if arena.numel() < replicas * stride:
        raise ValueError("grad_reduce: arena smaller than its images")
    _call("sn2_grad_reduce", _ptr(arena), n_flat, replicas, stride, _stream())


def fp_head_eval(d: FP, hd: Head):
    """include/strata_hip.h: sn2_fp_head_eval -- EVAL: the per-point layer and the head in one pass, no (B*N,36) buffer between."""
    _call("sn2_fp_head_eval", d, hd, _stream())


def head_forward(d: Head):
    _call("sn2_head_forward", d, _stream())


def head_backward(d: Head):
    _call("sn2_head_backward", d, _stream())


def head_bn_sums(d: Head, gamma, beta, mean, invstd, dgamma, dbeta, ok):
    """After head_backward: dgamma/dbeta (ACCUMULATED, complete) of the BatchNorm feeding lin1, from lin1's gradients (or by
    the kernel's own pass over the rows where a |gamma| is too small)."""
    for t, n in ((gamma, "gamma"), (beta, "beta"), (mean, "mean"), (invstd, "invstd"), (dgamma, "dgamma"), (dbeta, "dbeta")):
        _chk(t, F32, (d.cin,), n)
    _chk(ok, I32, (1,), "ok")
    _call("sn2_head_bn_sums", d, _ptr(gamma), _ptr(beta), _ptr(mean), _ptr(invstd), _ptr(dgamma), _ptr(dbeta), _ptr(ok), _stream())


# ---------------------------------------------------------------------------------------------- projections
def _xy_rows(clouds_dev: torch.Tensor):
    """clouds (B,C>=2,N) on device -> (pointer tensor, plot stride in floats)."""
    B, C, N = clouds_dev.shape
    _chk(clouds_dev, F32, (B, C, N), "clouds")
    if C < 2:
        raise ValueError("clouds needs at least the x and y rows")
    return clouds_dev, C * N


def plot_project_forward(pred_pointwise: torch.Tensor, clouds_dev: torch.Tensor, diam_pix: int):
    B, C, N = clouds_dev.shape
    _chk(pred_pointwise, F32, (B * N, 4), "pred_pointwise")
    t, stride = _xy_rows(clouds_dev)
    dev = pred_pointwise.device
    D = int(diam_pix)
    keys = torch.empty(B * D * D * 3, dtype=I64, device=dev)
    pix = torch.empty(B * N, dtype=I32, device=dev)
    arg = torch.empty(B * D * D * 3, dtype=I32, device=dev)
    nocc = torch.empty(B, dtype=I32, device=dev)
    pred = torch.empty(B, 4, dtype=F32, device=dev)
    _call("sn2_plot_project_forward", _ptr(pred_pointwise), _ptr(t), stride, B, N, D, _ptr(keys), _ptr(pix),
                                               _ptr(arg), _ptr(nocc), _ptr(pred), _stream())
    return pred, pix, arg, nocc


def p2_key_parts(N: int) -> int:
    """SN2_P2_KEY_PARTS of include/strata_hip.h."""
    return max(1, min(64, (N + 4095) // 4096))


def plot_pixels(clouds_dev: torch.Tensor, diam_pix: int, out=None):
    """clouds (B,C,N) on the device -> (mm (B,4) bounding boxes, pix (B*N) int32): the pixel ids of
    `project_to_plotwise_coverages` (project_to_2d.py:16-22), computed from the positions alone -- what a pipelined loop runs
    ahead of the feature pass (include/strata_hip.h: sn2_plot_pixels).  out = (mm, pix): caller-owned buffers."""
    B, C, N = clouds_dev.shape
    t, stride = _xy_rows(clouds_dev)
    if out is None:
        mm = torch.empty(B, 4, dtype=F32, device=clouds_dev.device)
        pix = torch.empty(B * N, dtype=I32, device=clouds_dev.device)
    else:
        mm, pix = out
        _chk(mm, F32, (B, 4), "out mm")
        _chk(pix, I32, (B * N,), "out pix")
    _call("sn2_plot_pixels", _ptr(t), stride, B, N, int(diam_pix), _ptr(mm), _ptr(pix), _stream())
    return mm, pix


def plot_project_forward_pix(pred_pointwise: torch.Tensor, pix: torch.Tensor, B: int, N: int, diam_pix: int):
    """`plot_project_forward` from pixel ids that `plot_pixels` computed ahead of time: two launches, no key table to clear."""
    _chk(pred_pointwise, F32, (B * N, 4), "pred_pointwise")
    _chk(pix, I32, (B * N,), "pix")
    dev = pred_pointwise.device
    D = int(diam_pix)
    keys = torch.empty(p2_key_parts(N) * B * D * D * 3, dtype=I64, device=dev)
    arg = torch.empty(B * D * D * 3, dtype=I32, device=dev)
    nocc = torch.empty(B, dtype=I32, device=dev)
    pred = torch.empty(B, 4, dtype=F32, device=dev)
    _call("sn2_plot_project_forward_pix", _ptr(pred_pointwise), _ptr(pix), B, N, D, _ptr(keys), _ptr(arg), _ptr(nocc), _ptr(pred),
          _stream(), key="sn2_plot_project_forward")
    return pred, pix, arg, nocc


def plot_project_backward(dpred, arg, nocc, pix, B, N, diam_pix):
    D = int(diam_pix)
    _chk(dpred, F32, (B, 4), "dpred")
    _chk(arg, I32, (B * D * D * 3,), "arg")
    _chk(nocc, I32, (B,), "nocc")
    _chk(pix, I32, (B * N,), "pix")
    dpw = torch.empty(B * N, 4, dtype=F32, device=dpred.device)
    _call("sn2_plot_project_backward", _ptr(dpred), _ptr(arg), _ptr(nocc), _ptr(pix), B, N, D, _ptr(dpw), _stream())
    return dpw


def raster_project(coverages: torch.Tensor, clouds_dev: torch.Tensor, diam_pix: int, diam_meters: int):
    """coverages (B*N,4), clouds (B,C,N) -> rasters (B,3,D,D) fp32 with NaN, pix (B*N) int32."""
    B, C, N = clouds_dev.shape
    _chk(coverages, F32, (B * N, 4), "coverages")
    t, stride = _xy_rows(clouds_dev)
    dev = coverages.device
    D = int(diam_pix)
    keys = torch.empty(B * D * D * 3, dtype=I64, device=dev)
    pix = torch.empty(B * N, dtype=I32, device=dev)
    rasters = torch.empty(B, 3, D, D, dtype=F32, device=dev)
    _call("sn2_raster_project", _ptr(coverages), _ptr(t), stride, B, N, D, int(diam_meters), _ptr(keys),
                                         _ptr(pix), _ptr(rasters), _stream())
    return rasters, pix


def mosaic_merge(rasters, weights, offsets, mean, wsum, window=None):
    """Fold plots 0..B-1 (in order) into the running mosaic (mean, wsum) (3,H,W), NaN = no data."""
    B, _, D, _ = rasters.shape
    _, H, W = mean.shape
    _chk(rasters, F32, (B, 3, D, D), "rasters")
    _chk(weights, F32, (D, D), "weights")
    _chk(offsets, I32, (B, 2), "offsets")
    _chk(mean, F32, (3, H, W), "mean")
    _chk(wsum, F32, (3, H, W), "wsum")
    y0, x0, wh, ww = (0, 0, H, W) if window is None else [int(v) for v in window]
    _call("sn2_mosaic_merge", _ptr(rasters), _ptr(weights), _ptr(offsets), B, D, H, W, _ptr(mean), _ptr(wsum),
          y0, x0, wh, ww, _stream())


def mosaic_finalize(mean: torch.Tensor, wsum: torch.Tensor):
    """mean (3,H,W), wsum (H,W) -> out (5,H,W) = [Vb, Vm_soft, Vh, Vm_hard, weights], thr (2,) = threshold, its index."""
    _, H, W = mean.shape
    _chk(mean, F32, (3, H, W), "mean")
    _chk(wsum, F32, (H, W), "wsum")
    dev = mean.device
    hist = torch.empty(10004, dtype=I32, device=dev)
    acc = torch.empty(1, dtype=F64, device=dev)
    thr = torch.empty(2, dtype=F32, device=dev)
    out = torch.empty(5, H, W, dtype=F32, device=dev)
    _call("sn2_mosaic_finalize", _ptr(mean), _ptr(wsum), H, W, _ptr(hist), _ptr(acc), _ptr(thr), _ptr(out), _stream())
    return out, thr


LOSS_BLOCKS = 1024


def kde_lookup(clouds_dev: torch.Tensor, z_max: float, X: torch.Tensor, Y: torch.Tensor, z_channel: int = 2) -> torch.Tensor:
    """clouds (B,C,N) fp32 on the device, X (K) / Y (3,K) fp64 interpolation tables -> pdf_all (B*N,3) fp64."""
    B, C, N = clouds_dev.shape
    _chk(clouds_dev, F32, (B, C, N), "clouds")
    K = X.shape[0]
    _chk(X, F64, (K,), "X")
    _chk(Y, F64, (3, K), "Y")
    if K < 2 or not (0 <= z_channel < C):
        raise ValueError("kde_lookup: need K >= 2 knots and a valid z channel")
    pdf = torch.empty(B * N, 3, dtype=F64, device=clouds_dev.device)
    _call("sn2_kde_lookup", _ptr(clouds_dev), B, C, N, int(z_channel), float(z_max), _ptr(X), _ptr(Y), K, _ptr(pdf), _stream())
    return pdf


def loss_forward(pred, gt, proba, pdf, m: float, e: float):
    """-> out (4,) fp64 = [total, absolute, NLL, entropy] (include/strata_hip.h: sn2_loss_forward)."""
    B, R = pred.shape[0], proba.shape[0]
    _chk(pred, F32, (B, 4), "pred")
    _chk(gt, F64, (B, 4), "gt")
    _chk(proba, F32, (R, 4), "proba")
    _chk(pdf, F64, (R, 3), "pdf")
    partials = torch.empty(2 * LOSS_BLOCKS, dtype=F64, device=pred.device)
    out = torch.empty(4, dtype=F64, device=pred.device)
    _call("sn2_loss_forward", _ptr(pred), _ptr(gt), B, _ptr(proba), _ptr(pdf), R, float(m), float(e), _ptr(partials),
          _ptr(out), _stream())
    return out


def loss_backward(pred, gt, proba, pdf, m: float, e: float, grad_total):
    B, R = pred.shape[0], proba.shape[0]
    _chk(grad_total, F64, None, "grad_total")
    dpred = torch.empty(B, 4, dtype=F32, device=pred.device)
    dproba = torch.empty(R, 4, dtype=F32, device=pred.device)
    _call("sn2_loss_backward", _ptr(pred), _ptr(gt), B, _ptr(proba), _ptr(pdf), R, float(m), float(e), _ptr(grad_total),
          _ptr(dpred), _ptr(dproba), _stream())
    return dpred, dproba


PROJECTED_LOSS_WS = 2 * 512 + 2      # SN2_PROJECTED_LOSS_WS


def projected_loss_forward(cov, pix, proba, pdf, gt, B: int, N: int, diam_pix: int, m: float, e: float):
    """include/strata_hip.h: sn2_projected_loss_forward -> (out (4,) fp64 = total, absolute, NLL, entropy; pred (B,4); arg, nocc)."""
    R, D = B * N, int(diam_pix)
    _chk(cov, F32, (R, 4), "coverages")
    _chk(pix, I32, (R,), "pix")
    _chk(proba, F32, (R, 4), "proba")
    _chk(pdf, F64, (R, 3), "pdf")
    _chk(gt, F64, (B, 4), "gt")
    dev = cov.device
    keys = torch.empty(p2_key_parts(N) * B * D * D * 3, dtype=I64, device=dev)
    arg = torch.empty(B * D * D * 3, dtype=I32, device=dev)
    nocc = torch.empty(B, dtype=I32, device=dev)
    pred = torch.empty(B, 4, dtype=F32, device=dev)
    partials = torch.empty(PROJECTED_LOSS_WS, dtype=F64, device=dev)
    out = torch.empty(4, dtype=F64, device=dev)
    _call("sn2_projected_loss_forward", _ptr(cov), _ptr(pix), _ptr(proba), _ptr(pdf), _ptr(gt), B, N, D, float(m), float(e), _ptr(keys),
          _ptr(arg), _ptr(nocc), _ptr(pred), _ptr(partials), _ptr(out), _stream())
    return out, pred, arg, nocc


def projected_loss_backward(pred, gt, proba, pdf, B: int, N: int, diam_pix: int, m: float, e: float, grad_total, arg, nocc, pix):
    """include/strata_hip.h: sn2_projected_loss_backward -> (d loss / d coverages, d loss / d proba), (B*N,4) each."""
    R = B * N
    _chk(grad_total, F64, None, "grad_total")
    dcov = torch.empty(R, 4, dtype=F32, device=pred.device)
    dproba = torch.empty(R, 4, dtype=F32, device=pred.device)
    _call("sn2_projected_loss_backward", _ptr(pred), _ptr(gt), B, _ptr(proba), _ptr(pdf), N, int(diam_pix), float(m), float(e),
          _ptr(grad_total), _ptr(arg), _ptr(nocc), _ptr(pix), _ptr(dcov), _ptr(dproba), _stream())
    return dcov, dproba


def adam_step_images(param, arena, replicas, stride, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, step_dev, grad_scale=1.0):
    """include/strata_hip.h: sn2_adam_step_images -- fold the gradient's images and take the Adam step in one launch."""
    n = param.numel()
    for t, nme in ((param, "param"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")):
        _chk(t, F32, (n,), nme)
    _chk(arena, F32, None, "arena")
    if arena.numel() < replicas * stride or stride < n:
        raise ValueError("adam_step_images: arena smaller than its images")
    _chk(step_dev, I32, (2,), "step_dev")
    _call("sn2_adam_step_images", _ptr(param), _ptr(arena), int(replicas), int(stride), _ptr(exp_avg), _ptr(exp_avg_sq), n, lr, beta1,
          beta2, eps, weight_decay, _ptr(step_dev), grad_scale, _stream())


def loss_term_forward(kind: int, x: torch.Tensor, y: Optional[torch.Tensor]):
    """ONE term of the loss (the reference's loop calls them one by one, learning/train.py:58-60) -> out (4,) fp64 with
    out[kind] = the term: kind 1 = absolute (x = pred (B,4), y = gt (B,4) fp64), 2 = NLL (x = proba (R,4), y = pdf (R,3) fp64),
    3 = entropy (x = proba (R,4)).  The other terms are skipped inside sn2_loss_forward (their inputs are not passed)."""
    out = torch.empty(4, dtype=F64, device=x.device)
    if kind == 1:
        B = x.shape[0]
        _chk(x, F32, (B, 4), "pred")
        _chk(y, F64, (B, 4), "gt")
        _call("sn2_loss_forward", _ptr(x), _ptr(y), B, None, None, 0, 0.0, 0.0, None, _ptr(out), _stream())
        return out
    R = x.shape[0]
    _chk(x, F32, (R, 4), "proba")
    partials = torch.empty(2 * LOSS_BLOCKS, dtype=F64, device=x.device)
    if kind == 2:
        _chk(y, F64, (R, 3), "pdf")
        _call("sn2_loss_forward", None, None, 0, _ptr(x), _ptr(y), R, 1.0, 0.0, _ptr(partials), _ptr(out), _stream())
    elif kind == 3:
        _call("sn2_loss_forward", None, None, 0, _ptr(x), None, R, 0.0, 1.0, _ptr(partials), _ptr(out), _stream())
    else:
        raise ValueError("loss term kind must be 1, 2 or 3")
    return out


def loss_term_backward(kind: int, x: torch.Tensor, y: Optional[torch.Tensor], grad: torch.Tensor):
    """d term / d x for the upstream gradient `grad` (fp64 device scalar) of that one term."""
    _chk(grad, F64, None, "grad")
    dx = torch.empty_like(x)
    if kind == 1:
        _call("sn2_loss_backward", _ptr(x), _ptr(y), x.shape[0], None, None, 0, 0.0, 0.0, _ptr(grad), _ptr(dx), None, _stream())
    elif kind == 2:
        _call("sn2_loss_backward", None, None, 0, _ptr(x), _ptr(y), x.shape[0], 1.0, 0.0, _ptr(grad), None, _ptr(dx), _stream())
    else:
        _call("sn2_loss_backward", None, None, 0, _ptr(x), None, x.shape[0], 0.0, 1.0, _ptr(grad), None, _ptr(dx), _stream())
    return dx


def adam_step(param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, step_dev, grad_scale=1.0):
    """step_dev: int32 device tensor (2,) = {number of steps taken so far (incremented by the call), 0}."""
    n = param.numel()
    for t, nme in ((param, "param"), (grad, "grad"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")):
        _chk(t, F32, (n,), nme)
    _chk(step_dev, I32, (2,), "step_dev")
    _call("sn2_adam_step", _ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), n, lr, beta1, beta2, eps,
          weight_decay, _ptr(step_dev), grad_scale, _stream())
