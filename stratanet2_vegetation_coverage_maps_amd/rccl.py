"""Direct ctypes binding of librccl.so for the ONE exchange step of the data-parallel path: a SUM all-reduce
of the flat 14 997-float gradient, enqueued with `ncclAllReduce` ON THE STEP'S OWN STREAM between the backward kernels and the
Adam kernel -- so it can be captured into the slot's hipGraph (one graph per step at any world size, no second stream, no
cross-stream event).  The reference has no counterpart (single GPU: `learning/train.py:46-66`); torch's process group stays
available as the fallback (`optim.allreduce_flat_grad`), and is still what carries rendezvous, barriers and the timing
reduction of bench.py.

RCCL is ROCm's NCCL: same API (`/opt/rocm/include/rccl/rccl.h`).  The library instance bound here is the one torch already
loaded (torch/lib/librccl.so) when there is one, so a process never holds two copies.
"""
import ctypes
import os
from ctypes import POINTER, Structure, byref, c_char, c_char_p, c_int, c_size_t, c_void_p

import torch

NCCL_UNIQUE_ID_BYTES = 128          # rccl.h:40
NCCL_SUM = 0                        # rccl.h: ncclRedOp_t
NCCL_FLOAT32 = 7                    # rccl.h: ncclDataType_t
NCCL_INT32 = 2


class RcclError(RuntimeError):
    pass


class _UniqueId(Structure):
    _fields_ = [("internal", c_char * NCCL_UNIQUE_ID_BYTES)]


_lib = None


def load():
    """librccl.so through ctypes; raises RcclError when no copy can be loaded."""
    global _lib
    if _lib is not None:
        return _lib
    cands = [os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), "/opt/rocm/lib/librccl.so", "librccl.so",
             "librccl.so.1"]
    err = None
    for path in cands:
        try:
            lib = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
        except OSError as exc:
            err = exc
            continue
        lib.ncclGetErrorString.restype = c_char_p
        lib.ncclGetErrorString.argtypes = [c_int]
        lib.ncclGetVersion.argtypes = [POINTER(c_int)]
        lib.ncclGetUniqueId.argtypes = [POINTER(_UniqueId)]
        lib.ncclCommInitRank.argtypes = [POINTER(c_void_p), c_int, _UniqueId, c_int]
        lib.ncclCommDestroy.argtypes = [c_void_p]
        lib.ncclCommCount.argtypes = [c_void_p, POINTER(c_int)]
        lib.ncclAllReduce.argtypes = [c_void_p, c_void_p, c_size_t, c_int, c_int, c_void_p, c_void_p]
        lib.ncclBroadcast.argtypes = [c_void_p, c_void_p, c_size_t, c_int, c_int, c_void_p, c_void_p]
        for fn in (lib.ncclGetVersion, lib.ncclGetUniqueId, lib.ncclCommInitRank, lib.ncclCommDestroy, lib.ncclCommCount,
                   lib.ncclAllReduce, lib.ncclBroadcast):
            fn.restype = c_int
        lib._sn2_path = path
        _lib = lib
        return lib
    raise RcclError(f"librccl.so not found ({err})")


def _check(rc, what):
    if rc != 0:
        raise RcclError(f"{what}: {load().ncclGetErrorString(rc).decode()} ({rc})")


def version() -> int:
    v = c_int(0)
    _check(load().ncclGetVersion(byref(v)), "ncclGetVersion")
    return v.value


def unique_id() -> bytes:
    """`ncclGetUniqueId`: 128 opaque bytes that rank 0 creates and every rank passes to `RcclComm`."""
    uid = _UniqueId()
    _check(load().ncclGetUniqueId(byref(uid)), "ncclGetUniqueId")
    return ctypes.string_at(byref(uid), NCCL_UNIQUE_ID_BYTES)       # (`.internal` as a c_char array would stop at the first NUL)


class RcclComm:
    """One RCCL communicator of `world` ranks (one process per GPU).  Every rank constructs it with the same `uid`
    (blocking: `ncclCommInitRank` is a rendezvous)."""

    def __init__(self, rank: int, world: int, uid: bytes, device):
        if len(uid) != NCCL_UNIQUE_ID_BYTES:
            raise RcclError("unique id must be 128 bytes")
        self.rank, self.world = int(rank), int(world)
        self.device = torch.device(device)
        u = _UniqueId()
        ctypes.memmove(byref(u), uid, NCCL_UNIQUE_ID_BYTES)
        self._comm = c_void_p()
        with torch.cuda.device(self.device):
            _check(load().ncclCommInitRank(byref(self._comm), self.world, u, self.rank), "ncclCommInitRank")
        n = c_int(0)
        _check(load().ncclCommCount(self._comm, byref(n)), "ncclCommCount")
        if n.value != self.world:
            raise RcclError(f"communicator reports {n.value} ranks, expected {self.world}")

    def all_reduce_sum_(self, t: torch.Tensor, stream=None):
        """In-place SUM all-reduce of a contiguous fp32 (or int32) device tensor on `stream` (default: torch's current
        stream -- inside `torch.cuda.graph(...)` that is the capturing stream, and the collective becomes a node of the
        graph).  Asynchronous."""
        if self._comm is None:
            raise RcclError("communicator destroyed")
        if not (t.is_cuda and t.is_contiguous() and t.device == self.device):
            raise RcclError("all_reduce_sum_: expected a contiguous tensor on the communicator's device")
        dt = {torch.float32: NCCL_FLOAT32, torch.int32: NCCL_INT32}.get(t.dtype)
        if dt is None:
            raise RcclError(f"all_reduce_sum_: unsupported dtype {t.dtype}")
        st = torch.cuda.current_stream(self.device) if stream is None else stream
        _check(load().ncclAllReduce(c_void_p(t.data_ptr()), c_void_p(t.data_ptr()), t.numel(), dt, NCCL_SUM, self._comm,
                                    c_void_p(st.cuda_stream)), "ncclAllReduce")

    def destroy(self):
        if self._comm is not None:
            comm, self._comm = self._comm, None
            _check(load().ncclCommDestroy(comm), "ncclCommDestroy")

    def __del__(self):
        try:
            self.destroy()
        except Exception:       # noqa: BLE001  (interpreter shutdown: the library may be gone)
            pass


def comm_from_torch_group(device, group=None) -> RcclComm:
    """A communicator over the ranks of torch's (default) process group: rank 0 draws the unique id and torch's group -- any
    backend -- carries its 128 bytes to the others.  Without an initialised process group: a one-rank communicator."""
    dist = torch.distributed
    if dist.is_available() and dist.is_initialized():
        uid, rank, world = exchange_unique_id(group)
        if uid is None:
            raise RcclError("rank 0 could not draw a unique id (ncclGetUniqueId)")
        return RcclComm(rank, world, uid, device)
    return RcclComm(0, 1, unique_id(), device)


def exchange_unique_id(group=None, make_uid=unique_id):
    """Rank 0 draws the communicator's unique id, torch's process group (any backend: it only moves 128 bytes through the
    store / a broadcast) hands it to every rank -> (uid or None, rank, world).  uid is None ON EVERY RANK when rank 0 could not
    draw one: the ranks fail together, nobody walks into `ncclCommInitRank` alone."""
    dist = torch.distributed
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    box = [None]
    if rank == 0:
        try:
            box[0] = make_uid()
        except Exception:                      # noqa: BLE001
            box[0] = None
    if world > 1:
        dist.broadcast_object_list(box, src=0, group=group)
    uid = box[0]
    if uid is not None and len(uid) != NCCL_UNIQUE_ID_BYTES:
        uid = None
    return uid, rank, world


def _all_agree(ok: bool, group, device) -> bool:
    """MIN over the ranks of a local success flag, through torch's group."""
    dist = torch.distributed
    on_dev = dist.get_backend(group) == "nccl"
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device if on_dev else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return int(flag.item()) == 1


def negotiate_comm(device, group=None, graph: bool = True, make_uid=unique_id, make_comm=None, test=None, log=None):
    """The direct-RCCL exchange for torch's (default) process group, or None when ANY rank cannot have it -- decided stage by
    stage, with a MIN all-reduce through torch's group between the stages, so that every RCCL collective (the rendezvous of
    `ncclCommInitRank`, the eager all-reduce of the self-test, the captured one) is entered by ALL ranks or by NONE: a rank
    whose librccl does not load, or whose eager result is wrong, can no longer leave its peers blocked inside the next
    collective (the agreement used to come after the whole sequence).
        stage 1  local: load the library                                -> agree
        stage 2  torch collective: rank 0's unique id to everyone       (None everywhere when rank 0 failed)
        stage 3  RCCL collective: ncclCommInitRank                      -> agree
        stage 4  RCCL collective: eager all-reduce with a known sum     -> agree on the RESULT
        stage 5  RCCL collective inside a captured graph (graph=True)   -> agree on the result
    make_uid / make_comm(rank, world, uid, device) / test(comm, graph) replace the library calls (tests: the staged agreement runs
    on CPU over gloo).  -> (communicator or None, reason the direct exchange is not used or "")."""
    make_comm = make_comm or RcclComm
    test = test or self_test
    say = log or (lambda m: None)
    ok, why = True, ""
    try:
        if make_uid is unique_id:
            load()
    except Exception as exc:                   # noqa: BLE001
        ok, why = False, f"stage 1 (load): {type(exc).__name__}: {exc}"
    if not _all_agree(ok, group, device):
        return None, why or "stage 1 (load): another rank could not load librccl"
    uid, rank, world = exchange_unique_id(group, make_uid)
    if uid is None:                            # the same on every rank: no agreement round needed
        return None, "stage 2 (unique id): rank 0 could not draw one"
    comm = None
    try:
        comm = make_comm(rank, world, uid, device)
    except Exception as exc:                   # noqa: BLE001
        ok, why = False, f"stage 3 (ncclCommInitRank): {type(exc).__name__}: {exc}"
    if not _all_agree(ok, group, device):
        _drop(comm)
        return None, why or "stage 3 (ncclCommInitRank): another rank failed"
    for stage, with_graph in ((4, False), (5, True)):
        if with_graph and not graph:
            break
        try:
            test(comm, graph=with_graph, eager=not with_graph)
        except Exception as exc:               # noqa: BLE001
            ok, why = False, f"stage {stage} ({'captured' if with_graph else 'eager'} all-reduce): {type(exc).__name__}: {exc}"
        if not _all_agree(ok, group, device):
            _drop(comm)
            return None, why or f"stage {stage}: another rank's self-test failed"
    say(f"direct RCCL exchange agreed on by {world} rank(s)")
    return comm, ""


def _drop(comm):
    try:
        if comm is not None:
            comm.destroy()
    except Exception:                          # noqa: BLE001
        pass


def self_test(comm: RcclComm, graph: bool = True, eager: bool = True):
    """Eager (eager=True) and hipGraph-captured (graph=True) all-reduce of a small buffer with a known answer; raises RcclError on
    a wrong result.  Run by every rank at start-up before the exchange is trusted with gradients (bench.py; `negotiate_comm`
    runs the two parts as separate stages)."""
    dev, w, r = comm.device, comm.world, comm.rank
    want = float(w * (w + 1) // 2)
    with torch.cuda.device(dev):
        if eager:
            x = torch.full((4096,), float(r + 1), device=dev)
            comm.all_reduce_sum_(x)
            torch.cuda.synchronize(dev)
            if not bool((x == want).all()):
                raise RcclError(f"eager all-reduce: got {float(x[0])}, expected {want}")
        if graph:
            buf = torch.empty(4096, device=dev)
            src = torch.full((4096,), float(r + 1), device=dev)
            from .hip_ops import shared_stream           # (no stream of its own: streams are a per-process resource, hip_ops.shared_stream)
            side = shared_stream(dev, "capture")
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                torch.add(src, 0.0, out=buf)                 # (a kernel, not a copy: captured memcpy / memset nodes are avoided everywhere)
                comm.all_reduce_sum_(buf)                    # warm-up on the capture stream (connections, lazy allocations)
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                torch.add(src, 0.0, out=buf)
                comm.all_reduce_sum_(buf)
                buf.mul_(2.0)
            for _ in range(3):
                g.replay()
            torch.cuda.synchronize(dev)
            if not bool((buf == 2.0 * want).all()):
                raise RcclError(f"captured all-reduce: got {float(buf[0])}, expected {2.0 * want}")
    return True
