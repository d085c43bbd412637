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
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(box, src=0, group=group)
        return RcclComm(rank, world, box[0], device)
    return RcclComm(0, 1, unique_id(), device)


def self_test(comm: RcclComm, graph: bool = True):
    """Eager and (graph=True) hipGraph-captured all-reduce of a small buffer with a known answer; raises RcclError on a wrong
    result.  Run by every rank at start-up before the exchange is trusted with gradients (bench.py)."""
    dev, w, r = comm.device, comm.world, comm.rank
    want = float(w * (w + 1) // 2)
    with torch.cuda.device(dev):
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
