"""RCCL through its C API (ctypes on the librccl.so PyTorch-ROCm has loaded): communicators of our own.

Why not ``torch.distributed.all_reduce`` for the data-parallel step (reference train.py:70,100-104)?  ProcessGroupNCCL
launches every collective on a stream of ITS OWN and hands the tensor over with events -- two cross-stream hand-offs
per call, ~20 us of idle GPU each, plus ~40 us of host time -- and a train step of this engine issues one latency-bound
SyncBN statistic exchange per BatchNorm layer and direction (44 / 76 / 108 per ResNet-18 / -34 / -50 step, SURVEY.md
§2b C2 / C3): +10 / +15 / +38 % on the step at world size 1 with the collectives forced (round 4).  Here
``ncclAllReduce(..., stream)`` is enqueued on the stream the producing kernel was launched on -- the exchange is one
more kernel in the data path's own queue, no event, no second stream -- and the gradient buckets go to a communication
stream this module forks and joins with plain events.  Both forms are ordinary stream work, so the whole
data-parallel step can be captured into a hipGraph (graph.GraphedTrainStep).

``torch.distributed`` stays the control plane: it carries the 128-byte ncclUniqueId from rank 0 to the others (any
backend), and gloo process groups (CPU tests) keep using its collectives.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import torch
import torch.distributed as dist

NCCL_UNIQUE_ID_BYTES = 128
SUM, AVG = 0, 4                                     # ncclRedOp_t (rccl.h:448-452)
_DTYPES = {torch.int8: 0, torch.uint8: 1, torch.int32: 2, torch.int64: 4, torch.float16: 6, torch.float32: 7,
           torch.float64: 8, torch.bfloat16: 9}     # ncclDataType_t (rccl.h:459-468)


class _UniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_char * NCCL_UNIQUE_ID_BYTES)]


_LIB = None


def lib():
    """librccl.so as loaded by torch (same soname => same handle: one RCCL per process)."""
    global _LIB
    if _LIB is None:
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        if not os.path.exists(path):
            path = "librccl.so"
        L = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
        L.ncclGetErrorString.restype = ctypes.c_char_p
        L.ncclGetErrorString.argtypes = [ctypes.c_int]
        L.ncclGetUniqueId.argtypes = [ctypes.POINTER(_UniqueId)]
        L.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, _UniqueId, ctypes.c_int]
        L.ncclCommDestroy.argtypes = [ctypes.c_void_p]
        L.ncclCommAbort.argtypes = [ctypes.c_void_p]
        L.ncclAllReduce.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
                                    ctypes.c_void_p, ctypes.c_void_p]
        L.ncclBroadcast.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
                                    ctypes.c_void_p, ctypes.c_void_p]
        for f in (L.ncclGetUniqueId, L.ncclCommInitRank, L.ncclCommDestroy, L.ncclCommAbort, L.ncclAllReduce,
                  L.ncclBroadcast):
            f.restype = ctypes.c_int
        _LIB = L
    return _LIB


def _chk(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f"{what}: {lib().ncclGetErrorString(rc).decode()} ({rc})")


def available() -> bool:
    try:
        lib()
        return torch.cuda.is_available()
    except OSError:
        return False


def _uid_bytes(uid: "_UniqueId") -> bytes:
    """All 128 bytes of an ncclUniqueId (a c_char ARRAY FIELD reads back as a NUL-terminated string: `bytes(uid.internal)`
    stops at the first zero byte, and the ids contain zeros)."""
    return ctypes.string_at(ctypes.addressof(uid), NCCL_UNIQUE_ID_BYTES)


def broadcast_id(raw: Optional[bytes], process_group=None) -> bytes:
    """The 128-byte id from the group's rank 0 (which passes `raw`) to every rank, as a CPU object over whatever backend
    the group has; returns it on every rank.  Collective."""
    world = dist.get_world_size(process_group)
    rank = dist.get_rank(process_group)
    if rank == 0 and (raw is None or len(raw) != NCCL_UNIQUE_ID_BYTES):
        raise ValueError("broadcast_id: rank 0 must pass the 128-byte id")
    if world == 1:
        return raw
    box = [raw if rank == 0 else None]
    src = dist.get_global_rank(process_group, 0) if process_group is not None else 0
    dist.broadcast_object_list(box, src=src, group=process_group)
    if not isinstance(box[0], (bytes, bytearray)) or len(box[0]) != NCCL_UNIQUE_ID_BYTES:
        raise RuntimeError("broadcast_id: did not receive a 128-byte id")
    return bytes(box[0])


class Communicator:
    """One ncclComm_t over the ranks of a torch.distributed process group, bound to this process's current device.
    Construction is collective (every rank of the group must call it, in the same order)."""

    def __init__(self, process_group=None):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised (it carries the ncclUniqueId)")
        L = lib()
        self.world = dist.get_world_size(process_group)
        self.rank = dist.get_rank(process_group)
        self.device = torch.cuda.current_device()
        uid = _UniqueId()
        if self.rank == 0:
            _chk(L.ncclGetUniqueId(ctypes.byref(uid)), "ncclGetUniqueId")
        raw = broadcast_id(_uid_bytes(uid) if self.rank == 0 else None, process_group)
        ctypes.memmove(ctypes.byref(uid), raw, NCCL_UNIQUE_ID_BYTES)
        self._comm = ctypes.c_void_p()
        _chk(L.ncclCommInitRank(ctypes.byref(self._comm), self.world, uid, self.rank), "ncclCommInitRank")

    def _check(self, t: torch.Tensor):
        if not t.is_cuda or t.device.index != self.device or not t.is_contiguous():
            raise RuntimeError("rccl: operands must be contiguous tensors on the communicator's device")
        if t.dtype not in _DTYPES:
            raise TypeError(f"rccl: unsupported dtype {t.dtype}")

    def all_reduce(self, t: torch.Tensor, op: int = SUM, stream: Optional[int] = None):
        """In-place all-reduce of `t`, enqueued on `stream` (a hipStream_t as an integer; default: torch's current
        stream).  Returns at once; ordinary stream order is the only synchronisation."""
        self._check(t)
        if stream is None:
            stream = torch.cuda.current_stream().cuda_stream
        p = ctypes.c_void_p(t.data_ptr())
        _chk(lib().ncclAllReduce(p, p, t.numel(), _DTYPES[t.dtype], op, self._comm, ctypes.c_void_p(stream)),
             "ncclAllReduce")

    def broadcast(self, t: torch.Tensor, root: int = 0, stream: Optional[int] = None):
        self._check(t)
        if stream is None:
            stream = torch.cuda.current_stream().cuda_stream
        p = ctypes.c_void_p(t.data_ptr())
        _chk(lib().ncclBroadcast(p, p, t.numel(), _DTYPES[t.dtype], root, self._comm, ctypes.c_void_p(stream)),
             "ncclBroadcast")

    def destroy(self):
        if self._comm:
            comm, self._comm = self._comm, ctypes.c_void_p()
            try:
                torch.cuda.synchronize(self.device)
                lib().ncclCommDestroy(comm)
            except Exception:  # noqa: BLE001  (interpreter shutdown: the runtime may already be gone)
                pass
