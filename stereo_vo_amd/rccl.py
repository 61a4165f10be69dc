"""ctypes mirror of include/svo_rccl.h (libsvo_rccl.so): the pose-record all-gather and the neighbour hand-over of the
multi-GPU path over RCCL, for hosts that hold `hip.Context`s.  Loaded on demand: a single-GPU host never needs it.
There is no fallback: a missing library or a failing RCCL call raises."""
import ctypes as C
import os

import numpy as np

from . import hip
from .abi import Result

EXPORTS = ["svo_group_create_local", "svo_group_unique_id", "svo_group_create_rank", "svo_group_destroy", "svo_group_size",
           "svo_group_last_error", "svo_group_allgather_results", "svo_group_send_frame", "svo_group_recv_frame",
           "svo_group_allgather_inplace", "svo_group_comm_count", "svo_group_comm_device"]
ID_BYTES = 128
_lib = None


def lib():
    global _lib
    if _lib is None:
        hip.lib()                                     # libsvo_rccl.so resolves its svo_* symbols from libsvo_hip.so
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libsvo_rccl.so")
        if not os.path.exists(path):
            raise hip.SvoError("libsvo_rccl.so is not built (make -C stereo_vo_amd/csrc)")
        L = C.CDLL(path)
        L.svo_group_last_error.restype = C.c_char_p
        L.svo_group_last_error.argtypes = [C.c_void_p]
        L.svo_group_destroy.restype = None
        L.svo_group_destroy.argtypes = [C.c_void_p]
        L.svo_group_size.argtypes = [C.c_void_p]
        L.svo_group_create_local.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]
        L.svo_group_unique_id.argtypes = [C.c_char_p]
        L.svo_group_create_rank.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.svo_group_allgather_results.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        L.svo_group_send_frame.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
        L.svo_group_recv_frame.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
        _lib = L
    return _lib


def unique_id():
    """128 bytes rank 0 ships to the other ranks (torch.distributed's store, a file, a socket)"""
    buf = C.create_string_buffer(ID_BYTES)
    rc = lib().svo_group_unique_id(buf)
    if rc != 0:
        raise hip.SvoError("svo_group_unique_id: %d" % rc)
    return buf.raw


class Group:
    """Ranks are GPUs: Group.local([0, 1, ...]) inside one process, Group.rank(id, n, rank, device) one per process."""

    def __init__(self, handle, rank=None):
        self.h, self.my_rank = handle, rank

    @classmethod
    def local(cls, devices):
        h = C.c_void_p()
        arr = (C.c_int * len(devices))(*devices)
        rc = lib().svo_group_create_local(arr, len(devices), C.byref(h))
        cls._ck(rc, h, "svo_group_create_local")
        return cls(h)

    @classmethod
    def rank(cls, id_bytes, n_ranks, rank, device):
        h = C.c_void_p()
        rc = lib().svo_group_create_rank(id_bytes, n_ranks, rank, device, C.byref(h))
        cls._ck(rc, h, "svo_group_create_rank")
        return cls(h, rank)

    @staticmethod
    def _ck(rc, h, what):
        if rc != 0:
            msg = lib().svo_group_last_error(h).decode() if h else ""
            if h:
                lib().svo_group_destroy(h)
            raise hip.SvoError("%s: %d %s" % (what, rc, msg))

    def _check(self, rc, what):
        if rc != 0:
            raise hip.SvoError("%s: %d %s" % (what, rc, lib().svo_group_last_error(self.h).decode()))

    @property
    def size(self):
        return lib().svo_group_size(self.h)

    def allgather_results(self, ctx, dev_ptr, nbytes, rank=None, stream=None):
        """all ranks' result records of the frame enqueued on `ctx` into device memory at dev_ptr (size x n_lanes records)"""
        r = self.my_rank if rank is None else rank
        self._check(lib().svo_group_allgather_results(self.h, r or 0, ctx.h, C.c_void_p(stream) if stream else None, C.c_void_p(dev_ptr), nbytes),
                    "svo_group_allgather_results")

    def send_frame(self, to_rank, dev_ptr, nbytes, stream, rank=None):
        r = self.my_rank if rank is None else rank
        self._check(lib().svo_group_send_frame(self.h, r or 0, to_rank, C.c_void_p(dev_ptr), nbytes, C.c_void_p(stream) if stream else None), "svo_group_send_frame")

    def recv_frame(self, from_rank, dev_ptr, nbytes, stream, rank=None):
        r = self.my_rank if rank is None else rank
        self._check(lib().svo_group_recv_frame(self.h, r or 0, from_rank, C.c_void_p(dev_ptr), nbytes, C.c_void_p(stream) if stream else None), "svo_group_recv_frame")

    def close(self):
        if self.h:
            lib().svo_group_destroy(self.h)
            self.h = None


def records_from_bytes(raw):
    """host bytes of a gathered table -> list of abi.Result"""
    n = len(raw) // C.sizeof(Result)
    return [Result.from_buffer_copy(raw, i * C.sizeof(Result)) for i in range(n)]
