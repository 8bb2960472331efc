"""ctypes mirror of include/volrend_gather.h (libvolrend_gather.so): the RGBA8 tile gather of the
screen-tile shard -- one grouped ncclSend / ncclRecv to the root per launch, the collective that
``volrend_headless --gpus N`` ships (volrend::internal::TileShardRenderer) and that
``bench.py --gpus N`` drives with one process per GPU.

There is no fallback inside this module: a missing library or a failing RCCL call raises.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VOLREND_GATHER_LIB") or os.path.join(HERE, "libvolrend_gather.so")
ID_BYTES = 128

PROTOTYPES = {
    "vr_gather_last_error": (C.c_char_p, []),
    "vr_gather_version": (C.c_int, []),
    "vr_gather_unique_id": (C.c_int, [C.c_void_p]),
    "vr_gather_init_rank": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "vr_gather_init_all": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_void_p)]),
    "vr_gather_free": (C.c_int, [C.c_void_p]),
    "vr_gather_rank": (C.c_int, [C.c_void_p]),
    "vr_gather_world": (C.c_int, [C.c_void_p]),
    "vr_gather_tiles": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int,
                                  C.c_void_p]),
    "vr_gather_group_begin": (C.c_int, []),
    "vr_gather_group_end": (C.c_int, []),
}

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it (`make gather`); the tile shard has no "
                               "other collective")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"{what}: {lib().vr_gather_last_error().decode(errors='replace')}")


def unique_id() -> bytes:
    buf = C.create_string_buffer(ID_BYTES)
    check(lib().vr_gather_unique_id(buf), "vr_gather_unique_id")
    return buf.raw


def version() -> str:
    v = lib().vr_gather_version()
    return f"{v // 10000}.{v // 100 % 100}.{v % 100}" if v >= 10000 else str(v)


class _Enqueued:
    """What TileGather.gather returns: the transfer is ordered on the stream it was enqueued on, so
    there is nothing to wait for on the host (torch.distributed's Work.wait() signature)."""

    def wait(self):
        return True


class TileGather:
    """One rank's handle on the shard's collective (one process per rank)."""

    def __init__(self, id_bytes: bytes, rank: int, world: int, device: int):
        if len(id_bytes) != ID_BYTES:
            raise ValueError("the unique id has 128 bytes")
        h = C.c_void_p()
        check(lib().vr_gather_init_rank(id_bytes, rank, world, device, C.byref(h)), "vr_gather_init_rank")
        self._h, self.rank, self.world = h, rank, world

    def tiles(self, send_ptr: int, recv_base_ptr: int, rank_stride: int, nbytes: int, stream_ptr,
              self_transfer: bool = False) -> None:
        check(lib().vr_gather_tiles(self._h, send_ptr, recv_base_ptr, rank_stride, nbytes,
                                    1 if self_transfer else 0, stream_ptr), "vr_gather_tiles")

    def gather(self, tensor, gather_list, dst=0, async_op=True):
        """torch.distributed.gather's call shape over vr_gather_tiles, for volrend_amd.dist.GatherPipeline:
        `tensor` = this rank's compact buffer of the launch; `gather_list` (rank 0) = the rows of ONE
        [world, ...] tensor, row r receiving rank r's buffer.  Enqueued on torch's current stream."""
        import torch
        assert dst == 0
        stream = int(torch.cuda.current_stream().cuda_stream)
        nbytes = tensor.numel() * tensor.element_size()
        base = stride = 0
        if self.rank == 0:
            base = gather_list[0].data_ptr()
            stride = (gather_list[1].data_ptr() - base) if len(gather_list) > 1 else nbytes
            for r, g in enumerate(gather_list):  # the rows must be one strided allocation
                assert g.data_ptr() == base + r * stride and g.numel() * g.element_size() >= nbytes
        self.tiles(tensor.data_ptr(), base, stride, nbytes, stream, self_transfer=(self.world == 1))
        return _Enqueued()

    def free(self):
        if self._h:
            lib().vr_gather_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
