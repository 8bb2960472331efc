"""Multi-GPU frame pipeline: screen-tile shard + gather of the RGBA8 tiles to rank 0.

One process per GPU.  The tree is replicated; every launch renders this rank's interleaved tiles
of ``n`` frames into a COMPACT buffer (``VrFrame.layout``), ONE gather per launch moves the
buffers to rank 0 -- each peer sends its ~1/world share straight to the root over its direct
xGMI link -- and rank 0 de-interleaves them into frames.  The gather of launch j overlaps the
rendering of launch j+1 (two buffer sets); with a ``stream_ctx`` the launches additionally
alternate between streams, so that the ramp-up / tail of one launch overlaps its neighbour.

This module is the PIPELINE (buffer sets, order of render / gather / assembly), not the
collective.  The transport is whatever object is handed in as ``dist`` -- anything with
``gather(tensor, gather_list, dst, async_op) -> work`` and ``work.wait()``:
  * on GPUs: ``volrend_amd.gather.TileGather`` -- libvolrend_gather.so, the grouped
    ncclSend / ncclRecv that ``volrend_headless --gpus N`` ships (bench.py --gpus N);
  * in the CPU tests (tests/test_dist_gloo.py) and the shared-GPU rehearsal:
    ``torch.distributed`` over gloo.
The rendering and the assembly are callbacks, so the same pipeline drives the HIP library on
GPUs and the CPU oracle in the gloo tests.
"""
from __future__ import annotations

import contextlib


class GatherPipeline:
    def __init__(self, dist, rank: int, world: int, make_buffer, make_gather_list,
                 force_collective: bool = False, stream_ctx=None):
        """``make_buffer()`` -> this rank's compact buffer (tensor) for one launch;
        ``make_gather_list()`` -> list of ``world`` tensors like it (rank 0 only);
        ``stream_ctx(j)`` -> context manager selecting the stream of launch j (optional;
        launches j and j+2 share buffers, so use an even number of alternating streams)."""
        self.dist, self.rank, self.world = dist, rank, world
        # world == 1 normally skips the collective; force_collective keeps it (tests the
        # RCCL path on a single-GPU box)
        self.collective = world > 1 or force_collective
        self.bufs = [make_buffer() for _ in range(2)]
        self.gathered = [make_gather_list() if rank == 0 else None for _ in range(2)]
        self.stream_ctx = stream_ctx or (lambda j: contextlib.nullcontext())
        self.works = {}
        self.sizes = {}

    def buffer(self, j: int):
        return self.bufs[j % 2]

    def submit(self, j: int, n: int) -> None:
        """Launch j has been enqueued into buffer(j): start its gather."""
        self.sizes[j] = n
        if not self.collective:
            return
        # only the n frames of this launch travel (row-sliced views of the [batch, bytes] buffers
        # are contiguous); a short last launch does not pay for the whole batch
        send = self.bufs[j % 2][:n] if getattr(self.bufs[j % 2], "ndim", 1) > 1 else self.bufs[j % 2]
        recv = None
        if self.rank == 0:
            recv = [g[:n] if getattr(g, "ndim", 1) > 1 else g for g in self.gathered[j % 2]]
        self.works[j] = self.dist.gather(send, recv, dst=0, async_op=True)

    def retire(self, j: int, assemble) -> None:
        """Wait for launch j's gather; rank 0 calls ``assemble(j, gathered_list, n)``.
        Must be called before buffer(j + 2) is written."""
        if j not in self.sizes:
            return
        with self.stream_ctx(j):
            n = self.sizes.pop(j)
            if self.collective:
                self.works.pop(j).wait()
            if self.rank == 0:
                assemble(j, self.gathered[j % 2] if self.collective else [self.bufs[j % 2]], n)

    def run(self, n_steps: int, batch: int, render, assemble, first_step: int = 0) -> int:
        """render(j, first_step, n, buffer) enqueues one launch; returns launches done."""
        j, done = 0, 0
        while done < n_steps:
            n = min(batch, n_steps - done)
            self.retire(j - 2, assemble)  # same stream as launch j (j - 2 = j mod 2)
            with self.stream_ctx(j):
                render(j, first_step + done, n, self.buffer(j))
                self.submit(j, n)
            done += n
            j += 1
        self.retire(j - 2, assemble)
        self.retire(j - 1, assemble)
        return j
