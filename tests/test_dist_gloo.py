"""N > 1 path on CPU: two processes, gloo backend, the same GatherPipeline bench.py uses
on RCCL.  Each rank renders its interleaved screen tiles (CPU oracle standing in for the
kernel), the compact RGBA8 buffers are gathered to rank 0 and de-interleaved; the result
must equal the single-process frame byte for byte."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tile_w, tile_h, n_frames, batch, out_path):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from oracle import binding as ob
    from tests import common
    from volrend_amd import tiles
    from volrend_amd.dist import GatherPipeline

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tree = common.small_scene(depth=4, basis_dim=4, seed=77)
    th = ob.TreeHandle(tree)
    w, h, focal = 72, 40, 100.0
    from volrend_amd import synth
    poses = synth.make_poses(8)
    npix = tiles.compact_pixels(w, h, tile_w, tile_h, world)
    frames = {}

    def render(j, first, n, buf):
        for i in range(n):
            cam = ob.make_camera(synth.c2w_to_transform(poses[(first + i) % 8]), w, h, focal)
            # a rank only traces the pixels it owns
            own = tiles.owner_map(w, h, tile_w, tile_h, world) == rank
            full, _, _ = ob.render(th, cam, ob.default_options(), nthreads=1, want_accum=False)
            full[~own] = 0
            buf[i] = torch.from_numpy(tiles.frame_to_compact(full, tile_w, tile_h, rank, world))

    def assemble(j, glist, n):
        g = np.stack([t.numpy() for t in glist])  # [world, batch, npix, 4]
        for i in range(n):
            frames[(j, i)] = tiles.assemble_tiles(g[:, i], w, h, tile_w, tile_h, world)

    pipe = GatherPipeline(dist, rank, world,
                          lambda: torch.zeros((batch, npix, 4), dtype=torch.uint8),
                          lambda: [torch.zeros((batch, npix, 4), dtype=torch.uint8)
                                   for _ in range(world)])
    pipe.run(n_frames, batch, render, assemble)
    if rank == 0:
        order = sorted(frames)
        np.save(out_path, np.stack([frames[k] for k in order]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("tile_w,tile_h,batch", [(72, 8, 2), (24, 16, 3)])
def test_two_rank_tile_shard_gather(tmp_path, tile_w, tile_h, batch):
    import torch.multiprocessing as mp
    from oracle import binding as ob
    from tests import common
    from volrend_amd import synth

    world, n_frames = 2, 5
    out = str(tmp_path / "frames.npy")
    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, world, port, tile_w, tile_h, n_frames, batch, out))
             for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    got = np.load(out)
    tree = common.small_scene(depth=4, basis_dim=4, seed=77)
    th = ob.TreeHandle(tree)
    poses = synth.make_poses(8)
    assert got.shape[0] == n_frames
    for i in range(n_frames):
        cam = ob.make_camera(synth.c2w_to_transform(poses[i % 8]), 72, 40, 100.0)
        want, _, _ = ob.render(th, cam, ob.default_options(), want_accum=False)
        assert np.array_equal(got[i], want), f"frame {i}"
