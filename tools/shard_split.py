#!/usr/bin/env python3
"""What does it cost a RANK to render its share of the driver's 20 poses as two launches of 10 on two
streams instead of one launch of 20?  (Two launches would let the gather of the first half run
under the rendering of the second; one launch leaves the whole gather exposed.)  Each rank's share
(world 2 / 4 / 8, 8-row bands, COMPACT layout) rendered alone on ONE GPU, host clock around
enqueue + synchronize, best of `reps`.  The render side only: no gather, no second GPU.

    python tools/shard_split.py [--config C1] [--frames 20] [--worlds 2,4,8]
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C1")
    ap.add_argument("--frames", type=int, default=20)
    ap.add_argument("--worlds", default="2,4,8")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--out", default="")
    args = ap.parse_args()

    import torch
    from volrend_amd import api, synth
    import bench as B

    cfg = synth.CONFIGS[args.config]
    W, H, focal = cfg["width"], cfg["height"], cfg["focal"]
    stree = B.load_or_make_tree(synth, args.config, 0, lambda: None)
    transforms = [synth.c2w_to_transform(p) for p in synth.make_poses(200)]
    tree = api.N3Tree.from_synth(stree)
    cam = api.Camera(W, H, focal, focal)
    opts = api.RenderOptions()
    nf = args.frames
    tr = [transforms[(5 + i) % 200] for i in range(nf)]
    tile_w = (W + 7) // 8 * 8
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    warm_imgs = [torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda") for _ in range(32)]
    warm = api.PreparedBatch(tree, cam, [transforms[100 + i] for i in range(32)], opts, warm_imgs, True)
    out = open(args.out, "a") if args.out else None

    def best(launches):
        """launches: [(PreparedBatch, stream index)] enqueued back to back."""
        for pb, si in launches:           # slot sizing
            pb.launch(streams[si])
        torch.cuda.synchronize()
        ms = []
        for _ in range(args.reps):
            warm.launch(streams[0])       # GPU clocks of a render loop
            streams[0].synchronize()
            t0 = time.perf_counter()
            for pb, si in launches:
                pb.launch(streams[si])
            torch.cuda.synchronize()
            ms.append((time.perf_counter() - t0) * 1e3)
        return min(ms)

    for world in [int(w) for w in args.worlds.split(",")]:
        one, two = [], []
        for r in range(world):
            shard = api.TileShard(tile_w, 8, r, world, compact=True)
            nbytes = api.compact_bytes(W, H, shard)
            buf = torch.zeros((nf, nbytes), dtype=torch.uint8, device="cuda")
            pb20 = api.PreparedBatch(tree, cam, tr, opts, [buf[i] for i in range(nf)], True, shard=shard)
            h = nf // 2
            pba = api.PreparedBatch(tree, cam, tr[:h], opts, [buf[i] for i in range(h)], True, shard=shard)
            pbb = api.PreparedBatch(tree, cam, tr[h:], opts, [buf[i] for i in range(h, nf)], True, shard=shard)
            one.append(best([(pb20, 0)]))
            two.append(best([(pba, 0), (pbb, 1)]))
            del buf
        rec = {"config": args.config, "world": world, "frames": nf,
               "one_launch_ms_per_rank": [round(x, 3) for x in one], "two_launches_two_streams_ms_per_rank": [round(x, 3) for x in two],
               "slowest_rank_one_launch_ms": round(max(one), 3), "slowest_rank_two_launches_ms": round(max(two), 3),
               "ratio": round(max(two) / max(one), 4),
               "note": "render side only (each rank's share alone on one GPU, host clock): what splitting costs before "
                       "the gather it would hide is counted"}
        print(json.dumps(rec), flush=True)
        if out:
            out.write(json.dumps(rec) + "\n")
            out.flush()
    tree.free_device()


if __name__ == "__main__":
    main()
