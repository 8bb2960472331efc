#!/usr/bin/env python3
"""Balance of the screen-tile shard, measured on ONE GPU: every rank's share of a launch (its
interleaved bands of `tile_rows` rows, VrFrame.rank / world) is rendered ALONE and timed, for a
set of band heights.  max / mean over the ranks is what the slowest GPU of an N-GPU job would
add to a perfectly balanced launch -- a measurement of the partition, NOT a scaling number (no
gather, no second GPU, every rank has the whole chip's caches to itself).

    python tools/shard_balance.py --config C3 --world 8 --tile-rows 8,16,32,64 --frames 64 \
           --out profiles/r04_shard_balance.jsonl
"""
from __future__ import annotations

import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C3")
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--tile-rows", default="8,16,32,64")
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--first-pose", type=int, default=64)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--out", default="")
    args = ap.parse_args()

    import torch
    from volrend_amd import _abi, api, synth
    import bench as B

    cfg = synth.CONFIGS[args.config]
    W, H, focal = cfg["width"], cfg["height"], cfg["focal"]
    stree = B.load_or_make_tree(synth, args.config, 0, lambda: None)
    transforms = [synth.c2w_to_transform(p) for p in synth.make_poses(200)]
    tree = api.N3Tree.from_synth(stree)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream()
    cam = api.Camera(W, H, focal, focal)
    opts = api.RenderOptions()
    nf = args.frames
    tr = [transforms[(args.first_pose + i) % 200] for i in range(nf)]
    tile_w = (W + 7) // 8 * 8
    out = open(args.out, "a") if args.out else None

    def timed(pb):
        pb.launch(stream)  # warm (also sizes the launch slot)
        torch.cuda.synchronize()
        ms = []
        for _ in range(args.reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            pb.launch(stream)
            e1.record(stream)
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1))
        return sum(ms) / len(ms)

    whole = torch.zeros((nf, H, W, 4), dtype=torch.uint8, device=dev)
    t_whole = timed(api.PreparedBatch(tree, cam, tr, opts, [whole[i] for i in range(nf)], True))
    for rows in [int(x) for x in args.tile_rows.split(",")]:
        per_rank = []
        for r in range(args.world):
            shard = api.TileShard(tile_w, rows, r, args.world, compact=True)
            nbytes = api.compact_bytes(W, H, shard)
            buf = torch.zeros((nf, nbytes), dtype=torch.uint8, device=dev)
            per_rank.append(timed(api.PreparedBatch(tree, cam, tr, opts, [buf[i] for i in range(nf)],
                                                    True, shard=shard)))
            del buf
        mean = sum(per_rank) / len(per_rank)
        rec = {"config": args.config, "world": args.world, "tile_rows": rows, "frames_per_launch": nf,
               "first_pose": args.first_pose, "whole_frame_launch_ms": round(t_whole, 3),
               "rank_launch_ms": [round(x, 3) for x in per_rank], "max_ms": round(max(per_rank), 3),
               "mean_ms": round(mean, 3), "max_over_mean": round(max(per_rank) / mean, 4),
               "sum_over_whole": round(sum(per_rank) / t_whole, 4),
               "note": "each rank's bands rendered alone on one GPU (no gather): the partition's balance, "
                       "not a scaling measurement"}
        print(json.dumps(rec), flush=True)
        if out:
            out.write(json.dumps(rec) + "\n")
            out.flush()
    tree.free_device()


if __name__ == "__main__":
    main()
