#!/usr/bin/env python3
"""Small launches on S alternating streams: does the tail of launch k run under the ramp-up and
the bulk of launch k + 1?

A launch of few frames is bound by its longest ray (the chip drains for ~0.3 ms while the last
rays finish); on ONE stream the next launch waits for that.  On S streams (the library keeps
one launch slot per stream) consecutive launches overlap.  Sustained time per frame over M
launches of F frames, fresh poses every launch, for S = 1, 2, 3, 4: host clock around
enqueue-all + synchronize (the launches are marshalled beforehand: one C call each).

    python tools/stream_overlap.py [--config C1] [--frames 1,2,4] [--streams 1,2,3,4] [--launches 48]
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
    ap.add_argument("--frames", default="1,2,4")
    ap.add_argument("--streams", default="1,2,3,4")
    ap.add_argument("--launches", type=int, default=48)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--variant", default="base")
    ap.add_argument("--tune", default="")
    ap.add_argument("--out", default="")
    args = ap.parse_args()

    import torch
    from volrend_amd import _abi, api, synth
    import bench as B

    if args.variant != "base":
        _abi._lib = None
        _abi.LIB_PATH = os.path.join(ROOT, "volrend_amd", f"libvolrend_hip_{args.variant}.so")
    cfg = synth.CONFIGS[args.config]
    W, H, focal = cfg["width"], cfg["height"], cfg["focal"]
    stree = B.load_or_make_tree(synth, args.config, 0, lambda: None)
    transforms = [synth.c2w_to_transform(p) for p in synth.make_poses(200)]
    tree = api.N3Tree.from_synth(stree)
    if args.tune:
        tree.set_tuning(**{k: int(x) for k, x in (kv.split("=") for kv in args.tune.split(","))})
    cam = api.Camera(W, H, focal, focal)
    opts = api.RenderOptions()
    smax = max(int(s) for s in args.streams.split(","))
    fmax = max(int(f) for f in args.frames.split(","))
    streams = [torch.cuda.Stream() for _ in range(smax)]
    # one frame set per stream: launches on different streams never share an output buffer
    imgs = torch.zeros((smax, fmax, H, W, 4), dtype=torch.uint8, device="cuda")
    out = open(args.out, "a") if args.out else None
    # GPU clocks of a running render loop (tools/lone_launch_probe.py)
    warm = api.PreparedBatch(tree, cam, [transforms[i] for i in range(64)], opts,
                             [torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda") for _ in range(64)], True)
    for nf in [int(f) for f in args.frames.split(",")]:
        for ns in [int(s) for s in args.streams.split(",")]:
            M = args.launches
            pbs = [api.PreparedBatch(tree, cam, [transforms[(7 + k * nf + i) % 200] for i in range(nf)],
                                     opts, [imgs[k % ns, i] for i in range(nf)], True) for k in range(M)]
            for k in range(2 * ns):  # every stream's launch slot owns its ray buffer
                pbs[k].launch(streams[k % ns])
            torch.cuda.synchronize()
            ms = []
            for _ in range(args.reps):
                warm.launch(streams[0])
                streams[0].synchronize()
                t0 = time.perf_counter()
                for k, pb in enumerate(pbs):
                    pb.launch(streams[k % ns])
                t_enq = time.perf_counter() - t0
                torch.cuda.synchronize()
                ms.append((time.perf_counter() - t0) * 1e3)
            rec = {"config": args.config, "variant": args.variant, "tune": args.tune, "frames_per_launch": nf,
                   "streams": ns, "launches": M,
                   "ms_per_frame_min": round(min(ms) / (M * nf), 5),
                   "ms_per_frame_mean": round(sum(ms) / len(ms) / (M * nf), 5),
                   "host_enqueue_ms_per_launch": round(t_enq * 1e3 / M, 4), "status": tree.status()}
            print(json.dumps(rec), flush=True)
            if out:
                out.write(json.dumps(rec) + "\n")
                out.flush()


if __name__ == "__main__":
    main()
