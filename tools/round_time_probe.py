#!/usr/bin/env python3
"""How long does ONE march round of a ray take when the chip is (almost) empty, and when it is full?

One-frame launches of the C1 tree at image sizes from 8x8 (one wave) to 800x800 (every lane of the
chip twice), same poses, same field of view.  The launch cannot finish before its longest ray has
marched all its samples one after the other, so  launch time / longest ray  is an upper bound of
the time per dependent round at that load; the 8x8 launch IS one chain.  The per-ray sample counts
come from the CPU oracle (measurement tooling, like every file under tools/).

    python tools/round_time_probe.py [--config C1] [--sizes 8,16,...] [--tunes ";refill_min=12"]
"""
from __future__ import annotations

import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C1")
    ap.add_argument("--sizes", default="8,16,32,64,128,256,400,800")
    ap.add_argument("--poses", default="10,60,110")
    ap.add_argument("--tunes", default="")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--out", default="")
    args = ap.parse_args()

    import torch
    from volrend_amd import api, synth
    from oracle import binding as ob
    import bench as B

    cfg = synth.CONFIGS[args.config]
    stree = B.load_or_make_tree(synth, args.config, 0, lambda: None)
    th = ob.TreeHandle(stree)
    poses = synth.make_poses(200)
    tree = api.N3Tree.from_synth(stree)
    stream = torch.cuda.current_stream()
    out = open(args.out, "a") if args.out else None
    for size in [int(s) for s in args.sizes.split(",")]:
        w = h = size
        f = cfg["focal"] * size / cfg["width"]
        img = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
        cam = api.Camera(w, h, f, f)
        longest, mean_s, n_in = [], [], []
        trs = []
        for pi in [int(p) for p in args.poses.split(",")]:
            tr = synth.c2w_to_transform(poses[pi])
            trs.append(tr)
            if size <= 256:
                samples, _, _ = ob.render_maps(th, ob.make_camera(tr, w, h, f), ob.default_options())
                longest.append(int(samples.max()))
                mean_s.append(float(samples.mean()))
                n_in.append(int((samples > 0).sum()))
        for tune in args.tunes.split(";"):
            if tune:
                tree.set_tuning(**{k: int(x) for k, x in (kv.split("=") for kv in tune.split(","))})
            per_pose = []
            for tr in trs:
                pb = api.PreparedBatch(tree, cam, [tr], api.RenderOptions(), [img], True)
                pb.launch(stream)
                torch.cuda.synchronize()
                ms = []
                for _ in range(args.reps):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(stream)
                    pb.launch(stream)
                    e1.record(stream)
                    torch.cuda.synchronize()
                    ms.append(e0.elapsed_time(e1))
                per_pose.append(min(ms))
            rec = {"config": args.config, "size": size, "tune": tune,
                   "launch_us_min_per_pose": [round(1e3 * m, 1) for m in per_pose],
                   "longest_ray_samples": longest or None,
                   "mean_samples": [round(m, 1) for m in mean_s] or None,
                   "us_per_round_bound": ([round(1e3 * m / l, 3) for m, l in zip(per_pose, longest)]
                                          if longest else None),
                   "status": tree.status()}
            print(json.dumps(rec), flush=True)
            if out:
                out.write(json.dumps(rec) + "\n")
                out.flush()


if __name__ == "__main__":
    main()
