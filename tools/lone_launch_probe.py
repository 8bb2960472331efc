#!/usr/bin/env python3
"""Why is the driver's ONE timed launch slower than the same launch in a running render loop?
Times the same 20-frame launch (poses 5..24, C1) under different pre-conditions: what ran right
before it (a 5-frame launch as `--warmup 5`, a 64-frame launch, the launch itself), how long the
GPU idled in between, and whether the poses were seen before.  HIP events on the launch stream.

    python tools/lone_launch_probe.py [--out profiles/r04_lone_launch_probe.jsonl]
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
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream()
    cam = api.Camera(W, H, focal, focal)
    opts = api.RenderOptions()
    nf = args.frames
    imgs = torch.zeros((64, H, W, 4), dtype=torch.uint8, device=dev)

    def batch(first, n):
        return api.PreparedBatch(tree, cam, [transforms[(first + i) % 200] for i in range(n)], opts,
                                 [imgs[i] for i in range(n)], True)

    target = batch(5, nf)
    warm5 = batch(0, 5)
    warm64 = batch(100, 64)
    other = [batch(40 + 23 * k, nf) for k in range(4)]
    warm64.launch(stream)  # sizes the launch slot
    torch.cuda.synchronize()

    def timed(pb):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        pb.launch(stream)
        e1.record(stream)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1)

    def flush_caches():  # other poses, twice the L2 + MALL footprint of a launch
        for o in other:
            o.launch(stream)
        torch.cuda.synchronize()

    cases = {
        "after_warmup5 (the driver's flags)": lambda: (flush_caches(), time.sleep(0.05), warm5.launch(stream), torch.cuda.synchronize()),
        "after_warmup5_no_idle_before": lambda: (flush_caches(), warm5.launch(stream), torch.cuda.synchronize()),
        "after_64_frames_of_other_poses": lambda: (flush_caches(), warm64.launch(stream), torch.cuda.synchronize()),
        "after_itself (same poses warm)": lambda: (target.launch(stream), torch.cuda.synchronize()),
        "after_itself_then_50ms_idle": lambda: (target.launch(stream), torch.cuda.synchronize(), time.sleep(0.05)),
        "after_other_poses_then_50ms_idle": lambda: (flush_caches(), time.sleep(0.05)),
        "after_other_poses_no_idle": lambda: (flush_caches(),),
    }
    # how much GPU work does it take to be back at full speed after an idle period?
    for n_busy in (5, 20, 64, 128, 256):
        pbs = [batch(100 + 7 * k, min(64, n_busy - 64 * k)) for k in range((n_busy + 63) // 64)]
        cases[f"idle_50ms_then_{n_busy}_frames_of_other_poses"] = (
            lambda pbs=pbs: (flush_caches(), time.sleep(0.05), [pb.launch(stream) for pb in pbs],
                             torch.cuda.synchronize()))
    for idle_ms in (0.2, 1, 5, 20):
        cases[f"idle_{idle_ms}ms"] = lambda idle_ms=idle_ms: (flush_caches(), time.sleep(idle_ms / 1e3))
    out = open(args.out, "a") if args.out else None
    for name, pre in cases.items():
        ms = []
        for _ in range(args.reps):
            pre()
            ms.append(timed(target))
        rec = {"config": args.config, "frames": nf, "precondition": name,
               "launch_ms": [round(x, 3) for x in ms], "ms_per_frame_mean": round(sum(ms) / len(ms) / nf, 5),
               "ms_per_frame_min": round(min(ms) / nf, 5)}
        print(json.dumps(rec), flush=True)
        if out:
            out.write(json.dumps(rec) + "\n")
    tree.free_device()


if __name__ == "__main__":
    main()
