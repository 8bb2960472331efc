#!/usr/bin/env python3
"""Fast A/B of library builds and scheduling knobs in ONE process (the tree is generated once,
every variant uploads it in ~0.3 s): per (variant, tune, frames-per-launch) the mean and minimum
duration of `reps` launches (HIP events on the launch stream), after one warm launch.

    python tools/quick_ab.py --config C1 --variants base,abl6 --tunes ";waves_per_cu=16" \
           --frames 64,20,1 --reps 4 [--check]

--check: the first frame of every (variant, tune) must equal the "base" frame byte for byte
(experiments that are meant to keep the results).  Measurement tooling, not the product.
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
    ap.add_argument("--config", default="C1")
    ap.add_argument("--variants", default="base")
    ap.add_argument("--tunes", default="", help="';'-separated k=v,k=v sets ('' = defaults)")
    ap.add_argument("--frames", default="64,20")
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--fp", default="strict")
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--rotate", action="store_true",
                    help="every timed launch renders DIFFERENT poses (as the bench's timed region does); "
                         "without it the same launch is repeated and finds its data warm in the caches")
    ap.add_argument("--out", default="")
    args = ap.parse_args()

    import torch
    from volrend_amd import _abi, api, synth
    import bench as B

    cfg = synth.CONFIGS[args.config]
    W, H, focal = cfg["width"], cfg["height"], cfg["focal"]
    stree = B.load_or_make_tree(synth, args.config, 0, lambda: None)
    transforms = [synth.c2w_to_transform(p) for p in synth.make_poses(200)]
    frames_list = [int(x) for x in args.frames.split(",")]
    nmax = max(frames_list)
    dev = torch.device("cuda", 0)
    imgs = torch.zeros((nmax, H, W, 4), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream()
    fp_mode = _abi.FP_FMA if args.fp == "fma" else _abi.FP_STRICT
    cam = api.Camera(W, H, focal, focal)
    opts = api.RenderOptions()
    ref_frame = None
    out = open(args.out, "a") if args.out else None
    for v in args.variants.split(","):
        _abi._lib = None
        _abi.LIB_PATH = (os.path.join(ROOT, "volrend_amd", "libvolrend_hip.so") if v == "base" else
                         os.path.join(ROOT, "volrend_amd", f"libvolrend_hip_{v}.so"))
        tree = api.N3Tree.from_synth(stree)
        # the first measurements of a process run while the GPU's clocks still ramp up (an MI355X out of idle
        # needs >= 30 ms of work, profiles/r04_lone_launch_probe.jsonl): ~250 ms of untimed frames first, or
        # whatever stands first in --tunes looks 3-5 % slower than it is (profiles/r06_frame_group.jsonl)
        npre = min(nmax, 64)
        pre = api.PreparedBatch(tree, cam, [transforms[(100 + i) % 200] for i in range(npre)], opts,
                                [imgs[i] for i in range(npre)], True, fp_mode=fp_mode)
        for _ in range(max(1, 1024 // npre)):
            pre.launch(stream)
        torch.cuda.synchronize()
        for tune in args.tunes.split(";"):
            if tune:
                tree.set_tuning(**{k: int(x) for k, x in (kv.split("=") for kv in tune.split(","))})
            for nf in frames_list:
                # poses offset like bench.py's timed region (warmup 64 / 5)
                first = 64 if nf >= 64 else 5
                def batch(at):
                    return api.PreparedBatch(tree, cam, [transforms[(at + i) % 200] for i in range(nf)],
                                             opts, [imgs[i] for i in range(nf)], True, fp_mode=fp_mode)
                pbs = [batch(first + (k * (nf + 7) if args.rotate else 0)) for k in range(args.reps + 1)]
                pbs[-1].launch(stream)
                torch.cuda.synchronize()
                ms = []
                for k in range(args.reps):
                    pb = pbs[k]
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(stream)
                    pb.launch(stream)
                    e1.record(stream)
                    torch.cuda.synchronize()
                    ms.append(e0.elapsed_time(e1))
                same = None
                if args.check:
                    pbs[0].launch(stream)
                    torch.cuda.synchronize()
                    f0 = imgs[0].clone()
                    if ref_frame is None or nf not in ref_frame:
                        ref_frame = ref_frame or {}
                        ref_frame[nf] = f0
                    same = bool(torch.equal(ref_frame[nf], f0))
                rec = {"variant": v, "tune": tune, "frames": nf, "config": args.config,
                       "ms_per_frame_mean": round(sum(ms) / len(ms) / nf, 5),
                       "ms_per_frame_min": round(min(ms) / nf, 5),
                       "launch_ms": [round(x, 3) for x in ms], "status": tree.status(),
                       "same_as_first": same}
                if os.environ.get("VR_TIMELINE"):
                    rec["sched_stats"] = list(tree.sched_stats().values())
                print(json.dumps(rec), flush=True)
                if out:
                    out.write(json.dumps(rec) + "\n")
                    out.flush()
        tree.free_device()
    _abi._lib = None


if __name__ == "__main__":
    main()
