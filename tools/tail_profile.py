#!/usr/bin/env python3
"""Time-resolved picture of a launch of the fused kernel (a -DVR_TIMELINE=3 build, variant "tl3"):
per 2^15 shader clocks (14.9 us at 2.2 GHz) since the start of each wave: the time spent in the
retire / refill block, marching and shading, the march rounds, the marching lanes and the waves
that ended.  From it: how many waves are alive, what they are doing, how long a round takes and
how full the waves are -- while the ray queue still feeds them and in the tail after it.

    python -m volrend_amd.build --variant tl3 -DVR_TIMELINE=3
    python tools/tail_profile.py [--config C1] [--frames 1] [--tunes ""]
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

NB, NR = 64, 6      # buckets; rows: refill / march / shade time (x16 clocks), rounds, lanes, waves ended
CLOCKS = float(1 << 15)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C1")
    ap.add_argument("--variant", default="tl3")
    ap.add_argument("--frames", default="1")
    ap.add_argument("--first-pose", type=int, default=5)
    ap.add_argument("--tunes", default="")
    ap.add_argument("--out", default="")
    args = ap.parse_args()

    import torch
    from volrend_amd import _abi, api, synth
    import bench as B

    cfg = synth.CONFIGS[args.config]
    W, H, focal = cfg["width"], cfg["height"], cfg["focal"]
    stree = B.load_or_make_tree(synth, args.config, 0, lambda: None)
    transforms = [synth.c2w_to_transform(p) for p in synth.make_poses(200)]
    _abi._lib = None
    _abi.LIB_PATH = os.path.join(ROOT, "volrend_amd", f"libvolrend_hip_{args.variant}.so")
    lib = _abi.lib()
    rd = lib.vr_exp_tl3_read
    rd.restype = C.c_int
    rd.argtypes = [C.POINTER(C.c_uint64 * (NR * NB)), C.c_int]
    frames_list = [int(x) for x in args.frames.split(",")]
    imgs = torch.zeros((max(frames_list), H, W, 4), dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream()
    cam = api.Camera(W, H, focal, focal)
    tree = api.N3Tree.from_synth(stree)
    out = open(args.out, "a") if args.out else None
    for tune in args.tunes.split(";"):
        if tune:
            tree.set_tuning(**{k: int(x) for k, x in (kv.split("=") for kv in tune.split(","))})
        for nf in frames_list:
            pb = api.PreparedBatch(tree, cam, [transforms[(args.first_pose + i) % 200] for i in range(nf)],
                                   api.RenderOptions(), [imgs[i] for i in range(nf)], True)
            pb.launch(stream)
            torch.cuda.synchronize()
            buf = (C.c_uint64 * (NR * NB))()
            assert rd(C.byref(buf), 1) == NB
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            pb.launch(stream)
            e1.record(stream)
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            assert rd(C.byref(buf), 1) == NB
            h = np.array(list(buf), dtype=np.float64).reshape(NR, NB)
            n_waves = h[5].sum()
            alive = n_waves - np.concatenate([[0], np.cumsum(h[5])[:-1]])   # at the start of the bucket
            last = int(np.nonzero(h[3] + h[5])[0][-1])
            rows = []
            for k in range(last + 1):
                t_ref, t_mar, t_sha, r, l, e = h[:, k]
                wave_clocks = CLOCKS * (alive[k] - 0.5 * e)     # wave time available in the bucket
                rows.append({"bucket": k, "rounds": int(r), "lanes_per_round": round(l / r, 1) if r else 0,
                             "waves_alive": int(alive[k]), "ended": int(e),
                             "frac_refill": round(16 * t_ref / wave_clocks, 3) if wave_clocks > 0 else None,
                             "frac_march": round(16 * t_mar / wave_clocks, 3) if wave_clocks > 0 else None,
                             "frac_shade": round(16 * t_sha / wave_clocks, 3) if wave_clocks > 0 else None,
                             "march_clocks_per_round": round(16 * t_mar / r) if r else None})
            rec = {"config": args.config, "tune": tune, "frames": nf, "launch_ms": round(ms, 4),
                   "waves": int(n_waves), "bucket_clocks": int(CLOCKS), "buckets": rows}
            print(json.dumps({k: v for k, v in rec.items() if k != "buckets"}))
            for row in rows:
                print("   bucket %2d  waves alive %4d  rounds %6d  lanes/round %4.1f  time: refill %5s march %5s shade %5s"
                      "  march clocks/round %s" % (row["bucket"], row["waves_alive"], row["rounds"], row["lanes_per_round"],
                                                   row["frac_refill"], row["frac_march"], row["frac_shade"],
                                                   row["march_clocks_per_round"]))
            if out:
                out.write(json.dumps(rec) + "\n")
                out.flush()


if __name__ == "__main__":
    main()
