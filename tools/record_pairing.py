#!/usr/bin/env python3
"""Which sibling should share a cache line?  An SH9 record is 54 bytes in a 64-byte slot, so a
128-byte line holds the records of two sibling slots -- today slots 2k and 2k + 1 of a node, i.e.
the two children that differ in z (slot = x << 2 | y << 1 | z).  A ray wants ONE of them; the line
is fetched whole.  This study renders frames with the instrumented flavour of a build whose
record bitmap has one bit per 64-byte slot (-DVR_TOUCH_LEAF_SHIFT=6, variant "t64"), reads the
bitmap (vr_touch_read) and counts, per frame, the distinct lines the touched records would occupy
if the pair were formed along z (as built), y or x -- and the floor: half the touched records.

    python -m volrend_amd.build --variant t64 -DVR_TOUCH_LEAF_SHIFT=6
    python tools/record_pairing.py --config C3 --poses 5,60,110 [--launch 16]
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
    ap.add_argument("--config", default="C3")
    ap.add_argument("--variant", default="t64")
    ap.add_argument("--poses", default="5,60,110")
    ap.add_argument("--launch", type=int, default=0, help="also: a launch of this many consecutive poses")
    ap.add_argument("--out", default="")
    args = ap.parse_args()

    import torch
    from volrend_amd import _abi, api, synth
    import bench as B

    _abi._lib = None
    _abi.LIB_PATH = os.path.join(ROOT, "volrend_amd", f"libvolrend_hip_{args.variant}.so")
    cfg = synth.CONFIGS[args.config]
    W, H, focal = cfg["width"], cfg["height"], cfg["focal"]
    stree = B.load_or_make_tree(synth, args.config, 0, lambda: None)
    transforms = [synth.c2w_to_transform(p) for p in synth.make_poses(200)]
    tree = api.N3Tree.from_synth(stree)
    info = tree.info()
    stride = int(info["leaf_stride"])
    cam = api.Camera(W, H, focal, focal)
    stream = torch.cuda.current_stream()
    tree.touch_enable(True)
    out = open(args.out, "a") if args.out else None

    def study(pose_list, label):
        n = len(pose_list)
        imgs = [torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda") for _ in range(n)]
        counters = torch.zeros((n, 7), dtype=torch.int64, device="cuda")
        tree.touch_count(reset=True)
        api.launch_renderer_batch(tree, cam, [transforms[p % 200] for p in pose_list], api.RenderOptions(), imgs,
                                  stream, True, counters=[counters[i] for i in range(n)])
        torch.cuda.synchronize()
        bm, gran = tree.touch_read(0)
        assert gran == 64 and stride == 64, (gran, stride)
        bits = np.unpackbits(bm.view(np.uint8), bitorder="little")          # bit i = slot i touched
        slots = np.nonzero(bits)[0].astype(np.int64)                        # slot = node * 8 + (x << 2 | y << 1 | z)
        rec = {"config": args.config, "what": label, "frames": n, "touched_records": int(slots.size),
               "hit_samples": int(counters[:, 4].sum().item())}
        for axis, bit in (("z (as built)", 1), ("y", 2), ("x", 4)):
            rec[f"distinct_lines_pair_{axis}"] = int(np.unique(slots & ~np.int64(bit)).size)
        rec["floor_half_the_records"] = int((slots.size + 1) // 2)
        # how full are the sibling groups?  (8 slots of a node = 4 lines as built)
        nodes, cnt = np.unique(slots >> 3, return_counts=True)
        rec["touched_nodes"] = int(nodes.size)
        rec["records_per_touched_node_mean"] = round(float(cnt.mean()), 3)
        print(json.dumps(rec), flush=True)
        if out:
            out.write(json.dumps(rec) + "\n")
            out.flush()

    for p in [int(x) for x in args.poses.split(",")]:
        study([p], f"one frame, pose {p}")
    if args.launch > 0:
        study(list(range(30, 30 + args.launch)), f"one launch of {args.launch} consecutive poses (30..)")
    tree.touch_enable(False)
    tree.free_device()


if __name__ == "__main__":
    main()
