#!/usr/bin/env python3
"""One bench-shaped launch against the oracle: N consecutive poses of a config in ONE
vr_render_batch launch (the shape bench.py times), every frame compared bit for bit.
    python tools/check_batch_parity.py [config] [n_frames] [first_pose]   (on the GPU box)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import binding as ob  # noqa: E402
from volrend_amd import api, synth  # noqa: E402


def main():
    import torch
    name = sys.argv[1] if len(sys.argv) > 1 else "C1"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    first = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    cfg = synth.CONFIGS[name]
    stree = bench.load_or_make_tree(synth, name, 0, lambda: None)
    W, H, focal = cfg["width"], cfg["height"], cfg["focal"]
    t = api.N3Tree.from_synth(stree)
    th = ob.TreeHandle(stree)
    poses = synth.make_poses(200)
    trs = [synth.c2w_to_transform(poses[(first + i) % 200]) for i in range(n)]
    imgs = torch.zeros((n, H, W, 4), dtype=torch.uint8, device="cuda")
    api.launch_renderer_batch(t, api.Camera(W, H, focal, focal), trs, api.RenderOptions(), list(imgs),
                              None, True)
    torch.cuda.synchronize()
    got = imgs.cpu().numpy()
    bad, t0 = [], time.time()
    for i in range(n):
        want, _, _ = ob.render(th, ob.make_camera(trs[i], W, H, focal), ob.default_options(),
                               want_accum=False)
        if not np.array_equal(got[i], want):
            bad.append(i)
    print(json.dumps({"config": name, "frames_in_one_launch": n, "first_pose": first,
                      "frames_bit_equal_to_oracle": n - len(bad), "mismatching_frames": bad,
                      "oracle_seconds": round(time.time() - t0, 1), "status": t.status()}))
    t.free_device()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
