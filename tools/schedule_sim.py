#!/usr/bin/env python3
"""How much a longest-first ray order could buy a ONE-frame launch: list scheduling of the real
per-ray sample counts (CPU oracle, C1) onto N lanes at a constant time per sample.  Orders: the
natural one (8x8 blocks in screen order), blocks sorted by the longest ray of the SAME block in an
earlier pose (what a temporal cost map could know), by their true longest ray, and single rays
longest first (the optimum).  Result (profiles/r03_schedule_sim.jsonl): a frame can never finish
before its longest ray (230-263 dependent samples), the natural order is within 7-19 % of that
bound, so the one-frame time is a critical-path figure -- no queue order fixes it.
    python tools/schedule_sim.py > profiles/r03_schedule_sim.jsonl     (CPU only, ~3 min)"""
import sys, json, heapq, numpy as np
sys.path.insert(0, '/root/repo')
from oracle import binding as ob
from volrend_amd import synth
cfg = synth.CONFIGS["C1"]
tree = synth.make_config_tree("C1")
th = ob.TreeHandle(tree)
W, H, focal = cfg["width"], cfg["height"], cfg["focal"]
poses = synth.make_poses(200)
def samples_of(pi):
    tr = synth.c2w_to_transform(poses[pi])
    s, hits, cnt = ob.render_maps(th, ob.make_camera(tr, W, H, focal), ob.default_options())
    return s.astype(np.int64)
def blocks(s):  # [Hb, Wb, 64]
    Hb, Wb = H // 8, W // 8
    return s.reshape(Hb, 8, Wb, 8).transpose(0, 2, 1, 3).reshape(Hb * Wb, 64)
def makespan(ray_lengths, n_lanes):
    # greedy: each lane takes the next ray of the queue when it is free (constant time per sample)
    heap = [0] * n_lanes
    end = 0
    for L in ray_lengths:
        t = heapq.heappop(heap) + int(L)
        if t > end: end = t
        heapq.heappush(heap, t)
    return end
prev = samples_of(4); cur = samples_of(5); far = samples_of(24)
for name, target, pred in (("pose5 from pose4", cur, prev), ("pose24 from pose4", far, prev)):
    b = blocks(target); bp = blocks(pred)
    total = int(b.sum())
    for lanes in (262144, 327680):
        nat = makespan(b.reshape(-1), lanes)
        order = np.argsort(-bp.max(1), kind="stable")          # blocks by the PREVIOUS launch's longest ray
        srt = makespan(b[order].reshape(-1), lanes)
        oracle_order = np.argsort(-b.max(1), kind="stable")
        best = makespan(b[oracle_order].reshape(-1), lanes)
        ray_sorted = makespan(np.sort(b.reshape(-1))[::-1], lanes)
        print(json.dumps({"case": name, "lanes": lanes, "ideal": total / lanes, "longest_ray": int(b.max()),
                          "natural": nat, "blocks_by_prev_max": srt, "blocks_by_true_max": best, "rays_longest_first": ray_sorted}), flush=True)
