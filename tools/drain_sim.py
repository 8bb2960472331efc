#!/usr/bin/env python3
"""What re-grouping the rays of the drain phase could buy a launch: a wave-level simulation of the
persistent kernel on the REAL per-ray sample counts of C1 frames (CPU oracle).

Model: W resident waves of 64 lanes march in lock step, one sample per live lane and tick; a wave
with >= 24 idle lanes takes the next rays of the queue; a tick costs  a + b * (waves alive)
shader clocks (fit of profiles/r03_tail_profile.jsonl: 2000 clocks per round in an empty chip,
5100 with 4600 waves marching -- a wave costs the same whether 3 or 60 of its lanes are alive).
Policies after the queue has run dry:
  none      every wave runs until its last ray ends (the kernel as it is)
  phased    waves hand their live rays to a list and exit -- when <= T of them are alive, or R
            rounds after the queue ran dry -- and a follow-up launch marches the list with full
            waves (launch gap G clocks); repeated until nothing is left
  ideal     live rays are re-packed into full waves every tick (an exchange that costs nothing)
    python tools/drain_sim.py [--frames 1] [--poses 5,24] > profiles/r03_drain_sim.jsonl
"""
import argparse
import json
import sys

import numpy as np

sys.path.insert(0, "/root/repo")

A, B = 1900.0, 0.7       # clocks per tick = A + B * waves alive
GHZ = 2.2


def tick_clocks(waves):
    return A + B * waves


def run_phase(rays, n_waves, policy, T, R, refill_min=24):
    """One launch over `rays` (remaining samples per ray, queue order).  Returns (clocks, leftover rays)."""
    n = len(rays)
    W = int(min(n_waves, max(1, (n + 63) // 64)))
    rem = np.zeros((W, 64), np.int32)
    head = 0
    clocks = 0.0
    dry_at = None
    tick = 0
    alive_w = np.ones(W, bool)
    posted = []
    while True:
        live = rem > 0
        idle = (~live).sum(1)
        # refill
        if head < n:
            want = alive_w & ((idle >= refill_min) | (idle == 64))
            if want.any():
                idx_w = np.nonzero(want)[0]
                cnt = idle[idx_w]
                start = head + np.concatenate([[0], np.cumsum(cnt)[:-1]])
                for w, s, c in zip(idx_w, start, cnt):
                    if s >= n:
                        break
                    c = int(min(c, n - s))
                    lanes = np.nonzero(~live[w])[0][:c]
                    rem[w, lanes] = rays[s:s + c]
                head = int(min(n, head + cnt.sum()))
                live = rem > 0
        if head >= n and dry_at is None:
            dry_at = tick
        n_alive = live.sum(1)
        if dry_at is not None:
            if policy == "phased":
                post = alive_w & (n_alive > 0) & ((n_alive <= T) | (tick - dry_at >= R)) & (tick > 0)
                if post.any():
                    posted.append(rem[post][rem[post] > 0])
                    rem[post] = 0
                    n_alive = (rem > 0).sum(1)
            alive_w &= n_alive > 0
            if policy == "ideal":
                left = rem[rem > 0]
                Wn = (len(left) + 63) // 64
                rem = np.zeros((max(Wn, 1), 64), np.int32)
                rem.reshape(-1)[:len(left)] = left
                alive_w = np.zeros(max(Wn, 1), bool)
                alive_w[:Wn] = True
        if not alive_w.any() and head >= n:
            break
        rem[rem > 0] -= 1
        clocks += tick_clocks(int(alive_w.sum()))
        tick += 1
    left = np.concatenate(posted) if posted else np.zeros(0, np.int32)
    return clocks, left


def simulate(rays, policy, T=24, R=10 ** 9, gap=22000.0, n_waves=5120):
    total = 0.0
    phases = 0
    t, r = T, R
    while len(rays):
        c, rays = run_phase(rays, n_waves, policy, t, r)
        total += c + (gap if phases else 0.0)
        phases += 1
        t = max(t // 2, 0) if policy == "phased" else t
    return total, phases


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--poses", default="5,24")
    ap.add_argument("--frames", default="1,4")
    args = ap.parse_args()
    from oracle import binding as ob
    from volrend_amd import synth
    cfg = synth.CONFIGS["C1"]
    tree = synth.make_config_tree("C1")
    th = ob.TreeHandle(tree)
    W, H, focal = cfg["width"], cfg["height"], cfg["focal"]
    poses = synth.make_poses(200)

    def frame(pi):
        tr = synth.c2w_to_transform(poses[pi])
        s, _, _ = ob.render_maps(th, ob.make_camera(tr, W, H, focal), ob.default_options())
        b = s.reshape(H // 8, 8, W // 8, 8).transpose(0, 2, 1, 3).reshape(-1, 64)
        return b[b.max(1) > 0].reshape(-1).astype(np.int32)     # blocks that enter the volume, queue order

    for first in [int(p) for p in args.poses.split(",")]:
        for nf in [int(f) for f in args.frames.split(",")]:
            rays = np.concatenate([frame(first + i) for i in range(nf)])
            rays = rays[rays > 0]
            base, _ = simulate(rays, "none")
            rec = {"first_pose": first, "frames": nf, "rays": int(len(rays)), "longest_ray": int(rays.max()),
                   "model": {"clocks_per_tick": [A, B], "ghz": GHZ},
                   "none_us": round(base / GHZ / 1e3, 1)}
            ideal, _ = simulate(rays, "ideal")
            rec["ideal_us"] = round(ideal / GHZ / 1e3, 1)
            for T, R in ((24, 10 ** 9), (32, 10 ** 9), (32, 16), (32, 32), (48, 24)):
                c, ph = simulate(rays, "phased", T, R)
                rec[f"phased_T{T}_R{R if R < 10**8 else 'inf'}_us"] = [round(c / GHZ / 1e3, 1), ph]
            print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
