#!/usr/bin/env python3
"""Small launches on TWO alternating streams, wave slot by wave slot: what would taking the drain
out of a launch buy the SUSTAINED rate?

tools/drain_sim.py models one launch alone and therefore never saw what a thinning wave costs its
neighbour: a wave that is down to a handful of live lanes still holds a full wave slot (96 VGPRs x
64 lanes + 7.5 KB of LDS), so the next launch ramps up into a half-occupied chip.  This model keeps
the chip's 5120 slots explicit and runs a sequence of launches the way the two-stream callers issue
them (launch k + 2 waits for launch k; launches k and k + 1 share the chip):

  * per launch: ray generation (needs free slots: its progress per tick is proportional to the
    free share of the chip), then the persistent render waves -- a wave starts when a slot is
    free, takes rays from its launch's queue when >= 20 of its lanes are idle, marches one sample
    per live lane and tick, and leaves when it holds nothing and the queue is dry;
  * a tick costs A + B * (occupied slots) shader clocks for everybody (fit: a lone wave 1900
    clocks per round all-in, a full chip 4950 = 0.236 ms per C1 frame at 64 frames per launch);
  * per-ray sample counts are the REAL ones of consecutive C1 poses (CPU oracle).

Policies for the drain phase of a launch (its queue is dry):
  none    every wave runs until its last ray has ended (the kernel as it is)
  tail    "resumable rays": a wave that is down to <= T live lanes writes them back (t, light,
          colour so far) and EXITS; when the launch's last wave has left, a follow-up kernel of the
          same launch marches the written-back rays in full waves (gap G ticks: kernel boundary +
          first refill).  The frame is complete when that kernel ends: same contract as today.
  ideal   live rays are re-packed into full waves every tick at no cost (bound)

    python tools/drain_sim2.py [--frames 1,4] [--launches 12] > profiles/r06_drain_sim2.jsonl   (CPU, ~4 min)
"""
import argparse
import json
import sys

import numpy as np

sys.path.insert(0, "/root/repo")

A, B, GHZ = 1900.0, 0.6, 2.2
SLOTS = 5120
REFILL_MIN = 20
RAYGEN_US_PER_FRAME = 16.0   # on an empty chip (kernel trace, profiles/r06_overlap_trace_r05kernel.jsonl)


class Launch:
    def __init__(self, idx, rays, nf):
        self.idx, self.rays, self.nf = idx, rays, nf
        self.head = 0
        self.want = int(min(SLOTS, max(256, (len(rays) // 64 + 1) // 2 * 1)))  # render grid ~ blocks / 2
        self.want = int(min(SLOTS, max(256, nf * 10000 // 2)))
        self.started = 0
        self.state = "wait"     # wait -> raygen -> render -> (tailgap -> tail) -> done
        self.raygen_left = RAYGEN_US_PER_FRAME * nf
        self.pool = []
        self.gap_left = 0
        self.t_start = self.t_end = None
        self.is_tail = False


def simulate(frames_rays, nf, n_launches, streams, policy, T=16, gap_ticks=5, raygen_eff=0.5):
    """frames_rays: list of per-frame ray-length arrays (queue order).  Returns per-launch (start, end) in us."""
    launches = []
    for k in range(n_launches):
        rays = np.concatenate([frames_rays[(k * nf + i) % len(frames_rays)] for i in range(nf)])
        launches.append(Launch(k, rays, nf))
    rem = np.zeros((SLOTS, 64), np.int32)
    owner = np.full(SLOTS, -1, np.int32)      # launch index per slot
    tailw = np.zeros(SLOTS, bool)             # slot runs a wave of a follow-up (tail) kernel
    now = 0.0
    done = 0
    while done < n_launches:
        # stream order: launch k may start once launch k - streams is done
        for L in launches:
            if L.state == "wait" and (L.idx < streams or launches[L.idx - streams].state == "done"):
                L.state, L.t_start = "raygen", now
        occupied = int((owner >= 0).sum())
        free = SLOTS - occupied
        dt_us = (A + B * occupied) / GHZ / 1e3
        # ray generation: progress with the free share of the chip (first come first served)
        share = free
        for L in launches:
            if L.state == "raygen":
                L.raygen_left -= dt_us * raygen_eff * share / SLOTS if occupied else dt_us
                share = 0
                if L.raygen_left <= 0:
                    L.state = "render"
        # follow-up kernels waiting for their gap
        for L in launches:
            if L.state == "tailgap":
                L.gap_left -= 1
                if L.gap_left <= 0:
                    L.rays = np.concatenate(L.pool) if L.pool else np.zeros(0, np.int32)
                    L.pool, L.head, L.started = [], 0, 0
                    L.want = int(min(SLOTS, (len(L.rays) + 63) // 64))
                    L.is_tail, L.state = True, "render"
        # new waves into free slots (older launch first)
        for L in launches:
            if L.state != "render" or L.started >= L.want or L.head >= len(L.rays):
                continue
            fs = np.nonzero(owner < 0)[0]
            n = int(min(len(fs), L.want - L.started, (len(L.rays) - L.head + 63) // 64))
            if n > 0:
                owner[fs[:n]] = L.idx
                tailw[fs[:n]] = L.is_tail
                rem[fs[:n]] = 0
                L.started += n
        # refill
        live = rem > 0
        idle = 64 - live.sum(1)
        for L in launches:
            if L.state != "render" or L.head >= len(L.rays):
                continue
            mine = np.nonzero((owner == L.idx) & ((idle >= REFILL_MIN) | (idle == 64)))[0]
            for w in mine:
                if L.head >= len(L.rays):
                    break
                c = int(min(idle[w], len(L.rays) - L.head))
                lanes = np.nonzero(rem[w] <= 0)[0][:c]
                rem[w, lanes] = L.rays[L.head:L.head + c]
                L.head += c
        live = rem > 0
        n_live = live.sum(1)
        # drain policies / exits
        for L in launches:
            if L.state != "render":
                continue
            dry = L.head >= len(L.rays)
            mine = owner == L.idx
            if dry and policy == "tail" and not L.is_tail:
                post = mine & (n_live > 0) & (n_live <= T)
                if post.any():
                    L.pool.append(rem[post][rem[post] > 0])
                    rem[post] = 0
                    n_live = (rem > 0).sum(1)
            if dry and policy == "ideal":
                idx = np.nonzero(mine)[0]
                left = rem[idx][rem[idx] > 0]
                need = (len(left) + 63) // 64
                rem[idx] = 0
                flat = np.zeros(need * 64, np.int32)
                flat[:len(left)] = left
                if need:
                    rem[idx[:need]] = flat.reshape(need, 64)
                n_live = (rem > 0).sum(1)
            if dry:
                gone = mine & (n_live == 0)
                owner[gone] = -1
                if not (owner == L.idx).any() and L.started > 0 or (dry and len(L.rays) == 0):
                    if L.pool and not L.is_tail:
                        L.state, L.gap_left = "tailgap", gap_ticks
                    else:
                        L.state, L.t_end = "done", now + dt_us
                        done += 1
        rem[rem > 0] -= 1
        now += dt_us
    return [(L.t_start, L.t_end) for L in launches]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", default="1,4")
    ap.add_argument("--launches", type=int, default=12)
    ap.add_argument("--first-pose", type=int, default=5)
    args = ap.parse_args()
    from oracle import binding as ob
    from volrend_amd import synth
    cfg = synth.CONFIGS["C1"]
    tree = synth.make_config_tree("C1")
    th = ob.TreeHandle(tree)
    W, H, focal = cfg["width"], cfg["height"], cfg["focal"]
    poses = synth.make_poses(200)

    def frame(pi):
        tr = synth.c2w_to_transform(poses[pi % 200])
        s, _, _ = ob.render_maps(th, ob.make_camera(tr, W, H, focal), ob.default_options())
        b = s.reshape(H // 8, 8, W // 8, 8).transpose(0, 2, 1, 3).reshape(-1, 64)
        r = b[b.max(1) > 0].reshape(-1).astype(np.int32)
        return r[r > 0]

    n_frames = 16
    frames_rays = [frame(args.first_pose + 7 * i) for i in range(n_frames)]
    for nf in [int(f) for f in args.frames.split(",")]:
        nl = args.launches
        rec = {"frames_per_launch": nf, "launches": nl, "model": {"tick_clocks": [A, B], "ghz": GHZ, "slots": SLOTS,
               "raygen_us_per_frame": RAYGEN_US_PER_FRAME}, "rays_per_frame": int(np.mean([len(r) for r in frames_rays])),
               "longest_ray": int(max(r.max() for r in frames_rays))}
        for streams in (1, 2):
            for policy, kw in (("none", {}), ("tail", dict(T=8)), ("tail", dict(T=16)), ("tail", dict(T=32)),
                               ("tail", dict(T=16, gap_ticks=10)), ("ideal", {})):
                se = simulate(frames_rays, nf, nl, streams, policy, **kw)
                ends = [e for _, e in se]
                sustained = (ends[-1] - ends[2]) / ((nl - 3) * nf)   # us per frame, steady state
                lat = float(np.median([e - s for s, e in se[2:]]))
                key = f"streams{streams}_{policy}" + ("" if not kw else "_" + "_".join(f"{k}{v}" for k, v in kw.items()))
                rec[key] = {"us_per_frame": round(sustained, 1), "launch_latency_us": round(lat, 1)}
                print(key, rec[key], file=sys.stderr, flush=True)
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
