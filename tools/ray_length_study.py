#!/usr/bin/env python3
"""Per-ray work of a frame on the CPU oracle: how long the longest rays are (the tail of a
persistent launch is one such chain) and how well cheap ray-generation-time quantities predict
them (for a longest-first ray order).   python tools/ray_length_study.py [config] [pose ...]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import binding as ob  # noqa: E402
from volrend_amd import synth  # noqa: E402


def chords(transform, W, H, focal, tree):
    """Length (tree units, after the _get_delta_scale normalisation) of each pixel's ray inside the unit cube."""
    tr = np.asarray(transform, np.float64).reshape(4, 3)
    ys, xs = np.mgrid[0:H, 0:W]
    d = np.stack([(xs - 0.5 * W) / focal, -(ys - 0.5 * H) / focal, -np.ones_like(xs, float)], -1)
    d = d @ tr[:3]          # rows of the column-major 4x3 are right / up / back
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    c = tr[3] * tree.invradius3.astype(np.float64) + tree.offset.astype(np.float64)
    d = d * tree.invradius3.astype(np.float64)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    inv = 1.0 / (d + 1e-9)
    t1, t2 = (0.0 - c) * inv, (1.0 - c) * inv
    tmin = np.maximum(np.minimum(t1, t2).max(-1), 0.0)
    tmax = np.maximum(t1, t2).min(-1)
    return np.maximum(tmax - tmin, 0.0)


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "C1"
    poses_idx = [int(a) for a in sys.argv[2:]] or [10, 60, 110]
    cfg = synth.CONFIGS[name]
    tree = synth.make_config_tree(name)
    th = ob.TreeHandle(tree)
    W, H, focal = cfg["width"], cfg["height"], cfg["focal"]
    poses = synth.make_poses(200)
    out = []
    for pi in poses_idx:
        tr = synth.c2w_to_transform(poses[pi])
        samples, hits, cnt = ob.render_maps(th, ob.make_camera(tr, W, H, focal), ob.default_options())
        s = samples[samples > 0].astype(np.float64)
        ch = chords(tr, W, H, focal, tree)[samples > 0]
        order = np.argsort(-s)
        top = order[: max(1, len(s) // 100)]          # the longest 1 % of the rays
        # how many of them a longest-chord-first order would have started in its first quarter
        ch_rank = np.argsort(np.argsort(-ch))
        rec = {"config": name, "pose": pi, "rays_in_volume": int(len(s)), "mean": round(float(s.mean()), 1),
               "p50": float(np.percentile(s, 50)), "p90": float(np.percentile(s, 90)),
               "p99": float(np.percentile(s, 99)), "p99.9": float(np.percentile(s, 99.9)),
               "max": float(s.max()),
               "corr_samples_vs_chord": round(float(np.corrcoef(s, ch)[0, 1]), 3),
               "top1pct_in_first_quarter_of_chord_order": round(float((ch_rank[top] < len(s) // 4).mean()), 3)}
        # block level (8x8 pixels, the unit a wave starts with): block max vs block mean chord
        Hb, Wb = H // 8, W // 8
        sm = samples[: Hb * 8, : Wb * 8].reshape(Hb, 8, Wb, 8).max((1, 3)).ravel().astype(float)
        cm = chords(tr, W, H, focal, tree)[: Hb * 8, : Wb * 8].reshape(Hb, 8, Wb, 8).mean((1, 3)).ravel()
        keep = sm > 0
        rec["corr_blockmax_vs_blockchord"] = round(float(np.corrcoef(sm[keep], cm[keep])[0, 1]), 3)
        out.append(rec)
        print(json.dumps(rec), flush=True)
    return out


if __name__ == "__main__":
    main()
