#!/usr/bin/env python3
"""What a small launch is made of: rocprofv3 --kernel-trace of lone launches (prepare_launch_kernel,
raygen_kernel, render_kernel with their start / end timestamps) -> per launch the three kernel
durations and the two gaps between them, for 8x8 ... 800x800 one-frame launches and a few frames
per launch.  Run from the repo root on the GPU box:

    python tools/launch_anatomy.py --out gpurun_out/x/launch_anatomy.jsonl

(parent: runs rocprofv3 on itself with --child; child: the launches.)
"""
from __future__ import annotations

import argparse
import csv
import glob
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(args):
    import torch
    from volrend_amd import _abi, api, synth
    import bench as B
    if args.variant != "base":
        _abi._lib = None
        _abi.LIB_PATH = os.path.join(ROOT, "volrend_amd", f"libvolrend_hip_{args.variant}.so")
    cfg = synth.CONFIGS[args.config]
    stree = B.load_or_make_tree(synth, args.config, 0, lambda: None)
    poses = synth.make_poses(200)
    tree = api.N3Tree.from_synth(stree)
    if args.tune:
        tree.set_tuning(**{k: int(x) for k, x in (kv.split("=") for kv in args.tune.split(","))})
    stream = torch.cuda.current_stream()
    shapes = []
    for tok in args.shapes.split(","):
        size, nf = tok.split("x")
        shapes.append((int(size), int(nf)))
    warm_imgs = [torch.zeros((cfg["height"], cfg["width"], 4), dtype=torch.uint8, device="cuda") for _ in range(32)]
    wcam = api.Camera(cfg["width"], cfg["height"], cfg["focal"], cfg["focal"])
    warm = api.PreparedBatch(tree, wcam, [synth.c2w_to_transform(poses[100 + i]) for i in range(32)],
                             api.RenderOptions(), warm_imgs, True)
    for size, nf in shapes:
        f = cfg["focal"] * size / cfg["width"]
        cam = api.Camera(size, size, f, f)
        imgs = [torch.zeros((size, size, 4), dtype=torch.uint8, device="cuda") for _ in range(nf)]
        for rep in range(args.reps):
            pb = api.PreparedBatch(tree, cam, [synth.c2w_to_transform(poses[(10 + 9 * rep + i) % 200])
                                               for i in range(nf)], api.RenderOptions(), imgs, True)
            if rep == 0:
                pb.launch(stream)  # slot sizing
                torch.cuda.synchronize()
            warm.launch(stream)     # the GPU's clocks of a render loop; also the marker between launches
            pb.launch(stream)
            torch.cuda.synchronize()
    print("STATUS", tree.status())


def parent(args):
    tmp = tempfile.mkdtemp(prefix="anat_", dir="/tmp")
    cmd = ["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", tmp, "-o", "t", "--",
           sys.executable, os.path.abspath(__file__), "--child", "--config", args.config, "--shapes", args.shapes,
           "--reps", str(args.reps), "--variant", args.variant, "--tune", args.tune]
    p = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=args.timeout)
    files = glob.glob(os.path.join(tmp, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        print(p.stderr.decode(errors="replace")[-2000:], file=sys.stderr)
        raise SystemExit("no kernel trace")
    rows = []
    for row in csv.DictReader(open(files[0])):
        name = row.get("Kernel_Name", "")
        kind = ("prepare" if "prepare_launch" in name else "raygen" if "raygen_kernel" in name else
                "render" if "render_kernel" in name else None)
        if kind:
            rows.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), kind,
                         int(row.get("Grid_Size", 0) or row.get("Grid_Size_X", 0) or 0)))
    rows.sort()
    # launches = consecutive (prepare.., raygen, render) triples; the 32-frame warm launch in front of
    # every measured one is recognised by its raygen grid
    launches, cur = [], []
    for r in rows:
        cur.append(r)
        if r[2] == "render":
            launches.append(cur)
            cur = []
    shapes = []
    for tok in args.shapes.split(","):
        size, nf = tok.split("x")
        shapes.append((int(size), int(nf)))
    # order in the child: per shape: [sizing launch], then reps x (warm, measured)
    out = open(args.out, "a") if args.out else None
    idx = 0
    for size, nf in shapes:
        idx += 1  # sizing launch
        recs = []
        for rep in range(args.reps):
            warm, meas = launches[idx], launches[idx + 1]
            idx += 2
            prep = [r for r in meas if r[2] == "prepare"]
            gen = [r for r in meas if r[2] == "raygen"][0]
            ren = [r for r in meas if r[2] == "render"][0]
            recs.append({"after_warm_gap_us": (prep[0][0] - warm[-1][1]) / 1e3,
                         "prepare_us": sum(r[1] - r[0] for r in prep) / 1e3,
                         "gap_prepare_raygen_us": (gen[0] - prep[-1][1]) / 1e3,
                         "raygen_us": (gen[1] - gen[0]) / 1e3,
                         "gap_raygen_render_us": (ren[0] - gen[1]) / 1e3,
                         "render_us": (ren[1] - ren[0]) / 1e3,
                         "total_us": (ren[1] - prep[0][0]) / 1e3,
                         "render_grid_threads": ren[3]})
        keys = recs[0].keys()
        rec = {"config": args.config, "variant": args.variant, "tune": args.tune, "size": size, "frames": nf,
               "reps": args.reps,
               "median": {k: round(sorted(r[k] for r in recs)[len(recs) // 2], 2) for k in keys},
               "min": {k: round(min(r[k] for r in recs), 2) for k in keys}}
        print(json.dumps(rec), flush=True)
        if out:
            out.write(json.dumps(rec) + "\n")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--config", default="C1")
    ap.add_argument("--shapes", default="8x1,32x1,128x1,800x1,800x2,800x4,800x20")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--variant", default="base")
    ap.add_argument("--tune", default="")
    ap.add_argument("--timeout", type=int, default=600)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    (child if args.child else parent)(args)


if __name__ == "__main__":
    main()
