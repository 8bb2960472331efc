#!/usr/bin/env python3
"""Kernel trace of small launches on alternating streams: WHEN do the three kernels of launch
k + 1 (prepare, ray generation, render) run relative to the render kernel of launch k?

rocprofv3 --kernel-trace of tools/stream_overlap.py (one row of it: F frames per launch, S streams),
then per launch of the last timed repetition: start / end of each kernel relative to the first
launch, and how much of the launch's render kernel ran beside its predecessor's.

    python tools/overlap_trace.py --frames 1 --streams 2 [--launches 24] [--variant base] --out x.jsonl
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C1")
    ap.add_argument("--frames", type=int, default=1)
    ap.add_argument("--streams", type=int, default=2)
    ap.add_argument("--launches", type=int, default=24)
    ap.add_argument("--variant", default="base")
    ap.add_argument("--tune", default="")
    ap.add_argument("--timeout", type=int, default=600)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    tmp = tempfile.mkdtemp(prefix="ovl_", dir="/tmp")
    cmd = ["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", tmp, "-o", "t", "--",
           sys.executable, os.path.join(ROOT, "tools", "stream_overlap.py"), "--config", args.config,
           "--frames", str(args.frames), "--streams", str(args.streams), "--launches", str(args.launches),
           "--reps", "1", "--variant", args.variant, "--tune", args.tune]
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
            rows.append({"kind": kind, "start": int(row["Start_Timestamp"]), "end": int(row["End_Timestamp"]),
                         "queue": row.get("Queue_Id", ""), "grid": int(row.get("Grid_Size", 0) or 0)})
    # a launch = prepare.., raygen, render in ONE queue (stream); per queue in start order
    by_q = {}
    for r in sorted(rows, key=lambda r: r["start"]):
        by_q.setdefault(r["queue"], []).append(r)
    launches = []
    for q, rs in by_q.items():
        cur = []
        for r in rs:
            cur.append(r)
            if r["kind"] == "render":
                launches.append({"queue": q, "prepare": [x for x in cur if x["kind"] == "prepare"],
                                 "raygen": [x for x in cur if x["kind"] == "raygen"][-1], "render": r})
                cur = []
    launches.sort(key=lambda L: L["render"]["start"])
    # the timed repetition = the last `launches` launches whose raygen grid is the F-frame one
    small = [L for L in launches if L["raygen"]["grid"] == launches[-1]["raygen"]["grid"]][-args.launches:]
    t0 = small[0]["prepare"][0]["start"] if small[0]["prepare"] else small[0]["raygen"]["start"]
    out = open(args.out, "a") if args.out else None
    recs = []
    for i, L in enumerate(small):
        prev = small[i - 1] if i else None
        rec = {"i": i, "queue": L["queue"],
               "prepare_start_us": round((L["prepare"][0]["start"] - t0) / 1e3, 1) if L["prepare"] else None,
               "raygen_start_us": round((L["raygen"]["start"] - t0) / 1e3, 1),
               "raygen_us": round((L["raygen"]["end"] - L["raygen"]["start"]) / 1e3, 1),
               "render_start_us": round((L["render"]["start"] - t0) / 1e3, 1),
               "render_end_us": round((L["render"]["end"] - t0) / 1e3, 1),
               "render_us": round((L["render"]["end"] - L["render"]["start"]) / 1e3, 1)}
        if prev:
            rec["raygen_start_before_prev_render_end_us"] = round(
                (prev["render"]["end"] - L["raygen"]["start"]) / 1e3, 1)
            rec["render_start_before_prev_render_end_us"] = round(
                (prev["render"]["end"] - L["render"]["start"]) / 1e3, 1)
            rec["end_to_end_us"] = round((L["render"]["end"] - prev["render"]["end"]) / 1e3, 1)
        recs.append(rec)
    mid = recs[4:-2] if len(recs) > 8 else recs[1:]

    def med(k):
        v = sorted(r[k] for r in mid if r.get(k) is not None)
        return v[len(v) // 2] if v else None

    summary = {"config": args.config, "variant": args.variant, "tune": args.tune, "frames_per_launch": args.frames,
               "streams": args.streams, "launches": len(small),
               "sustained_ms_per_frame": round((small[-1]["render"]["end"] - small[2]["render"]["end"]) / 1e6 /
                                               ((len(small) - 3) * args.frames), 5),
               "median": {k: med(k) for k in ("raygen_us", "render_us", "raygen_start_before_prev_render_end_us",
                                              "render_start_before_prev_render_end_us", "end_to_end_us")},
               "timeline": recs[4:12]}
    print(json.dumps(summary), flush=True)
    if out:
        out.write(json.dumps(summary) + "\n")


if __name__ == "__main__":
    main()
