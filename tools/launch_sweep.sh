#!/bin/bash
# Frames-per-launch sweep of the C1 bench: ms/frame at 1/4/16/20/64 poses
# per vr_render_batch launch.   tools/launch_sweep.sh <tag>   -> gpurun_out/sweep_<tag>.jsonl
set -u
TAG=${1:-run}
OUT=gpurun_out/sweep_$TAG.jsonl
: > $OUT
for spec in "1 64 8" "4 128 16" "16 256 32" "20 20 5" "64 256 64"; do
  set -- $spec
  timeout 300 python bench.py --batch $1 --steps $2 --warmup $3 --no-cpu-baseline ${SWEEP_ARGS:-} \
      2>> gpurun_out/sweep_$TAG.log | python -c '
import json,sys
for l in sys.stdin:
    d=json.loads(l)
    print(json.dumps({"frames_per_launch": d["config"]["frames_per_launch"], "steps": d["steps"],
                      "ms_per_frame": d["ms_per_step"], "fps": d["fps"], "mrays": d["value"],
                      "kernel_ms_per_frame": d["roofline"]["kernel_ms_per_frame"],
                      "roofline_frac": d["roofline"]["frac"]}))' >> $OUT
done
cat $OUT
