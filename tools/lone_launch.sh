#!/bin/bash
# What a lone launch costs, as the driver times it: K frames in ONE timed launch (bench.py --steps K
# --warmup W --batch 128), at different pose offsets and sizes.
set -u
OUT=gpurun_out/lone_${1:-x}.jsonl; : > $OUT
for spec in "20 5" "20 5" "20 105" "64 5" "40 5" "10 5" "4 5" "1 5" "1 5"; do set -- $spec
  timeout 300 python bench.py --steps $1 --warmup $2 --batch 128 --no-cpu-baseline --no-parity --live-traffic 0 --cold 0 --repeats 0 2>/dev/null | python -c '
import json,sys
for l in sys.stdin:
    d=json.loads(l); r=d["roofline"]
    print(json.dumps({"steps": d["steps"], "warmup": d["warmup"], "ms_per_frame": d["ms_per_step"], "launch_ms": round(r["kernel_ms_mean"],3), "frac": r["frac"], "kernel": r["kernel"].split("<")[0], "samples_per_ray": r["samples_per_ray"], "hits_per_ray": r["hit_samples_per_ray"]}))' >> $OUT
done
cat $OUT
