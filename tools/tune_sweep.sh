#!/bin/bash
# Scheduling-knob sweep on the C1 bench: tools/tune_sweep.sh <tag> "k=v,k=v" "k=v" ...
set -u
TAG=$1; shift
OUT=gpurun_out/tune_$TAG.jsonl
: > $OUT
for t in "$@"; do
  timeout 300 python bench.py --no-cpu-baseline --tune "$t" ${SWEEP_ARGS:-} 2>> gpurun_out/tune_$TAG.log | python -c '
import json,sys
t=sys.argv[1]
for l in sys.stdin:
    d=json.loads(l)
    print(json.dumps({"tune": t, "ms_per_frame": d["ms_per_step"], "fps": d["fps"]}))' "$t" >> $OUT
done
cat $OUT
