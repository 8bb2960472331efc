#!/bin/bash
# tools/tune_cfg.sh <tag> <config> "k=v,..." ... : tuning sweep on a config (128 steps)
set -u
TAG=$1; CFG=$2; shift 2
OUT=gpurun_out/tunec_$TAG.jsonl; : > $OUT
for t in "$@"; do
  timeout 600 python bench.py --config $CFG --steps 128 --warmup 64 --no-cpu-baseline --tune "$t" 2>/dev/null | python -c '
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(json.dumps({"tune": sys.argv[1], "ms": d["ms_per_step"], "frac": d["roofline"]["frac"]}))' "$t" >> $OUT
done
cat $OUT
