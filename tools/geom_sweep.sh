#!/bin/bash
# Lookup-structure geometry sweep: ms/frame for (top_levels, brick_levels) pairs.
#   tools/geom_sweep.sh <tag> <config> "<g0,bl> ..."   -> gpurun_out/geom_<tag>.jsonl
set -u
TAG=$1; CFG=$2; shift 2
OUT=gpurun_out/geom_$TAG.jsonl
: > $OUT
for pair in $*; do
  g0=${pair%,*}; bl=${pair#*,}
  VR_TOP_LEVELS=$g0 VR_BRICK_LEVELS=$bl timeout 300 python bench.py --config $CFG --no-cpu-baseline ${SWEEP_ARGS:-} \
      2>> gpurun_out/geom_$TAG.log | python -c '
import json,sys
g0,bl=sys.argv[1:3]
for l in sys.stdin:
    d=json.loads(l)
    print(json.dumps({"top_levels": int(g0), "brick_levels": int(bl), "ms_per_frame": d["ms_per_step"], "fps": d["fps"],
                      "kernel_ms_per_frame": d["roofline"]["kernel_ms_per_frame"], "frac": d["roofline"]["frac"]}))' $g0 $bl >> $OUT
done
cat $OUT
