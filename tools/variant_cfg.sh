#!/bin/bash
# tools/variant_cfg.sh <tag> "<configs>" <variant> ...  : bench (64 frames / launch) of library variants on several configs
set -u
TAG=$1; CFGS=$2; shift 2
OUT=gpurun_out/vcfg_$TAG.jsonl; : > $OUT
for c in $CFGS; do for v in "$@"; do
  lib=""; [ "$v" != base ] && lib="$PWD/volrend_amd/libvolrend_hip_$v.so"
  VOLREND_HIP_LIB=$lib timeout 600 python bench.py --config $c --steps 128 --warmup 64 --no-cpu-baseline 2>/dev/null | python -c '
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(json.dumps({"config": sys.argv[2], "variant": sys.argv[1], "ms": d["ms_per_step"], "frac": d["roofline"]["frac"]}))' $v $c >> $OUT
done; done
cat $OUT
