#!/bin/bash
# A/B of library builds, interleaved to average out box drift:  tools/ab.sh <tag> <reps> <variant> ...
set -u
TAG=$1; REPS=$2; shift 2
OUT=gpurun_out/ab_$TAG.jsonl; : > $OUT
for rep in $(seq $REPS); do for v in "$@"; do
  lib=""; [ "$v" != base ] && lib="$PWD/volrend_amd/libvolrend_hip_$v.so"
  for spec in "256 64" "20 5"; do set -- $spec $@; s=$1; w=$2; shift 2
    VOLREND_HIP_LIB=$lib timeout 300 python bench.py --steps $s --warmup $w --no-cpu-baseline ${SWEEP_ARGS:-} 2>/dev/null | python -c '
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(json.dumps({"variant": sys.argv[1], "steps": d["steps"], "ms": d["ms_per_step"], "kernel_ms": d["roofline"]["kernel_ms_per_frame"]}))' $v >> $OUT
  done
done; done
python - <<PY
import json,collections
r=collections.defaultdict(list)
for l in open("$OUT"):
    d=json.loads(l); r[(d["variant"],d["steps"])].append(d["ms"])
for k,v in sorted(r.items()): print(k, " ".join(f"{x:.4f}" for x in v), " mean %.4f min %.4f"%(sum(v)/len(v), min(v)))
PY
