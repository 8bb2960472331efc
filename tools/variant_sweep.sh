#!/bin/bash
# Experiment builds side by side: tools/variant_sweep.sh <tag> <variant> ...   ("base" = product library)
#   (python -m volrend_amd.build --variant NAME -D... builds volrend_amd/libvolrend_hip_NAME.so)
# Per variant: the C1 bench at 64 frames per launch and with the driver's flags (--steps 20 --warmup 5).
set -u
TAG=$1; shift
OUT=gpurun_out/variants_$TAG.jsonl
: > $OUT
for v in "$@"; do
  lib=""; [ "$v" != base ] && lib="$PWD/volrend_amd/libvolrend_hip_$v.so"
  for spec in "256 64" "20 5"; do
    set -- $spec
    VR_TIMELINE=$([ "$v" = tl ] && echo 1) VOLREND_HIP_LIB=$lib timeout 300 python bench.py --steps $1 --warmup $2 --no-cpu-baseline ${SWEEP_ARGS:-} \
        2>> gpurun_out/variants_$TAG.log | python -c '
import json,sys
v=sys.argv[1]
for l in sys.stdin:
    d=json.loads(l)
    print(json.dumps({"variant": v, "steps": d["steps"], "ms_per_frame": d["ms_per_step"], "fps": d["fps"],
                      "kernel_ms_per_frame": d["roofline"]["kernel_ms_per_frame"]}))' $v >> $OUT
  done
done
cat $OUT
grep -h timeline gpurun_out/variants_$TAG.log
