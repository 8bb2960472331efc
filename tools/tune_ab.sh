#!/bin/bash
# Interleaved A/B of tuning strings on C1 (256 steps and the driver's 20):  tools/tune_ab.sh <tag> <reps> "k=v,.." ...
set -u
TAG=$1; REPS=$2; shift 2
OUT=gpurun_out/tuneab_$TAG.jsonl; : > $OUT
for rep in $(seq $REPS); do for t in "$@"; do for spec in "256 64" "20 5"; do set -- $spec "$@"; s=$1; w=$2; shift 2
  timeout 300 python bench.py --steps $s --warmup $w --no-cpu-baseline --tune "$t" 2>/dev/null | python -c '
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(json.dumps({"tune": sys.argv[1], "steps": d["steps"], "ms": d["ms_per_step"]}))' "$t" >> $OUT
done; done; done
python - <<PY
import json,collections
r=collections.defaultdict(list)
for l in open("$OUT"):
    d=json.loads(l); r[(d["tune"],d["steps"])].append(d["ms"])
for k,v in sorted(r.items()): print(k, " ".join(f"{x:.4f}" for x in v), " mean %.4f"%(sum(v)/len(v)))
PY
