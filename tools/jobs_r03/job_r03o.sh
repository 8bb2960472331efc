set -u
mkdir -p gpurun_out/r03o
O=gpurun_out/r03o
timeout 900 python tools/quick_ab.py --config C2 --variants base,sh25w5 --tunes "split=0" --frames 16,1 --reps 3 --rotate --check --out $O/ab_c2.jsonl > $O/ab_c2.log 2>&1
cat $O/*.jsonl | python -c '
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d["config"], d["variant"], d["tune"], d["frames"], d["ms_per_frame_mean"], d["ms_per_frame_min"], d.get("same_as_first"), d.get("status"))'
