set -u
mkdir -p gpurun_out/r03n
O=gpurun_out/r03n
timeout 900 python tools/quick_ab.py --config C3 --variants base,sh9w7 --tunes "split=0" --frames 16,1 --reps 4 --rotate --check --out $O/ab_c3.jsonl > $O/ab_c3.log 2>&1
timeout 900 python tools/quick_ab.py --config C1 --variants base,sh16w6,sh16w6r56 --tunes "split=0" --frames 64,20 --reps 4 --rotate --check --out $O/ab_c1.jsonl > $O/ab_c1.log 2>&1
cat $O/*.jsonl | python -c '
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d["config"], d["variant"], d["tune"], d["frames"], d["ms_per_frame_mean"], d["ms_per_frame_min"], d.get("same_as_first"), d.get("status"))'
