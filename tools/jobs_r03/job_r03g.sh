set -u
mkdir -p gpurun_out/r03g
O=gpurun_out/r03g
timeout 900 python tools/quick_ab.py --config C1 --variants base,sl1,sl2,sl8,sl16 --tunes "split=1,refill_min=12" --frames 64,20 --reps 3 --check --out $O/ab_c1.jsonl > $O/ab_c1.log 2>&1
timeout 900 python tools/quick_ab.py --config C1 --variants base --tunes "split=0" --frames 64,20 --reps 3 --check --out $O/ab_c1.jsonl >> $O/ab_c1.log 2>&1
timeout 900 python tools/quick_ab.py --config C3 --variants base,sl8 --tunes "split=1,refill_min=12;split=0,refill_min=24" --frames 16 --reps 3 --check --out $O/ab_c3.jsonl > $O/ab_c3.log 2>&1
cat $O/*.jsonl | python -c '
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d["config"], d["variant"], d["tune"], d["frames"], d["ms_per_frame_mean"], d["ms_per_frame_min"], d.get("same_as_first"), d.get("status"))'
tail -2 $O/ab_c1.log
