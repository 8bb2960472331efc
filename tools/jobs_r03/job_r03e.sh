set -u
mkdir -p gpurun_out/r03e
O=gpurun_out/r03e
VR_TIMELINE=1 timeout 600 python tools/quick_ab.py --config C1 --variants tl2 --tunes "split=1;split=1,refill_min=8" --frames 64 --reps 2 --out $O/tl2.jsonl > $O/tl2.log 2>&1
cat $O/*.jsonl | python -c '
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d["config"], d["variant"], d["tune"], d["frames"], d["ms_per_frame_mean"], d["ms_per_frame_min"], d.get("sched_stats"))'
tail -3 $O/tl2.log
