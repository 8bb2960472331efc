set -u
mkdir -p gpurun_out/r03c
O=gpurun_out/r03c
VR_TIMELINE=1 timeout 600 python tools/quick_ab.py --config C1 --variants tl --tunes "split=1;split=0" --frames 64 --reps 2 --out $O/tl_c1.jsonl > $O/tl_c1.log 2>&1
VR_TIMELINE=1 timeout 600 python tools/quick_ab.py --config C3 --variants tl --tunes "split=1" --frames 16 --reps 2 --out $O/tl_c3.jsonl > $O/tl_c3.log 2>&1
timeout 900 python tools/quick_ab.py --config C1 --variants base,mc32,mc16,abl7 --tunes "split=1" --frames 64,1 --reps 3 --out $O/ab_var.jsonl > $O/ab_var.log 2>&1
timeout 900 python tools/quick_ab.py --config C1 --variants base --tunes "split=1,waves_per_cu=24;split=1,waves_per_cu=28;split=1,waves_per_cu=32;split=1,waves_per_cu=32,refill_min=8;split=1,waves_per_cu=32,refill_min=16;split=1,waves_per_cu=32,refill_min=16,march_max=64" --frames 64 --reps 3 --out $O/ab_tune.jsonl > $O/ab_tune.log 2>&1
cat $O/*.jsonl | python -c '
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d["config"], d["variant"], d["tune"], d["frames"], d["ms_per_frame_mean"], d["ms_per_frame_min"], d.get("sched_stats"))'
tail -3 $O/ab_var.log
