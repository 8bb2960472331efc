# refill threshold / queue layout at small launches (library with the 8192-ray steal threshold)
set -u
mkdir -p gpurun_out/r03y
O=gpurun_out/r03y
rm -f $O/*
timeout 900 python tools/quick_ab.py --config C1 --variants base --tunes "split=0,refill_min=24;split=0,refill_min=16;split=0,refill_min=12;split=0,refill_min=8;split=0,refill_min=24,xcd_queues=0;split=0,refill_min=24,xcd_queues=1,march_max=8;split=1,refill_min_split=12,march_max=16;split=1,refill_min_split=8;split=1,refill_min_split=20;split=1,refill_min_split=12,march_max=8" --frames 1,2,4 --reps 8 --rotate --check --out $O/ab_c1.jsonl > $O/ab_c1.log 2>&1
cat $O/ab_c1.jsonl | python -c '
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d["config"], d["variant"], d["tune"], d["frames"], d["ms_per_frame_mean"], d["ms_per_frame_min"], d.get("same_as_first"), d.get("status"))'
