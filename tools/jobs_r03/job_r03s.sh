set -u
mkdir -p gpurun_out/r03s
O=gpurun_out/r03s
timeout 900 python tools/quick_ab.py --config C1 --variants base --tunes "split=0,chunk_max=4096,refill_min=24;split=0,chunk_max=1024;split=0,chunk_max=256;split=0,chunk_max=4096,xcd_queues=0;split=0,xcd_queues=1,refill_min=16;split=0,refill_min=32;split=0,refill_min=24,march_max=8;split=0,march_max=32;split=0,march_max=16,waves_per_cu=18;split=0,waves_per_cu=20" --frames 20 --reps 6 --rotate --out $O/ab_c1_20.jsonl > $O/ab.log 2>&1
cat $O/*.jsonl | python -c '
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d["config"], d["tune"], d["frames"], d["ms_per_frame_mean"], d["ms_per_frame_min"], d["launch_ms"])'
