set -u
mkdir -p gpurun_out/r03a
O=gpurun_out/r03a
timeout 900 python tools/quick_ab.py --config C1 --variants base --tunes "waves_per_cu=12;waves_per_cu=16;waves_per_cu=20" --frames 64,20,1 --reps 3 --out $O/ab_base.jsonl > $O/ab_base.log 2>&1
timeout 900 python tools/quick_ab.py --config C1 --variants abl6 --tunes "waves_per_cu=12;waves_per_cu=16;waves_per_cu=20;waves_per_cu=24;waves_per_cu=32;waves_per_cu=40" --frames 64,20,1 --reps 3 --out $O/ab_abl6.jsonl > $O/ab_abl6.log 2>&1
timeout 900 python tools/quick_ab.py --config C3 --variants base,abl6 --tunes "waves_per_cu=16;waves_per_cu=24;waves_per_cu=32" --frames 16 --reps 3 --out $O/ab_c3.jsonl > $O/ab_c3.log 2>&1
cat $O/ab_base.jsonl $O/ab_abl6.jsonl $O/ab_c3.jsonl | python -c '
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d["config"], d["variant"], d["tune"], d["frames"], d["ms_per_frame_mean"], d["ms_per_frame_min"])'
tail -3 $O/ab_base.log
