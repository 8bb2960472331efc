# steal threshold (rays a foreign queue must still hold): old library vs 2048 / 8192 (base) / 32768
set -u
mkdir -p gpurun_out/r03x
O=gpurun_out/r03x
rm -f $O/*
timeout 600 python tools/tail_profile.py --frames 1 --tunes "split=0" --out $O/tail_profile_s8k.jsonl > $O/tail_s8k.log 2>&1
timeout 900 python tools/quick_ab.py --config C1 --variants old,s2k,base,s32k,old,s2k,base,s32k --tunes "split=1;split=0" --frames 1,2,4,20,64 --reps 8 --rotate --check --out $O/ab_c1.jsonl > $O/ab_c1.log 2>&1
timeout 900 python tools/quick_ab.py --config C3 --variants old,base,s32k,old,base,s32k --tunes "split=1;split=0" --frames 1,16 --reps 4 --rotate --check --out $O/ab_c3.jsonl > $O/ab_c3.log 2>&1
cat $O/ab_c1.jsonl $O/ab_c3.jsonl | python -c '
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d["config"], d["variant"], d["tune"], d["frames"], d["ms_per_frame_mean"], d["ms_per_frame_min"], d.get("same_as_first"), d.get("status"))'
grep "bucket  [89] \|bucket 1[0-9] \|launch_ms" $O/tail_s8k.log
