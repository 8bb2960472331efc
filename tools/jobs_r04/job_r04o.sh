# r04o: refill threshold x march steps per pass again, with the 19-word ray records (a refill reads 40 % less)
set -u
O=gpurun_out/r04o; mkdir -p $O; rm -f $O/*
T="refill_min=24,march_max=16;refill_min=16,march_max=16;refill_min=16,march_max=8;refill_min=12,march_max=8;refill_min=20,march_max=12;refill_min=32,march_max=16;refill_min=24,march_max=16"
timeout 900 python tools/quick_ab.py --config C1 --variants base,base --tunes "$T" --frames 64,20 --reps 5 --rotate --out $O/ab_c1.jsonl > $O/ab_c1.log 2>&1
timeout 900 python tools/quick_ab.py --config C3 --variants base --tunes "$T" --frames 16 --reps 4 --rotate --out $O/ab_c3.jsonl > $O/ab_c3.log 2>&1
cat $O/ab_c1.jsonl $O/ab_c3.jsonl | python -c '
import json,sys,collections
r=collections.OrderedDict()
for l in sys.stdin:
    d=json.loads(l); k=(d["config"], d["tune"], d["frames"]); r.setdefault(k,[]).append((d["ms_per_frame_mean"], d["ms_per_frame_min"]))
for k,v in r.items(): print(*k, " ".join("%.4f/%.4f"%(a,b) for a,b in v))'
