# r04n: where the time goes now -- timing ablations of the current kernel (wrong pictures on purpose): abl4 = no record
# fetch, abl5 = every record out of a 128 KB window (always cached), abl6 = no colour work at all (march only);
# waves per CU for the march-only build and the product
set -u
O=gpurun_out/r04n; mkdir -p $O; rm -f $O/*
timeout 900 python tools/quick_ab.py --config C1 --variants base,abl4,abl5,abl6,base,abl4,abl5,abl6 --tunes "" --frames 64 --reps 5 --rotate --out $O/abl_c1.jsonl > $O/abl_c1.log 2>&1
timeout 900 python tools/quick_ab.py --config C1 --variants base,abl6 --tunes "waves_per_cu=12;waves_per_cu=16;waves_per_cu=20" --frames 64 --reps 4 --rotate --out $O/waves_c1.jsonl > $O/waves_c1.log 2>&1
timeout 900 python tools/quick_ab.py --config C3 --variants base,abl4,abl5,abl6 --tunes "" --frames 16 --reps 4 --rotate --out $O/abl_c3.jsonl > $O/abl_c3.log 2>&1
cat $O/abl_c1.jsonl $O/waves_c1.jsonl $O/abl_c3.jsonl | python -c '
import json,sys,collections
r=collections.OrderedDict()
for l in sys.stdin:
    d=json.loads(l); k=(d["config"], d["variant"], d["tune"], d["frames"]); r.setdefault(k,[]).append((d["ms_per_frame_mean"], d["ms_per_frame_min"]))
for k,v in r.items(): print(*k, " ".join("%.4f/%.4f"%(a,b) for a,b in v))'
