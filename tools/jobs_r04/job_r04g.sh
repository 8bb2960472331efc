# r04g: 16 bytes of padding between the row groups of the record stage (LDS bank conflicts are half of the LDS cycles, r04d)
set -u
O=gpurun_out/r04g; mkdir -p $O; rm -f $O/*
V=pk0,pad16,pk0,pad16,pk0,pad16
timeout 900 python tools/quick_ab.py --config C1 --variants $V --tunes "split=0" --frames 64,1 --reps 5 --rotate --check --out $O/ab_c1.jsonl > $O/ab_c1.log 2>&1
timeout 900 python tools/quick_ab.py --config C2 --variants pk0,pad16,pk0,pad16 --tunes "split=0" --frames 8 --reps 3 --rotate --check --out $O/ab_c2.jsonl > $O/ab_c2.log 2>&1
for f in ab_c1 ab_c2; do cat $O/$f.jsonl | python -c '
import json,sys,collections
r=collections.OrderedDict()
for l in sys.stdin:
    d=json.loads(l); k=(d["config"], d["variant"], d["tune"], d["frames"]); r.setdefault(k,[]).append((d["ms_per_frame_mean"], d["ms_per_frame_min"], d.get("same_as_first")))
for k,v in r.items(): print(*k, " ".join("%.4f/%.4f"%(a,b) for a,b,_ in v), all(x[2] for x in v))'; done
