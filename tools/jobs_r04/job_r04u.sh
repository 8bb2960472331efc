# r04u: wave priority -- march rounds (memory-issuing) above shade rounds (pm1 / pm3) or the other way round (ps1)
set -u
O=gpurun_out/r04u; mkdir -p $O; rm -f $O/*
V=base,pm1,ps1,pm3,base,pm1,ps1,pm3
timeout 900 python tools/quick_ab.py --config C1 --variants $V --tunes "" --frames 64,20,1 --reps 5 --rotate --check --out $O/ab_c1.jsonl > $O/ab_c1.log 2>&1
timeout 900 python tools/quick_ab.py --config C3 --variants base,pm1,ps1,pm3 --tunes "" --frames 16 --reps 4 --rotate --check --out $O/ab_c3.jsonl > $O/ab_c3.log 2>&1
cat $O/ab_c1.jsonl $O/ab_c3.jsonl | python -c '
import json,sys,collections
r=collections.OrderedDict()
for l in sys.stdin:
    d=json.loads(l); k=(d["config"], d["variant"], d["frames"]); r.setdefault(k,[]).append((d["ms_per_frame_mean"], d["ms_per_frame_min"], d.get("same_as_first")))
for k,v in r.items(): print(*k, " ".join("%.4f/%.4f"%(a,b) for a,b,_ in v), all(x[2] for x in v))'
