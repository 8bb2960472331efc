# r04e: the CU's vector L1 stalls on pending lines two cycles out of three (r04d).  Cache policy of the three streams
# of the fused kernel: records (DMA) sc0 / sc1 / sc0 sc1, lookups (top + brick words) sc1 / nt, ray words sc1 / nt, all sc1
set -u
O=gpurun_out/r04e; mkdir -p $O; rm -f $O/*
V=base,r16,r1,r17,l1,l2,y1,y2,a1
timeout 900 python tools/quick_ab.py --config C1 --variants $V,$V --tunes "split=0" --frames 64,20,1 --reps 4 --rotate --check --out $O/ab_c1.jsonl > $O/ab_c1.log 2>&1
timeout 900 python tools/quick_ab.py --config C3 --variants $V,base,a1 --tunes "split=0;split=0,records_nt=0" --frames 16 --reps 4 --rotate --check --out $O/ab_c3.jsonl > $O/ab_c3.log 2>&1
timeout 900 python tools/quick_ab.py --config C2 --variants base,r16,l1,a1,base,r16,l1,a1 --tunes "split=0" --frames 8 --reps 3 --rotate --check --out $O/ab_c2.jsonl > $O/ab_c2.log 2>&1
for f in ab_c1 ab_c3 ab_c2; do cat $O/$f.jsonl | python -c '
import json,sys,collections
r=collections.OrderedDict()
for l in sys.stdin:
    d=json.loads(l); k=(d["config"], d["variant"], d["tune"], d["frames"]); r.setdefault(k,[]).append((d["ms_per_frame_mean"], d["ms_per_frame_min"], d.get("same_as_first")))
for k,v in r.items(): print(*k, " ".join("%.4f/%.4f"%(a,b) for a,b,_ in v), all(x[2] for x in v))'; done
