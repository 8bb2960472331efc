# r04c: what does an instruction cost?  Round-3 fused kernel with n extra no-ops per march round (vector: mv10 / mv20,
# scalar: ms10 / ms20) or per shade round (vector: sv40); plus strict vs fma (48 fewer vector instructions per shade round)
set -u
O=gpurun_out/r04c; mkdir -p $O; rm -f $O/*
V=base,mv10,mv20,ms10,ms20,sv40,base,mv10,mv20,ms10,ms20,sv40
timeout 900 python tools/quick_ab.py --config C1 --variants $V --tunes "split=0" --frames 64,1 --reps 5 --rotate --out $O/ab_c1.jsonl > $O/ab_c1.log 2>&1
timeout 900 python tools/quick_ab.py --config C1 --variants base,base --fp fma --tunes "split=0" --frames 64,1 --reps 5 --rotate --out $O/ab_c1_fma.jsonl > $O/ab_c1_fma.log 2>&1
timeout 900 python tools/quick_ab.py --config C3 --variants base,mv20,ms20,sv40,base,mv20,ms20,sv40 --tunes "split=0" --frames 16 --reps 4 --rotate --out $O/ab_c3.jsonl > $O/ab_c3.log 2>&1
for f in ab_c1 ab_c1_fma ab_c3; do echo $f; cat $O/$f.jsonl | python -c '
import json,sys,collections
r=collections.OrderedDict()
for l in sys.stdin:
    d=json.loads(l); k=(d["config"], d["variant"], d["tune"], d["frames"]); r.setdefault(k,[]).append((d["ms_per_frame_mean"], d["ms_per_frame_min"]))
for k,v in r.items(): print(*k, " ".join("%.4f/%.4f"%(a,b) for a,b in v))'; done
