# r04b: pipelined fused kernel with the two-step lookup (no repeated rounds) against round 3 ("old"), the
# one-word-per-round variant ("retry") and three diagnostic builds that serialise the requests again
set -u
O=gpurun_out/r04b; mkdir -p $O; rm -f $O/*
timeout 600 python -m pytest tests/test_gpu_chain.py tests/test_gpu_parity.py -x -q --timeout 600 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
V=old,base,retry,ser1,ser2,ser3,old,base,retry,ser1,ser2,ser3
timeout 900 python tools/quick_ab.py --config C1 --variants $V --tunes "split=0;split=0,refill_min=12;split=0,refill_min=16" --frames 64,20,1 --reps 4 --rotate --check --out $O/ab_c1.jsonl > $O/ab_c1.log 2>&1
timeout 900 python tools/quick_ab.py --config C3 --variants old,base,retry,ser3,old,base --tunes "split=0;split=0,refill_min=16" --frames 16 --reps 4 --rotate --check --out $O/ab_c3.jsonl > $O/ab_c3.log 2>&1
timeout 900 python tools/quick_ab.py --config C2 --variants old,base,old,base --tunes "split=0;split=0,refill_min=16" --frames 8 --reps 3 --rotate --check --out $O/ab_c2.jsonl > $O/ab_c2.log 2>&1
cat $O/ab_c1.jsonl $O/ab_c3.jsonl $O/ab_c2.jsonl | python -c '
import json,sys,collections
r=collections.OrderedDict()
for l in sys.stdin:
    d=json.loads(l); k=(d["config"], d["variant"], d["tune"], d["frames"]); r.setdefault(k,[]).append((d["ms_per_frame_mean"], d["ms_per_frame_min"], d.get("same_as_first"), d.get("status")))
for k,v in r.items(): print(*k, " ".join("%.4f/%.4f"%(a,b) for a,b,_,_ in v), all(x[2] for x in v), max(x[3] for x in v))'
