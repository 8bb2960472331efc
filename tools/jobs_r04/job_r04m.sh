# r04m: ray records of SH trees carry the view direction (19 words) instead of the basis values (up to 41): base,
# against vd0 = records with the basis values.  Lone launches (driver shape) via bench.py, steady state via quick_ab
set -u
O=gpurun_out/r04m; mkdir -p $O; rm -f $O/*
timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_parity.py tests/test_gpu_probe.py tests/test_gpu_fullsize.py -x -q --timeout 800 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
V=vd0,base,vd0,base,vd0,base
timeout 900 python tools/quick_ab.py --config C1 --variants $V --tunes "" --frames 64,20,1 --reps 5 --rotate --check --out $O/ab_c1.jsonl > $O/ab_c1.log 2>&1
timeout 900 python tools/quick_ab.py --config C3 --variants vd0,base,vd0,base --tunes "" --frames 16 --reps 4 --rotate --check --out $O/ab_c3.jsonl > $O/ab_c3.log 2>&1
timeout 900 python tools/quick_ab.py --config C2 --variants vd0,base,vd0,base --tunes "" --frames 8 --reps 3 --rotate --check --out $O/ab_c2.jsonl > $O/ab_c2.log 2>&1
for f in ab_c1 ab_c3 ab_c2; do cat $O/$f.jsonl | python -c '
import json,sys,collections
r=collections.OrderedDict()
for l in sys.stdin:
    d=json.loads(l); k=(d["config"], d["variant"], d["tune"], d["frames"]); r.setdefault(k,[]).append((d["ms_per_frame_mean"], d["ms_per_frame_min"], d.get("same_as_first")))
for k,v in r.items(): print(*k, " ".join("%.4f/%.4f"%(a,b) for a,b,_ in v), all(x[2] for x in v))'; done
