# fused kernel: "alive" and "stopped" derived from t / tmax instead of loop-carried booleans (18 scalar
# instructions per march round fewer, 3 vector ones more): parity, then A/B against the library before
set -u
mkdir -p gpurun_out/r03ab
O=gpurun_out/r03ab
rm -f $O/*
timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_parity.py tests/test_gpu_probe.py -x -q --timeout 600 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
timeout 900 python tools/quick_ab.py --config C1 --variants old,base,old,base,old,base --tunes "split=0" --frames 64,20,1 --reps 6 --rotate --check --out $O/ab_c1.jsonl > $O/ab_c1.log 2>&1
timeout 900 python tools/quick_ab.py --config C3 --variants old,base,old,base --tunes "split=0" --frames 16 --reps 4 --rotate --check --out $O/ab_c3.jsonl > $O/ab_c3.log 2>&1
timeout 900 python tools/quick_ab.py --config C2 --variants old,base,old,base --tunes "split=0" --frames 8 --reps 3 --rotate --check --out $O/ab_c2.jsonl > $O/ab_c2.log 2>&1
cat $O/ab_c1.jsonl $O/ab_c3.jsonl $O/ab_c2.jsonl | python -c '
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d["config"], d["variant"], d["tune"], d["frames"], d["ms_per_frame_mean"], d["ms_per_frame_min"], d.get("same_as_first"), d.get("status"))'
