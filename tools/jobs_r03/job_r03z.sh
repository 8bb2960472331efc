# ray exchange (drain-phase consolidation) in the fused kernel: parity, then launch shapes against the old library
set -u
mkdir -p gpurun_out/r03z
O=gpurun_out/r03z
rm -f $O/*
timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_parity.py -x -q --timeout 600 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 900 python tools/quick_ab.py --config C1 --variants old --tunes "split=0" --frames 1,2,4,20,64 --reps 8 --rotate --out $O/ab_c1.jsonl > $O/ab_c1_old.log 2>&1
timeout 900 python tools/quick_ab.py --config C1 --variants base --tunes "split=0,consolidate=0;split=0,consolidate=16;split=0,consolidate=24;split=0,consolidate=32" --frames 1,2,4,20,64 --reps 8 --rotate --check --out $O/ab_c1.jsonl > $O/ab_c1.log 2>&1
timeout 900 python tools/quick_ab.py --config C1 --variants old --tunes "split=0" --frames 1,2,4,20,64 --reps 8 --rotate --out $O/ab_c1.jsonl >> $O/ab_c1_old.log 2>&1
cat $O/ab_c1.jsonl | python -c '
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d["config"], d["variant"], d["tune"], d["frames"], d["ms_per_frame_mean"], d["ms_per_frame_min"], d.get("same_as_first"), d.get("status"))'
tail -3 $O/ab_c1.log
