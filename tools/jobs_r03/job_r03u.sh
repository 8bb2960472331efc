set -u
mkdir -p gpurun_out/r03u
O=gpurun_out/r03u
timeout 600 python -m pytest tests/test_gpu_chain.py -x -q --timeout 300 > $O/pytest_chain.log 2>&1; echo "chain rc=$?"; tail -2 $O/pytest_chain.log
timeout 900 python tools/quick_ab.py --config C1 --variants base,flat,base,flat --tunes "split=1" --frames 64,20,4,2,1 --reps 6 --rotate --check --out $O/ab_c1.jsonl > $O/ab_c1.log 2>&1
timeout 900 python tools/quick_ab.py --config C1 --variants base --tunes "split=0" --frames 64,20,4,2,1 --reps 6 --rotate --check --out $O/ab_c1.jsonl >> $O/ab_c1.log 2>&1
timeout 900 python tools/quick_ab.py --config C3 --variants base,flat --tunes "split=1;split=0" --frames 16,1 --reps 4 --rotate --check --out $O/ab_c3.jsonl > $O/ab_c3.log 2>&1
cat $O/*.jsonl | python -c '
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d["config"], d["variant"], d["tune"], d["frames"], d["ms_per_frame_mean"], d["ms_per_frame_min"], d.get("same_as_first"), d.get("status"))'
