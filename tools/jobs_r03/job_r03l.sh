set -u
mkdir -p gpurun_out/r03l
O=gpurun_out/r03l
VR_FLUSH_WAIT=12 timeout 600 python -m pytest tests/test_gpu_chain.py tests/test_gpu_parity.py -x -q --timeout 300 > $O/pytest_flush.log 2>&1; echo "flush pytest rc=$?"; tail -3 $O/pytest_flush.log
timeout 900 python tools/quick_ab.py --config C1 --variants base --tunes "split=0,flush_wait=0;split=0,flush_wait=8;split=0,flush_wait=16;split=0,flush_wait=24;split=0,flush_wait=32;split=0,flush_wait=16,refill_min=12;split=0,flush_wait=24,refill_min=16" --frames 64,20,1 --reps 3 --check --out $O/ab_c1.jsonl > $O/ab_c1.log 2>&1
timeout 900 python tools/quick_ab.py --config C3 --variants base --tunes "split=0,flush_wait=0;split=0,flush_wait=16;split=0,flush_wait=32" --frames 16 --reps 3 --check --out $O/ab_c3.jsonl > $O/ab_c3.log 2>&1
timeout 900 python tools/quick_ab.py --config C2 --variants base --tunes "split=0,flush_wait=0;split=0,flush_wait=16;split=0,flush_wait=32" --frames 8 --reps 2 --check --out $O/ab_c2.jsonl > $O/ab_c2.log 2>&1
cat $O/*.jsonl | python -c '
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d["config"], d["variant"], d["tune"], d["frames"], d["ms_per_frame_mean"], d["ms_per_frame_min"], d.get("same_as_first"), d.get("status"))'
