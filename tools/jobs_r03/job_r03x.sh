# steal threshold (an eighth of a queue, at most 8192 rays, is left to the queue's own waves): parity, then the
# old library (round-2 protocol: steals down to the last chunk) against the new one, both kernels
set -u
mkdir -p gpurun_out/r03x
O=gpurun_out/r03x
rm -f $O/*
timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_parity.py tests/test_gpu_streams.py -x -q --timeout 600 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
timeout 900 python tools/quick_ab.py --config C1 --variants old,base,old,base --tunes "split=1;split=0" --frames 1,2,4,20,64 --reps 8 --rotate --check --out $O/ab_c1.jsonl > $O/ab_c1.log 2>&1
timeout 900 python tools/quick_ab.py --config C3 --variants old,base,old,base --tunes "split=1;split=0" --frames 1,16 --reps 4 --rotate --check --out $O/ab_c3.jsonl > $O/ab_c3.log 2>&1
cat $O/ab_c1.jsonl $O/ab_c3.jsonl | python -c '
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d["config"], d["variant"], d["tune"], d["frames"], d["ms_per_frame_mean"], d["ms_per_frame_min"], d.get("same_as_first"), d.get("status"))'
