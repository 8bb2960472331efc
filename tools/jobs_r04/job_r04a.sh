# r04a: the pipelined fused kernel (one wait per round: lookup word + ray refill + record DMA) -- parity first,
# then A/B against the round-3 library (libvolrend_hip_old.so) with a refill_min sweep
set -u
O=gpurun_out/r04a; mkdir -p $O; rm -f $O/*
timeout 1200 python -m pytest tests/test_gpu_chain.py tests/test_gpu_parity.py tests/test_gpu_probe.py -x -q --timeout 900 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
T="split=0;split=0,refill_min=8;split=0,refill_min=12;split=0,refill_min=16;split=0,refill_min=32;split=0,refill_min=48"
timeout 900 python tools/quick_ab.py --config C1 --variants old,base,old,base --tunes "$T" --frames 64,20 --reps 4 --rotate --check --out $O/ab_c1.jsonl > $O/ab_c1.log 2>&1
timeout 600 python tools/quick_ab.py --config C1 --variants old,base,old,base --tunes "split=-1;split=0;split=0,refill_min=8" --frames 1,4 --reps 6 --rotate --check --out $O/ab_c1_small.jsonl > $O/ab_c1_small.log 2>&1
timeout 900 python tools/quick_ab.py --config C3 --variants old,base,old,base --tunes "split=0;split=0,refill_min=12" --frames 16 --reps 4 --rotate --check --out $O/ab_c3.jsonl > $O/ab_c3.log 2>&1
timeout 900 python tools/quick_ab.py --config C2 --variants old,base --tunes "split=0;split=0,refill_min=12" --frames 8 --reps 3 --rotate --check --out $O/ab_c2.jsonl > $O/ab_c2.log 2>&1
cat $O/ab_c1.jsonl $O/ab_c1_small.jsonl $O/ab_c3.jsonl $O/ab_c2.jsonl | python -c '
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d["config"], d["variant"], d["tune"], d["frames"], d["ms_per_frame_mean"], d["ms_per_frame_min"], d.get("same_as_first"), d.get("status"))'
tail -5 $O/ab_c1.log
