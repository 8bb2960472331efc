# r06h: cost-ordered ray queues (blocks of every queue's region sorted by the previous launches' longest rays).
# Parity first (the parity / chain / streams / status files + the new sequence test), then the A/B in ONE process:
# cost_order=0 / 1 at 64, 20, 8, 4, 2, 1 frames per launch (fresh poses every launch), the sustained rate of small
# launches on one / two streams, C3 and C2 at 64.
set -u
O=gpurun_out/r06h; mkdir -p $O; rm -f $O/*
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_chain.py tests/test_gpu_streams.py tests/test_gpu_status.py tests/test_gpu_fullsize.py -x -q --timeout 900 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 900 python tools/quick_ab.py --config C1 --variants base --tunes "cost_order=0;cost_order=1" --frames 64,20,8,4,2,1 --reps 6 --rotate --check --out $O/cost_order_ab.jsonl 2>/dev/null | cut -c1-210
for t in "cost_order=0" "cost_order=1"; do
  timeout 300 python tools/stream_overlap.py --frames 1,2,4,8 --streams 1,2 --tune "$t" --out $O/cost_order_streams.jsonl 2>/dev/null | cut -c1-190
done
timeout 900 python tools/quick_ab.py --config C3 --variants base --tunes "cost_order=0;cost_order=1" --frames 64,16,1 --reps 3 --rotate --check --out $O/cost_order_ab.jsonl 2>/dev/null | cut -c1-210
timeout 900 python tools/quick_ab.py --config C2 --variants base --tunes "cost_order=0;cost_order=1" --frames 32,1 --reps 3 --rotate --check --out $O/cost_order_ab.jsonl 2>/dev/null | cut -c1-210
