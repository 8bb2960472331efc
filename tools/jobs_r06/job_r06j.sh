# r06j: cost-ordered ray queues, third build (order kernel: LDS-staged separable dilation, parallel scan).
set -u
O=gpurun_out/r06j; mkdir -p $O; rm -f $O/*
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "cost_ordered or random_launch or raygen_workgroup" --timeout 800 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
timeout 600 python tools/launch_anatomy.py --shapes 800x1,800x4,800x20 --reps 5 --out $O/anatomy.jsonl 2>/dev/null | cut -c1-330
timeout 900 python tools/quick_ab.py --config C1 --variants base --tunes "cost_order=0;cost_order=1" --frames 64,20,8,4,2,1 --reps 6 --rotate --check --out $O/cost_order_ab.jsonl 2>/dev/null | cut -c1-210
for t in "cost_order=0" "cost_order=1"; do
  timeout 300 python tools/stream_overlap.py --frames 1,2,4 --streams 1,2 --tune "$t" --out $O/cost_order_streams.jsonl 2>/dev/null | cut -c1-190
done
