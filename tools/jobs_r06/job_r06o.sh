# r06o: thin waves march with a padded EXEC mask (a vector instruction with <= 8 active lanes costs 3.3-4.5x on this
# chip: tools/ubench/exec_mask_rate.hip).  Parity first, then the A/B against the library without the padding in ONE
# process (the candidate second and first), small launches on one / two streams, the 8x8-pixel probe.
set -u
O=gpurun_out/r06o; mkdir -p $O; rm -f $O/*
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_chain.py tests/test_gpu_probe.py tests/test_gpu_status.py -x -q --timeout 900 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
timeout 900 python tools/quick_ab.py --config C1 --variants nopad,base,nopad,base --tunes "" --frames 1,2,4,8,20,64 --reps 6 --rotate --check --out $O/pad_ab.jsonl 2>/dev/null | cut -c1-200
for v in nopad base; do
  timeout 300 python tools/stream_overlap.py --frames 1,2,4 --streams 1,2 --variant $v --out $O/pad_streams.jsonl 2>/dev/null | cut -c1-190
done
timeout 600 python tools/quick_ab.py --config C3 --variants nopad,base --tunes "" --frames 1,16,64 --reps 3 --rotate --check --out $O/pad_ab.jsonl 2>/dev/null | cut -c1-200
