# r06c: ray generation in small workgroups + per-queue ray regions (so that the ray generation of launch k + 1
# fits beside the tail of launch k): parity first (new tests + chain + parity files), then the sustained rate of
# small launches on 1 / 2 streams for raygen_waves = 16 / 4 / 1, the kernel trace at 1 and 4 frames per launch,
# and the batch shapes (64 / 20 frames per launch) for the same three.
set -u
O=gpurun_out/r06c; mkdir -p $O; rm -f $O/*
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_chain.py tests/test_gpu_streams.py tests/test_gpu_status.py -x -q --timeout 900 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for g in 16 4 1; do
  timeout 300 python tools/stream_overlap.py --frames 1,2,4,8 --streams 1,2 --tune raygen_waves=$g --out $O/stream_overlap_raygen.jsonl 2>/dev/null | cut -c1-200
done
for g in 4 1; do for fs in "1 2" "4 2"; do set -- $fs
  timeout 300 python tools/overlap_trace.py --frames $1 --streams $2 --tune raygen_waves=$g --out $O/overlap_trace.jsonl > $O/trace_$1_$2_$g.log 2>&1; tail -1 $O/trace_$1_$2_$g.log | cut -c1-420
done; done
timeout 600 python tools/quick_ab.py --config C1 --variants base --tunes "raygen_waves=16;raygen_waves=4;raygen_waves=1" --frames 64,20,8,1 --reps 4 --rotate --check --out $O/raygen_ab.jsonl 2>/dev/null | cut -c1-200
