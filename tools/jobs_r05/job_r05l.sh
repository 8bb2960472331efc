# r05l: which sibling should share a cache line with an SH9 record?  (tools/record_pairing.py, C3; a build whose
# record bitmap has one bit per 64-byte slot); the new vr_touch_read through the test-suite's touch test
set -u
O=gpurun_out/r05l; mkdir -p $O; rm -f $O/*
timeout 300 python -m pytest tests/test_gpu_touch.py -x -q > $O/pytest_touch.log 2>&1; tail -1 $O/pytest_touch.log
timeout 900 python tools/record_pairing.py --config C3 --poses 5,60,110 --launch 16 --out $O/r05_record_pairing.jsonl 2>&1 | grep "^{" | cut -c1-700
