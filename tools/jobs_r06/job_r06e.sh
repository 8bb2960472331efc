# r06e: one gather path -- bench.py's tile mode and the CLI both through libvolrend_gather.so: the bench tests
# (forced gather on one rank, shared-GPU rehearsal, plain `python bench.py --gpus 2` self-launch), the CLI tests
# (--gpus 1 through RCCL, --share_gpu rehearsal), renderer + status tests; then the forced-gather bench line and
# the CLI benchmark's RCCL row.
set -u
O=gpurun_out/r06e; mkdir -p $O; rm -f $O/*
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_gpu_bench.py tests/test_gpu_cli.py tests/test_gpu_renderer.py tests/test_gpu_status.py tests/test_gpu_cpp_api.py -x -q --timeout 900 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
VOLREND_FORCE_GATHER=1 timeout 300 python bench.py --no-cpu-baseline > $O/bench_forced_gather.json 2> $O/bench_forced_gather.log; echo "forced rc=$?"
timeout 300 python bench.py --no-cpu-baseline > $O/bench_plain.json 2> $O/bench_plain.log; echo "plain rc=$?"
python - <<PY
import json
for f in ("bench_forced_gather","bench_plain"):
    d=json.load(open("$O/%s.json"%f)); print(f, d["ms_per_step"], d["fps"], d["parity"]["rgba8_equal"], (d.get("rccl") or {}).get("gather"), d["repeats"]["ms_per_step"])
PY
