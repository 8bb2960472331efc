# quick GPU check of a kernel change: parity suites + the C1 bench at 64 frames / launch and with the driver's flags
set -u
mkdir -p gpurun_out
TAG=${1:-q}
timeout 1500 python -m pytest tests -m gpu -x -q ${PYTEST_ARGS:-} > gpurun_out/pytest_$TAG.log 2>&1; tail -3 gpurun_out/pytest_$TAG.log
for spec in "256 64" "20 5"; do set -- $spec
timeout 300 python bench.py --steps $1 --warmup $2 --no-cpu-baseline 2>gpurun_out/bench_$TAG.log | python -c '
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d["steps"], d["ms_per_step"], d["fps"], d["roofline"]["kernel_ms_per_frame"], d["roofline"]["frac"])'
done
