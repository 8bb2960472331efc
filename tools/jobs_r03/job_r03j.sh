set -u
mkdir -p gpurun_out/r03j
O=gpurun_out/r03j
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 600 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_gpu.log
timeout 900 python tools/upload_bench.py > $O/upload_bench.json 2> $O/upload_bench.log; echo "upload rc=$?"; cat $O/upload_bench.json
timeout 900 python tools/cli_bench.py > $O/cli_bench.json 2> $O/cli_bench.log; echo "cli rc=$?"; cat $O/cli_bench.json; tail -3 $O/cli_bench.log
