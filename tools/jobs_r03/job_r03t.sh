set -u
mkdir -p gpurun_out/r03t
timeout 1800 python -m pytest tests -m gpu -x -q --timeout 900 > gpurun_out/r03t/pytest_gpu.log 2>&1; tail -3 gpurun_out/r03t/pytest_gpu.log
