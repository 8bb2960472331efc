set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_r02_a.log 2>&1; tail -3 gpurun_out/pytest_r02_a.log
bash tools/launch_sweep.sh r02_baseline
