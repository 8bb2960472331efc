set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_r02_a.log 2>&1; tail -3 gpurun_out/pytest_r02_a.log
timeout 400 python bench.py > gpurun_out/r02a_bench.json 2> gpurun_out/r02a_bench.log; echo "bench rc=$?"; cut -c1-300 gpurun_out/r02a_bench.json
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02a_bench_driverflags.json 2> gpurun_out/r02a_bench_driverflags.log; cut -c1-300 gpurun_out/r02a_bench_driverflags.json
bash tools/launch_sweep.sh r02a > /dev/null 2>&1; cat gpurun_out/sweep_r02a.jsonl
timeout 300 python bench.py --batch 20 --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | cut -c100-260
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r02a -o stats --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 128 --warmup 64 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_r02a_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof_r02a.log ); echo "prof rc=$?"
