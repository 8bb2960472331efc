set -u
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu5.log 2>&1; tail -2 gpurun_out/pytest_gpu5.log
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 400 python bench.py > gpurun_out/bench_final5.json 2> gpurun_out/bench_final5.log; echo "bench rc=$?"; cut -c1-400 gpurun_out/bench_final5.json
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_final5 -o stats --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 128 --warmup 64 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_final5_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof_final5.log ); echo "prof rc=$?"
PMC_TIMEOUT=150 PMC_GROUPS="fetch write rdsize tcc sq1 sq2" bash tools/pmc.sh final5 > gpurun_out/pmc_final5.log 2>&1; echo "pmc rc=$?"
for c in C2 C3; do timeout 400 python bench.py --config $c --no-cpu-baseline > gpurun_out/bench_$c.json 2> gpurun_out/bench_$c.log; cut -c1-330 gpurun_out/bench_$c.json; echo; done
timeout 300 python bench.py --fp fma --no-cpu-baseline 2>/dev/null | cut -c100-210
timeout 300 python bench.py --readback --no-cpu-baseline 2>/dev/null | cut -c100-210
