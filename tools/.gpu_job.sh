set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu3.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu3.log
timeout 400 python bench.py > gpurun_out/bench_final3.json 2> gpurun_out/bench_final3.log; echo "bench rc=$?"; cat gpurun_out/bench_final3.json
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_final3 -o stats --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 128 --warmup 64 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_final3_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof_final3.log ); echo "prof rc=$?"
timeout 500 python tools/upload_bench.py C1 > gpurun_out/upload_bench.json 2> gpurun_out/upload_bench.log; echo "upload rc=$?"; cat gpurun_out/upload_bench.json
PMC_GROUPS="fetch write rdsize tcc sq1 sq2" bash tools/pmc.sh final3 > gpurun_out/pmc_final3.log 2>&1; echo "pmc rc=$?"; tail -40 gpurun_out/pmc_final3.log
