set -u
mkdir -p gpurun_out/r03q
O=gpurun_out/r03q
timeout 900 python tools/upload_bench.py > $O/r03_upload_bench.json 2> $O/upload_bench.log; cat $O/r03_upload_bench.json
VR_SWEEP_SEEDS=600 timeout 1500 python -m pytest tests/test_gpu_chain.py -q -x --timeout 1400 -k "random_sweep" > $O/seed_sweep_chain.log 2>&1; tail -2 $O/seed_sweep_chain.log
VR_SWEEP_SEEDS=600 timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x --timeout 1400 -k "sweep or random" > $O/seed_sweep_parity.log 2>&1; tail -2 $O/seed_sweep_parity.log
