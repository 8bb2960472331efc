set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_touch.py tests/test_gpu_bench.py tests/test_gpu_cli.py tests/test_gpu_streams.py -x -q > gpurun_out/pytest_r02_c.log 2>&1; tail -15 gpurun_out/pytest_r02_c.log
timeout 900 python tools/measure_traffic.py --config C1 --groups rdsize write fetch tcc sq1 sq2 --out gpurun_out/r02_traffic_C1.json > /dev/null 2> gpurun_out/traffic_C1.log; tail -60 gpurun_out/traffic_C1.log | cut -c1-400
cp gpurun_out/r02_traffic_C1.json profiles/r02_traffic_C1.json
timeout 600 python bench.py > gpurun_out/r02c_bench.json 2> gpurun_out/r02c_bench.log; echo "bench rc=$?"; tail -3 gpurun_out/r02c_bench.log
python - <<'PY'
import json
d=json.load(open("gpurun_out/r02c_bench.json"))
print(json.dumps(d["roofline"])[:1500]); print(json.dumps(d["cpu_baseline"])[:1200])
PY
