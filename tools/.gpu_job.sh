set -u
mkdir -p gpurun_out/sweep
run() { tag=$1; shift; timeout 300 env "$@" python bench.py --no-cpu-baseline > gpurun_out/sweep/$tag.json 2> gpurun_out/sweep/$tag.log; python - <<PY
import json
try:
    d=json.load(open("gpurun_out/sweep/$tag.json")); print("$tag", d["ms_per_step"], d["value"], d["roofline"]["kernel_ms_mean"])
except Exception as e: print("$tag failed", e)
PY
}
run new X=1
run new_m2 VR_MARCH_MAX=2
run new_m4 VR_MARCH_MAX=4
run new_g7 VR_GRID_LEVELS=7
timeout 300 python bench.py --batch 128 --steps 512 --warmup 128 --no-cpu-baseline 2>/dev/null | cut -c1-220
timeout 300 python bench.py --batch 32 --no-cpu-baseline 2>/dev/null | cut -c1-220
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -2
