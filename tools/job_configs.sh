set -u
mkdir -p gpurun_out
for c in C2 C3 C1r C1t; do
  timeout 900 python bench.py --config $c --no-cpu-baseline > gpurun_out/r02_bench_$c.json 2> gpurun_out/r02_bench_$c.log
  python - <<PY
import json
d=json.load(open("gpurun_out/r02_bench_$c.json")); r=d["roofline"]
print("$c", d["ms_per_step"], d["fps"], d["value"], "frac", r["frac"], "samples", r["samples_per_ray"], "hits", r["hit_samples_per_ray"], "uniq/frame MB", r["unique_bytes_per_frame"]/1e6, d["config"]["workload"][:90])
PY
  grep "sched per frame" gpurun_out/r02_bench_$c.log | cut -c1-250
done
