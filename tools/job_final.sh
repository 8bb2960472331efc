# Round-end measurement set (run through gpurun from the repo root):  bash tools/job_final.sh <round tag>
set -u
R=${1:-r02}
mkdir -p gpurun_out/$R
export TMPDIR=/tmp
O=gpurun_out/$R
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
# traffic first: bench.py then reports it (source hash verified)
for c in C1 C2 C3; do
  timeout 900 python tools/measure_traffic.py --config $c --groups rdsize write fetch tcc sq1 sq2 --out $O/${R}_traffic_$c.json > /dev/null 2> $O/traffic_$c.log
  cp $O/${R}_traffic_$c.json profiles/${R}_traffic_$c.json; tail -2 $O/traffic_$c.log
done
timeout 600 python bench.py > $O/${R}_final_bench.json 2> $O/final_bench.log; echo "bench rc=$?"
timeout 400 python bench.py --steps 20 --warmup 5 > $O/${R}_final_bench_driverflags.json 2> $O/final_bench_driverflags.log
for c in C2 C3 C1r C1t; do timeout 900 python bench.py --config $c --no-cpu-baseline > $O/${R}_bench_$c.json 2> $O/bench_$c.log; done
timeout 300 python bench.py --fp fma --no-cpu-baseline > $O/${R}_bench_C1_fma.json 2>/dev/null
timeout 300 python bench.py --readback --no-cpu-baseline > $O/${R}_bench_C1_readback.json 2>/dev/null
VOLREND_FORCE_GATHER=1 timeout 300 python bench.py --no-cpu-baseline > $O/${R}_bench_C1_forced_gather.json 2>/dev/null
bash tools/launch_sweep.sh $R > /dev/null 2>&1; cp gpurun_out/sweep_$R.jsonl $O/${R}_launch_sweep.jsonl
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o stats --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 128 --warmup 64 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof_bench.json 2> $GRAFT_REPO_ROOT/$O/prof.log ); cp $O/prof/stats_kernel_stats.csv $O/${R}_final_kernel_stats.csv
VR_TIMELINE=1 VOLREND_HIP_LIB=$PWD/volrend_amd/libvolrend_hip_tl.so timeout 300 python bench.py --no-cpu-baseline 2>&1 >/dev/null | grep timeline > $O/${R}_timeline.txt
for c in C2 C3; do VR_TIMELINE=1 VOLREND_HIP_LIB=$PWD/volrend_amd/libvolrend_hip_tl.so timeout 600 python bench.py --config $c --no-cpu-baseline 2>&1 >/dev/null | grep timeline | sed -e "s/^/$c: /" >> $O/${R}_timeline.txt; done
bash tools/kernel_resources.sh > $O/${R}_kernel_resources.txt 2>&1
timeout 900 python tools/cli_bench.py > $O/${R}_cli_bench.json 2> $O/cli_bench.log
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/${R}_*bench*.json")):
    try:
        d=json.load(open(f)); r=d.get("roofline",{})
        print(f.split("/")[-1], d.get("ms_per_step"), d.get("fps"), d.get("value"), "frac", r.get("frac"), "traffic", r.get("traffic"))
    except Exception as e: print(f, "ERR", e)
PY
