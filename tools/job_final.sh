# Round-end measurement set (run through gpurun from the repo root):  bash tools/job_final.sh <round tag>
# ~8 GPU-minutes.  Everything lands in gpurun_out/<tag>/; the traffic JSONs are also copied to
# profiles/ at once so that the bench lines taken afterwards can report them (source hash verified).
set -u
R=${1:-r06}
mkdir -p gpurun_out/$R
export TMPDIR=/tmp
O=gpurun_out/$R
timeout 1800 python -m pytest tests -m gpu -x -q --timeout 900 > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
# --- PMC traffic first: C1 at 64, 20 (the driver's launch shape) and 1 frames per launch, C2 / C3 read sizes + L2
timeout 900 python tools/measure_traffic.py --config C1 --groups rdsize write fetch tcc sq1 sq2 --out $O/${R}_traffic_C1.json > /dev/null 2> $O/traffic_C1.log; tail -1 $O/traffic_C1.log
timeout 900 python tools/measure_traffic.py --config C1 --batch 20 --groups rdsize write tcc sq1 sq2 --out $O/${R}_traffic_C1_20.json > /dev/null 2> $O/traffic_C1_20.log; tail -1 $O/traffic_C1_20.log
timeout 900 python tools/measure_traffic.py --config C1 --batch 1 --groups rdsize write sq1 sq2 --out $O/${R}_traffic_C1_1.json > /dev/null 2> $O/traffic_C1_1.log; tail -1 $O/traffic_C1_1.log
for c in C2 C3; do
  timeout 900 python tools/measure_traffic.py --config $c --groups rdsize write tcc sq1 sq2 --out $O/${R}_traffic_$c.json > /dev/null 2> $O/traffic_$c.log; tail -1 $O/traffic_$c.log
done
cp $O/${R}_traffic_*.json profiles/
# --- bench lines
timeout 600 python bench.py > $O/${R}_final_bench.json 2> $O/final_bench.log; echo "bench rc=$?"
timeout 400 python bench.py --steps 20 --warmup 5 > $O/${R}_final_bench_driverflags.json 2> $O/final_bench_driverflags.log
# the same with the traffic measured by the run itself (two rocprofv3 --pmc passes of a child process after the
# timed region: opt-in since round 6, the default line reports the committed hash-verified profile)
timeout 400 python bench.py --steps 20 --warmup 5 --live-traffic 1 --no-cpu-baseline > $O/${R}_final_bench_driverflags_live_traffic.json 2> $O/final_bench_driverflags_live.log
for c in C2 C3; do timeout 900 python bench.py --config $c --no-cpu-baseline > $O/${R}_bench_$c.json 2> $O/bench_$c.log; done
timeout 300 python bench.py --fp fma --no-cpu-baseline > $O/${R}_bench_C1_fma.json 2>/dev/null
timeout 300 python bench.py --readback --no-cpu-baseline > $O/${R}_bench_C1_readback.json 2>/dev/null
VOLREND_FORCE_GATHER=1 timeout 300 python bench.py --no-cpu-baseline > $O/${R}_bench_C1_forced_gather.json 2>/dev/null
# --- launch shape: frames per launch (steady state, quick_ab with fresh poses per launch) and lone launches as the driver times them
timeout 600 python tools/quick_ab.py --config C1 --variants base --tunes "" --frames 64,20,8,4,2,1 --reps 4 --rotate --out $O/${R}_launch_shape.jsonl > $O/launch_shape.log 2>&1
bash tools/lone_launch.sh $R > /dev/null 2>&1; cp gpurun_out/lone_$R.jsonl $O/${R}_lone_launch.jsonl
# --- small launches: anatomy (kernel trace), one / two alternating streams, the dependent round in an empty / full chip
timeout 600 python tools/launch_anatomy.py --out $O/${R}_launch_anatomy.jsonl > $O/anatomy.log 2>&1
timeout 600 python tools/stream_overlap.py --frames 1,2,4,8 --streams 1,2 --out $O/${R}_stream_overlap.jsonl > $O/overlap.log 2>&1
for fs in "1 2" "2 2" "4 2"; do set -- $fs; timeout 300 python tools/overlap_trace.py --frames $1 --streams $2 --out $O/${R}_overlap_trace_final.jsonl > $O/trace_$1_$2.log 2>&1; done
# the N > 1 bookkeeping of the driver's command with 8 ranks SHARING this one GPU (gloo, host-staged gather: a rehearsal of
# the code path -- launcher, shards, pipeline, assembly, self-check, rank-0 line --, never a measurement)
VOLREND_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 8 --steps 20 --warmup 5 --no-cpu-baseline > $O/${R}_rehearsal_8ranks_one_gpu.json 2> $O/rehearsal8.log; echo "rehearsal8 rc=$?"
# C3: where the time goes (timing ablations in one process)
timeout 900 python tools/quick_ab.py --config C3 --variants base,abl4,abl5,abl6 --tunes "" --frames 64 --reps 3 --rotate --out $O/${R}_C3_ablation_final.jsonl > $O/c3_abl.log 2>&1
timeout 600 python tools/round_time_probe.py --out $O/${R}_round_time_probe.jsonl > $O/round_time.log 2>&1
VOLREND_HIP_LIB=$PWD/volrend_amd/libvolrend_hip_tl3.so VR_TIMELINE=3 timeout 300 python tools/tail_profile.py --frames 1 --out $O/${R}_tail_profile_final.jsonl > $O/tail.log 2>&1
# --- rocprofv3 kernel stats of the default and of the driver's command
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o stats --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 128 --warmup 64 --no-cpu-baseline --no-parity > $GRAFT_REPO_ROOT/$O/prof_bench.json 2> $GRAFT_REPO_ROOT/$O/prof.log ); cp $O/prof/stats_kernel_stats.csv $O/${R}_final_kernel_stats.csv
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof20 -o stats --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity > $GRAFT_REPO_ROOT/$O/prof20_bench.json 2> $GRAFT_REPO_ROOT/$O/prof20.log ); cp $O/prof20/stats_kernel_stats.csv $O/${R}_driverflags_kernel_stats.csv
# --- per-phase timeline (profiling build) of the render kernel (C1, C3)
VR_TIMELINE=1 VOLREND_HIP_LIB=$PWD/volrend_amd/libvolrend_hip_tl.so timeout 300 python bench.py --no-cpu-baseline --no-parity --repeats 0 2>&1 >/dev/null | grep timeline > $O/${R}_timeline.txt
VR_TIMELINE=1 VOLREND_HIP_LIB=$PWD/volrend_amd/libvolrend_hip_tl.so timeout 600 python bench.py --config C3 --no-cpu-baseline --no-parity 2>&1 >/dev/null | grep timeline | sed -e "s/^/C3: /" >> $O/${R}_timeline.txt
bash tools/kernel_resources.sh > $O/${R}_kernel_resources.txt 2>&1
# --- bench-shaped launches against the CPU oracle, frame by frame, and the wide random sweep
rm -f $O/${R}_batch_parity.jsonl
timeout 600 python tools/check_batch_parity.py C1 64 64 >> $O/${R}_batch_parity.jsonl 2>/dev/null
timeout 600 python tools/check_batch_parity.py C1 1 17 >> $O/${R}_batch_parity.jsonl 2>/dev/null
timeout 900 python tools/check_batch_parity.py C3 16 30 >> $O/${R}_batch_parity.jsonl 2>/dev/null
timeout 900 python tools/check_batch_parity.py C2 8 10 >> $O/${R}_batch_parity.jsonl 2>/dev/null
VR_SWEEP_SEEDS=600 timeout 1500 python -m pytest tests/test_gpu_chain.py -q -x --timeout 1400 -k "random_sweep" > $O/seed_sweep_chain.log 2>&1
VR_SWEEP_SEEDS=600 timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x --timeout 1400 -k "sweep or random" > $O/seed_sweep_parity.log 2>&1
( echo "VR_SWEEP_SEEDS=600 pytest tests/test_gpu_chain.py -k random_sweep:"; tail -1 $O/seed_sweep_chain.log; echo "VR_SWEEP_SEEDS=600 pytest tests/test_gpu_parity.py -k 'sweep or random':"; tail -1 $O/seed_sweep_parity.log ) > $O/${R}_seed_sweep.txt
timeout 900 python tools/cli_bench.py > $O/${R}_cli_bench.json 2> $O/cli_bench.log
VR_UPLOAD_TIMING=1 timeout 900 python tools/upload_bench.py > $O/${R}_upload_bench.json 2> $O/upload_bench.log
# --- balance of the 8-rank tile shard (each rank's bands rendered alone on this one GPU)
rm -f $O/${R}_shard_balance.jsonl
timeout 900 python tools/shard_balance.py --config C3 --world 8 --tile-rows 8,16,32,64 --frames 64 --out $O/${R}_shard_balance.jsonl > /dev/null 2>&1
# the driver's launch shape: 20 poses per launch, 2 / 4 / 8 ranks (C1: the bench's workload; C3: BASELINE config 3)
for c in C1 C3; do for w in 2 4 8; do
  timeout 600 python tools/shard_balance.py --config $c --world $w --tile-rows 8 --frames 20 --first-pose 5 --out $O/${R}_shard_balance.jsonl > /dev/null 2>&1
done; done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/${R}_*bench*.json")):
    try:
        d=json.load(open(f)); r=d.get("roofline",{})
        print(f.split("/")[-1], d.get("ms_per_step"), d.get("fps"), d.get("value"), "frac", r.get("frac"), "traffic", r.get("traffic"), "parity", (d.get("parity") or {}).get("rgba8_equal"))
    except Exception as e: print(f, "ERR", e)
PY
