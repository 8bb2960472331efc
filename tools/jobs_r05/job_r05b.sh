# r05b: the drain-phase flush rule (a ray blocked by its full colour queue gets a partial shade round at once when
# few lanes of the wave still march): parity under the rule, sweep of the lane threshold at 64..1 frames per launch,
# the time-resolved one-frame launch with it, and small launches on 1 / 2 streams with it
set -u
O=gpurun_out/r05b; mkdir -p $O; rm -f $O/*
export TMPDIR=/tmp
VR_DRAIN_FLUSH=16 timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_parity.py tests/test_gpu_streams.py -x -q --timeout 600 > $O/pytest_drain16.log 2>&1; tail -2 $O/pytest_drain16.log
VR_DRAIN_FLUSH=64 timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_fullsize.py -x -q --timeout 600 > $O/pytest_drain64.log 2>&1; tail -2 $O/pytest_drain64.log
timeout 900 python tools/quick_ab.py --config C1 --variants base --tunes ";drain_flush=4;drain_flush=8;drain_flush=16;drain_flush=32;drain_flush=64;drain_flush=0" --frames 64,20,4,2,1 --reps 5 --rotate --check --out $O/ab_c1.jsonl > $O/ab_c1.log 2>&1
python - <<PY
import json
rows=[json.loads(l) for l in open("$O/ab_c1.jsonl")]
for r in rows: print(r["tune"] or "default", r["frames"], r["ms_per_frame_mean"], r["ms_per_frame_min"], r["same_as_first"], r["status"])
PY
timeout 600 python tools/quick_ab.py --config C3 --variants base --tunes ";drain_flush=16;drain_flush=64" --frames 16,2 --reps 3 --rotate --check --out $O/ab_c3.jsonl > $O/ab_c3.log 2>&1; cut -c1-200 $O/ab_c3.log | grep variant
for d in 16 64; do
  VR_TIMELINE=3 timeout 300 python tools/tail_profile.py --variant tl3 --frames 1 --tunes "drain_flush=$d" --out $O/tail_tl3_d$d.jsonl > $O/tail_d$d.log 2>&1; head -1 $O/tail_d$d.log
done
timeout 600 python tools/stream_overlap.py --frames 1,2,4 --streams 1,2 --tune drain_flush=16 --out $O/stream_overlap_d16.jsonl > $O/overlap16.log 2>&1; cut -c1-250 $O/overlap16.log | tail -6
timeout 600 python tools/stream_overlap.py --frames 1,2,4 --streams 1,2 --tune drain_flush=64 --out $O/stream_overlap_d64.jsonl > $O/overlap64.log 2>&1; cut -c1-250 $O/overlap64.log | tail -6
