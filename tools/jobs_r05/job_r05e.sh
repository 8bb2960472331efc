# r05e: the tree-wise brick entry order (blocked for lookup structures beyond 128 MB) + scalar round counters:
# full GPU suite, then new default against the previous commit's library ("prev") on C1 / C2 / C3, same process;
# C1 / C2 with the blocked order forced on and C3 with it forced off (VR_BRICK_BLOCKED, upload-time); SH9 at seven
# waves per SIMD; balance of the tile shard at the driver's launch shape (20 poses per launch)
set -u
O=gpurun_out/r05e; mkdir -p $O; rm -f $O/*
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -x -q --timeout 900 > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
timeout 900 python tools/quick_ab.py --config C1 --variants prev,base,prev,base --tunes "" --frames 64,20,1 --reps 5 --rotate --check --out $O/ab_c1.jsonl > $O/ab_c1.log 2>&1; cut -c1-200 $O/ab_c1.log | grep variant
timeout 900 python tools/quick_ab.py --config C3 --variants prev,base,prev,base --tunes "" --frames 16 --reps 4 --rotate --check --out $O/ab_c3.jsonl > $O/ab_c3.log 2>&1; cut -c1-200 $O/ab_c3.log | grep variant
timeout 900 python tools/quick_ab.py --config C2 --variants prev,base,prev,base --tunes "" --frames 8 --reps 3 --rotate --check --out $O/ab_c2.jsonl > $O/ab_c2.log 2>&1; cut -c1-200 $O/ab_c2.log | grep variant
VR_BRICK_BLOCKED=0 timeout 600 python tools/quick_ab.py --config C3 --variants base --tunes "" --frames 16 --reps 4 --rotate --out $O/ab_c3_linear.jsonl 2>&1 | cut -c1-200 | grep variant
timeout 600 python tools/quick_ab.py --config C3 --variants base,sh9w7,base,sh9w7 --tunes "" --frames 16 --reps 4 --rotate --check --out $O/ab_c3_w7.jsonl 2>&1 | cut -c1-200 | grep variant
for c in C1 C3; do for w in 2 4 8; do
  timeout 600 python tools/shard_balance.py --config $c --world $w --tile-rows 8 --frames 20 --first-pose 5 --out $O/r05_shard_balance.jsonl > /dev/null 2>&1
done; done
python - <<PY
import json
for l in open("$O/r05_shard_balance.jsonl"):
    r=json.loads(l); print(r["config"], "world", r["world"], "whole", r["whole_frame_launch_ms"], "max rank", r["max_ms"], "mean", r["mean_ms"], "speedup bound", round(r["whole_frame_launch_ms"]/r["max_ms"],2))
PY
