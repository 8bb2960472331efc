# r05a: first GPU job of round 5.  (1) the new status tests + a sanity subset of the GPU suite on the changed
# sources; (2) anatomy of small launches (kernel trace: three kernels and their gaps); (3) one-frame launches on
# 1..4 alternating streams; (4) what a march round costs without its memory latency (timeline builds with the
# lookup loads ablated) -- the floor of the drain phase
set -u
O=gpurun_out/r05a; mkdir -p $O; rm -f $O/*
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_status.py tests/test_gpu_streams.py tests/test_gpu_renderer.py tests/test_gpu_cli.py -x -q --timeout 600 > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 600 python tools/launch_anatomy.py --out $O/launch_anatomy.jsonl > $O/anatomy.log 2>&1; tail -8 $O/anatomy.log | cut -c1-400
timeout 600 python tools/stream_overlap.py --frames 1,2,4 --streams 1,2,3,4 --out $O/stream_overlap.jsonl > $O/overlap.log 2>&1; tail -12 $O/overlap.log | cut -c1-300
for v in tl3 tl3a6 tl3a9 tl3a10; do
  VR_TIMELINE=3 timeout 300 python tools/tail_profile.py --variant $v --frames 1 --out $O/tail_$v.jsonl > $O/tail_$v.log 2>&1
  python - <<PY
import json
r=json.loads(open("$O/tail_$v.jsonl").readline())
b=r["buckets"]
lo=[x for x in b if x["rounds"]>50 and x["lanes_per_round"]<=4]
hi=[x for x in b if x["lanes_per_round"]>=55 and x["waves_alive"]>4000 and x["march_clocks_per_round"]]
print("$v", "launch_ms", r["launch_ms"], "buckets", len(b), "floor clocks/round (<=4 lanes):", [x["march_clocks_per_round"] for x in lo][:12], "loaded:", [x["march_clocks_per_round"] for x in hi][:8])
PY
done
