# r05i: the CLI again with its launch slots sized before the clock starts (r05h: the first launches' multi-GB
# hipMallocs sat inside the timed loop -> outliers of 2-4x); the time-resolved 20-frame launch (how long is the tail
# the driver's one-launch region pays?)
set -u
O=gpurun_out/r05i; mkdir -p $O; rm -f $O/*
for i in 1 2; do
timeout 900 python tools/cli_bench.py > $O/r05_cli_bench_$i.json 2> $O/cli_bench_$i.log; python - <<PY
import json
d=json.load(open("$O/r05_cli_bench_$i.json"))
print(" ".join("%s=%s" % (k, v["ms_per_frame"]) for k,v in d.items() if isinstance(v,dict)))
PY
done
VR_TIMELINE=3 timeout 300 python tools/tail_profile.py --variant tl3 --frames 20,4 --out $O/tail_20.jsonl > $O/tail_20.log 2>&1; grep -v "^   bucket" $O/tail_20.log | cut -c1-200
python - <<PY
import json
for l in open("$O/tail_20.jsonl"):
    r=json.loads(l); b=r["buckets"]
    print("frames", r["frames"], "launch_ms", r["launch_ms"], "buckets", len(b))
    for x in b:
        if x["bucket"] % 8 == 0 or x["waves_alive"] < 4600: print("  b%3d alive %4d lanes %4.1f march %s shade %s refill %s clk/round %s" % (x["bucket"], x["waves_alive"], x["lanes_per_round"], x["frac_march"], x["frac_shade"], x["frac_refill"], x["march_clocks_per_round"]))
PY
