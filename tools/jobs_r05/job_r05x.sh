# r05x: last validation of the final tree: full GPU suite, smoke(), both bench lines, and both lines three more times
# on this one box (what a box repeats to)
set -u
O=gpurun_out/r05x; mkdir -p $O; rm -f $O/*
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 900 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py > $O/r05_final_bench.json 2> $O/final_bench.log; echo "bench rc=$?"
timeout 400 python bench.py --steps 20 --warmup 5 > $O/r05_final_bench_driverflags.json 2> $O/final_bench_driverflags.log; echo "bench20 rc=$?"
for i in 1 2 3; do
  timeout 300 python bench.py --no-cpu-baseline --live-traffic 0 2>/dev/null >> $O/repeats_tmp.jsonl
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --live-traffic 0 2>/dev/null >> $O/repeats_tmp.jsonl
done
python - <<PY
import json
out=open("$O/r05_bench_repeats.jsonl","w")
for l in open("$O/repeats_tmp.jsonl"):
    d=json.loads(l); out.write(json.dumps({"steps": d["steps"], "warmup": d["warmup"], "ms_per_step": d["ms_per_step"], "ms_per_step_cold": d.get("ms_per_step_cold"), "fps": d["fps"], "value": d["value"], "frac": d["roofline"]["frac"], "frac_of_model": (d["roofline"].get("model") or {}).get("frac_of_model"), "repeats": d["repeats"]["ms_per_step"], "parity": d["parity"]["rgba8_equal"]})+"\n")
out.close()
for f in ("r05_final_bench.json","r05_final_bench_driverflags.json"):
    d=json.load(open("$O/"+f)); r=d["roofline"]
    print(f, d["ms_per_step"], d.get("ms_per_step_cold"), d["fps"], d["value"], "frac", r["frac"], "model", (r.get("model") or {}).get("frac_of_model"), "traffic_frac", r["traffic_frac"], "parity", d["parity"]["rgba8_equal"], "sched", d["sched"]["valu_insts_per_frame"] is not None)
print(open("$O/r05_bench_repeats.jsonl").read())
PY
