# r06x: last validation of the final tree: full GPU suite, smoke(), both bench lines, and the driver's line three more times
set -u
O=gpurun_out/r06x; mkdir -p $O; rm -f $O/*
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q --timeout 900 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py > $O/r06_final_bench_lastbox.json 2> $O/final_bench.log; echo "bench rc=$?"
( time timeout 400 python bench.py --steps 20 --warmup 5 > $O/r06_final_bench_driverflags_lastbox.json 2> $O/final_bench_driverflags.log ) 2>&1 | grep real; echo "bench20 rc=$?"
for i in 1 2 3; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null >> $O/repeats_tmp.jsonl
done
python - <<PY
import json
out=open("$O/r06_bench_repeats.jsonl","w")
for l in open("$O/repeats_tmp.jsonl"):
    d=json.loads(l); out.write(json.dumps({"steps": d["steps"], "warmup": d["warmup"], "ms_per_step": d["ms_per_step"], "ms_per_step_cold": d.get("ms_per_step_cold"), "fps": d["fps"], "value": d["value"], "frac": d["roofline"]["frac"], "traffic_frac": d["roofline"]["traffic_frac"], "frac_of_model": (d["roofline"].get("model") or {}).get("frac_of_model"), "repeats": d["repeats"]["ms_per_step"], "parity": d["parity"]["rgba8_equal"]})+"\n")
out.close()
for f in ("r06_final_bench_lastbox.json","r06_final_bench_driverflags_lastbox.json"):
    d=json.load(open("$O/"+f)); r=d["roofline"]
    print(f, d["ms_per_step"], d.get("ms_per_step_cold"), d["fps"], d["value"], "frac", r["frac"], "model", (r.get("model") or {}).get("frac_of_model"), "traffic_frac", r["traffic_frac"], "parity", d["parity"]["rgba8_equal"])
print(open("$O/r06_bench_repeats.jsonl").read())
PY
