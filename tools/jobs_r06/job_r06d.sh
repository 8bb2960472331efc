# r06d: the whole GPU suite on the tree of commit 00ed2e8 (raygen workgroups, per-queue regions, status on a
# stream, underlay producer stream, bench self-launch), smoke(), and both bench lines.
set -u
O=gpurun_out/r06d; mkdir -p $O; rm -f $O/*
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q --timeout 900 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench20.json 2> $O/bench20.log ) 2>&1 | grep real; echo "bench20 rc=$?"
python - <<PY
import json
d=json.load(open("$O/bench20.json")); r=d["roofline"]
print(d["ms_per_step"], d.get("ms_per_step_cold"), d["fps"], d["value"], "frac", r["frac"], "model", (r.get("model") or {}).get("frac_of_model"), "traffic_frac", r["traffic_frac"], r["traffic_source"][:80], "parity", d["parity"]["rgba8_equal"], d["repeats"]["ms_per_step"])
PY
