set -u
mkdir -p gpurun_out/r03b
O=gpurun_out/r03b
# smoke first (tiny): a hang shows here cheaply
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout 900 python -m pytest tests -m gpu -x -q --timeout 180 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_gpu.log
timeout 600 python tools/quick_ab.py --config C1 --variants base --tunes "split=0;split=1" --frames 64,20,1 --reps 3 --check --out $O/ab_c1.jsonl > $O/ab_c1.log 2>&1
timeout 600 python tools/quick_ab.py --config C3 --variants base --tunes "split=0;split=1" --frames 16 --reps 3 --check --out $O/ab_c3.jsonl > $O/ab_c3.log 2>&1
timeout 600 python tools/quick_ab.py --config C2 --variants base --tunes "split=0;split=1" --frames 8 --reps 2 --check --out $O/ab_c2.jsonl > $O/ab_c2.log 2>&1
cat $O/ab_c1.jsonl $O/ab_c3.jsonl $O/ab_c2.jsonl | python -c '
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d["config"], d["variant"], d["tune"], d["frames"], d["ms_per_frame_mean"], d["ms_per_frame_min"], "status", d["status"], "same", d["same_as_first"])'
tail -3 $O/ab_c1.log
