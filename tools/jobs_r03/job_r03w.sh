# dry-queue hints in the ray queue (grab_chunk): old library vs new, both kernels, 1..64 frames per launch
set -u
mkdir -p gpurun_out/r03w
O=gpurun_out/r03w
rm -f $O/*
timeout 600 python -m pytest tests/test_gpu_chain.py -x -q --timeout 300 > $O/pytest_chain.log 2>&1; echo "chain rc=$?"; tail -2 $O/pytest_chain.log
timeout 600 python tools/tail_profile.py --frames 1 --tunes "split=0" --out $O/tail_profile_new.jsonl > $O/tail_new.log 2>&1
timeout 900 python tools/quick_ab.py --config C1 --variants old,base,old,base --tunes "split=1;split=0" --frames 1,2,4,20,64 --reps 8 --rotate --check --out $O/ab_c1.jsonl > $O/ab_c1.log 2>&1
timeout 900 python tools/quick_ab.py --config C3 --variants old,base,old,base --tunes "split=1;split=0" --frames 1,16 --reps 4 --rotate --check --out $O/ab_c3.jsonl > $O/ab_c3.log 2>&1
cat $O/ab_c1.jsonl $O/ab_c3.jsonl | python -c '
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d["config"], d["variant"], d["tune"], d["frames"], d["ms_per_frame_mean"], d["ms_per_frame_min"], d.get("same_as_first"), d.get("status"))'
grep -v "^   " $O/tail_new.log | tail -3
python - <<'PY'
import json
for l in open("gpurun_out/r03w/tail_profile_new.jsonl"):
    d=json.loads(l)
    for r in d["buckets"][10:26]: print(r["t_us"], r["rounds"], r["lanes_per_round"], r["waves_alive"], r["us_per_round_per_wave"])
PY
