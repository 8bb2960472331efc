set -u
mkdir -p gpurun_out/r03m
O=gpurun_out/r03m
export TMPDIR=/tmp
for uc in 0 1; do
VR_RECORDS_UC=$uc timeout 900 python tools/quick_ab.py --config C3 --variants base --tunes "split=0;split=0,records_nt=0" --frames 16 --reps 3 --check --out $O/ab_c3_uc$uc.jsonl > $O/ab_c3_uc$uc.log 2>&1
VR_RECORDS_UC=$uc timeout 900 python tools/quick_ab.py --config C1 --variants base --tunes "split=0;split=0,records_nt=1" --frames 64 --reps 3 --check --out $O/ab_c1_uc$uc.jsonl > $O/ab_c1_uc$uc.log 2>&1
done
for f in $O/ab_*.jsonl; do echo $f; python -c '
import json,sys
for l in open(sys.argv[1]):
    d=json.loads(l); print("  ", d["config"], d["tune"], d["frames"], d["ms_per_frame_mean"], d["ms_per_frame_min"], d.get("same_as_first"), d.get("status"))' $f; done
VR_RECORDS_UC=1 timeout 900 python tools/measure_traffic.py --config C3 --split 0 --groups rdsize tcc --out $O/pmc_C3_uc1.json > /dev/null 2> $O/pmc_C3_uc1.log
python - <<PY
import json
d=json.load(open("$O/pmc_C3_uc1.json"))
for k in ("read_bytes_per_frame","frac_requests_128B","l2_hit_rate","kernel_ms_under_pmc","failed_groups","raw_counters_per_launch"):
    print("uc1", k, d.get(k))
PY
