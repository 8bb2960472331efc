set -u
mkdir -p gpurun_out/r03h
O=gpurun_out/r03h
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_chain.py -x -q --timeout 300 > $O/pytest_chain.log 2>&1; echo "chain rc=$?"; tail -3 $O/pytest_chain.log
timeout 900 python tools/quick_ab.py --config C1 --variants base --tunes "split=1,refill_min=12;split=0,refill_min=24" --frames 64,20,1 --reps 3 --check --out $O/ab_c1.jsonl > $O/ab_c1.log 2>&1
timeout 900 python tools/quick_ab.py --config C3 --variants base --tunes "split=1,refill_min=12;split=0,refill_min=24" --frames 16 --reps 3 --check --out $O/ab_c3.jsonl > $O/ab_c3.log 2>&1
cat $O/*.jsonl | python -c '
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d["config"], d["variant"], d["tune"], d["frames"], d["ms_per_frame_mean"], d["ms_per_frame_min"], d.get("same_as_first"), d.get("status"))'
timeout 900 python tools/measure_traffic.py --config C1 --split 1 --groups sq1 sq2 --out $O/pmc_C1_split1.json > /dev/null 2> $O/pmc_C1_split1.log
python - <<PY
import json
d=json.load(open("$O/pmc_C1_split1.json"))
for k in ("valu_insts_per_frame","salu_insts_per_frame","lds_insts_per_frame","valu_lane_utilisation","wave_wait_fraction","valu_issue_cycles_per_simd_over_kernel_cycles_at_2p4GHz","kernel_ms_under_pmc","failed_groups"):
    print("split1", k, d.get(k))
PY
