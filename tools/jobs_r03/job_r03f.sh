set -u
mkdir -p gpurun_out/r03f
O=gpurun_out/r03f
export TMPDIR=/tmp
for sp in 1 0; do
  timeout 900 python tools/measure_traffic.py --config C1 --split $sp --groups sq1 sq2 --out $O/pmc_C1_split$sp.json > /dev/null 2> $O/pmc_C1_split$sp.log
  tail -3 $O/pmc_C1_split$sp.log
  python - <<PY
import json
d=json.load(open("$O/pmc_C1_split$sp.json"))
for k in ("valu_insts_per_frame","salu_insts_per_frame","lds_insts_per_frame","vmem_read_insts_per_frame","valu_lane_utilisation","wave_wait_fraction","valu_issue_cycles_per_simd_over_kernel_cycles_at_2p4GHz","kernel_ms_under_pmc","failed_groups"):
    print("split$sp", k, d.get(k))
print(d.get("raw_counters_per_launch"))
PY
done
