# r04d: where do the waves wait?  PMC passes of the round-3 fused kernel on C1 (64 frames per launch): vector L1 (TCP)
# stall / latency counters, LDS, scalar
set -u
O=gpurun_out/r04d; mkdir -p $O; rm -f $O/*
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 1500 python tools/measure_traffic.py --config C1 --batch 64 --groups tcp1 tcp2 tcp3 tcp4 sq1 sq2 sq3 sq4 tcc --out $O/pmc_C1.json > $O/pmc_C1.log 2>&1
echo rc=$?; tail -5 $O/pmc_C1.log
python - <<PY
import json
d=json.load(open("$O/pmc_C1.json"))
print(d.get("failed_groups"))
for k,v in d["raw_counters_per_launch"].items(): print(k, "%.4g"%v)
PY
