# r05g: (1) SH25 records on a 256-byte stride (two whole lines per record) against the 160-byte stride: time and
# fabric traffic of C2; (2) knob sweep on the final kernel (refill_min x march_max, drain_flush) at 20 / 64 frames
# per launch; (3) the CLI with its new defaults (batch 64, equal launches, auto streams)
set -u
O=gpurun_out/r05g; mkdir -p $O; rm -f $O/*
export TMPDIR=/tmp
timeout 900 python tools/quick_ab.py --config C2 --variants base,sh25s256,base,sh25s256 --tunes "" --frames 8 --reps 3 --rotate --check --out $O/ab_c2_stride.jsonl > $O/ab_c2_stride.log 2>&1; cut -c1-200 $O/ab_c2_stride.log | grep variant
for v in base sh25s256; do
  L=$PWD/volrend_amd/libvolrend_hip.so; [ $v != base ] && L=$PWD/volrend_amd/libvolrend_hip_$v.so
  VOLREND_HIP_LIB=$L timeout 900 python tools/measure_traffic.py --config C2 --batch 16 --groups rdsize tcc --out $O/traffic_C2_$v.json > /dev/null 2> $O/traffic_C2_$v.log; tail -1 $O/traffic_C2_$v.log
done
python - <<PY
import json
for v in ("base","sh25s256"):
    d=json.load(open("$O/traffic_C2_%s.json"%v))
    print(v, "GB/frame", round(d.get("read_bytes_per_frame",0)/1e9,3), "L2 hit", round(d.get("l2_hit_rate",0),3), "kernel ms", d.get("kernel_ms_under_pmc"))
PY
timeout 900 python tools/quick_ab.py --config C1 --variants base --tunes ";refill_min=16;refill_min=24;march_max=8;march_max=16;refill_min=16,march_max=8;refill_min=24,march_max=16;drain_flush=8;drain_flush=32;" --frames 64,20 --reps 5 --rotate --out $O/ab_knobs.jsonl > $O/ab_knobs.log 2>&1
python - <<PY
import json
rows=[json.loads(l) for l in open("$O/ab_knobs.jsonl")]
for r in rows: print(r["tune"] or "default", r["frames"], r["ms_per_frame_mean"], r["ms_per_frame_min"])
PY
timeout 900 python tools/cli_bench.py > $O/r05_cli_bench.json 2> $O/cli_bench.log; cut -c1-1600 $O/r05_cli_bench.json
