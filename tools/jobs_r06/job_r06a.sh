# r06a: C3 -- where does the time go?  The table C1 has had since round 4 (DESIGN "Where the time goes"):
# timing ablations of the round-5 kernel on C3 at 64 frames per launch in ONE process (quick_ab), then the PMC
# passes (read sizes, L2, SQ issue / wait) of every build: base, no record fetch (4), records out of a 128 KB
# window (5), march only (6), march without the brick load (9).
set -u
O=gpurun_out/r06a; mkdir -p $O; rm -f $O/*
export TMPDIR=/tmp
timeout 900 python tools/quick_ab.py --config C3 --variants base,abl4,abl5,abl6,abl9 --tunes "" --frames 64 --reps 3 --rotate --out $O/C3_ablation_ab.jsonl > $O/ab.log 2>&1; cat $O/C3_ablation_ab.jsonl | cut -c1-200
for v in base abl4 abl5 abl6 abl9; do
  L=$PWD/volrend_amd/libvolrend_hip.so; [ $v != base ] && L=$PWD/volrend_amd/libvolrend_hip_$v.so
  VOLREND_HIP_LIB=$L timeout 900 python tools/measure_traffic.py --config C3 --batch 64 --groups rdsize tcc sq1 sq2 --out $O/traffic_C3_$v.json > /dev/null 2> $O/traffic_C3_$v.log; tail -1 $O/traffic_C3_$v.log
done
python - <<PY
import json
for v in ("base","abl4","abl5","abl6","abl9"):
    try:
        d=json.load(open("$O/traffic_C3_%s.json"%v))
    except Exception as e:
        print(v, "ERR", e); continue
    rc=d.get("raw_counters_per_launch",{})
    print(v, "GB/frame", round(d.get("read_bytes_per_frame",0)/1e9,3), "L2 hit", round(d.get("l2_hit_rate",0) or 0,3), "kernel ms", d.get("kernel_ms_under_pmc"), "wait", d.get("wave_wait_fraction"), "valu/frame M", round((d.get("valu_insts_per_frame") or 0)/1e6,1))
PY
