# r05p: does the ray order move C3's L2<->fabric traffic (every record line crosses the fabric 2.1x per frame, every
# brick line ~8x)?  PMC passes (read sizes, L2) under the ray-order knobs: frame-major instead of frame-minor ids,
# 4x4 / 8x8 super-blocks of 8x8-pixel blocks, one ray queue for the chip instead of one per XCD
set -u
O=gpurun_out/r05p; mkdir -p $O; rm -f $O/*
export TMPDIR=/tmp
i=0
for tune in "" "frame_group=1" "super_block=4" "frame_group=1,super_block=4" "frame_group=8,super_block=8" "xcd_queues=0"; do
  i=$((i+1))
  if [ -z "$tune" ]; then BA=""; else BA="--tune $tune"; fi
  timeout 600 python tools/measure_traffic.py --config C3 --batch 64 --groups rdsize tcc --bench-args "$BA" --out $O/traffic_C3_$i.json > /dev/null 2> $O/traffic_$i.log
  python - <<PY
import json
d=json.load(open("$O/traffic_C3_$i.json"))
print("tune=[$tune]", "GB/frame", round(d.get("read_bytes_per_frame",0)/1e9,3), "L2 hit", round(d.get("l2_hit_rate",0),3), "L2 req/frame M", round(d.get("l2_requests_per_frame",0)/1e6,1), "kernel ms per 64-frame launch", d.get("kernel_ms_under_pmc"))
d["tune"]="$tune"
json.dump(d,open("$O/traffic_C3_$i.json","w"))
PY
done
