# r05d: bricks stored in 4 x 4 x 2 line blocks instead of 1 x 4 x 8 slabs (-DVR_BRICK_BLOCKED=1, variant "blk"):
# parity of the variant, A/B on C1 / C3 / C2, fabric traffic of C3 with it
set -u
O=gpurun_out/r05d; mkdir -p $O; rm -f $O/*
export TMPDIR=/tmp
VOLREND_HIP_LIB=$PWD/volrend_amd/libvolrend_hip_blk.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_chain.py tests/test_gpu_fullsize.py -x -q --timeout 600 > $O/pytest_blk.log 2>&1; tail -2 $O/pytest_blk.log
timeout 900 python tools/quick_ab.py --config C3 --variants base,blk,base,blk --tunes "" --frames 16 --reps 4 --rotate --check --out $O/ab_c3.jsonl > $O/ab_c3.log 2>&1; cut -c1-230 $O/ab_c3.log | grep variant
timeout 900 python tools/quick_ab.py --config C1 --variants base,blk,base,blk --tunes "" --frames 64,20,1 --reps 4 --rotate --check --out $O/ab_c1.jsonl > $O/ab_c1.log 2>&1; cut -c1-230 $O/ab_c1.log | grep variant
timeout 900 python tools/quick_ab.py --config C2 --variants base,blk --tunes "" --frames 8 --reps 3 --rotate --check --out $O/ab_c2.jsonl > $O/ab_c2.log 2>&1; cut -c1-230 $O/ab_c2.log | grep variant
VOLREND_HIP_LIB=$PWD/volrend_amd/libvolrend_hip_blk.so timeout 900 python tools/measure_traffic.py --config C3 --batch 64 --groups rdsize tcc --out $O/traffic_C3_blk.json > /dev/null 2> $O/traffic_C3_blk.log; tail -1 $O/traffic_C3_blk.log
python - <<PY
import json
d=json.load(open("$O/traffic_C3_blk.json"))
print("blk GB/frame", round(d.get("read_bytes_per_frame",0)/1e9,3), "L2 hit", round(d.get("l2_hit_rate",0),3), "L2 req/frame M", round(d.get("l2_requests_per_frame",0)/1e6,1), d.get("kernel_ms_under_pmc"))
PY
