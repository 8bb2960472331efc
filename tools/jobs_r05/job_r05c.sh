# r05c: (1) C3: where do the 4.45 GB of L2<->fabric traffic per frame come from?  PMC passes of the kernel as it is,
# without the record fetch (-DVR_ABLATE=4) and with every record out of a 128 KB window (=5): records vs everything
# else; the vector L1's view (tcp1) of the three builds.  (2) the CLI at 1 / 4 / 32 poses per launch on 1 / 2 streams,
# VolumeRenderer with overlapping render() calls (tests).
set -u
O=gpurun_out/r05c; mkdir -p $O; rm -f $O/*
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_renderer.py tests/test_gpu_cli.py tests/test_gpu_status.py -x -q --timeout 600 > $O/pytest.log 2>&1; tail -2 $O/pytest.log
timeout 900 python tools/cli_bench.py > $O/r05_cli_bench.json 2> $O/cli_bench.log; cut -c1-1500 $O/r05_cli_bench.json
cd /tmp && cd - > /dev/null
for v in base abl4 abl5; do
  L=$PWD/volrend_amd/libvolrend_hip.so; [ $v != base ] && L=$PWD/volrend_amd/libvolrend_hip_$v.so
  VOLREND_HIP_LIB=$L timeout 900 python tools/measure_traffic.py --config C3 --batch 64 --groups rdsize tcc tcp1 --out $O/traffic_C3_$v.json > /dev/null 2> $O/traffic_C3_$v.log; tail -1 $O/traffic_C3_$v.log
done
python - <<PY
import json
for v in ("base","abl4","abl5"):
    d=json.load(open("$O/traffic_C3_%s.json"%v))
    rc=d.get("raw_counters_per_launch",{})
    print(v, "GB/frame", round(d.get("read_bytes_per_frame",0)/1e9,3), "L2 hit", round(d.get("l2_hit_rate",0),3), "L2 req/frame M", round(d.get("l2_requests_per_frame",0)/1e6,1), "kernel ms", d.get("kernel_ms_under_pmc"), "TCP->TCC rd/frame M", round(rc.get("TCP_TCC_READ_REQ_sum",0)/64/1e6,1), "TCP accesses/frame M", round(rc.get("TCP_TOTAL_CACHE_ACCESSES_sum",0)/64/1e6,1))
PY
