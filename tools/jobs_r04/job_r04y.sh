# r04y: data for the next round on the final kernel: time of a dependent march round in an empty / full chip
# (tools/round_time_probe.py), time-resolved one-frame launch (tools/tail_profile.py, -DVR_TIMELINE=3 build),
# vector-L1 / LDS counters of C3 and C2
set -u
O=gpurun_out/r04y; mkdir -p $O; rm -f $O/*
timeout 600 python tools/round_time_probe.py --out $O/r04_round_time_probe.jsonl > $O/round_time.log 2>&1; tail -3 $O/round_time.log | cut -c1-300
VOLREND_HIP_LIB=$PWD/volrend_amd/libvolrend_hip_tl3.so VR_TIMELINE=3 timeout 600 python tools/tail_profile.py --frames 1 --out $O/r04_tail_profile.jsonl > $O/tail.log 2>&1; tail -2 $O/tail.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 900 python tools/measure_traffic.py --config C3 --batch 64 --groups tcp1 tcp2 sq1 sq3 tcc --out $O/r04_pmc_C3_tcp.json > /dev/null 2> $O/pmc_C3.log; tail -1 $O/pmc_C3.log
timeout 900 python tools/measure_traffic.py --config C2 --batch 64 --groups tcp1 tcp2 sq1 sq3 tcc --out $O/r04_pmc_C2_tcp.json > /dev/null 2> $O/pmc_C2.log; tail -1 $O/pmc_C2.log
