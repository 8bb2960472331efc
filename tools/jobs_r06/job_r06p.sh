# r06p: padded EXEC, second build (pad lanes sample where the first marching ray samples: no misses of their own)
set -u
O=gpurun_out/r06p; mkdir -p $O; rm -f $O/*
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_chain.py -x -q --timeout 800 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
timeout 900 python tools/quick_ab.py --config C1 --variants nopad,base,nopad,base --tunes "" --frames 1,4,20,64 --reps 6 --rotate --check --out $O/pad_ab.jsonl 2>/dev/null | cut -c1-200
timeout 600 python tools/round_time_probe.py --out $O/round_time_probe.jsonl 2>/dev/null | cut -c1-300
VOLREND_HIP_LIB=$PWD/volrend_amd/libvolrend_hip_nopad.so timeout 600 python tools/round_time_probe.py --out $O/round_time_probe_nopad.jsonl 2>/dev/null | cut -c1-300
