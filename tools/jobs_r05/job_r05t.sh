# r05t: eight colour items per ray instead of four (-DVR_OWNER_Q=8, variant "q8": the ring positions of a ray's
# outstanding items in a register pair; fits SH16 without scratch since round 5 freed three registers): parity,
# A/B on C1 at 64 / 20 / 4 / 2 / 1 frames per launch, C2, C3
set -u
O=gpurun_out/r05t; mkdir -p $O; rm -f $O/*
VOLREND_HIP_LIB=$PWD/volrend_amd/libvolrend_hip_q8.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_chain.py tests/test_gpu_fullsize.py -x -q --timeout 600 > $O/pytest_q8.log 2>&1; tail -1 $O/pytest_q8.log
timeout 900 python tools/quick_ab.py --config C1 --variants base,q8,base,q8 --tunes "" --frames 64,20,4,2,1 --reps 5 --rotate --check --out $O/ab_c1.jsonl > $O/ab_c1.log 2>&1
timeout 900 python tools/quick_ab.py --config C2 --variants base,q8,base,q8 --tunes "" --frames 8 --reps 3 --rotate --check --out $O/ab_c2.jsonl > $O/ab_c2.log 2>&1
timeout 900 python tools/quick_ab.py --config C3 --variants base,q8,base,q8 --tunes "" --frames 16,2 --reps 4 --rotate --check --out $O/ab_c3.jsonl > $O/ab_c3.log 2>&1
python - <<PY
import json
for f in ("ab_c1","ab_c2","ab_c3"):
    rows=[json.loads(l) for l in open("$O/%s.jsonl"%f)]
    for fr in sorted({r["frames"] for r in rows}, reverse=True):
        for v in ("base","q8"):
            xs=[r for r in rows if r["frames"]==fr and r["variant"]==v]
            print(f, fr, v, [r["ms_per_frame_mean"] for r in xs], "same", all(r["same_as_first"] in (True,None) for r in xs))
PY
