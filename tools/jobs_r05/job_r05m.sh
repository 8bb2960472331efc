# r05m: every distinct leaf of a shade round fetched once (-DVR_DEDUP=1, variant "dd": the items elect one of
# themselves per leaf through a hash table in LDS; the build still spills 16-48 bytes): parity, A/B on C1 / C2 / C3;
# the touch test with vr_touch_read
set -u
O=gpurun_out/r05m; mkdir -p $O; rm -f $O/*
timeout 300 python -m pytest tests/test_gpu_touch.py -x -q > $O/pytest_touch.log 2>&1; tail -1 $O/pytest_touch.log
VOLREND_HIP_LIB=$PWD/volrend_amd/libvolrend_hip_dd.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_chain.py -x -q --timeout 600 > $O/pytest_dd.log 2>&1; tail -1 $O/pytest_dd.log
timeout 900 python tools/quick_ab.py --config C1 --variants base,dd,base,dd --tunes "" --frames 64,20,1 --reps 5 --rotate --check --out $O/ab_c1.jsonl > $O/ab_c1.log 2>&1
timeout 900 python tools/quick_ab.py --config C2 --variants base,dd,base,dd --tunes "" --frames 8 --reps 3 --rotate --check --out $O/ab_c2.jsonl > $O/ab_c2.log 2>&1
timeout 900 python tools/quick_ab.py --config C3 --variants base,dd,base,dd --tunes "" --frames 16 --reps 4 --rotate --check --out $O/ab_c3.jsonl > $O/ab_c3.log 2>&1
python - <<PY
import json
for f in ("ab_c1","ab_c2","ab_c3"):
    rows=[json.loads(l) for l in open("$O/%s.jsonl"%f)]
    for fr in sorted({r["frames"] for r in rows}, reverse=True):
        for v in ("base","dd"):
            xs=[r for r in rows if r["frames"]==fr and r["variant"]==v]
            print(f, fr, v, [r["ms_per_frame_mean"] for r in xs], "same", all(r["same_as_first"] in (True,None) for r in xs))
PY
