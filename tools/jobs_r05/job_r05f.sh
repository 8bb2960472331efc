# r05f: brick order as a template parameter of the production flavours (no branch in the march round), the lane's
# ray id in LDS, SH9 at seven waves per SIMD without scratch: full GPU suite; new default against the library of the
# drain-flush commit ("prev") on C1 / C2 / C3, same process
set -u
O=gpurun_out/r05f; mkdir -p $O; rm -f $O/*
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -x -q --timeout 900 > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
timeout 900 python tools/quick_ab.py --config C1 --variants prev,base,prev,base,prev,base --tunes "" --frames 64,20,1 --reps 5 --rotate --check --out $O/ab_c1.jsonl > $O/ab_c1.log 2>&1
timeout 900 python tools/quick_ab.py --config C3 --variants prev,base,prev,base --tunes "" --frames 16 --reps 4 --rotate --check --out $O/ab_c3.jsonl > $O/ab_c3.log 2>&1
timeout 900 python tools/quick_ab.py --config C2 --variants prev,base,prev,base --tunes "" --frames 8 --reps 3 --rotate --check --out $O/ab_c2.jsonl > $O/ab_c2.log 2>&1
python - <<PY
import json
for f in ("ab_c1","ab_c3","ab_c2"):
    rows=[json.loads(l) for l in open("$O/%s.jsonl"%f)]
    for fr in sorted({r["frames"] for r in rows}, reverse=True):
        for v in ("prev","base"):
            xs=[r for r in rows if r["frames"]==fr and r["variant"]==v]
            print(f, fr, v, [r["ms_per_frame_mean"] for r in xs], "same", all(r["same_as_first"] in (True,None) for r in xs))
PY
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench20.json 2> $O/bench20.log; python -c "
import json; d=json.load(open('$O/bench20.json')); print('bench20', d['ms_per_step'], d.get('ms_per_step_cold'), d['repeats']['ms_per_step'], d['roofline'].get('model'), d['parity']['rgba8_equal'])"
