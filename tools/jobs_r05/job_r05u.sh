# r05u: the kernel as it is, compiled with other scheduler settings of the AMDGPU backend (no source change, so
# bit-identical by construction -- checked anyway): trk = -mllvm --amdgpu-use-amdgpu-trackers=1 (SH16 at 88-89
# VGPRs instead of 96), ilp = --amdgpu-sched-strategy=max-ilp, clause = max-memory-clause, trkilp = both
set -u
O=gpurun_out/r05u; mkdir -p $O; rm -f $O/*
V=base,trk,ilp,clause,trkilp
timeout 1000 python tools/quick_ab.py --config C1 --variants $V,$V --tunes "" --frames 64,20,4,1 --reps 4 --rotate --check --out $O/ab_c1.jsonl > $O/ab_c1.log 2>&1
timeout 600 python tools/quick_ab.py --config C3 --variants $V,$V --tunes "" --frames 16 --reps 4 --rotate --check --out $O/ab_c3.jsonl > $O/ab_c3.log 2>&1
timeout 600 python tools/quick_ab.py --config C2 --variants $V,$V --tunes "" --frames 8 --reps 3 --rotate --check --out $O/ab_c2.jsonl > $O/ab_c2.log 2>&1
python - <<PY
import json
for f in ("ab_c1","ab_c3","ab_c2"):
    try: rows=[json.loads(l) for l in open("$O/%s.jsonl"%f)]
    except Exception as e: print(f, e); continue
    for fr in sorted({r["frames"] for r in rows}, reverse=True):
        for v in "$V".split(","):
            xs=[r for r in rows if r["frames"]==fr and r["variant"]==v]
            print(f, fr, v, [r["ms_per_frame_mean"] for r in xs], "same", all(r["same_as_first"] in (True,None) for r in xs))
PY
