# r05o: do small launches want other scheduling knobs than batches?  (waves per CU, refill threshold, march pass
# length at 1 / 2 / 4 frames per launch; all runtime knobs)
set -u
O=gpurun_out/r05o; mkdir -p $O; rm -f $O/*
timeout 900 python tools/quick_ab.py --config C1 --variants base --tunes ";waves_per_cu=10;waves_per_cu=14;waves_per_cu=16;waves_per_cu=0,refill_min=8;refill_min=12;refill_min=32;refill_min=20,march_max=4;march_max=8;march_max=24;march_max=12,drain_flush=24;drain_flush=16,chunk_max=64;chunk_max=256;chunk_max=4096" --frames 1,2,4 --reps 6 --rotate --out $O/ab_small.jsonl > $O/ab_small.log 2>&1
python - <<PY
import json
rows=[json.loads(l) for l in open("$O/ab_small.jsonl")]
tunes=[]
for r in rows:
    if r["tune"] not in tunes: tunes.append(r["tune"])
for t in tunes:
    print((t or "default").ljust(40), " ".join("%d: %.4f/%.4f" % (r["frames"], r["ms_per_frame_mean"], r["ms_per_frame_min"]) for r in rows if r["tune"]==t))
PY
