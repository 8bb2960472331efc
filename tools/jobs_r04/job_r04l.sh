# r04l: refill threshold x march steps per pass on the current kernel (runtime knobs; the refill batch averages 37
# lanes at the defaults 24 / 16, i.e. 29 % of the march lanes idle)
set -u
O=gpurun_out/r04l; mkdir -p $O; rm -f $O/*
T=""
for r in 24 16 12 8; do for m in 16 8 4; do T="$T;refill_min=$r,march_max=$m"; done; done
T="${T#;}"
timeout 900 python tools/quick_ab.py --config C1 --variants base --tunes "$T;refill_min=24,march_max=16" --frames 64,20 --reps 5 --rotate --check --out $O/ab_c1.jsonl > $O/ab_c1.log 2>&1
timeout 900 python tools/quick_ab.py --config C3 --variants base --tunes "refill_min=24,march_max=16;refill_min=16,march_max=8;refill_min=12,march_max=8;refill_min=8,march_max=4;refill_min=24,march_max=16" --frames 16 --reps 4 --rotate --check --out $O/ab_c3.jsonl > $O/ab_c3.log 2>&1
cat $O/ab_c1.jsonl $O/ab_c3.jsonl | python -c '
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d["config"], d["tune"], d["frames"], d["ms_per_frame_mean"], d["ms_per_frame_min"], d.get("same_as_first"))'
