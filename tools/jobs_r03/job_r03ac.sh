# split kernel: alive / stopped derived from t < tmax in the march wave too (tools/patches/r04_split_kernel_scalar_diet.patch,
# NOT applied to the product: no GPU budget left for the measurement set): pictures and timing against the product library
set -u
mkdir -p gpurun_out/r03ac
O=gpurun_out/r03ac
rm -f $O/*
timeout 300 python tools/quick_ab.py --config C1 --variants base,sd,base,sd --tunes "split=1" --frames 1,4,64 --reps 6 --rotate --check --out $O/ab_c1.jsonl > $O/ab_c1.log 2>&1
cat $O/ab_c1.jsonl | python -c '
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d["config"], d["variant"], d["tune"], d["frames"], d["ms_per_frame_mean"], d["ms_per_frame_min"], d.get("same_as_first"), d.get("status"))'
