# r04r: ray-order / chunk knobs on the final kernel (runtime knobs): frames per ray-order group, chunk size, super-blocks
set -u
O=gpurun_out/r04r; mkdir -p $O; rm -f $O/*
T=";frame_group=1;frame_group=4;frame_group=16;chunk_max=1024;chunk_max=16384;super_block=2;super_block=4;xcd_queues=0;"
timeout 900 python tools/quick_ab.py --config C1 --variants base --tunes "$T" --frames 64,20 --reps 5 --rotate --out $O/ab_c1.jsonl > $O/ab_c1.log 2>&1
timeout 900 python tools/quick_ab.py --config C3 --variants base --tunes ";frame_group=1;frame_group=4;chunk_max=1024;super_block=2;" --frames 16 --reps 4 --rotate --out $O/ab_c3.jsonl > $O/ab_c3.log 2>&1
cat $O/ab_c1.jsonl $O/ab_c3.jsonl | python -c '
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d["config"], repr(d["tune"]), d["frames"], d["ms_per_frame_mean"], d["ms_per_frame_min"])'
