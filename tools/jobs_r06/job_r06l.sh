# r06l: poses per ray-order group (frame_group: ids are block-major, frame-minor INSIDE a group of G consecutive poses):
# C3 and C1 at 64 frames per launch, C1 at 20, in one process each.
set -u
O=gpurun_out/r06l; mkdir -p $O; rm -f $O/*
export TMPDIR=/tmp
timeout 900 python tools/quick_ab.py --config C3 --variants base --tunes ";frame_group=4;frame_group=8;frame_group=16;frame_group=32;frame_group=1" --frames 64 --reps 3 --rotate --check --out $O/frame_group.jsonl 2>/dev/null | cut -c1-200
timeout 900 python tools/quick_ab.py --config C1 --variants base --tunes ";frame_group=4;frame_group=8;frame_group=16;frame_group=32" --frames 64,20 --reps 4 --rotate --check --out $O/frame_group.jsonl 2>/dev/null | cut -c1-200
