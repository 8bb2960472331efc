# r06m: frame_group (poses per ray-order group) -- finer sweep, default interleaved between the candidates to rule out drift;
# C1 at 20 and 64 frames per launch, C3 and C2 at 64 / 32.
set -u
O=gpurun_out/r06m; mkdir -p $O; rm -f $O/*
export TMPDIR=/tmp
timeout 900 python tools/quick_ab.py --config C1 --variants base --tunes ";frame_group=8;;frame_group=6;frame_group=10;;frame_group=12;frame_group=5;;frame_group=8;frame_group=2" --frames 20,64 --reps 6 --rotate --check --out $O/frame_group.jsonl 2>/dev/null | cut -c1-175
timeout 900 python tools/quick_ab.py --config C3 --variants base --tunes ";frame_group=8;;frame_group=12;frame_group=6;;frame_group=8" --frames 64 --reps 4 --rotate --check --out $O/frame_group.jsonl 2>/dev/null | cut -c1-175
timeout 900 python tools/quick_ab.py --config C2 --variants base --tunes ";frame_group=8;;frame_group=4;frame_group=16" --frames 32 --reps 3 --rotate --check --out $O/frame_group.jsonl 2>/dev/null | cut -c1-175
