# r06k: cost-ordered ray queues with a PERFECT prediction (the same poses launched again and again: the cost map is the
# previous run of the very same frames): is the concept worth anything on the chip?
set -u
O=gpurun_out/r06k; mkdir -p $O; rm -f $O/*
export TMPDIR=/tmp
timeout 900 python tools/quick_ab.py --config C1 --variants base --tunes "cost_order=0;cost_order=1;cost_order=0;cost_order=1" --frames 1,2,4,20 --reps 10 --check --out $O/cost_order_same_poses.jsonl 2>/dev/null | cut -c1-260
