# r06b: (1) kernel trace of one-frame launches on two alternating streams: when do prepare / ray generation /
# render of launch k + 1 start relative to the render kernel of launch k?  (2) instruction-issue priority by
# thinness (s_setprio, -DVR_PRIO=1/2/3): lone launches and the two-stream sustained rate.
set -u
O=gpurun_out/r06b; mkdir -p $O; rm -f $O/*
export TMPDIR=/tmp
for fs in "1 2" "1 1" "4 2"; do set -- $fs
  timeout 300 python tools/overlap_trace.py --frames $1 --streams $2 --out $O/overlap_trace.jsonl > $O/trace_$1_$2.log 2>&1; tail -1 $O/trace_$1_$2.log | cut -c1-900
done
for v in base prio1 prio2 prio3; do
  timeout 300 python tools/stream_overlap.py --frames 1,4 --streams 1,2 --variant $v --out $O/stream_overlap_prio.jsonl 2>/dev/null | cut -c1-220
done
timeout 600 python tools/quick_ab.py --config C1 --variants base,prio1,prio2,prio3 --tunes "" --frames 64,20,1 --reps 4 --rotate --check --out $O/prio_ab.jsonl 2>/dev/null | cut -c1-200
