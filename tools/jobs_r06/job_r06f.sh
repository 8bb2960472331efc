# r06f: does capping the render kernel of a SMALL launch below the chip's wave slots (16 / 18 / 12 of 20 waves per CU)
# leave room for the next launch on the other stream -- its ray generation and its first render waves -- while this
# one is still in its busy phase?  Sustained ms per frame on two streams (and one, for the cost of the cap).
set -u
O=gpurun_out/r06f; mkdir -p $O; rm -f $O/*
export TMPDIR=/tmp
for t in "" "waves_per_cu=18" "waves_per_cu=16" "waves_per_cu=12"; do
  timeout 300 python tools/stream_overlap.py --frames 1,2,4,8 --streams 2,1 --tune "$t" --out $O/stream_overlap_waves.jsonl 2>/dev/null | cut -c1-190
done
for t in "waves_per_cu=16"; do for fs in "1 2" "4 2"; do set -- $fs
  timeout 300 python tools/overlap_trace.py --frames $1 --streams $2 --tune "$t" --out $O/overlap_trace_waves.jsonl > $O/trace_$1_$2.log 2>&1; tail -1 $O/trace_$1_$2.log | cut -c1-420
done; done
