# r04h: the driver's flags (--steps 20 --warmup 5) as one 20-frame launch (default) against 2 x 10 / 4 x 5 frames on two
# alternating streams (the tail of one launch under the ramp-up of the next); same for the default flags
set -u
O=gpurun_out/r04h; mkdir -p $O; rm -f $O/*
for rep in 1 2 3; do
for spec in "64 1" "10 2" "5 2" "10 1" "7 3"; do set -- $spec
  timeout 300 python bench.py --steps 20 --warmup 5 --batch $1 --streams $2 --no-cpu-baseline 2>/dev/null | python -c '
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(json.dumps({"steps": d["steps"], "batch": sys.argv[1], "streams": sys.argv[2], "ms": d["ms_per_step"], "kernel_ms": d["roofline"]["kernel_ms_per_frame"], "parity": d["parity"]["rgba8_equal"]}))' $1 $2 | tee -a $O/driverflags.jsonl
done; done
for spec in "64 1" "64 2" "32 2"; do set -- $spec
  timeout 300 python bench.py --batch $1 --streams $2 --no-cpu-baseline 2>/dev/null | python -c '
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(json.dumps({"steps": d["steps"], "batch": sys.argv[1], "streams": sys.argv[2], "ms": d["ms_per_step"], "kernel_ms": d["roofline"]["kernel_ms_per_frame"], "parity": d["parity"]["rgba8_equal"]}))' $1 $2 | tee -a $O/default.jsonl
done
