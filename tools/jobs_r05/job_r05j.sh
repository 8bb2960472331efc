# r05j: with the lookup lines cheaper (blocked bricks), is the non-temporal record stream still the right policy
# for C3 (and the cached one for C2)?  And the lookup geometry of C3 again with the blocked order.  Plus the CLI set
# with the VolumeRenderer render() loop.
set -u
O=gpurun_out/r05j; mkdir -p $O; rm -f $O/*
timeout 600 python tools/quick_ab.py --config C3 --variants base --tunes ";records_nt=0;records_nt=1;" --frames 16 --reps 4 --rotate --check --out $O/ab_c3_nt.jsonl 2>&1 | cut -c1-220 | grep variant
timeout 600 python tools/quick_ab.py --config C2 --variants base --tunes ";records_nt=1;records_nt=0" --frames 8 --reps 3 --rotate --check --out $O/ab_c2_nt.jsonl 2>&1 | cut -c1-220 | grep variant
for g in "6 3" "8 3" "7 3"; do set -- $g
  VR_TOP_LEVELS=$1 VR_BRICK_LEVELS=$2 timeout 600 python tools/quick_ab.py --config C3 --variants base --tunes "" --frames 16 --reps 4 --rotate --out $O/ab_c3_geom.jsonl 2>&1 | cut -c1-220 | grep variant | sed -e "s/^/G0=$1 BL=$2 /"
done
timeout 900 python tools/cli_bench.py > $O/r05_cli_bench.json 2> $O/cli_bench.log; python - <<PY
import json
d=json.load(open("$O/r05_cli_bench.json"))
print(" ".join("%s=%s" % (k, v.get("ms_per_frame")) for k,v in d.items() if isinstance(v,dict)))
print(d.get("volume_renderer_render_loop"))
PY
