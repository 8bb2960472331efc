# r04j: validation after the clean-up (split kernel removed, hooks header, RenderOptions order, VolumeRenderer facade,
# ADVICE fixes): full GPU suite; upload phases of three fresh processes; bench line with its repeats
set -u
O=gpurun_out/r04j; mkdir -p $O; rm -f $O/*
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 900 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
VR_UPLOAD_TIMING=1 timeout 600 python tools/upload_bench.py > $O/upload.json 2> $O/upload.log; python - <<PY
import json
d=json.load(open("$O/upload.json"))
for k in ("plain","quant_device_decode"):
    print(k, d[k]["all_upload_ms"])
    for ph in d[k].get("all_phases") or []: print("   ", ph)
PY
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench20.json 2> $O/bench20.log; python -c "
import json; d=json.load(open('$O/bench20.json')); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['repeats'], d['sched'], d['parity']['rgba8_equal'])"
