# r05h: the CLI at 64 / 32 / 4 / 1 poses per launch on one and two render streams (which default?)
set -u
O=gpurun_out/r05h; mkdir -p $O; rm -f $O/*
timeout 900 python tools/cli_bench.py > $O/r05_cli_bench.json 2> $O/cli_bench.log; python - <<PY
import json
d=json.load(open("$O/r05_cli_bench.json"))
for k,v in d.items():
    if isinstance(v,dict): print(k, v["ms_per_frame"], v["fps"])
PY
timeout 900 python tools/cli_bench.py > $O/r05_cli_bench_2.json 2> $O/cli_bench2.log; python - <<PY
import json
d=json.load(open("$O/r05_cli_bench_2.json"))
for k,v in d.items():
    if isinstance(v,dict): print(k, v["ms_per_frame"], v["fps"])
PY
