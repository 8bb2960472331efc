# r04i: full GPU suite (incl. the VolumeRenderer facade), split vs fused kernel at one frame per launch after the
# march-round diet, balance of the 8-rank tile shard on C3, upload timing
set -u
O=gpurun_out/r04i; mkdir -p $O; rm -f $O/*
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 900 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
timeout 600 python tools/quick_ab.py --config C1 --variants base,base,base --tunes "split=1;split=0" --frames 1,2 --reps 8 --rotate --check --out $O/split_c1.jsonl > $O/split_c1.log 2>&1
timeout 600 python tools/quick_ab.py --config C3 --variants base,base --tunes "split=1;split=0" --frames 1 --reps 6 --rotate --check --out $O/split_c3.jsonl > $O/split_c3.log 2>&1
cat $O/split_c1.jsonl $O/split_c3.jsonl | python -c '
import json,sys,collections
r=collections.OrderedDict()
for l in sys.stdin:
    d=json.loads(l); k=(d["config"], d["variant"], d["tune"], d["frames"]); r.setdefault(k,[]).append((d["ms_per_frame_mean"], d["ms_per_frame_min"], d.get("same_as_first")))
for k,v in r.items(): print(*k, " ".join("%.4f/%.4f"%(a,b) for a,b,_ in v), all(x[2] for x in v))'
timeout 900 python tools/shard_balance.py --config C3 --world 8 --tile-rows 8,16,32,64 --frames 64 --out $O/shard_balance.jsonl > $O/shard_balance.log 2>&1; tail -4 $O/shard_balance.log | cut -c1-400
timeout 900 python tools/shard_balance.py --config C3 --world 8 --tile-rows 8,16,32,64 --frames 16 --first-pose 5 --out $O/shard_balance.jsonl >> $O/shard_balance.log 2>&1
timeout 600 env VR_UPLOAD_TIMING=1 python tools/upload_bench.py > $O/upload.log 2>&1; tail -30 $O/upload.log
