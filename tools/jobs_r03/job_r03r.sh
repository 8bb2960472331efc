set -u
mkdir -p gpurun_out/r03r
O=gpurun_out/r03r
: > $O/r03_batch_parity.jsonl
timeout 600 python tools/check_batch_parity.py C1 64 64 0 >> $O/r03_batch_parity.jsonl 2>/dev/null
timeout 600 python tools/check_batch_parity.py C1 64 64 1 >> $O/r03_batch_parity.jsonl 2>/dev/null
timeout 600 python tools/check_batch_parity.py C1 1 17 -1 >> $O/r03_batch_parity.jsonl 2>/dev/null
timeout 900 python tools/check_batch_parity.py C3 16 30 0 >> $O/r03_batch_parity.jsonl 2>/dev/null
timeout 900 python tools/check_batch_parity.py C3 16 30 1 >> $O/r03_batch_parity.jsonl 2>/dev/null
timeout 900 python tools/check_batch_parity.py C2 8 10 -1 >> $O/r03_batch_parity.jsonl 2>/dev/null
cat $O/r03_batch_parity.jsonl
