# r05q: the whole GPU suite with the blocked brick order forced on for every tree (VR_BRICK_BLOCKED=1: the production
# flavours' second template instance across all formats / options / sizes), and with the drain rule off / at its
# widest (VR_DRAIN_FLUSH=0 / 64)
set -u
O=gpurun_out/r05q; mkdir -p $O; rm -f $O/*
VR_BRICK_BLOCKED=1 timeout 1500 python -m pytest tests -m gpu -x -q --timeout 900 > $O/pytest_blocked.log 2>&1; echo "blocked: $(tail -1 $O/pytest_blocked.log)"
VR_DRAIN_FLUSH=64 timeout 1500 python -m pytest tests -m gpu -x -q --timeout 900 > $O/pytest_drain64.log 2>&1; echo "drain 64: $(tail -1 $O/pytest_drain64.log)"
VR_DRAIN_FLUSH=0 timeout 1500 python -m pytest tests -m gpu -x -q --timeout 900 -k "chain or parity or fullsize" > $O/pytest_drain0.log 2>&1; echo "drain 0: $(tail -1 $O/pytest_drain0.log)"
