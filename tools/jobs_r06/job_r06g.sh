# r06g: random launch shapes (image size, poses per launch, tile shard, raygen workgroup size, queues) against the
# oracle: the suite's 10 seeds, then 400.
set -u
O=gpurun_out/r06g; mkdir -p $O; rm -f $O/*
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "random_launch_shapes or raygen_workgroup" --timeout 800 > $O/pytest10.log 2>&1; tail -2 $O/pytest10.log
VR_SHAPE_SEEDS=400 timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -k "random_launch_shapes" --timeout 1400 > $O/pytest400.log 2>&1; tail -3 $O/pytest400.log
