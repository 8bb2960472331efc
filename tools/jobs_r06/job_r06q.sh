# r06q: wide seeded sweeps on the final sources, both FP models: 3000 random configurations (kernel == oracle, RGBA8 and
# fp32 accumulators) and 1500 random launch shapes.
set -u
O=gpurun_out/r06q; mkdir -p $O; rm -f $O/*
export TMPDIR=/tmp
VR_SWEEP_SEEDS=3000 timeout 3000 python -m pytest tests/test_gpu_chain.py -q -x --timeout 2900 -k "random_sweep" > $O/seed_sweep_chain.log 2>&1; tail -1 $O/seed_sweep_chain.log
VR_SHAPE_SEEDS=1500 timeout 3000 python -m pytest tests/test_gpu_parity.py -q -x --timeout 2900 -k "random_launch_shapes" > $O/shape_sweep.log 2>&1; tail -1 $O/shape_sweep.log
