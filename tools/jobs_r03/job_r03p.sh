set -u
mkdir -p gpurun_out/r03p
O=gpurun_out/r03p
timeout 900 python -m pytest tests/test_abi.py tests/test_gpu_parity.py tests/test_gpu_cli.py tests/test_gpu_streams.py -q -x --timeout 600 -m "gpu or not gpu" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
timeout 900 python tools/upload_bench.py > $O/upload_bench.json 2> $O/upload_bench.log; cat $O/upload_bench.json
python - <<'PY'
import os, sys, subprocess
sys.path.insert(0, os.getcwd())
import bench
from volrend_amd import synth
t = bench.load_or_make_tree(synth, "C1", 0, lambda: None)
work = "/dev/shm/volrend_amd_upload2"; os.makedirs(work, exist_ok=True)
pose = synth.write_pose_dir(work, synth.make_poses(8)[:1], 64, 90.0)[0]
npz = os.path.join(work, "plain.npz"); synth.save_npz(t, npz)
for rep in range(3):
    r = subprocess.run(["volrend_amd/bin/volrend_headless", npz, pose, "-w", "64", "-h", "64"], capture_output=True, text=True, env=dict(os.environ, VR_UPLOAD_TIMING="1"))
    print(r.stderr[-900:])
PY
