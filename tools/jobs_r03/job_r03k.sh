set -u
mkdir -p gpurun_out/r03k
O=gpurun_out/r03k
VR_UPLOAD_TIMING=1 timeout 900 python tools/upload_bench.py > $O/upload_bench.json 2> $O/upload_bench.log; echo "upload rc=$?"; cat $O/upload_bench.json
# the CLI's stderr is swallowed by upload_bench: run the CLI directly once for the phase prints
python - <<'PY'
import os, sys, subprocess
sys.path.insert(0, os.getcwd())
import bench
from volrend_amd import synth
t = bench.load_or_make_tree(synth, "C1", 0, lambda: None)
work = "/dev/shm/volrend_amd_upload2"; os.makedirs(work, exist_ok=True)
pose = synth.write_pose_dir(work, synth.make_poses(8)[:1], 64, 90.0)[0]
npz = os.path.join(work, "plain.npz"); synth.save_npz(t, npz)
for rep in range(2):
    r = subprocess.run(["volrend_amd/bin/volrend_headless", npz, pose, "-w", "64", "-h", "64"], capture_output=True, text=True, env=dict(os.environ, VR_UPLOAD_TIMING="1"))
    print(r.stderr[-1500:])
PY
