#!/usr/bin/env python3
"""volrend_headless end to end on the C1 workload (200 poses, 800x800): the reference's own timing
lines with and without PNG output.  Run on the GPU box from the repo root; prints one JSON object."""
import json
import os
import re
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from volrend_amd import synth  # noqa: E402

CLI = os.path.join(ROOT, "volrend_amd", "bin", "volrend_headless")


def run(args):
    t0 = time.perf_counter()
    r = subprocess.run([CLI, *args], capture_output=True, text=True, timeout=900)
    wall = time.perf_counter() - t0
    if r.returncode != 0:
        raise RuntimeError(r.stderr[-2000:])
    ms = float(re.search(r"([0-9.]+) ms per frame", r.stdout).group(1))
    fps = float(re.search(r"([0-9.]+) fps", r.stdout).group(1))
    return {"ms_per_frame": round(ms, 4), "fps": round(fps, 1), "process_wall_s": round(wall, 2)}


def render_loop(npz, cfg, work):
    """volrend::VolumeRenderer::render() called frame after frame (the interactive path: composited,
    cleared frame, one launch per frame on the facade's two alternating frame streams), 200 frames
    over 8 orbit cameras: tests/cpp/renderer_check.cpp in its "loop" mode."""
    import numpy as np
    import tempfile
    exe = os.path.join(tempfile.mkdtemp(prefix="vr_loop_", dir="/tmp"), "renderer_check")  # (/dev/shm is noexec)
    subprocess.check_call(["make", "-C", ROOT, "host"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"),
                           "-I", "/opt/rocm/include", os.path.join(ROOT, "tests", "cpp", "renderer_check.cpp"),
                           os.path.join(ROOT, "volrend_amd", "libvolrend_host.a"), "-L", os.path.join(ROOT, "volrend_amd"),
                           "-lvolrend_hip", "-L/opt/rocm/lib", "-lamdhip64", "-lz", "-pthread",
                           "-Wl,-rpath," + os.path.join(ROOT, "volrend_amd"), "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    spec = [f"size {cfg['width']} {cfg['height']} {cfg['focal']!r} {cfg['focal']!r}", "background_brightness 1.0",
            "loop 200"]
    for i in range(8):
        a = 2.0 * np.pi * i / 8.0
        c = np.array([4.0 * np.cos(a) * np.cos(0.5), 4.0 * np.sin(a) * np.cos(0.5), 4.0 * np.sin(0.5)])
        b = c / np.linalg.norm(c)
        spec.append("cam " + " ".join(repr(float(x)) for x in (*c, *b)))
    sp = os.path.join(work, "loop_spec.txt")
    open(sp, "w").write("\n".join(spec) + "\n")
    r = subprocess.run([exe, npz, sp, os.path.join(work, "loop.raw")], capture_output=True, text=True, timeout=600)
    if r.returncode != 0:
        return {"error": (r.stdout + r.stderr)[-400:]}
    ms = float(re.search(r"loop_ms_per_frame ([0-9.]+)", r.stdout).group(1))
    return {"ms_per_frame": round(ms, 4), "fps": round(1000.0 / ms, 1), "frames": 200,
            "what": "VolumeRenderer::render() frame after frame, 800x800 of C1, composited over the cleared frame"}


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "C1"
    cfg = synth.CONFIGS[name]
    t = bench.load_or_make_tree(synth, name, 0, lambda: None)
    work = "/dev/shm/volrend_amd_cli"
    shutil.rmtree(work, ignore_errors=True)
    os.makedirs(work)
    npz = os.path.join(work, "tree.npz")
    synth.save_npz(t, npz)
    poses = synth.write_pose_dir(work, synth.make_poses(200), cfg["width"], cfg["focal"])
    common = [npz, *poses, "-i", os.path.join(work, "intrinsics.txt"), "-w", str(cfg["width"]),
              "-h", str(cfg["height"])]
    out = {"config": name, "poses": len(poses)}
    out["timing_only"] = run(common)  # defaults: --batch 64 (200 poses -> 4 equal launches of 50), --streams auto
    for b in (64, 32):
        for st in (1, 2):
            out[f"batch{b}_streams{st}"] = run(common + ["--batch", str(b), "--streams", str(st)])
    # the reference's own launch shape -- one pose per launch -- and small launches, on one stream and
    # on two alternating ones (--streams 0 = auto picks 2 below 48 poses per launch)
    for b in (1, 4):
        for st in (1, 2):
            out[f"batch{b}_streams{st}"] = run(common + ["--batch", str(b), "--streams", str(st)])
    # the native tile-shard path on this one-GPU box: 1 rank through RCCL (ncclCommInitAll + a
    # grouped self send/recv per launch), and a 2-rank REHEARSAL sharing the GPU (never a measurement
    # of scaling: both ranks compete for the same chip)
    out["gpus1_rccl_self"] = run(common + ["--batch", "64", "--gpus", "1"])
    out["gpus2_shared_gpu_rehearsal"] = run(common + ["--batch", "64", "--gpus", "2", "--share_gpu"])
    out["write_png"] = run(common + ["-o", os.path.join(work, "out")])
    out["volume_renderer_render_loop"] = render_loop(npz, cfg, work)
    out["png_files"] = len(os.listdir(os.path.join(work, "out")))
    shutil.rmtree(work, ignore_errors=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
