"""-m gpu: the sample guard's status word reaches every product caller.

vr_render* only enqueues, so a launch whose rays hit the guard (vr_kernels.hip: a wave that
marches `max_iter` rounds without retiring a ray cuts what is still marching) cannot fail the
call: the sticky word of vr_tree_status carries it, and every render loop of the product checks
it once its last launch has run -- volrend_headless (message + exit 1, the reference's
abort-on-error convention, src/cuda/common.cu:8-21), volrend::VolumeRenderer::read_frame and
TileShardRenderer::sync (exceptions), bench.py.  The guard is lowered from its 2^22 rounds to a
handful (tuning key `max_iter` / VR_MAX_ITER) so that ordinary rays trip it."""
import os
import subprocess

import numpy as np
import pytest

from tests import common
from volrend_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "volrend_amd", "bin", "volrend_headless")


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return torch


def test_status_word_through_the_api(torch_cuda):
    torch = torch_cuda
    from volrend_amd import api
    tree = common.small_scene(depth=6, basis_dim=9, seed=941)
    t = api.N3Tree.from_synth(tree)
    tr, w, h, f = common.camera_for(pose_idx=3, size=96)
    cam = api.Camera(w, h, f, f)
    cam.transform = np.asarray(tr, np.float32)
    img = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
    want, _, _ = common.oracle_frame(tree, tr, w, h, f, 0)
    api.launch_renderer(t, cam, api.RenderOptions(), img, None, None, True)
    torch.cuda.synchronize()
    assert t.status() == 0 and np.array_equal(img.cpu().numpy(), want)
    t.set_tuning(max_iter=2)  # every wave whose rays outlive its first pass of march rounds trips
    api.launch_renderer(t, cam, api.RenderOptions(), img, None, None, True)
    torch.cuda.synchronize()
    assert t.status() & 1, "rays were cut by the guard but the status word says nothing"
    assert not np.array_equal(img.cpu().numpy(), want), "cut rays cannot give the right picture"
    assert t.status(reset=True) & 1 and t.status() == 0  # sticky until reset
    # the same word read ON the launch's stream (vr_tree_status_on: that stream alone is waited for)
    side = torch.cuda.Stream()
    api.launch_renderer(t, cam, api.RenderOptions(), img, None, side, True)
    assert t.status(stream=side) & 1, "the read on the launch's own stream must see its bit"
    assert t.status(reset=True, stream=side) & 1 and t.status(stream=side) == 0 and t.status() == 0
    t.set_tuning(max_iter=1 << 22)
    api.launch_renderer(t, cam, api.RenderOptions(), img, None, None, True)
    torch.cuda.synchronize()
    assert t.status() == 0 and np.array_equal(img.cpu().numpy(), want)
    t.free_device()


def _scene_files(tmp_path, n_poses=3):
    tree = common.small_scene(depth=6, basis_dim=9, seed=942)
    npz = str(tmp_path / "tree.npz")
    synth.save_npz(tree, npz, compressed=False)
    w, h, focal = 96, 80, 130.0
    paths = synth.write_pose_dir(str(tmp_path), synth.make_poses(8)[:n_poses], w, focal)
    return npz, paths, w, h


@pytest.mark.parametrize("gpus", [0, 2], ids=["single", "tile_shard_rehearsal"])
def test_headless_cli_fails_loudly_on_a_guard_trip(torch_cuda, tmp_path, gpus):
    if not os.path.exists(CLI):
        subprocess.check_call(["make", "-C", ROOT, "cli"], stdout=subprocess.DEVNULL)
    npz, paths, w, h = _scene_files(tmp_path)
    cmd = [CLI, npz, *paths, "-w", str(w), "-h", str(h), "-i", str(tmp_path / "intrinsics.txt")]
    if gpus:
        cmd += ["--gpus", str(gpus), "--share_gpu"]
    ok = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert ok.returncode == 0, ok.stderr
    bad = subprocess.run(cmd, capture_output=True, text=True, timeout=300,
                         env={**os.environ, "VR_MAX_ITER": "2"})
    assert bad.returncode == 1, (bad.returncode, bad.stdout, bad.stderr)
    assert "sample guard" in bad.stderr and "ERROR" in bad.stderr
    # step_size = 0 (the reference spins forever, rt_core.cuh:108): refused before any launch
    zero = subprocess.run(cmd + ["-s", "0"], capture_output=True, text=True, timeout=300)
    assert zero.returncode == 1 and "step_size" in zero.stderr


def test_volume_renderer_fails_loudly_on_a_guard_trip(torch_cuda, tmp_path):
    subprocess.check_call(["make", "-C", ROOT, "host"], stdout=subprocess.DEVNULL)
    exe = str(tmp_path / "renderer_check")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-D__HIP_PLATFORM_AMD__",
                           "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
                           os.path.join(ROOT, "tests", "cpp", "renderer_check.cpp"),
                           os.path.join(ROOT, "volrend_amd", "libvolrend_host.a"),
                           "-L", os.path.join(ROOT, "volrend_amd"), "-lvolrend_hip",
                           "-L/opt/rocm/lib", "-lamdhip64", "-lz", "-pthread",
                           "-Wl,-rpath," + os.path.join(ROOT, "volrend_amd"),
                           "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    npz, _, w, h = _scene_files(tmp_path, 1)
    f = w * 1111.111 / 800.0
    sp = str(tmp_path / "spec.txt")
    open(sp, "w").write(f"size {w} {h} {f!r} {f!r}\nbackground_brightness 1.0\n"
                        "cam -3.0 0.4 2.2 -0.75 0.1 0.55\n")
    raw = str(tmp_path / "out.raw")
    ok = subprocess.run([exe, npz, sp, raw], capture_output=True, text=True, timeout=300)
    assert ok.returncode == 0, ok.stdout + ok.stderr
    bad = subprocess.run([exe, npz, sp, raw], capture_output=True, text=True, timeout=300,
                         env={**os.environ, "VR_MAX_ITER": "2"})
    assert bad.returncode == 3 and "EXCEPTION" in bad.stdout and "sample guard" in bad.stdout, \
        (bad.returncode, bad.stdout, bad.stderr)
    # step_size = 0 through the facade: launch_renderer refuses it (the reference would hang)
    open(sp, "a").write("step_size 0\n")
    zero = subprocess.run([exe, npz, sp, raw], capture_output=True, text=True, timeout=300)
    assert zero.returncode == 3 and "step_size" in zero.stdout, zero.stdout + zero.stderr
