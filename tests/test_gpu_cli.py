"""-m gpu: the volrend_headless CLI end to end (tree.npz + pose/*.txt + intrinsics.txt in,
PNGs + the reference's two result lines out), checked against the CPU oracle."""
import os
import re
import subprocess

import numpy as np
import pytest

from tests import common
from tests.common import ob
from volrend_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "volrend_amd", "bin", "volrend_headless")


@pytest.fixture(scope="module")
def cli():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    if not os.path.exists(CLI):
        subprocess.check_call(["make", "-C", ROOT, "cli"], stdout=subprocess.DEVNULL)
    return CLI


def test_headless_cli_matches_oracle(cli, tmp_path):
    from PIL import Image
    tree = common.small_scene(depth=5, basis_dim=16, seed=401)
    npz = str(tmp_path / "tree.npz")
    synth.save_npz(tree, npz, compressed=True)
    poses = synth.make_poses(8)[:5]
    w, h, focal = 96, 72, 130.0
    paths = synth.write_pose_dir(str(tmp_path), poses[:3], w, focal)
    multi = str(tmp_path / "pose" / "multi.txt")  # two stacked 4x4 -> multi_000000/1
    np.savetxt(multi, np.concatenate([poses[3], poses[4]], axis=0))
    out_dir = str(tmp_path / "out" / "frames")
    r = subprocess.run([cli, npz, *paths, multi, "-w", str(w), "-h", str(h), "-i",
                        str(tmp_path / "intrinsics.txt"), "-o", out_dir, "--batch", "4"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().splitlines()
    assert re.fullmatch(r"\d+\.\d{10} ms per frame", lines[-3])
    assert re.fullmatch(r"\d+\.\d{10} fps", lines[-2])
    assert "INFO: Use NeRF camera convention" in r.stdout
    names = ["0000", "0001", "0002", "multi_000000", "multi_000001"]
    th = ob.TreeHandle(tree)
    for name, pose in zip(names, poses):
        img = np.asarray(Image.open(os.path.join(out_dir, name + ".png")))
        cam = ob.make_camera(synth.c2w_to_transform(pose), w, h, focal)
        want, _, _ = ob.render(th, cam, ob.default_options(), want_accum=False)
        assert img.shape == (h, w, 4)
        assert np.array_equal(img, want), name


def test_headless_cli_flags(cli, tmp_path):
    from PIL import Image
    tree = common.small_scene(depth=4, basis_dim=4, seed=402)
    npz = str(tmp_path / "t.npz")
    synth.save_npz(tree, npz)
    pose = synth.make_poses(8)[2]
    # OpenCV convention file: flip y/z columns, ask the CLI to flip back with -r
    cv = pose.copy()
    cv[:, 1] *= -1
    cv[:, 2] *= -1
    pf = str(tmp_path / "cv.txt")
    np.savetxt(pf, cv[:3])  # 3x4 form
    out_dir = str(tmp_path / "o")
    r = subprocess.run([cli, npz, pf, "-r", "--width", "64", "--height", "64", "--fx", "90",
                        "--bg", "0.25", "-s", "1e-3", "-e", "0.05", "-a", "0.5", "--scale", "0.5",
                        "-o", out_dir, "--fp", "fma"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "INFO: Use OpenCV camera convention" in r.stdout
    img = np.asarray(Image.open(os.path.join(out_dir, "cv.png")))
    th = ob.TreeHandle(tree)
    cam = ob.make_camera(synth.c2w_to_transform(pose), 32, 32, 45.0)  # --scale 0.5
    opt = ob.default_options(background_brightness=0.25, step_size=1e-3, stop_thresh=0.05,
                             sigma_thresh=0.5)
    want, _, _ = ob.render(th, cam, opt, ob.FP_FMA, want_accum=False)
    assert np.array_equal(img, want)
    # no poses -> the reference's warning and exit code 1
    r = subprocess.run([cli, npz], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "No camera poses specified" in r.stderr


def test_headless_cli_quantised_tree(cli, tmp_path):
    """A compress_octree.py-style file goes through the C++ loader's device decode."""
    from PIL import Image
    tree = common.small_scene(depth=4, basis_dim=9, seed=403)
    npz = str(tmp_path / "q.npz")
    common.write_quantised_npz(tree, npz, n_retain=1)
    pose = synth.make_poses(8)[5]
    paths = synth.write_pose_dir(str(tmp_path), [pose], 64, 90.0)
    out_dir = str(tmp_path / "o")
    r = subprocess.run([cli, npz, *paths, "-w", "64", "-h", "64", "--fx", "90", "-o", out_dir],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "Decoding quantized colors" in r.stderr + r.stdout
    img = np.asarray(Image.open(os.path.join(out_dir, "0000.png")))
    cam = ob.make_camera(synth.c2w_to_transform(pose), 64, 64, 90.0)
    want, _, _ = ob.render(ob.TreeHandle(tree), cam, ob.default_options(), want_accum=False)
    assert np.array_equal(img, want)


def test_headless_cli_tile_shard(cli, tmp_path):
    """--gpus N: interleaved screen tiles, one tree replica per rank (vr_tree_clone), gather to
    the root, batch assembly -- must write the same PNGs as the oracle.  On this one-GPU box:
    --gpus 1 goes through RCCL (ncclCommInitAll + a grouped self send/recv), --gpus 2/3 with
    --share_gpu rehearse the N-rank bookkeeping on one device; with >= 2 GPUs visible the real
    2-rank RCCL path runs as well."""
    import torch
    from PIL import Image
    tree = common.small_scene(depth=5, basis_dim=9, seed=404)
    npz = str(tmp_path / "tree.npz")
    synth.save_npz(tree, npz)
    poses = synth.make_poses(8)[:7]
    w, h, focal = 100, 76, 120.0     # neither a multiple of the tile size
    paths = synth.write_pose_dir(str(tmp_path), poses, w, focal)
    th = ob.TreeHandle(tree)
    want = [ob.render(th, ob.make_camera(synth.c2w_to_transform(p), w, h, focal),
                      ob.default_options(), want_accum=False)[0] for p in poses]
    runs = [(["--gpus", "1"], "RCCL"),
            (["--gpus", "2", "--share_gpu"], "REHEARSAL"),
            (["--gpus", "3", "--share_gpu", "--tile", "16"], "REHEARSAL")]
    if torch.cuda.device_count() >= 2:
        runs.append((["--gpus", "2"], "RCCL"))
    for k, (flags, label) in enumerate(runs):
        out_dir = str(tmp_path / f"o{k}")
        r = subprocess.run([cli, npz, *paths, "-w", str(w), "-h", str(h), "--fx", str(focal), "-o",
                            out_dir, "--batch", "3", *flags], capture_output=True, text=True,
                           timeout=600)
        assert r.returncode == 0, r.stderr + r.stdout
        assert "screen-tile shard" in r.stdout and label in r.stdout, r.stdout
        assert re.search(r"\d+\.\d{10} fps", r.stdout)
        for i in range(len(poses)):
            img = np.asarray(Image.open(os.path.join(out_dir, f"{i:04d}.png")))
            assert np.array_equal(img, want[i]), (flags, i)
    if torch.cuda.device_count() < 2:  # more ranks than GPUs without --share_gpu: a clean error
        r = subprocess.run([cli, npz, paths[0], "-w", "64", "-h", "64", "--gpus", "2"],
                           capture_output=True, text=True, timeout=120)
        assert r.returncode == 1 and "need more GPUs" in r.stderr
