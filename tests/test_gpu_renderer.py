"""-m gpu: volrend::VolumeRenderer, the reference's renderer facade
(include/volrend/renderer.hpp:11-42, src/cuda_renderer.cpp:83-195) rebuilt without OpenGL --
render() / set() / clear() / resize() / get_backend(), public camera / options -- against the CPU
oracle's compositing path (offscreen = 0: the ray march composited over the RGBA8 already in the
frame, rays ended by the R32F depth image; volrend.cu:142-147,152-165)."""
import os
import subprocess

import numpy as np
import pytest

from tests import common
from volrend_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    subprocess.check_call(["make", "-C", ROOT, "host"], stdout=subprocess.DEVNULL)
    out = str(tmp_path_factory.mktemp("bin") / "renderer_check")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-D__HIP_PLATFORM_AMD__",
                           "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
                           os.path.join(ROOT, "tests", "cpp", "renderer_check.cpp"),
                           os.path.join(ROOT, "volrend_amd", "libvolrend_host.a"),
                           "-L", os.path.join(ROOT, "volrend_amd"), "-lvolrend_hip",
                           "-L/opt/rocm/lib", "-lamdhip64", "-lz", "-pthread",
                           "-Wl,-rpath," + os.path.join(ROOT, "volrend_amd"),
                           "-Wl,-rpath,/opt/rocm/lib", "-o", out])
    return out


CAMS = [(-3.0, 0.4, 2.2, -0.75, 0.1, 0.55), (2.5, -2.5, 1.0, 0.66, -0.66, 0.26), (0.3, 3.6, 0.8, 0.05, 0.97, 0.2)]


@pytest.mark.parametrize("underlay", [False, True, "stream"],
                         ids=["cleared_frame", "mesh_underlay", "mesh_underlay_produced_on_a_stream"])
@pytest.mark.parametrize("brightness", [1.0, 0.3])
def test_volume_renderer_matches_oracle_compositing(exe, tmp_path, underlay, brightness):
    """underlay == "stream": the underlay images are produced on a stream of the caller's right
    before every render() and overwritten with junk right after it, with no host synchronisation
    in between -- set_underlay(..., producer_stream) must order render()'s copies behind the
    producer's work AND the producer's next writes behind the copies (the two frames alternate
    between two streams, so "the stream of the last render()" orders nothing)."""
    basis_dim = 9
    tree = common.small_scene(depth=6, basis_dim=basis_dim, seed=501)
    w, h = 136, 96
    f = w * 1111.111 / 800.0
    npz = str(tmp_path / "t.npz")
    synth.save_npz(tree, npz, compressed=False)
    spec = [f"size {w} {h} {f!r} {f!r}", f"background_brightness {brightness!r}"]
    if underlay and brightness < 1.0:
        spec.append("burst 3")  # render() three times before the frame is read: both frame streams in flight
    rng = np.random.default_rng(7)
    if underlay:  # what a mesh pass would leave behind: colour + a depth that ends part of the rays
        rgba0 = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
        rgba0[..., 3] = 255
        depth0 = np.full((h, w), 1e9, dtype=np.float32)
        y0, y1, x0, x1 = h // 4, 3 * h // 4, w // 3, 2 * w // 3
        depth0[y0:y1, x0:x1] = rng.uniform(2.0, 5.0, size=(y1 - y0, x1 - x0)).astype(np.float32)
        rgba0.tofile(str(tmp_path / "u_rgba.raw"))
        depth0.tofile(str(tmp_path / "u_depth.raw"))
        spec.append(f"underlay {tmp_path / 'u_rgba.raw'} {tmp_path / 'u_depth.raw'}")
        if underlay == "stream":
            spec.append("underlay_stream 1")
    else:  # glClear: (b, b, b, 1) as RGBA8, depth 1e9 (cuda_renderer.cpp:85-92)
        c = int(np.floor(min(max(brightness, 0.0), 1.0) * 255.0 + 0.5))
        rgba0 = np.empty((h, w, 4), dtype=np.uint8)
        rgba0[..., :3] = c
        rgba0[..., 3] = 255
        depth0 = np.full((h, w), 1e9, dtype=np.float32)
    for c in CAMS:
        spec.append("cam " + " ".join(repr(float(x)) for x in c))
    sp = str(tmp_path / "spec.txt")
    open(sp, "w").write("\n".join(spec) + "\n")
    raw = str(tmp_path / "out.raw")
    r = subprocess.run([exe, npz, sp, raw], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l and not l.startswith("INFO:")]
    transforms = [np.array([float(x) for x in l.split()[1:]], dtype=np.float32)
                  for l in lines if l.startswith("transform")]
    assert len(transforms) == len(CAMS)
    tail = lines[-1].split()
    # set() narrowed the range to the tree's basis; the backend says what it is
    assert tail[:3] == ["basis_minmax", "0", str(basis_dim - 1)] and tail[-1] == "HIP"
    got = np.fromfile(raw, dtype=np.uint8).reshape(len(CAMS) + 1, h, w, 4)
    hit = 0
    for i, tr in enumerate(transforms):
        want, _, _ = common.oracle_frame(tree, tr, w, h, f, offscreen=False, rgba_init=rgba0,
                                         depth_init=depth0, background_brightness=brightness,
                                         basis_minmax=(0, basis_dim - 1))
        assert np.array_equal(got[i], want), f"frame {i}: differs from the oracle's compositing path"
        hit += int((want != rgba0).any())
    assert hit == len(CAMS), "every camera must see the volume"
    # clear(): render() without a tree leaves the cleared frame / the underlay
    assert np.array_equal(got[-1], rgba0)
