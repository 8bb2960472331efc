"""-m gpu: the kept C++ API -- volrend::N3Tree::open -> volrend::launch_renderer /
launch_renderer_batch (reference include/volrend/cuda/renderer_kernel.hpp:9-12) with every
RenderOptions field (render_options.hpp:11-53) off its default -- against the CPU oracle.
renderer.cpp maps the 13 option fields to the C ABI by hand: a swapped or dropped field shows
here (each case moves fields that change the picture on their own)."""
import os
import subprocess

import numpy as np
import pytest

from tests import common
from tests.common import ob
from volrend_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    subprocess.check_call(["make", "-C", ROOT, "host"], stdout=subprocess.DEVNULL)
    out = str(tmp_path_factory.mktemp("bin") / "gpu_check")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-D__HIP_PLATFORM_AMD__",
                           "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
                           os.path.join(ROOT, "tests", "cpp", "gpu_check.cpp"),
                           os.path.join(ROOT, "volrend_amd", "libvolrend_host.a"),
                           "-L", os.path.join(ROOT, "volrend_amd"), "-lvolrend_hip",
                           "-L/opt/rocm/lib", "-lamdhip64", "-lz", "-pthread",
                           "-Wl,-rpath," + os.path.join(ROOT, "volrend_amd"),
                           "-Wl,-rpath,/opt/rocm/lib", "-o", out])
    return out


CASES = {
    "march": dict(step_size=1e-3, sigma_thresh=0.5, stop_thresh=0.1, background_brightness=0.25),
    "bbox_basis_rot": dict(render_bbox=(0.1, 0.2, 0.05, 0.8, 0.9, 0.7), basis_minmax=(1, 5),
                           rot_dirs=(0.3, -0.2, 0.9)),
    "depth": dict(render_depth=1, step_size=2e-4),
    "probe": dict(enable_probe=1, probe=(0.1, -0.2, 0.3), probe_disp_size=40),
    "defaults_with_inert_fields": dict(show_grid=1, grid_max_depth=7),  # carried, ignored by the kernel
}


@pytest.mark.parametrize("mode", ["single", "batch"])
@pytest.mark.parametrize("case", list(CASES))
def test_cpp_launch_renderer_matches_oracle(exe, tmp_path, case, mode):
    tree = common.small_scene(depth=6, basis_dim=9, seed=500)
    poses = [synth.c2w_to_transform(p) for p in synth.make_poses(3)]
    opts = dict(CASES[case])
    oracle_opts = {k: v for k, v in opts.items() if k not in ("show_grid", "grid_max_depth")}
    w, h = 120, 88
    f = w * 1111.111 / 800.0
    # the spec carries every field; the oracle is given the ones it knows
    got = _run(exe, tmp_path, tree, poses, w, h, f, mode, opts, oracle_opts)
    assert (got[..., :3] != 255).any()


def _run(exe, tmp_path, tree, poses, w, h, f, mode, spec_opts, oracle_opts):
    npz = str(tmp_path / "t.npz")
    synth.save_npz(tree, npz, compressed=False)
    spec = [f"size {w} {h} {f!r} {f!r}", f"mode {mode}"]
    for k, v in spec_opts.items():
        vals = v if isinstance(v, (tuple, list)) else (v,)
        spec.append(k + " " + " ".join(repr(float(x)) if isinstance(x, float) else str(int(x))
                                       for x in vals))
    for p in poses:
        spec.append("pose " + " ".join(repr(float(x)) for x in p))
    sp = str(tmp_path / "spec.txt")
    open(sp, "w").write("\n".join(spec) + "\n")
    raw = str(tmp_path / "out.raw")
    r = subprocess.run([exe, npz, sp, raw], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    got = np.fromfile(raw, dtype=np.uint8).reshape(len(poses), h, w, 4)
    for i, p in enumerate(poses):
        want, _, _ = common.oracle_frame(tree, p, w, h, f, **oracle_opts)
        assert np.array_equal(got[i], want), f"{mode} frame {i}: differs from the oracle"
    return got
