"""-m gpu: the lumisphere probe path on the HIP kernels against the CPU oracle.

Reference: the probe circle overlay of render_kernel (src/cuda/volrend.cu:100-134), the
coefficient fetch retrieve_cursor_lumisphere_kernel (volrend.cu:175-191) and the pre-kernel
launch of launch_renderer (volrend.cu:202-209).  Here: probe_overlay_kernel, probe_kernel and
vr_probe_coeffs (volrend_amd/csrc/vr_kernels.hip).  Bar: RGBA8 and fp32 accumulators bit-equal.
"""
import ctypes as C

import numpy as np
import pytest

from tests import common
from tests.common import ob

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return torch


def _oracle(tree, tr, w, h, f, fp_mode, offscreen=True, rgba_init=None, depth_init=None, **kw):
    return common.oracle_frame(tree, tr, w, h, f, fp_mode, offscreen=offscreen,
                               rgba_init=rgba_init, depth_init=depth_init, **kw)


def _gpu(torch, tree, tr, w, h, f, fp_mode, **kw):
    from tests.test_gpu_parity import gpu_frame
    return gpu_frame(torch, tree, tr, w, h, f, fp_mode, **kw)


def _assert_same(rgba_g, acc_g, rgba_o, acc_o, what=""):
    assert np.array_equal(rgba_g, rgba_o), f"{what}: {(rgba_g != rgba_o).any(-1).sum()} px differ"
    assert np.array_equal(acc_g.view(np.uint32), acc_o.view(np.uint32)), f"{what}: accumulators"


PROBE_CASES = [
    # (format, basis_dim, option kwargs)
    ("SH", 4, dict(probe=(0.1, 0.0, 0.2), probe_disp_size=30, basis_minmax=(0, 3))),
    ("SH", 9, dict(probe=(-0.2, 0.15, 0.05), probe_disp_size=41)),            # default {0, 24}
    ("SH", 16, dict(probe=(0.0, 0.0, 0.0), probe_disp_size=24, basis_minmax=(1, 9))),
    ("SH", 25, dict(probe=(0.3, -0.3, 0.3), probe_disp_size=50)),
    ("SH", 1, dict(probe=(0.05, 0.1, -0.1), probe_disp_size=20)),
    ("SG", 9, dict(probe=(0.1, 0.1, 0.1), probe_disp_size=33)),
    ("ASG", 4, dict(probe=(-0.1, 0.2, 0.0), probe_disp_size=28, basis_minmax=(0, 2))),
    ("RGBA", 0, dict(probe=(0.0, 0.1, 0.2), probe_disp_size=36)),
]


@pytest.mark.parametrize("fp_mode", [0, 1])
@pytest.mark.parametrize("fmt,basis_dim,kw", PROBE_CASES,
                         ids=[f"{c[0]}{c[1]}" for c in PROBE_CASES])
def test_probe_overlay_bit_exact(torch_cuda, fmt, basis_dim, kw, fp_mode):
    tree = common.small_scene(depth=5, basis_dim=basis_dim, fmt=fmt, seed=181 + basis_dim)
    tr, w, h, f = common.camera_for(pose_idx=5, size=88)
    kw = dict(kw, enable_probe=1)
    rgba_o, acc_o, _ = _oracle(tree, tr, w, h, f, fp_mode, **kw)
    kw_g = dict(kw, enable_probe=True)
    rgba_g, acc_g = _gpu(torch_cuda, tree, tr, w, h, f, fp_mode, **kw_g)
    _assert_same(rgba_g, acc_g, rgba_o, acc_o, f"{fmt}{basis_dim}")
    # the overlay really is in the frame: inside the circle alpha is 1 whatever the scene holds
    side = kw["probe_disp_size"]
    cy, cx = 5 + side // 2, w - side + side // 2 - 5
    assert acc_g[cy, cx, 3] == 1.0


def test_probe_overlay_wider_than_the_image(torch_cuda):
    """probe_disp_size + 5 larger than the frame: the corner square is clipped, not wrapped."""
    tree = common.small_scene(depth=4, basis_dim=4, seed=183)
    tr, w, h, f = common.camera_for(pose_idx=2, size=40)
    kw = dict(probe=(0.1, 0.2, 0.0), probe_disp_size=60, basis_minmax=(0, 3))
    for ww, hh in [(40, 40), (37, 21)]:
        rgba_o, acc_o, _ = _oracle(tree, tr, ww, hh, f, 0, enable_probe=1, **kw)
        rgba_g, acc_g = _gpu(torch_cuda, tree, tr, ww, hh, f, 0, enable_probe=True, **kw)
        _assert_same(rgba_g, acc_g, rgba_o, acc_o, f"{ww}x{hh}")


@pytest.mark.parametrize("fp_mode", [0, 1])
def test_probe_with_compositing(torch_cuda, fp_mode):
    """offscreen = 0: the probe circle ends with alpha 1 over whatever was in the target."""
    tree = common.small_scene(depth=5, basis_dim=9, seed=185)
    tr, w, h, f = common.camera_for(pose_idx=4, size=72)
    rng = np.random.default_rng(6)
    init = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
    depth = rng.uniform(2.0, 6.0, size=(h, w)).astype(np.float32)
    kw = dict(probe=(0.15, -0.1, 0.1), probe_disp_size=32, basis_minmax=(0, 8))
    rgba_o, acc_o, _ = _oracle(tree, tr, w, h, f, fp_mode, offscreen=False, rgba_init=init,
                               depth_init=depth, enable_probe=1, **kw)
    rgba_g, acc_g = _gpu(torch_cuda, tree, tr, w, h, f, fp_mode, offscreen=False, rgba_init=init,
                         depth_init=depth, enable_probe=True, **kw)
    _assert_same(rgba_g, acc_g, rgba_o, acc_o)


@pytest.mark.parametrize("fp_mode", [0, 1])
def test_probe_batch_and_tile_shards(torch_cuda, fp_mode):
    """Several poses in one launch (every frame gets its own overlay: the circle shows the
    lumisphere in that frame's camera axes), FRAME and COMPACT layouts of every rank of a
    tile-sharded launch, re-assembled."""
    torch = torch_cuda
    from volrend_amd import api
    tree = common.small_scene(depth=5, basis_dim=9, seed=187)
    w, h = 104, 80
    f = w * 1111.111 / 800.0
    poses = [common.camera_for(pose_idx=i, size=w)[0] for i in (1, 3, 6)]
    kw = dict(probe=(0.1, 0.05, 0.2), probe_disp_size=38)
    want = [_oracle(tree, tr, w, h, f, fp_mode, enable_probe=1, **kw) for tr in poses]
    t = api.N3Tree.from_synth(tree)
    cam = api.Camera(w, h, f, f)
    opts = api.RenderOptions(enable_probe=True, **kw)
    n = len(poses)
    # one launch, FRAME layout
    imgs = torch.zeros((n, h, w, 4), dtype=torch.uint8, device="cuda")
    accs = torch.zeros((n, h, w, 4), dtype=torch.float32, device="cuda")
    api.launch_renderer_batch(t, cam, poses, opts, list(imgs), None, True, accums=list(accs),
                              fp_mode=fp_mode)
    torch.cuda.synchronize()
    for i in range(n):
        _assert_same(imgs[i].cpu().numpy(), accs[i].cpu().numpy(), want[i][0], want[i][1],
                     f"batch frame {i}")
    # tile shards: the corner square straddles several tiles / ranks
    for world, tw, th in [(2, 104, 8), (3, 32, 16), (8, 8, 8)]:
        sh0 = api.TileShard(tw, th, 0, world, compact=True)
        nbytes = api.compact_bytes(w, h, sh0)
        gathered = torch.zeros((world, n, nbytes), dtype=torch.uint8, device="cuda")
        frames = torch.zeros((n, h, w, 4), dtype=torch.uint8, device="cuda")
        for r in range(world):
            api.launch_renderer_batch(t, cam, poses, opts, list(frames), None, True,
                                      shard=api.TileShard(tw, th, r, world, compact=False),
                                      fp_mode=fp_mode)
            api.launch_renderer_batch(t, cam, poses, opts, list(gathered[r]), None, True,
                                      shard=api.TileShard(tw, th, r, world, compact=True),
                                      fp_mode=fp_mode)
        out = torch.zeros((n, h, w, 4), dtype=torch.uint8, device="cuda")
        api.assemble_tiles_batch(out, gathered, n, w, h, sh0)
        torch.cuda.synchronize()
        for i in range(n):
            assert np.array_equal(frames[i].cpu().numpy(), want[i][0]), (world, tw, th, i)
            assert np.array_equal(out[i].cpu().numpy(), want[i][0]), (world, tw, th, i)
    t.free_device()


@pytest.mark.parametrize("fmt,basis_dim", [("SH", 16), ("SH", 9), ("SH", 25), ("RGBA", 0), ("SG", 4)])
def test_probe_coeffs_match_oracle(torch_cuda, fmt, basis_dim):
    """vr_probe_coeffs == retrieve_cursor_lumisphere_kernel (volrend.cu:175-191): the
    data_dim - 1 values of the leaf that holds opt.probe, as floats -- compared with the
    oracle and, where it was built, with the reference's own kernel compiled for the host."""
    torch = torch_cuda
    from volrend_amd import _abi, api
    tree = common.small_scene(depth=6, basis_dim=basis_dim, fmt=fmt, seed=189)
    t = api.N3Tree.from_synth(tree)
    th = ob.TreeHandle(tree)
    rng = np.random.default_rng(17)
    pts = [(0.0, 0.0, 0.0), (0.37, -0.41, 0.12)] + [tuple(rng.uniform(-0.6, 0.6, 3)) for _ in range(30)]
    L = _abi.lib()
    n = tree.data_dim - 1
    nonzero = 0
    for p in pts:
        opt_o = ob.default_options(enable_probe=1, probe=p)
        want = np.zeros(n, np.float32)
        ob.lib().or_probe_coeffs(C.byref(th.struct), C.byref(opt_o), want.ctypes.data)
        if ob.ref_lib() is not None:
            ref = np.zeros(n, np.float32)
            ob.ref_lib().ref_probe_coeffs(C.byref(th.struct), C.byref(opt_o), ref.ctypes.data)
            assert np.array_equal(ref.view(np.uint32), want.view(np.uint32))
        out = torch.full((n,), -7.0, dtype=torch.float32, device="cuda")
        o = api.RenderOptions(enable_probe=True, probe=p).to_c()
        _abi.check(L.vr_probe_coeffs(t.handle, C.byref(o), out.data_ptr(), None))
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), p
        nonzero += int(np.any(want != 0))
    assert nonzero > 3  # some probes landed in occupied leaves
    t.free_device()
