"""The oracle's pin: the C restatement (oracle/vr_oracle.c, strict mode) against the
reference's OWN device code compiled for the host (oracle/_ref/libvolrend_ref.so, built
by oracle/ref_build/Makefile from /root/reference/src/cuda/volrend.cu and the headers it
includes).  Bit equality of the fp32 accumulators and of RGBA8 is required.

Needs the _ref library: built here by __graft_entry__.build() (reference mount present);
it travels with the snapshot to the GPU box.  Skipped only if neither exists."""
import ctypes as C

import numpy as np
import pytest

from tests import common
from tests.common import ob

ref_available = ob.ref_lib() is not None
pytestmark = pytest.mark.skipif(not ref_available, reason="reference host build unavailable")


def both(tree, tr, w, h, f, ndc=None, **opt_kw):
    th = ob.TreeHandle(tree, ndc=ndc)
    cam = ob.make_camera(tr, w, h, f)
    opt = ob.default_options(**opt_kw)
    rgba_o, acc_o, _ = ob.render(th, cam, opt, ob.FP_STRICT)
    rgba_r = ob.ref_render(th, cam, opt)
    return rgba_o, acc_o, rgba_r, (th, cam, opt)


@pytest.mark.parametrize("fmt,basis_dim", [("SH", 1), ("SH", 4), ("SH", 9), ("SH", 16), ("SH", 25),
                                            ("RGBA", 0), ("SG", 9), ("SG", 25), ("ASG", 4)])
def test_formats_bit_exact(fmt, basis_dim):
    tree = common.small_scene(depth=5, basis_dim=basis_dim, fmt=fmt, seed=100 + basis_dim)
    tr, w, h, f = common.camera_for(pose_idx=2, size=64)
    rgba_o, acc_o, rgba_r, (th, cam, opt) = both(tree, tr, w, h, f)
    assert np.array_equal(rgba_o, rgba_r)
    acc_r = ob.ref_trace(th, cam, opt)
    assert np.array_equal(acc_o.view(np.uint32), acc_r.view(np.uint32))
    assert (rgba_o[..., :3] != 255).any(), "scene must not be empty"


@pytest.mark.parametrize("kw", [
    dict(step_size=1e-3, sigma_thresh=0.5, stop_thresh=0.1, background_brightness=0.25),
    dict(render_bbox=(0.1, 0.2, 0.0, 0.8, 0.9, 0.7)),
    dict(basis_minmax=(1, 5)),
    dict(rot_dirs=(0.3, -0.2, 0.9)),
    dict(render_depth=1),
    dict(step_size=1e-5, stop_thresh=1e-4),
    dict(background_brightness=0.0, sigma_thresh=5.0),
])
def test_options_bit_exact(kw):
    tree = common.small_scene(depth=6, basis_dim=9, seed=141)
    tr, w, h, f = common.camera_for(pose_idx=3, size=56)
    rgba_o, acc_o, rgba_r, (th, cam, opt) = both(tree, tr, w, h, f, **kw)
    assert np.array_equal(rgba_o, rgba_r)
    if not kw.get("rot_dirs"):  # ref_trace takes the un-rotated prologue of render_kernel too
        pass
    acc_r = ob.ref_trace(th, cam, opt)
    assert np.array_equal(acc_o.view(np.uint32), acc_r.view(np.uint32))


def test_ndc_bit_exact():
    tree = common.small_scene(depth=5, basis_dim=4, seed=151)
    tr = np.array([1, 0, 0, 0, 1, 0, 0, 0, 1, 0.05, -0.02, 0.3], dtype=np.float32)
    rgba_o, acc_o, rgba_r, (th, cam, opt) = both(tree, tr, 96, 72, 80.0, ndc=(96.0, 72.0, 80.0))
    assert np.array_equal(rgba_o, rgba_r)
    assert np.array_equal(acc_o.view(np.uint32), ob.ref_trace(th, cam, opt).view(np.uint32))


def test_compositing_over_existing_frame():
    tree = common.small_scene(depth=5, basis_dim=9, seed=171)
    tr, w, h, f = common.camera_for(pose_idx=4, size=48)
    rng = np.random.default_rng(5)
    init = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
    depth = rng.uniform(2.0, 6.0, size=(h, w)).astype(np.float32)
    th = ob.TreeHandle(tree)
    cam = ob.make_camera(tr, w, h, f)
    opt = ob.default_options()
    rgba_o, _, _ = ob.render(th, cam, opt, offscreen=False, rgba_init=init, depth_init=depth)
    rgba_r = ob.ref_render(th, cam, opt, offscreen=False, rgba_init=init, depth_init=depth)
    assert np.array_equal(rgba_o, rgba_r)


@pytest.mark.parametrize("fmt,bd,minmax,offscreen", [
    ("SH", 4, (0, 3), True), ("SH", 16, (2, 11), True), ("SH", 25, (0, 24), True),
    ("SG", 9, (0, 8), True), ("ASG", 4, (0, 3), False), ("RGBA", 0, (0, 24), True)])
def test_probe_overlay(fmt, bd, minmax, offscreen):
    tree = common.small_scene(depth=5, basis_dim=bd, fmt=fmt, seed=181)
    tr, w, h, f = common.camera_for(pose_idx=5, size=72)
    th = ob.TreeHandle(tree)
    cam = ob.make_camera(tr, w, h, f)
    # basis_minmax narrowed to the coefficients that exist (what VolumeRenderer::set does,
    # src/cuda_renderer.cpp:176-177); upstream reads out of bounds otherwise
    opt = ob.default_options(enable_probe=1, probe=(0.1, 0.0, 0.2), probe_disp_size=30,
                             basis_minmax=minmax)
    kw = {}
    if not offscreen:
        rng = np.random.default_rng(8)
        kw = dict(offscreen=False, rgba_init=rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8),
                  depth_init=rng.uniform(2.0, 6.0, size=(h, w)).astype(np.float32))
    rgba_o, _, _ = ob.render(th, cam, opt, **kw)
    rgba_r = ob.ref_render(th, cam, opt, **kw)
    assert np.array_equal(rgba_o, rgba_r)
    # probe coefficients: retrieve_cursor_lumisphere_kernel (volrend.cu:175-191)
    n = tree.data_dim - 1
    a, b = np.zeros(n, np.float32), np.zeros(n, np.float32)
    ob.lib().or_probe_coeffs(C.byref(th.struct), C.byref(opt), a.ctypes.data)
    ob.ref_lib().ref_probe_coeffs(C.byref(th.struct), C.byref(opt), b.ctypes.data)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_point_query_and_basis():
    tree = common.small_scene(depth=6, basis_dim=25, seed=191)
    th = ob.TreeHandle(tree)
    rng = np.random.default_rng(3)
    L, R = ob.lib(), ob.ref_lib()
    for _ in range(500):
        p = rng.uniform(-0.1, 1.1, size=3).astype(np.float32)
        a = (C.c_float * 3)(*p)
        b = (C.c_float * 3)(*p)
        ca, cb, depth = C.c_float(), C.c_float(), C.c_int()
        la = L.or_query(C.byref(th.struct), C.byref(a), C.byref(ca), C.byref(depth))
        lb = R.ref_query(C.byref(th.struct), C.byref(b), C.byref(cb))
        assert la == lb and ca.value == cb.value and list(a) == list(b)
        d = rng.standard_normal(3).astype(np.float32)
        d /= np.linalg.norm(d)
        dv = (C.c_float * 3)(*d)
        oa, obuf = (C.c_float * 25)(), (C.c_float * 25)()
        L.or_basis(C.byref(th.struct), C.byref(dv), ob.FP_STRICT, C.byref(oa))
        R.ref_basis(C.byref(th.struct), C.byref(dv), C.byref(obuf))
        assert list(oa) == list(obuf)


def test_libm_expf_build_is_close():
    """With glibc's expf instead of the deterministic one the reference build differs by
    rounding noise only: bounds the cost of fixing the exp implementation."""
    tree = common.small_scene(depth=6, basis_dim=16, seed=201)
    tr, w, h, f = common.camera_for(pose_idx=1, size=96)
    th = ob.TreeHandle(tree)
    cam = ob.make_camera(tr, w, h, f)
    opt = ob.default_options()
    a = ob.ref_render(th, cam, opt).astype(np.int32)
    b = ob.ref_render(th, cam, opt, libm_expf=True).astype(np.int32)
    diff = np.abs(a - b)
    assert diff.max() <= 1
    assert (diff > 0).any(-1).mean() < 0.02


def test_fma_model_noise_floor():
    """strict vs nvcc-like FMA contraction: the parity noise floor reported in DESIGN.md."""
    tree = common.small_scene(depth=6, basis_dim=16, seed=211)
    tr, w, h, f = common.camera_for(pose_idx=6, size=96)
    a, acc_a, _ = common.oracle_frame(tree, tr, w, h, f, ob.FP_STRICT)
    b, acc_b, _ = common.oracle_frame(tree, tr, w, h, f, ob.FP_FMA)
    d = np.abs(a.astype(np.int32) - b.astype(np.int32))
    assert d.max() <= 1
    mse = float((d.astype(np.float64) ** 2).mean())
    psnr = 10 * np.log10(255.0 ** 2 / mse) if mse > 0 else np.inf
    assert psnr > 60.0


@pytest.mark.parametrize("N,depth,fmt,basis_dim", [(4, 3, "SH", 4), (3, 3, "RGBA", 0), (8, 2, "SH", 9)])
def test_general_branching_factor_bit_exact(N, depth, fmt, basis_dim):
    """N != 2: the float recurrence of query_single_from_root is no longer exact for every N
    (N = 3), so this pins the literal restatement, rounding included."""
    tree = common.random_tree_general_n(N, depth, basis_dim, fmt, seed=500 + N)
    tr, w, h, f = common.camera_for(pose_idx=3, size=48)
    rgba_o, acc_o, rgba_r, (th, cam, opt) = both(tree, tr, w, h, f)
    assert np.array_equal(rgba_o, rgba_r)
    assert np.array_equal(acc_o.view(np.uint32), ob.ref_trace(th, cam, opt).view(np.uint32))
    assert (rgba_o[..., :3] != 255).any()


@pytest.mark.parametrize("seed", range(24))
def test_random_configurations_bit_exact(seed):
    """Seeded sweep of the pin: random format / basis size / tree / camera / options (including
    odd image sizes, cameras inside the volume, degenerate thresholds and NDC) -- RGBA8 and fp32
    accumulators of the oracle equal the reference host build's bit for bit."""
    tree, tr, w, h, focal, ndc, kw, (fmt, bd) = common.random_configuration(seed)
    rgba_o, acc_o, rgba_r, (th, cam, opt) = both(tree, tr, w, h, focal, ndc=ndc, **kw)
    assert np.array_equal(rgba_o, rgba_r), (fmt, bd, kw)
    acc_r = ob.ref_trace(th, cam, opt)
    assert np.array_equal(acc_o.view(np.uint32), acc_r.view(np.uint32)), (fmt, bd, kw)


@pytest.mark.parametrize("depth,step", [(26, 1e-8), (28, 1e-4), (30, 1e-8)])
def test_deep_n2_chain_tree(depth, step):
    """Trees deeper than 24 levels (the kernels' float-descent fallback for N = 2,
    tests/test_gpu_parity.py): the oracle's root descent equals the reference's
    query_single_from_root (n3tree_query.hpp:22-47) at every depth the format allows -- the camera
    sits inside the deepest leaf of a 26-30 level chain and every ray samples every level."""
    tree, T = common.deep_chain_tree_n2(depth=depth, basis_dim=4, seed=depth)
    tr, w, h, f = common.camera_at(T)
    th = ob.TreeHandle(tree)
    cam = ob.make_camera(tr, w, h, f)
    opt = ob.default_options(step_size=step)
    rgba_o, acc_o, cnt = ob.render(th, cam, opt, ob.FP_STRICT)
    assert cnt["hit_samples"] > 5000 and cnt["child_reads"] > 8 * cnt["samples"]
    assert np.array_equal(rgba_o, ob.ref_render(th, cam, opt))
    assert np.array_equal(acc_o.view(np.uint32), ob.ref_trace(th, cam, opt).view(np.uint32))
