"""-m gpu: the HIP kernel (through the C ABI) against the CPU oracle.

Bar (BASELINE.json north_star): fp32 accumulators within 1 ulp -- we require
and get bit equality -- and RGBA8 bit-exact, for both FP models.
"""
import os

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


def gpu_frame(torch, tree, transform, w, h, focal, fp_mode=0, ndc=None, offscreen=True,
              rgba_init=None, depth_init=None, shard=None, **opt_kw):
    from volrend_amd import api
    t = api.N3Tree.from_synth(tree, ndc=ndc)
    cam = api.Camera(w, h, focal, focal)
    cam.transform = np.asarray(transform, dtype=np.float32)
    opts = api.RenderOptions(**opt_kw)
    if rgba_init is not None:
        img = torch.from_numpy(rgba_init.copy()).cuda()
    else:
        img = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
    acc = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    dep = torch.from_numpy(depth_init).cuda() if depth_init is not None else None
    api.launch_renderer(t, cam, opts, img, dep, torch.cuda.current_stream(), offscreen,
                        accum=acc, shard=shard, fp_mode=fp_mode)
    torch.cuda.synchronize()
    out = img.cpu().numpy(), acc.cpu().numpy()
    t.free_device()
    return out


def assert_parity(rgba_g, acc_g, rgba_o, acc_o):
    ulp = common.ulp_diff(acc_g, acc_o)
    bad_px = int((rgba_g != rgba_o).any(-1).sum())
    assert ulp.max() <= 1, f"accumulators differ by up to {ulp.max()} ulp"
    assert bad_px == 0, f"{bad_px} RGBA8 pixels differ"
    assert np.array_equal(acc_g.view(np.uint32), acc_o.view(np.uint32)), "accumulators not bit-equal"


@pytest.mark.parametrize("fp_mode", [0, 1])
@pytest.mark.parametrize("basis_dim", [1, 4, 9, 16, 25])
def test_sh_bit_exact(torch_cuda, basis_dim, fp_mode):
    tree = common.small_scene(depth=5, basis_dim=basis_dim, seed=20 + basis_dim)
    tr, w, h, f = common.camera_for(pose_idx=2, size=96)
    rgba_o, acc_o, cnt = common.oracle_frame(tree, tr, w, h, f, fp_mode)
    assert cnt["hit_samples"] > 1000
    rgba_g, acc_g = gpu_frame(torch_cuda, tree, tr, w, h, f, fp_mode)
    assert_parity(rgba_g, acc_g, rgba_o, acc_o)


@pytest.mark.parametrize("fp_mode", [0, 1])
@pytest.mark.parametrize("fmt,basis_dim", [("RGBA", 0), ("SG", 9), ("SG", 25), ("ASG", 4), ("SG", 7),
                                           ("SG", 23), ("ASG", 25)])  # 23, 25: records beyond one line (head + tail)
def test_other_formats_bit_exact(torch_cuda, fmt, basis_dim, fp_mode):
    tree = common.small_scene(depth=5, basis_dim=basis_dim, fmt=fmt, seed=31)
    tr, w, h, f = common.camera_for(pose_idx=5, size=80)
    rgba_o, acc_o, _ = common.oracle_frame(tree, tr, w, h, f, fp_mode)
    rgba_g, acc_g = gpu_frame(torch_cuda, tree, tr, w, h, f, fp_mode)
    assert_parity(rgba_g, acc_g, rgba_o, acc_o)


@pytest.mark.parametrize("fp_mode", [0, 1])
def test_options_bit_exact(torch_cuda, fp_mode):
    tree = common.small_scene(depth=6, basis_dim=9, seed=41)
    tr, w, h, f = common.camera_for(pose_idx=3, size=72)
    cases = [
        dict(step_size=1e-3, sigma_thresh=0.5, stop_thresh=0.1, background_brightness=0.25),
        dict(render_bbox=(0.1, 0.2, 0.0, 0.8, 0.9, 0.7)),
        dict(basis_minmax=(1, 5)),
        dict(rot_dirs=(0.3, -0.2, 0.9)),
        dict(render_depth=1),
        dict(step_size=1e-5, stop_thresh=1e-4),
    ]
    for kw in cases:
        rgba_o, acc_o, _ = common.oracle_frame(tree, tr, w, h, f, fp_mode, **kw)
        rgba_g, acc_g = gpu_frame(torch_cuda, tree, tr, w, h, f, fp_mode, **kw)
        assert_parity(rgba_g, acc_g, rgba_o, acc_o)


@pytest.mark.parametrize("fp_mode", [0, 1])
def test_ndc_bit_exact(torch_cuda, fp_mode):
    tree = common.small_scene(depth=5, basis_dim=4, seed=51)
    ndc = (96.0, 72.0, 80.0)
    # a forward-facing camera looking down -z from z>0
    tr = np.array([1, 0, 0, 0, 1, 0, 0, 0, 1, 0.05, -0.02, 0.3], dtype=np.float32)
    rgba_o, acc_o, _ = common.oracle_frame(tree, tr, 96, 72, 80.0, fp_mode, ndc=ndc)
    rgba_g, acc_g = gpu_frame(torch_cuda, tree, tr, 96, 72, 80.0, fp_mode, ndc=ndc)
    assert_parity(rgba_g, acc_g, rgba_o, acc_o)


def test_ragged_image_and_miss(torch_cuda):
    """Width/height not multiples of the 8x8 wave tile; a pose that misses the box."""
    tree = common.small_scene(depth=4, basis_dim=4, seed=61)
    tr, _, _, f = common.camera_for(pose_idx=1, size=61)
    rgba_o, acc_o, _ = common.oracle_frame(tree, tr, 61, 37, f)
    rgba_g, acc_g = gpu_frame(torch_cuda, tree, tr, 61, 37, f)
    assert_parity(rgba_g, acc_g, rgba_o, acc_o)
    # camera far away looking away from the volume: every ray misses
    tr2 = tr.copy()
    tr2[9:12] = [50.0, 50.0, 50.0]
    rgba_o, acc_o, cnt = common.oracle_frame(tree, tr2, 40, 40, f)
    rgba_g, acc_g = gpu_frame(torch_cuda, tree, tr2, 40, 40, f)
    assert_parity(rgba_g, acc_g, rgba_o, acc_o)


def test_compositing_over_existing_frame(torch_cuda):
    """offscreen=False: composite over the RGBA8 + R32F mesh depth already in the
    target (the interactive caller, src/cuda_renderer.cpp:115-118)."""
    tree = common.small_scene(depth=5, basis_dim=9, seed=71)
    tr, w, h, f = common.camera_for(pose_idx=4, size=64)
    rng = np.random.default_rng(5)
    init = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
    depth = rng.uniform(2.0, 6.0, size=(h, w)).astype(np.float32)
    rgba_o, acc_o, _ = common.oracle_frame(tree, tr, w, h, f, offscreen=False, rgba_init=init,
                                           depth_init=depth)
    rgba_g, acc_g = gpu_frame(torch_cuda, tree, tr, w, h, f, offscreen=False, rgba_init=init,
                              depth_init=depth)
    assert_parity(rgba_g, acc_g, rgba_o, acc_o)


def test_tile_shards_reassemble(torch_cuda):
    """Screen-tile shards (FRAME and COMPACT layouts) reproduce the single-GPU frame."""
    torch = torch_cuda
    from volrend_amd import api
    tree = common.small_scene(depth=5, basis_dim=4, seed=81)
    tr, w, h, f = common.camera_for(pose_idx=6, size=100)
    rgba_ref, _ = gpu_frame(torch, tree, tr, w, h, f)
    t = api.N3Tree.from_synth(tree)
    cam = api.Camera(w, h, f, f)
    cam.transform = tr
    opts = api.RenderOptions()
    for world, tw, th in [(2, 104, 8), (3, 32, 16), (8, 104, 8), (4, 8, 8)]:
        frame = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
        sh0 = api.TileShard(tw, th, 0, world, compact=True)
        nbytes = api.compact_bytes(w, h, sh0)
        gathered = torch.zeros((world, nbytes), dtype=torch.uint8, device="cuda")
        for r in range(world):
            api.launch_renderer(t, cam, opts, frame, None, None, True,
                                shard=api.TileShard(tw, th, r, world, compact=False))
            api.launch_renderer(t, cam, opts, gathered[r], None, None, True,
                                shard=api.TileShard(tw, th, r, world, compact=True))
        out = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
        api.assemble_tiles(out, gathered, w, h, sh0)
        torch.cuda.synchronize()
        assert np.array_equal(frame.cpu().numpy(), rgba_ref), (world, tw, th)
        assert np.array_equal(out.cpu().numpy(), rgba_ref), (world, tw, th)
    t.free_device()


@pytest.mark.parametrize("fmt,basis_dim", [("SH", 16), ("RGBA", 0), ("SG", 4)])
def test_access_counters_match_oracle(torch_cuda, fmt, basis_dim):
    """The instrumented kernel flavour (VrFrame.counters) is the algorithmic-bytes meter of
    bench.py's roofline: it must agree with the oracle's counters to the integer, and it
    must not change the image."""
    torch = torch_cuda
    from volrend_amd import _abi, api
    tree = common.small_scene(depth=6, basis_dim=basis_dim, fmt=fmt, seed=91)
    tr, w, h, f = common.camera_for(pose_idx=2, size=88)
    rgba_o, acc_o, cnt_o = common.oracle_frame(tree, tr, w, h, f)
    t = api.N3Tree.from_synth(tree)
    cam = api.Camera(w, h, f, f)
    cam.transform = tr
    img = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
    counters = torch.zeros(7, dtype=torch.int64, device="cuda")
    api.launch_renderer(t, cam, api.RenderOptions(), img, None, None, True, counters=counters)
    torch.cuda.synchronize()
    cnt_g = dict(zip(_abi.COUNTER_FIELDS, [int(v) for v in counters.cpu().tolist()]))
    assert cnt_g == cnt_o
    assert np.array_equal(img.cpu().numpy(), rgba_o)
    t.free_device()


def test_api_errors_are_codes_not_crashes(torch_cuda):
    """cuda_assert prints and exits (src/cuda/common.cu:8-21); the library returns codes."""
    torch = torch_cuda
    import ctypes as C
    from volrend_amd import _abi, api
    tree = common.small_scene(depth=3, basis_dim=4, seed=7)
    t = api.N3Tree.from_synth(tree)
    img = torch.zeros((32, 32, 4), dtype=torch.uint8, device="cuda")
    cam = api.Camera(32, 32, 50.0)
    L = _abi.lib()

    def rc_of(mutate_cam=None, mutate_frame=None, n=1):
        c = cam.to_c()
        f = _abi.VrFrame()
        L.vr_default_frame(C.byref(f))
        f.rgba = img.data_ptr()
        if mutate_cam:
            mutate_cam(c)
        if mutate_frame:
            mutate_frame(f)
        o = api.RenderOptions().to_c()
        return L.vr_render_batch(t.handle, n, C.byref(c), C.byref(o), C.byref(f), None)

    assert rc_of() == 0
    assert rc_of(mutate_cam=lambda c: setattr(c, "width", 70000)) == 1
    assert rc_of(mutate_cam=lambda c: setattr(c, "fx", 0.0)) == 1
    assert rc_of(mutate_frame=lambda f: setattr(f, "rgba", None)) == 1
    assert rc_of(mutate_frame=lambda f: setattr(f, "fp_mode", 7)) == 1
    assert rc_of(mutate_frame=lambda f: (setattr(f, "world", 2), setattr(f, "rank", 2))) == 1
    assert rc_of(mutate_frame=lambda f: (setattr(f, "tile_w", 12), setattr(f, "tile_h", 8))) == 1
    assert rc_of(mutate_frame=lambda f: setattr(f, "pitch", 16)) == 1
    assert rc_of(n=0) == 1 and rc_of(n=1000) == 1
    assert b"n_frames" in L.vr_last_error()
    torch.cuda.synchronize()
    t.free_device()
    with pytest.raises(RuntimeError):
        _ = t.handle  # freed trees are not usable


@pytest.mark.parametrize("fp_mode", [0, 1])
@pytest.mark.parametrize("N,depth,fmt,basis_dim", [(4, 3, "SH", 4), (3, 3, "RGBA", 0), (8, 2, "SH", 9)])
def test_general_branching_factor_bit_exact(torch_cuda, N, depth, fmt, basis_dim, fp_mode):
    """N != 2 takes the GENERIC kernel flavour (literal float descent)."""
    tree = common.random_tree_general_n(N, depth, basis_dim, fmt, seed=500 + N)
    tr, w, h, f = common.camera_for(pose_idx=3, size=48)
    rgba_o, acc_o, _ = common.oracle_frame(tree, tr, w, h, f, fp_mode)
    rgba_g, acc_g = gpu_frame(torch_cuda, tree, tr, w, h, f, fp_mode)
    assert_parity(rgba_g, acc_g, rgba_o, acc_o)


@pytest.mark.parametrize("n_retain,basis_dim", [(1, 4), (0, 9), (4, 16)])
def test_quantised_tree_device_decode(torch_cuda, tmp_path, n_retain, basis_dim):
    """Quantised tree.npz: the codebook decode runs on the device during the upload
    (vr_tree_upload_quantized) and must reproduce the reference's host loop
    (src/n3tree.cpp:310-339) bit for bit -- as a data array and as a rendered frame."""
    from volrend_amd import api
    torch = torch_cuda
    tree = common.small_scene(depth=4, basis_dim=basis_dim, seed=700 + basis_dim)
    p = str(tmp_path / "q.npz")
    common.write_quantised_npz(tree, p, n_retain=n_retain, compressed=(n_retain == 1))
    t = api.N3Tree(p)                       # device decode
    assert t.data_ is None and t.is_device_loaded()
    on_dev = t.decode_host(on_device=True)  # vr_decode_quantized
    assert np.array_equal(on_dev.view(np.uint16), tree.data.view(np.uint16))
    t2 = api.N3Tree()
    t2.open(p, upload=True, device_decode=False)  # numpy decode on the host, plain upload
    assert np.array_equal(t2.data_.view(np.uint16), tree.data.view(np.uint16))
    tr, w, h, f = common.camera_for(pose_idx=1, size=80)
    rgba_o, acc_o, cnt = common.oracle_frame(tree, tr, w, h, f, 0)
    assert cnt["hit_samples"] > 500
    cam = api.Camera(w, h, f, f)
    cam.transform = np.asarray(tr, dtype=np.float32)
    for tt in (t, t2):
        img = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
        acc = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
        api.launch_renderer(tt, cam, api.RenderOptions(), img, None, torch.cuda.current_stream(),
                            True, accum=acc)
        torch.cuda.synchronize()
        assert_parity(img.cpu().numpy(), acc.cpu().numpy(), rgba_o, acc_o)


def test_quantised_upload_rejects_bad_descriptors(torch_cuda):
    import ctypes as C
    from volrend_amd import _abi
    L = _abi.lib()
    tree = common.small_scene(depth=2, basis_dim=4, seed=5)
    d = _abi.VrTreeDesc()
    L.vr_default_tree_desc(C.byref(d))
    child = np.ascontiguousarray(tree.child)
    d.child, d.N, d.capacity, d.data_dim = child.ctypes.data, 2, tree.capacity, tree.data_dim
    d.format, d.basis_dim = _abi.FORMAT_SH, 4
    n_slots = tree.capacity * 8
    sig = np.zeros(n_slots, np.float16)
    qm = np.zeros((4, n_slots), np.uint16)
    qc = np.zeros((4, 65536, 3), np.float16)
    h = C.c_void_p()
    q = _abi.VrQuantDesc()
    q.sigma, q.quant_map, q.quant_colors = sig.ctypes.data, qm.ctypes.data, qc.ctypes.data
    q.n_quant = 5                                       # 3*5+1 > data_dim
    assert L.vr_tree_upload_quantized(C.byref(d), C.byref(q), C.byref(h)) == 1
    q.n_quant, q.n_retained = 3, 1                      # retained without the array
    assert L.vr_tree_upload_quantized(C.byref(d), C.byref(q), C.byref(h)) == 1
    q.n_quant, q.n_retained, q.sigma = 4, 0, None
    assert L.vr_tree_upload_quantized(C.byref(d), C.byref(q), C.byref(h)) == 1
    q.sigma = sig.ctypes.data
    assert L.vr_tree_upload_quantized(C.byref(d), C.byref(q), C.byref(h)) == 0
    L.vr_tree_free(h)


@pytest.mark.parametrize("fp_mode", [0, 1])
def test_random_configurations_bit_exact(torch_cuda, fp_mode):
    """The seeded sweep of tests/test_oracle_vs_ref.py (random format / basis size / tree /
    camera / options / NDC, odd image sizes, cameras inside the volume) on the GPU."""
    # VR_SWEEP_SEEDS=N widens the sweep for a one-off hunt (round 2: 600 seeds, both models, clean)
    for seed in range(int(os.environ.get("VR_SWEEP_SEEDS", "24"))):
        tree, tr, w, h, focal, ndc, kw, what = common.random_configuration(seed)
        rgba_o, acc_o, _ = common.oracle_frame(tree, tr, w, h, focal, fp_mode, ndc=ndc, **kw)
        rgba_g, acc_g = gpu_frame(torch_cuda, tree, tr, w, h, focal, fp_mode, ndc=ndc, **kw)
        try:
            assert_parity(rgba_g, acc_g, rgba_o, acc_o)
        except AssertionError as e:
            raise AssertionError(f"seed {seed} {what} {kw}: {e}") from None


@pytest.mark.parametrize("fp_mode", [0, 1])
@pytest.mark.parametrize("top_levels,brick_levels,blocked",
                         [(1, 1, 0), (1, 3, 0), (1, 4, 0), (2, 2, 0), (2, 3, 0), (2, 4, 0), (3, 3, 0), (3, 4, 0),
                          (4, 1, 0), (5, 2, 0), (6, 3, 0), (8, 3, 0),
                          # bricks in 4 x 4 x 2 line blocks (the entry order large trees get at upload)
                          (1, 3, 1), (2, 3, 1), (3, 3, 1), (4, 3, 1), (6, 3, 1), (2, 4, 1)])
def test_lookup_structure_geometries(torch_cuda, top_levels, brick_levels, blocked, fp_mode):
    """The N == 2 lookup structure (top grid of 2^G0 cells per axis + bricks of 2^BL entries per
    axis + child words below, vr_kernels.hip) is a pure index: whatever its geometry, every
    sample must land in the leaf the reference's root descent finds (n3tree_query.hpp:13-48).
    Small geometries push a depth-7 tree through every branch: top leaves, brick leaves of all
    three depths, and the child-word walk below the bricks -- in both entry orders of the bricks
    (brick_blocked applies to 8^3 bricks only: (2, 4, 1) must quietly stay x-major)."""
    from volrend_amd import api
    tree = common.small_scene(depth=7, basis_dim=4, seed=1201)
    tr, w, h, f = common.camera_for(pose_idx=3, size=72)
    rgba_o, acc_o, cnt = common.oracle_frame(tree, tr, w, h, f, fp_mode)
    assert cnt["hit_samples"] > 1000
    api.set_tuning(top_levels=top_levels, brick_levels=brick_levels, brick_blocked=blocked)
    try:
        rgba_g, acc_g = gpu_frame(torch_cuda, tree, tr, w, h, f, fp_mode)
        # instrumented flavour: child_reads = sum of leaf depths must match the oracle's walk
        import torch
        t = api.N3Tree.from_synth(tree)
        cam = api.Camera(w, h, f, f)
        cam.transform = np.asarray(tr, np.float32)
        img = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
        counters = torch.zeros(7, dtype=torch.int64, device="cuda")
        api.launch_renderer(t, cam, api.RenderOptions(), img, None, None, True, counters=counters,
                            fp_mode=fp_mode)
        torch.cuda.synchronize()
        from volrend_amd import _abi
        cnt_g = dict(zip(_abi.COUNTER_FIELDS, [int(v) for v in counters.cpu().tolist()]))
        t.free_device()
    finally:
        api.set_tuning(top_levels=0, brick_levels=3, brick_blocked=-1)
    assert_parity(rgba_g, acc_g, rgba_o, acc_o)
    assert cnt_g == cnt
    assert np.array_equal(img.cpu().numpy(), rgba_o)


@pytest.mark.parametrize("frame_group,super_block", [(1, 1), (2, 2), (3, 4), (0, 3), (4, 64)])
def test_ray_order_is_scheduling_only(torch_cuda, frame_group, super_block):
    """The ray-id order (frame groups x super-blocks of 8x8 pixel blocks, vr_kernels.hip locate())
    decides which rays march together, never what they compute: a 5-pose batch of a ragged
    image -- whole frames and 3-way tile shards with odd tile sizes -- must equal the oracle
    under every order."""
    torch = torch_cuda
    from volrend_amd import api
    tree = common.small_scene(depth=5, basis_dim=9, seed=1301)
    w, h, f = 150, 107, 170.0
    trs = [common.camera_for(pose_idx=i, size=64)[0] for i in range(5)]
    want = [common.oracle_frame(tree, tr, w, h, f, 0)[0] for tr in trs]
    t = api.N3Tree.from_synth(tree)
    cam = api.Camera(w, h, f, f)
    t.set_tuning(frame_group=frame_group, super_block=super_block)  # this tree only
    try:
        imgs = torch.zeros((5, h, w, 4), dtype=torch.uint8, device="cuda")
        api.launch_renderer_batch(t, cam, trs, api.RenderOptions(), list(imgs), None, True)
        world, tw, th = 3, 48, 24
        sh0 = api.TileShard(tw, th, 0, world, compact=True)
        nbytes = api.compact_bytes(w, h, sh0)
        gathered = torch.zeros((world, 5, nbytes), dtype=torch.uint8, device="cuda")
        for r in range(world):
            api.launch_renderer_batch(t, cam, trs, api.RenderOptions(),
                                      [gathered[r, i] for i in range(5)], None, True,
                                      shard=api.TileShard(tw, th, r, world, compact=True))
        outs = torch.zeros((5, h, w, 4), dtype=torch.uint8, device="cuda")
        api.assemble_tiles_batch(outs, gathered, 5, w, h, sh0, torch.cuda.current_stream())
        torch.cuda.synchronize()
    finally:
        pass
    for i in range(5):
        assert np.array_equal(imgs[i].cpu().numpy(), want[i]), ("frame", i)
        assert np.array_equal(outs[i].cpu().numpy(), want[i]), ("sharded", i)
    t.free_device()


@pytest.mark.parametrize("fp_mode", [0, 1])
@pytest.mark.parametrize("depth,step", [(26, 1e-8), (28, 1e-4), (30, 1e-8)])
def test_n2_tree_beyond_the_integer_lookup_takes_the_float_descent(torch_cuda, depth, step, fp_mode):
    """An N = 2 tree deeper than 24 levels cannot use the integer lookup (exact digits of a
    binary32 coordinate): vr_tree_upload must route it to the literal float descent, which the
    reference runs for every tree (n3tree_query.hpp:22-47).  A chain tree of 26-30 levels with
    the camera INSIDE its deepest leaf (tests/common.py deep_chain_tree_n2): every ray starts at
    depth `depth` and samples every level on its way out.  Kernel == oracle (both FP models,
    accumulators, RGBA8, access counters) == the reference's own render_kernel / trace_ray
    compiled for the host."""
    import torch
    from volrend_amd import _abi, api
    tree, T = common.deep_chain_tree_n2(depth=depth, basis_dim=4, seed=depth)
    tr, w, h, f = common.camera_at(T)
    rgba_o, acc_o, cnt = common.oracle_frame(tree, tr, w, h, f, fp_mode, step_size=step)
    assert cnt["hit_samples"] > 5000 and cnt["child_reads"] > 8 * cnt["samples"]
    t = api.N3Tree.from_synth(tree)
    info = t.info()
    assert info["N"] == 2 and info["max_depth"] == depth - 1
    assert info["query_mode"] == _abi.QUERY_DESCENT and info["top_levels"] == 0  # the fallback WAS taken
    cam = api.Camera(w, h, f, f)
    cam.transform = np.asarray(tr, np.float32)
    img = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
    acc = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    counters = torch.zeros(7, dtype=torch.int64, device="cuda")
    api.launch_renderer(t, cam, api.RenderOptions(step_size=step), img, None, None, True, accum=acc,
                        counters=counters, fp_mode=fp_mode)
    torch.cuda.synchronize()
    assert t.status() == 0
    cnt_g = dict(zip(_abi.COUNTER_FIELDS, [int(v) for v in counters.cpu().tolist()]))
    # and once more without counters (the uninstrumented GENERIC flavour is the one a product launch takes)
    img2 = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
    api.launch_renderer(t, cam, api.RenderOptions(step_size=step), img2, None, None, True, fp_mode=fp_mode)
    torch.cuda.synchronize()
    t.free_device()
    assert_parity(img.cpu().numpy(), acc.cpu().numpy(), rgba_o, acc_o)
    assert np.array_equal(img2.cpu().numpy(), rgba_o)
    assert cnt_g == cnt
    if fp_mode == 0 and common.ob.ref_lib() is not None:
        th = common.ob.TreeHandle(tree)
        ocam = common.ob.make_camera(tr, w, h, f)
        opt = common.ob.default_options(step_size=step)
        assert np.array_equal(img.cpu().numpy(), common.ob.ref_render(th, ocam, opt)), "kernel != reference"
        assert np.array_equal(acc.cpu().numpy().view(np.uint32),
                              common.ob.ref_trace(th, ocam, opt).view(np.uint32)), "kernel != reference (fp32)"


@pytest.mark.parametrize("raygen_waves", [16, 4, 1])
@pytest.mark.parametrize("xcd_queues", [1, 0])
def test_raygen_workgroup_size_is_scheduling_only(torch_cuda, raygen_waves, xcd_queues):
    """Ray generation runs in workgroups of 16, 4 or 1 waves (small launches: one) and compacts the
    rays of each of the 8 (or 1) ray queues to the front of the queue's own region, whose
    boundaries lie at multiples of 16 blocks.  A 5-pose batch of a ragged image (block counts
    that are no multiple of 16, queues of unequal length, rays that miss the volume) -- whole
    frames and 3-way tile shards -- must equal the oracle whatever the workgroup size."""
    torch = torch_cuda
    from volrend_amd import api, synth
    tree = common.small_scene(depth=5, basis_dim=9, seed=1311)
    w, h = 150, 91
    f = 1.9 * w   # narrow view from far away: many rays miss the volume
    trs = [synth.c2w_to_transform(p) for p in synth.make_poses(5, radius=5.5)]
    want = [common.oracle_frame(tree, tr, w, h, f)[0] for tr in trs]
    t = api.N3Tree.from_synth(tree)
    t.set_tuning(raygen_waves=raygen_waves, xcd_queues=xcd_queues)
    cam = api.Camera(w, h, f, f)
    imgs = torch.zeros((5, h, w, 4), dtype=torch.uint8, device="cuda")
    api.launch_renderer_batch(t, cam, trs, api.RenderOptions(), [imgs[i] for i in range(5)], None, True)
    torch.cuda.synchronize()
    assert t.status() == 0
    got = imgs.cpu().numpy()
    for i in range(5):
        assert np.array_equal(got[i], want[i]), f"pose {i}"
    # tile shards: every rank's COMPACT buffer, assembled, equals the frame
    from volrend_amd import tiles
    for world in (3,):
        parts = []
        for rank in range(world):
            shard = api.TileShard(16, 8, rank, world, compact=True)
            nb = api.compact_bytes(w, h, shard)
            buf = torch.zeros((5, nb), dtype=torch.uint8, device="cuda")
            api.launch_renderer_batch(t, cam, trs, api.RenderOptions(), [buf[i] for i in range(5)], None, True,
                                      shard=shard)
            torch.cuda.synchronize()
            parts.append(buf.cpu().numpy())
        for i in range(5):
            g = np.stack([parts[r][i].reshape(-1, 4) for r in range(world)])  # [world, compact pixels, 4]
            frame = tiles.assemble_tiles(g, w, h, 16, 8, world)
            assert np.array_equal(frame, want[i]), f"shard world {world} pose {i}"
    t.free_device()


def test_random_launch_shapes(torch_cuda):
    """Seeded random LAUNCH shapes (what the ray queues and the ray generation see, not what a ray
    computes): image size, poses per launch, tile shard (world, tile height), raygen workgroup size,
    one or eight queues, camera distance (share of rays that miss the volume).  Every rank's COMPACT
    buffer, assembled, must equal the oracle's frames.  VR_SHAPE_SEEDS=N widens the sweep."""
    torch = torch_cuda
    from volrend_amd import api, synth, tiles
    rng0 = np.random.default_rng(4242)
    trees = [common.small_scene(depth=d, basis_dim=b, seed=1400 + d) for d, b in ((4, 4), (5, 9), (6, 16))]
    handles = [api.N3Tree.from_synth(t) for t in trees]
    oracle_cache = {}
    for seed in range(int(os.environ.get("VR_SHAPE_SEEDS", "10"))):
        rng = np.random.default_rng(7000 + seed)
        ti = int(rng.integers(len(trees)))
        w, h = int(rng.integers(17, 200)), int(rng.integers(9, 140))
        nf = int(rng.integers(1, 10))
        world = int(rng.integers(1, 5))
        tile_h = int(rng.choice([8, 16, 24]))
        radius = float(rng.choice([3.0, 4.0, 6.5]))
        f = float(rng.uniform(0.8, 2.2)) * w
        gw = int(rng.choice([0, 1, 4, 16]))
        xq = int(rng.integers(2))
        poses = synth.make_poses(16, radius=radius)
        trs = [synth.c2w_to_transform(poses[int(rng.integers(16))]) for _ in range(nf)]
        t = handles[ti]
        t.set_tuning(raygen_waves=gw, xcd_queues=xq)
        cam = api.Camera(w, h, f, f)
        tile_w = (w + 7) // 8 * 8
        parts = []
        for rank in range(world):
            shard = api.TileShard(tile_w, tile_h, rank, world, compact=True)
            nb = api.compact_bytes(w, h, shard)
            buf = torch.zeros((nf, nb), dtype=torch.uint8, device="cuda")
            api.launch_renderer_batch(t, cam, trs, api.RenderOptions(), [buf[i] for i in range(nf)], None, True,
                                      shard=shard)
            parts.append(buf)
        torch.cuda.synchronize()
        assert t.status() == 0
        parts = [p.cpu().numpy() for p in parts]
        what = f"seed {seed}: tree {ti} {w}x{h} x{nf} world {world} tile_h {tile_h} raygen_waves {gw} queues {8 if xq else 1} r {radius}"
        for i, tr in enumerate(trs):
            want = common.oracle_frame(trees[ti], tr, w, h, f)[0]
            g = np.stack([parts[r][i].reshape(-1, 4) for r in range(world)])
            got = tiles.assemble_tiles(g, w, h, tile_w, tile_h, world)
            assert np.array_equal(got, want), what + f" pose {i}"
    for t in handles:
        t.free_device()
