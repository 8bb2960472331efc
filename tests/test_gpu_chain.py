"""-m gpu: the whole parity chain in ONE record -- HIP kernel == CPU oracle == the reference's
own render_kernel / trace_ray compiled for the host (oracle/_ref/libvolrend_ref.so, which travels
to the GPU box with the snapshot).  tests/test_oracle_vs_ref.py pins oracle == reference on the
CPU; this file repeats that comparison next to the kernel's output so that the GPU test record
alone shows kernel == oracle == reference.
"""
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


def kernel_frame(torch, tree, tr, w, h, f, fp_mode=0, ndc=None, **opt_kw):
    from volrend_amd import api
    t = api.N3Tree.from_synth(tree, ndc=ndc)
    cam = api.Camera(w, h, f, f)
    cam.transform = np.asarray(tr, dtype=np.float32)
    img = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
    acc = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    api.launch_renderer(t, cam, api.RenderOptions(**opt_kw), img, None, torch.cuda.current_stream(),
                        True, accum=acc, fp_mode=fp_mode)
    torch.cuda.synchronize()
    assert t.status() == 0, "the kernel reported a guard trip"
    out = img.cpu().numpy(), acc.cpu().numpy()
    t.free_device()
    return out


CASES = [
    ("SH", 16, dict()),
    ("SH", 9, dict(step_size=1e-3, sigma_thresh=0.5, stop_thresh=0.1, background_brightness=0.25)),
    ("SH", 25, dict(render_bbox=(0.1, 0.2, 0.0, 0.8, 0.9, 0.7))),
    ("SH", 4, dict(basis_minmax=(1, 3), rot_dirs=(0.3, -0.2, 0.9))),
    ("SH", 1, dict(step_size=1e-5, stop_thresh=1e-4)),
    ("RGBA", 0, dict(background_brightness=0.0)),
    ("SG", 9, dict()),
    ("SH", 16, dict(render_depth=1)),
]


@pytest.mark.parametrize("fmt,basis_dim,kw", CASES,
                         ids=[f"{c[0]}{c[1]}-{'-'.join(c[2]) or 'default'}" for c in CASES])
def test_kernel_equals_oracle_equals_reference(torch_cuda, fmt, basis_dim, kw):
    if ob.ref_lib() is None:
        pytest.skip("oracle/_ref/libvolrend_ref.so missing (build() where the reference is mounted)")
    tree = common.small_scene(depth=6, basis_dim=basis_dim, fmt=fmt, seed=300 + basis_dim)
    tr, w, h, f = common.camera_for(pose_idx=2, size=104)
    th = ob.TreeHandle(tree)
    cam = ob.make_camera(tr, w, h, f)
    opt = ob.default_options(**kw)
    rgba_o, acc_o, cnt = ob.render(th, cam, opt, ob.FP_STRICT)
    rgba_r = ob.ref_render(th, cam, opt)            # the reference's render_kernel
    acc_r = ob.ref_trace(th, cam, opt)              # the reference's trace_ray
    rgba_k, acc_k = kernel_frame(torch_cuda, tree, tr, w, h, f, 0, **kw)
    assert cnt["samples"] > 10000
    assert np.array_equal(rgba_o, rgba_r), "oracle != reference (RGBA8)"
    assert np.array_equal(acc_o.view(np.uint32), acc_r.view(np.uint32)), "oracle != reference (fp32)"
    assert np.array_equal(rgba_k, rgba_r), "kernel != reference (RGBA8)"
    assert np.array_equal(acc_k.view(np.uint32), acc_r.view(np.uint32)), "kernel != reference (fp32)"


def test_ndc_chain(torch_cuda):
    if ob.ref_lib() is None:
        pytest.skip("oracle/_ref/libvolrend_ref.so missing")
    tree = common.small_scene(depth=5, basis_dim=9, seed=351)
    tr = np.array([1, 0, 0, 0, 1, 0, 0, 0, 1, 0.05, -0.02, 0.3], dtype=np.float32)
    ndc = (96.0, 72.0, 80.0)
    th = ob.TreeHandle(tree, ndc=ndc)
    cam = ob.make_camera(tr, 96, 72, 80.0)
    opt = ob.default_options()
    rgba_r = ob.ref_render(th, cam, opt)
    acc_r = ob.ref_trace(th, cam, opt)
    rgba_k, acc_k = kernel_frame(torch_cuda, tree, tr, 96, 72, 80.0, 0, ndc=ndc)
    assert np.array_equal(rgba_k, rgba_r)
    assert np.array_equal(acc_k.view(np.uint32), acc_r.view(np.uint32))


@pytest.mark.parametrize("fp_mode", [0, 1])
def test_random_sweep(torch_cuda, fp_mode):
    """The seeded random configurations of the CPU pin (formats, odd sizes, cameras inside the
    volume, degenerate thresholds, bbox, basis range, rotation, depth mode, NDC), through the
    kernel in both FP models."""
    # VR_SWEEP_SEEDS=N widens the sweep for a one-off hunt (every round's final kernel: 600 seeds,
    # both FP models, clean -- profiles/r0N_seed_sweep.txt)
    import os
    for seed in range(int(os.environ.get("VR_SWEEP_SEEDS", "24"))):
        tree, tr, w, h, f, ndc, kw, tag = common.random_configuration(seed)
        rgba_o, acc_o, _ = common.oracle_frame(tree, tr, w, h, f, fp_mode, ndc=ndc, **kw)
        rgba_k, acc_k = kernel_frame(torch_cuda, tree, tr, w, h, f, fp_mode, ndc=ndc, **kw)
        assert np.array_equal(rgba_k, rgba_o), f"seed {seed} {tag}: RGBA8 differs"
        assert np.array_equal(acc_k.view(np.uint32), acc_o.view(np.uint32)), \
            f"seed {seed} {tag}: accumulators differ"


def test_batch_of_poses(torch_cuda):
    """A bench-shaped launch (several poses, one queue, rays of different frames in one wave;
    enough rays that waves refill many times) frame by frame against the oracle."""
    torch = torch_cuda
    from volrend_amd import api
    tree = common.small_scene(depth=7, basis_dim=16, seed=377)
    w = h = 200
    f = w * 1111.111 / 800.0
    from volrend_amd import synth
    trs = [synth.c2w_to_transform(p) for p in synth.make_poses(12)]
    t = api.N3Tree.from_synth(tree)
    cam = api.Camera(w, h, f, f)
    imgs = torch.zeros((len(trs), h, w, 4), dtype=torch.uint8, device="cuda")
    accs = torch.zeros((len(trs), h, w, 4), dtype=torch.float32, device="cuda")
    api.launch_renderer_batch(t, cam, trs, api.RenderOptions(), [imgs[i] for i in range(len(trs))],
                              torch.cuda.current_stream(), True,
                              accums=[accs[i] for i in range(len(trs))])
    torch.cuda.synchronize()
    assert t.status() == 0
    got, got_acc = imgs.cpu().numpy(), accs.cpu().numpy()
    t.free_device()
    for i, tr in enumerate(trs):
        rgba_o, acc_o, _ = common.oracle_frame(tree, tr, w, h, f)
        assert np.array_equal(got[i], rgba_o), f"pose {i}: RGBA8 differs"
        assert np.array_equal(got_acc[i].view(np.uint32), acc_o.view(np.uint32)), f"pose {i}"


def test_tree_with_backward_links(torch_cuda):
    """A valid tree whose node order is scrambled (children at LOWER indices than their parents:
    relative links may be negative in the file format): the upload's index-order sweep gives up
    at the first backward link and the general walk takes over; the picture is the oracle's."""
    from volrend_amd import synth
    tree = common.small_scene(depth=5, basis_dim=9, seed=391)
    cap = tree.capacity
    rng = np.random.default_rng(7)
    new_of = np.concatenate([[0], 1 + rng.permutation(cap - 1)])        # root stays node 0
    child = tree.child.reshape(cap, 8).astype(np.int64)
    tgt = np.where(child != 0, np.arange(cap)[:, None] + child, -1)     # absolute child index
    new_child = np.zeros_like(child)
    new_child[new_of] = np.where(tgt >= 0, new_of[np.maximum(tgt, 0)] - new_of[:, None], 0)
    data = np.empty_like(tree.data.reshape(cap, 8, -1))
    data[new_of] = tree.data.reshape(cap, 8, -1)
    assert (new_child < 0).any()
    scr = synth.SynthTree(new_child.astype(np.int32).reshape(cap, 2, 2, 2),
                          data.reshape(cap, 2, 2, 2, -1), tree.offset, tree.invradius3,
                          tree.data_format, tree.extra, tree.depth)
    tr, w, h, f = common.camera_for(pose_idx=3, size=88)
    rgba_o, acc_o, _ = common.oracle_frame(tree, tr, w, h, f)           # the ORIGINAL tree
    rgba_s, acc_s, _ = common.oracle_frame(scr, tr, w, h, f)            # numbering is not geometry
    assert np.array_equal(rgba_o, rgba_s)
    rgba_k, acc_k = kernel_frame(torch_cuda, scr, tr, w, h, f)
    assert np.array_equal(rgba_k, rgba_o)
    assert np.array_equal(acc_k.view(np.uint32), acc_o.view(np.uint32))


@pytest.mark.parametrize("xcd_queues", [1, 0], ids=["8 queues", "1 queue"])
def test_every_ray_queue_is_drained(torch_cuda, xcd_queues):
    """The ray buffer is cut into 8 queues, one per XCD; a wave steals from a foreign queue only
    while that queue holds a good part of its rays and leaves the rest to the queue's own waves
    (vr_kernels.hip, grab_chunk).  A launch of fewer waves than queues has queues WITHOUT waves of
    their own and must steal to the end; launches of a few, of about eight and of many waves, and
    the one-queue layout, all have to deliver every pixel the oracle delivers."""
    from volrend_amd import api
    tree = common.small_scene(depth=6, basis_dim=9, seed=407)
    t = api.N3Tree.from_synth(tree)
    t.set_tuning(xcd_queues=xcd_queues)
    torch = torch_cuda
    try:
        for w, h in ((8, 8), (17, 9), (24, 24), (40, 24), (64, 56), (128, 120), (264, 200)):
            f = w * 1111.111 / 800.0 * 1.3
            tr, _, _, _ = common.camera_for(pose_idx=5, size=w)
            cam = api.Camera(w, h, f, f)
            cam.transform = np.asarray(tr, dtype=np.float32)
            img = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
            acc = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
            api.launch_renderer(t, cam, api.RenderOptions(), img, None, torch.cuda.current_stream(),
                                True, accum=acc)
            torch.cuda.synchronize()
            assert t.status() == 0
            rgba_o, acc_o, cnt = common.oracle_frame(tree, tr, w, h, f)
            assert np.array_equal(img.cpu().numpy(), rgba_o), f"{w}x{h}: RGBA8 differs"
            assert np.array_equal(acc.cpu().numpy().view(np.uint32), acc_o.view(np.uint32)), f"{w}x{h}"
    finally:
        t.free_device()
