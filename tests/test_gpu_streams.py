"""-m gpu: launches in flight on several streams (the launch-slot ring of vr_api.cpp).

Every launch carries per-launch scratch in device memory; a slot is reused only behind the
event of the launch that held it last.  3 streams x 6 launches (> 8 slots, different poses and
image sizes, probe on some) must all reproduce the oracle."""
import ctypes as C

import numpy as np
import pytest

from tests import common

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return torch


def test_three_streams_six_launches_each(torch_cuda):
    torch = torch_cuda
    from volrend_amd import api
    tree = common.small_scene(depth=6, basis_dim=9, seed=931)
    t = api.N3Tree.from_synth(tree)
    streams = [torch.cuda.Stream() for _ in range(3)]
    jobs = []
    for j in range(18):
        size = [96, 64, 120][j % 3] + 8 * (j // 6)
        tr, w, h, f = common.camera_for(pose_idx=j % 8, size=size)
        kw = {}
        if j % 4 == 1:
            kw = dict(enable_probe=True, probe=(0.05 * j - 0.3, 0.1, 0.2), probe_disp_size=24,
                      basis_minmax=(0, 8))
        n = 1 + (j % 3)  # batch of n identical poses: every frame must equal the oracle's
        imgs = torch.zeros((n, h, w, 4), dtype=torch.uint8, device="cuda")
        jobs.append((tr, w, h, f, kw, n, imgs))
    torch.cuda.synchronize()
    for rnd in range(6):          # round-robin: launch j of stream s is job 3*rnd + s
        for si, st in enumerate(streams):
            tr, w, h, f, kw, n, imgs = jobs[3 * rnd + si]
            cam = api.Camera(w, h, f, f)
            with torch.cuda.stream(st):
                api.launch_renderer_batch(t, cam, [tr] * n, api.RenderOptions(**kw), list(imgs),
                                          st, True)
    torch.cuda.synchronize()
    for j, (tr, w, h, f, kw, n, imgs) in enumerate(jobs):
        okw = {k: (1 if v is True else v) for k, v in kw.items()}
        rgba_o, _, _ = common.oracle_frame(tree, tr, w, h, f, 0, **okw)
        got = imgs.cpu().numpy()
        for i in range(n):
            assert np.array_equal(got[i], rgba_o), f"job {j} frame {i}"
    t.free_device()


def test_reserve_status_and_step_size(torch_cuda):
    torch = torch_cuda
    from volrend_amd import _abi, api
    tree = common.small_scene(depth=4, basis_dim=4, seed=933)
    t = api.N3Tree.from_synth(tree)
    t.reserve(200, 120, 6)
    with pytest.raises(_abi.VolrendError):
        t.reserve(200, 120, _abi.MAX_BATCH + 1)
    tr, w, h, f = common.camera_for(pose_idx=2, size=64)
    cam = api.Camera(w, h, f, f)
    cam.transform = np.asarray(tr, np.float32)
    img = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
    for bad in (0.0, -1e-3, float("nan")):
        with pytest.raises(_abi.VolrendError) as e:
            api.launch_renderer(t, cam, api.RenderOptions(step_size=bad), img, None, None, True)
        assert e.value.code == 1 and "step_size" in str(e.value)
    api.launch_renderer(t, cam, api.RenderOptions(), img, None, None, True)
    torch.cuda.synchronize()
    assert t.status() == 0
    rgba_o, _, _ = common.oracle_frame(tree, tr, w, h, f, 0)
    assert np.array_equal(img.cpu().numpy(), rgba_o)
    t.free_device()


def test_tree_clone_renders_the_same_frames(torch_cuda):
    """vr_tree_clone: the device-to-device replica (here onto the same device -- the box has one)
    renders bit for bit what the original renders, both stay usable side by side, and freeing
    one leaves the other intact."""
    torch = torch_cuda
    from volrend_amd import api
    tree = common.small_scene(depth=6, basis_dim=16, seed=414)
    t = api.N3Tree.from_synth(tree)
    r = t.clone_to(0)
    assert r.info()["device_bytes"] == t.info()["device_bytes"] and r.info()["device"] == 0
    tr, w, h, f = common.camera_for(pose_idx=2, size=104)
    cam = api.Camera(w, h, f, f)
    rgba_o, acc_o, _ = common.oracle_frame(tree, tr, w, h, f, 0)
    imgs = torch.zeros((2, h, w, 4), dtype=torch.uint8, device="cuda")
    accs = torch.zeros((2, h, w, 4), dtype=torch.float32, device="cuda")
    s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
    for k, (tt, st) in enumerate(((t, s0), (r, s1))):
        with torch.cuda.stream(st):
            api.launch_renderer_batch(tt, cam, [tr], api.RenderOptions(), [imgs[k]], st, True,
                                      accums=[accs[k]])
    torch.cuda.synchronize()
    for k in range(2):
        assert np.array_equal(imgs[k].cpu().numpy(), rgba_o)
        assert np.array_equal(accs[k].cpu().numpy().view(np.uint32), acc_o.view(np.uint32))
    t.free_device()
    img = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
    api.launch_renderer_batch(r, cam, [tr], api.RenderOptions(), [img], torch.cuda.current_stream(),
                              True)
    torch.cuda.synchronize()
    assert np.array_equal(img.cpu().numpy(), rgba_o)
    with pytest.raises(api._abi.VolrendError):
        r.clone_to(99)
    r.free_device()


def test_per_tree_tuning_and_tile_reserve(torch_cuda):
    """Knobs live in the tree (vr_tree_set_tuning): two trees with opposite kernel organisations
    render the same bits side by side, a clone inherits its source's knobs, upload-time keys and
    unknown keys are refused.  vr_reserve_tiles sizes the slots with the launch's own rounding
    (tile rows that do not divide the height), several slots at once."""
    torch = torch_cuda
    from volrend_amd import _abi, api
    tree = common.small_scene(depth=6, basis_dim=9, seed=515)
    a = api.N3Tree.from_synth(tree)
    b = api.N3Tree.from_synth(tree)
    a.set_tuning(drain_flush=6, refill_min=8, march_max=4)
    b.set_tuning(refill_min=40, waves_per_cu=12)
    c = a.clone_to(0)  # inherits drain_flush=6, refill_min=8, march_max=4
    for bad_key in ("top_levels", "brick_levels", "no_such_knob"):
        with pytest.raises(_abi.VolrendError):
            a.set_tuning(**{bad_key: 3})
    w, h = 200, 100  # 100 rows, 24-row tiles: 5 tile rows, the last one ragged
    f = 260.0
    trs = [common.camera_for(pose_idx=i, size=64)[0] for i in range(3)]
    want = [common.oracle_frame(tree, tr, w, h, f, 0)[0] for tr in trs]
    cam = api.Camera(w, h, f, f)
    world, tw, th = 2, 200, 24
    for t in (a, b, c):
        t.reserve(w, h, len(trs), shard=api.TileShard(tw, th, 0, world, compact=True), n_slots=3)
        with pytest.raises(_abi.VolrendError):
            t.reserve(w, h, len(trs), n_slots=9)
        # whole frames (one launch of 3 poses, then a one-frame launch)
        imgs = torch.zeros((len(trs), h, w, 4), dtype=torch.uint8, device="cuda")
        api.launch_renderer_batch(t, cam, trs, api.RenderOptions(), list(imgs), None, True)
        single = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
        cam.transform = np.asarray(trs[1], np.float32)
        api.launch_renderer(t, cam, api.RenderOptions(), single, None, None, True)
        # the tile shard with ragged tile rows, gathered and assembled
        sh0 = api.TileShard(tw, th, 0, world, compact=True)
        nbytes = api.compact_bytes(w, h, sh0)
        gathered = torch.zeros((world, len(trs), nbytes), dtype=torch.uint8, device="cuda")
        for r in range(world):
            api.launch_renderer_batch(t, cam, trs, api.RenderOptions(),
                                      [gathered[r, i] for i in range(len(trs))], None, True,
                                      shard=api.TileShard(tw, th, r, world, compact=True))
        outs = torch.zeros((len(trs), h, w, 4), dtype=torch.uint8, device="cuda")
        api.assemble_tiles_batch(outs, gathered, len(trs), w, h, sh0, torch.cuda.current_stream())
        torch.cuda.synchronize()
        assert t.status() == 0
        assert np.array_equal(single.cpu().numpy(), want[1])
        for i in range(len(trs)):
            assert np.array_equal(imgs[i].cpu().numpy(), want[i]), ("frame", i)
            assert np.array_equal(outs[i].cpu().numpy(), want[i]), ("sharded", i)
    for t in (a, b, c):
        t.free_device()
