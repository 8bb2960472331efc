"""-m gpu: the distinct-line meter behind bench.py's B_unique (vr_touch_enable / vr_touch_count).

The bitmaps live in the device layout, which the oracle does not have, so the checks are the
properties a set-of-lines count must satisfy plus the cases that can be counted by hand."""
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


def _render_counted(torch, api, t, tr, w, h, f):
    img = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(7, dtype=torch.int64, device="cuda")
    cam = api.Camera(w, h, f, f)
    api.launch_renderer_batch(t, cam, [tr], api.RenderOptions(), [img],
                              torch.cuda.current_stream(), True, counters=[cnt])
    torch.cuda.synchronize()
    return img.cpu().numpy(), dict(zip(api._abi.COUNTER_FIELDS, cnt.cpu().tolist()))


def test_touch_counts_are_a_set_measure(torch_cuda):
    torch = torch_cuda
    from volrend_amd import api
    tree = common.small_scene(depth=6, basis_dim=16, seed=77)
    t = api.N3Tree.from_synth(tree)
    stride = t.info()["leaf_stride"]
    assert stride == 128  # SH16: one record per 128-byte line
    tr_a, w, h, f = common.camera_for(pose_idx=1, size=96)
    tr_b = common.camera_for(pose_idx=5, size=96)[0]

    # off: instrumented launches work and nothing can be counted
    _render_counted(torch, api, t, tr_a, w, h, f)
    with pytest.raises(api._abi.VolrendError):
        t.touch_count()

    t.touch_enable(True)
    img_a, cnt_a = _render_counted(torch, api, t, tr_a, w, h, f)
    a = t.touch_count(reset=False)
    # the bitmaps themselves (vr_touch_read): one bit per 128-byte line, as many set as counted
    for which, key in enumerate(("leaves", "nodes", "top", "bricks")):
        bm, gran = t.touch_read(which)
        assert gran == 128
        assert int(np.unpackbits(bm.view(np.uint8)).sum()) == a[key], key
    _render_counted(torch, api, t, tr_a, w, h, f)        # the same frame again: same set
    assert t.touch_count(reset=True) == a
    assert t.touch_count(reset=False) == dict(leaves=0, nodes=0, top=0, bricks=0)  # reset worked
    _render_counted(torch, api, t, tr_b, w, h, f)
    b = t.touch_count(reset=False)
    _render_counted(torch, api, t, tr_a, w, h, f)        # union of both frames
    u = t.touch_count(reset=True)
    for k in a:
        assert max(a[k], b[k]) <= u[k] <= a[k] + b[k], k
    # a record line is touched only by a hit sample; there are no more distinct ones than hits
    assert 0 < a["leaves"] <= cnt_a["hit_samples"]
    occupied = int((tree.data[..., -1].astype(np.float32) > 1e-2).sum())
    assert a["leaves"] <= occupied
    # every sample reads the top grid or a brick; the whole structure is small
    assert a["top"] > 0 and a["top"] + a["bricks"] + a["nodes"] <= cnt_a["samples"]
    # the meter does not change the picture
    rgba_o, _, _ = common.oracle_frame(tree, tr_a, w, h, f, 0)
    assert np.array_equal(img_a, rgba_o)
    t.touch_enable(False)
    t.free_device()


def test_touch_count_of_a_tree_counted_by_hand(torch_cuda):
    """One level: the root's 8 leaves.  The lookup structure is a 2^3 top grid (64 bytes = one
    line) that resolves every leaf; a camera that sees the whole cube touches every occupied
    leaf's record (SH16: one line each) and nothing else."""
    torch = torch_cuda
    from volrend_amd import api
    tree = common.small_scene(depth=1, basis_dim=16, seed=5)
    occupied = int((tree.data[..., -1].astype(np.float32) > 1e-2).sum())
    assert occupied > 0
    t = api.N3Tree.from_synth(tree)
    t.touch_enable(True)
    tr, w, h, f = common.camera_for(pose_idx=1, size=64, focal=40.0)
    _render_counted(torch, api, t, tr, w, h, f)
    got = t.touch_count()
    assert got["top"] == 1 and got["bricks"] == 0 and got["nodes"] == 0
    assert 0 < got["leaves"] <= occupied
    t.free_device()
