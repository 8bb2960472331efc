"""-m gpu: parity at BASELINE.json's full sizes.  The GPU box has enough host cores to run
the CPU oracle on whole 800x800 frames in well under a second, so the check is exact
(RGBA8 + fp32 accumulators bit-equal), complemented by size-independent properties:
counters identical to the oracle's, batch == single launches, tile shards == whole frame."""
import os
import sys

import numpy as np
import pytest

from tests.common import ob
from volrend_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return torch


def config_tree(name):
    sys.path.insert(0, ROOT)
    import bench
    return bench.load_or_make_tree(synth, name, 0, lambda: None)


@pytest.mark.parametrize("name,poses", [("C1", (0, 57, 133)), ("C2", (20,)), ("C3", (5,))])
def test_full_size_frames_bit_exact(torch_cuda, name, poses):
    torch = torch_cuda
    from volrend_amd import _abi, api
    cfg = synth.CONFIGS[name]
    stree = config_tree(name)
    W, H, focal = cfg["width"], cfg["height"], cfg["focal"]
    t = api.N3Tree.from_synth(stree)
    th = ob.TreeHandle(stree)
    cam = api.Camera(W, H, focal, focal)
    all_poses = synth.make_poses(200)
    trs = [synth.c2w_to_transform(all_poses[i]) for i in poses]
    n = len(trs)
    imgs = [torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda") for _ in range(n)]
    accs = [torch.zeros((H, W, 4), dtype=torch.float32, device="cuda") for _ in range(n)]
    cnts = [torch.zeros(7, dtype=torch.int64, device="cuda") for _ in range(n)]
    api.launch_renderer_batch(t, cam, trs, api.RenderOptions(), imgs, None, True, accums=accs,
                              counters=cnts)
    # the production flavour (no counters) in one batch, and as single launches
    fast = [torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda") for _ in range(n)]
    api.launch_renderer_batch(t, cam, trs, api.RenderOptions(), fast, None, True)
    single = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
    cam.transform = trs[0]
    api.launch_renderer(t, cam, api.RenderOptions(), single, None, None, True)
    # two interleaved tile shards into one frame
    sharded = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
    for r in range(2):
        api.launch_renderer(t, cam, api.RenderOptions(), sharded, None, None, True,
                            shard=api.TileShard((W + 7) // 8 * 8, 8, r, 2, compact=False))
    torch.cuda.synchronize()
    for i in range(n):
        ocam = ob.make_camera(trs[i], W, H, focal)
        rgba_o, acc_o, cnt_o = ob.render(th, ocam, ob.default_options())
        got = imgs[i].cpu().numpy()
        assert np.array_equal(got, rgba_o), f"{name} pose {poses[i]}: RGBA8 differs"
        assert np.array_equal(accs[i].cpu().numpy().view(np.uint32), acc_o.view(np.uint32))
        assert dict(zip(_abi.COUNTER_FIELDS, cnts[i].cpu().tolist())) == cnt_o
        assert np.array_equal(fast[i].cpu().numpy(), rgba_o)
        if i == 0:
            assert np.array_equal(single.cpu().numpy(), rgba_o)
            assert np.array_equal(sharded.cpu().numpy(), rgba_o)
            assert cnt_o["hit_samples"] > 1_000_000
    t.free_device()
