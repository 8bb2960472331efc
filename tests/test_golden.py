"""Golden vectors (tests/golden/golden_v1.npz, made by tests/golden/make_golden.py from
the host build of the reference's own device code).

CPU: the oracle reproduces them bit for bit (strict = reference, fma = contraction model).
GPU (-m gpu): the HIP kernel, through the C ABI, reproduces them bit for bit."""
import json
import os

import numpy as np
import pytest

from tests.common import ob
from volrend_amd import synth

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_v1.npz")
Z = np.load(PATH)
INDEX = json.loads(bytes(Z["index"]).decode())


def load_case(c):
    n = c["name"]
    child = Z[n + "/child"]
    data = Z[n + "/data"].view(np.float16)
    extra = Z[n + "/extra"] if (n + "/extra") in Z.files else None
    tree = synth.SynthTree(child, data, np.asarray(c["offset"], np.float32),
                           np.asarray(c["invradius3"], np.float32), c["data_format"], extra)
    return tree


@pytest.mark.parametrize("case", INDEX, ids=[c["name"] for c in INDEX])
@pytest.mark.parametrize("mode", ["strict", "fma"])
def test_oracle_reproduces_golden(case, mode):
    tree = load_case(case)
    th = ob.TreeHandle(tree, ndc=tuple(case["ndc"]) if case["ndc"] else None)
    cam = ob.make_camera(case["transform"], case["width"], case["height"], case["focal"])
    opt = ob.default_options(**case["opts"])
    rgba, acc, cnt = ob.render(th, cam, opt, ob.FP_STRICT if mode == "strict" else ob.FP_FMA)
    assert np.array_equal(rgba, Z[f"{case['name']}/rgba_{mode}"])
    assert np.array_equal(acc.view(np.uint32), Z[f"{case['name']}/accum_{mode}"])
    if mode == "strict":
        assert cnt == case["counters"]


@pytest.mark.gpu
@pytest.mark.parametrize("case", INDEX, ids=[c["name"] for c in INDEX])
@pytest.mark.parametrize("mode", ["strict", "fma"])
def test_kernel_reproduces_golden(case, mode):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    from volrend_amd import _abi, api
    tree = load_case(case)
    t = api.N3Tree.from_synth(tree, ndc=tuple(case["ndc"]) if case["ndc"] else None)
    w, h = case["width"], case["height"]
    cam = api.Camera(w, h, case["focal"], case["focal"])
    cam.transform = np.asarray(case["transform"], dtype=np.float32)
    o = {k: (tuple(v) if isinstance(v, list) else v) for k, v in case["opts"].items()}
    if "render_depth" in o:
        o["render_depth"] = bool(o["render_depth"])
    opts = api.RenderOptions(**o)
    img = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
    acc = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    api.launch_renderer(t, cam, opts, img, None, torch.cuda.current_stream(), True, accum=acc,
                        fp_mode=_abi.FP_STRICT if mode == "strict" else _abi.FP_FMA)
    torch.cuda.synchronize()
    assert np.array_equal(img.cpu().numpy(), Z[f"{case['name']}/rgba_{mode}"])
    assert np.array_equal(acc.cpu().numpy().view(np.uint32), Z[f"{case['name']}/accum_{mode}"])
    t.free_device()
