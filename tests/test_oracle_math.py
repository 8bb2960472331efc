"""Building blocks of the oracle: deterministic expf, fp16 conversion, counters."""
import numpy as np

from tests import common
from tests.common import ob


def test_half_conversion_exhaustive():
    L = ob.lib()
    bits = np.arange(65536, dtype=np.uint16)
    want = bits.view(np.float16).astype(np.float32)
    got = np.array([L.or_half2float(int(b)) for b in bits], dtype=np.float32)
    nan = np.isnan(want)
    assert np.array_equal(np.isnan(got), nan)
    assert np.array_equal(got[~nan].view(np.uint32), want[~nan].view(np.uint32))


def test_det_expf_accuracy_and_edges():
    L = ob.lib()
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.uniform(-90, 88, 20000), rng.uniform(-1, 1, 20000),
                         np.array([0.0, -0.0, 1.0, -1.0, 88.7, -87.3, -100.0, -103.9])]).astype(
                             np.float32)
    got = np.array([L.or_expf(float(x)) for x in xs], dtype=np.float32)
    ref = np.exp(xs.astype(np.float64))
    ok = (ref > 1.2e-38) & (ref < 3.0e38)
    ulp = np.abs(got[ok].astype(np.float64) - ref[ok]) / np.spacing(ref[ok].astype(np.float32))
    assert ulp.max() < 1.0, ulp.max()
    assert L.or_expf(-200.0) == 0.0
    assert np.isinf(L.or_expf(100.0))
    assert np.isnan(L.or_expf(float("nan")))
    assert L.or_expf(0.0) == 1.0


def test_counters_consistent():
    tree = common.small_scene(depth=6, basis_dim=16, seed=5)
    tr, w, h, f = common.camera_for(pose_idx=2, size=64)
    _, _, c = common.oracle_frame(tree, tr, w, h, f)
    dd = tree.data_dim
    assert c["rays"] == w * h
    assert c["child_reads"] >= c["samples"] >= c["hit_samples"] >= c["early_stops"]
    assert c["alg_bytes"] == 4 * c["child_reads"] + 2 * c["samples"] + \
        2 * (dd - 1) * c["hit_samples"] + 4 * w * h
    # single-threaded and multi-threaded runs agree
    th = ob.TreeHandle(tree)
    cam = ob.make_camera(tr, w, h, f)
    a, _, c1 = ob.render(th, cam, ob.default_options(), nthreads=1)
    b, _, c2 = ob.render(th, cam, ob.default_options(), nthreads=5)
    assert np.array_equal(a, b) and c1 == c2


def test_region_render_matches_full_frame():
    tree = common.small_scene(depth=5, basis_dim=4, seed=6)
    tr, w, h, f = common.camera_for(pose_idx=1, size=50)
    full, _, _ = common.oracle_frame(tree, tr, w, h, f)
    part, _, _ = common.oracle_frame(tree, tr, w, h, f, region=(10, 20, 30, 17))
    assert np.array_equal(part[20:37, 10:40], full[20:37, 10:40])
    assert (part[:20] == 0).all()
