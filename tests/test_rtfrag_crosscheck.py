"""Cross-check of the oracle against the reference's OTHER backend: shaders/rt.frag (read
from the reference mount at run time) executed on a software GL rasteriser
(oracle/ref_build/rtfrag_baseline.py).  The GLSL path differs from the CUDA path by
documented rounding details (UNORM8 rounds to nearest, exp/pow in the
rasteriser's own precision), so agreement is PSNR-level: > 45 dB, no pixel off by more than 2.
Skipped where the reference mount or SwiftShader is missing (e.g. on the GPU box)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPT = os.path.join(ROOT, "oracle", "ref_build", "rtfrag_baseline.py")


def available():
    if not os.path.exists("/root/reference/shaders/rt.frag"):
        return False
    sys.path.insert(0, os.path.join(ROOT, "oracle", "ref_build"))
    import rtfrag_baseline
    return rtfrag_baseline.find_swiftshader() is not None


@pytest.mark.skipif(not available(), reason="needs /root/reference and a software GL")
def test_rt_frag_agrees_with_oracle(tmp_path):
    out = str(tmp_path / "r.json")
    subprocess.check_call([sys.executable, SCRIPT, "--size", "64", "--pose", "37", "--out", out],
                          stdout=subprocess.DEVNULL, timeout=600)
    r = json.load(open(out))
    assert r["gl_error"] == 0
    assert r["psnr_rt_frag_vs_oracle_db"] > 45.0
    assert r["max_abs_diff"] <= 2 and r["pixels_differing_by_more_than_2"] == 0
    assert r["samples"] > 10000
