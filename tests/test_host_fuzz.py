"""Hostile tree.npz files: the C++ loader (own zip / npy reader + N3Tree::load_npz) must reject
truncated and corrupted archives with an error, never read out of bounds.  The check binary is
built with AddressSanitizer + UBSan from the host sources; mutations are seeded."""
import os
import random
import subprocess

import pytest

from tests import common
from volrend_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "volrend_amd", "csrc", "host")


@pytest.fixture(scope="module")
def asan_exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("asan") / "host_check_asan")
    srcs = [os.path.join(HOST, f) for f in ("npz.cpp", "n3tree.cpp", "camera.cpp", "opts.cpp",
                                            "imwrite.cpp", "renderer.cpp")]
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined",
           "-fno-omit-frame-pointer", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", "host_check.cpp"), *srcs,
           "-L", os.path.join(ROOT, "volrend_amd"), "-lvolrend_hip", "-lz", "-lpthread",
           "-Wl,-rpath," + os.path.join(ROOT, "volrend_amd"), "-Wl,-rpath,/opt/rocm/lib", "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("no sanitizer build here: " + r.stderr[-300:])
    return out


def mutations(data: bytes, rng: random.Random, n: int):
    size = len(data)
    for cut in list(range(size - 140, size, 14)) + [0, 1, 4, 10, 22, 30, 64, 100, size // 3, size // 2]:
        yield data[:max(cut, 0)]
    for _ in range(n):
        b = bytearray(data)
        for _ in range(rng.choice([1, 1, 2, 4])):
            r = rng.random()
            if r < 0.45:    # local headers / npy preambles of the first members
                i = rng.randrange(min(600, size))
            elif r < 0.85:  # central directory, zip64 records, end record
                i = size - 1 - rng.randrange(min(700, size))
            else:
                i = rng.randrange(size)
            b[i] = rng.choice([0, 1, 0xFF, 0x7F, 0x80, rng.randrange(256)])
        yield bytes(b)


def test_loader_survives_corrupted_archives(asan_exe, tmp_path):
    tree = synth.make_tree(depth=3, basis_dim=4, seed=5)
    files = []
    for name, writer in (("c.npz", lambda p: synth.save_npz(tree, p, compressed=True)),
                         ("s.npz", lambda p: synth.save_npz(tree, p, compressed=False)),
                         ("q.npz", lambda p: common.write_quantised_npz(tree, p, n_retain=1))):
        p = str(tmp_path / name)
        writer(p)
        files.append(p)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")
    rng = random.Random(20260924)
    case = str(tmp_path / "case.npz")
    n_cases = n_rejected = 0
    for f in files:
        data = open(f, "rb").read()
        ok = subprocess.run([asan_exe, "tree", f], capture_output=True, env=env, timeout=120)
        assert ok.returncode == 0, ok.stderr.decode("latin1")[-500:]
        for mutated in mutations(data, rng, 90):
            open(case, "wb").write(mutated)
            r = subprocess.run([asan_exe, "tree", case], capture_output=True, env=env, timeout=120)
            err = r.stderr.decode("latin1")
            n_cases += 1
            n_rejected += r.returncode != 0
            assert r.returncode >= 0, f"loader died with signal {-r.returncode}"
            assert "AddressSanitizer" not in err and "runtime error" not in err, err[-1500:]
    assert n_cases > 300 and n_rejected > 50


def test_loader_rejects_wrong_rank_arrays(asan_exe, tmp_path):
    """Well-formed archives whose arrays have the wrong RANK (a byte-flip fuzzer never makes
    these): 1-D quant_map, 0-d data / quant_colors / data_retained, short offset / invradius3.
    Every one must be refused with an error -- no out-of-bounds shape read, no null deref."""
    import numpy as np
    tree = synth.make_tree(depth=3, basis_dim=4, seed=6)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")
    plain = str(tmp_path / "plain.npz")
    synth.save_npz(tree, plain, compressed=False)
    quant = str(tmp_path / "quant.npz")
    common.write_quantised_npz(tree, quant, n_retain=1, compressed=False)

    def variant(src, name, **replace):
        z = dict(np.load(src))
        for k, v in replace.items():
            if v is None:
                z.pop(k, None)
            else:
                z[k] = v
        p = str(tmp_path / name)
        np.savez(p, **z)
        return p

    zq = np.load(quant)
    cases = [
        variant(plain, "data_0d.npz", data=np.float16(1.0)),
        variant(plain, "data_1d.npz", data=np.zeros(7, np.float16)),
        variant(plain, "data_4d.npz", data=np.zeros((tree.capacity, 2, 2, 2), np.float16)),
        variant(plain, "offset_short.npz", offset=np.zeros(2, np.float32)),
        variant(plain, "offset_0d.npz", offset=np.float32(0.5)),
        variant(plain, "invradius3_short.npz", invradius3=np.zeros(1, np.float32)),
        variant(plain, "child_3d.npz", child=np.zeros((tree.capacity, 2, 2), np.int32)),
        variant(quant, "qm_1d.npz", quant_map=zq["quant_map"].reshape(-1)),
        variant(quant, "qm_0d.npz", quant_map=np.uint16(3)),
        variant(quant, "qc_0d.npz", quant_colors=np.float16(0.5)),
        variant(quant, "qc_2d.npz", quant_colors=zq["quant_colors"].reshape(-1, 3)),
        variant(quant, "ret_0d.npz", data_retained=np.float16(0.25)),
        variant(quant, "ret_2d.npz", data_retained=zq["data_retained"].reshape(-1, 3)),
        variant(quant, "sigma_0d.npz", sigma=np.float16(2.0)),
    ]
    for ok in (plain, quant):
        r = subprocess.run([asan_exe, "tree", ok], capture_output=True, env=env, timeout=120)
        assert r.returncode == 0, r.stderr.decode("latin1")[-500:]
    for p in cases:
        r = subprocess.run([asan_exe, "tree", p], capture_output=True, env=env, timeout=120)
        err = r.stderr.decode("latin1")
        assert r.returncode > 0, f"{os.path.basename(p)} was accepted (rc {r.returncode})"
        assert "AddressSanitizer" not in err and "runtime error" not in err, \
            os.path.basename(p) + ": " + err[-1500:]
