"""The kept C++ host layer (include/volrend/*.hpp, volrend_amd/csrc/host): npz reader,
N3Tree loader (plain / deflated / legacy / quantised / NDC sidecar), option parser, PNG
writer.  No device: N3Tree::upload_on_open is cleared by the helper."""
import os
import subprocess

import numpy as np
import pytest

from tests import common

from volrend_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def fnv(b: bytes) -> int:
    h = 1469598103934665603
    for byte in np.frombuffer(b, dtype=np.uint8).tolist():
        h = ((h ^ byte) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def fnv_np(a: np.ndarray) -> int:
    # vectorised FNV is awkward; arrays in these tests are small
    return fnv(np.ascontiguousarray(a).tobytes())


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    subprocess.check_call(["make", "-C", ROOT, "host"], stdout=subprocess.DEVNULL)
    out = str(tmp_path_factory.mktemp("bin") / "host_check")
    # the loader references vr_tree_upload & co: link the real library (never called here)
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "host_check.cpp"),
                           os.path.join(ROOT, "volrend_amd", "libvolrend_host.a"),
                           "-L", os.path.join(ROOT, "volrend_amd"), "-lvolrend_hip", "-lz",
                           "-Wl,-rpath," + os.path.join(ROOT, "volrend_amd"),
                           "-Wl,-rpath,/opt/rocm/lib", "-o", out])
    return out


def run(exe, *args):
    out = subprocess.check_output([exe, *args], text=True, stderr=subprocess.DEVNULL)
    return "\n".join(l for l in out.splitlines() if not l.startswith("INFO:"))


def parse_kv(line):
    return dict(tok.split("=", 1) for tok in line.split())


@pytest.mark.parametrize("compressed", [False, True])
def test_npz_members(exe, tmp_path, compressed):
    t = synth.make_tree(depth=3, basis_dim=4, seed=21)
    p = str(tmp_path / "t.npz")
    synth.save_npz(t, p, compressed=compressed)
    z = np.load(p)
    lines = {l.split()[0]: l for l in run(exe, "npz", p).splitlines()}
    assert set(lines) == set(z.files)
    for k in z.files:
        a = z[k]
        kv = parse_kv(lines[k].split(" ", 1)[1])
        assert kv["kind"] == a.dtype.kind
        assert int(kv["word"]) == a.dtype.itemsize
        assert [int(x) for x in kv["shape"].split(",") if x] == list(a.shape)
        assert int(kv["fnv"]) == fnv_np(a)


def test_tree_loader_plain_and_legacy_and_ndc(exe, tmp_path):
    t = synth.make_tree(depth=3, basis_dim=9, seed=22)
    p = str(tmp_path / "a.npz")
    synth.save_npz(t, p, compressed=True)
    out = run(exe, "tree", p).splitlines()
    facts = parse_kv(out[0])
    assert facts == dict(N="2", capacity=str(t.capacity), data_dim="28", format="SH9", loaded="1")
    assert "ndc=0" in out[2]
    kv = parse_kv(out[3])
    assert int(kv["child_fnv"]) == fnv_np(t.child) and int(kv["data_fnv"]) == fnv_np(t.data)
    assert out[4] == "pack=3,1,0,1"
    # legacy keys + LLFF sidecar
    q = str(tmp_path / "old.npz")
    np.savez(q, data_dim=np.int64(t.data_dim), child=t.child, data=t.data,
             invradius=np.float64(0.25), offset=t.offset)
    pb = np.zeros((3, 17))
    pb[:, 4], pb[:, 9], pb[:, 14] = 378.0, 504.0, 410.0
    pb[:, 15], pb[:, 16] = 1.0, 9.0
    pb[:, 0:15:5] = 0.1
    np.save(str(tmp_path / "old_poses_bounds.npy"), pb)
    out = run(exe, "tree", q).splitlines()
    assert parse_kv(out[0])["format"] == "SH9"
    assert out[1].startswith("scale=0.25,0.25,0.25")
    assert out[2].split()[0:4] == ["ndc=1", "504", "378", "410"]


def test_tree_loader_quantised(exe, tmp_path):
    t = synth.make_tree(depth=3, basis_dim=4, seed=23)
    p = str(tmp_path / "q.npz")
    common.write_quantised_npz(t, p, n_retain=1)
    out = run(exe, "tree", p).splitlines()
    assert int(parse_kv(out[3])["data_fnv"]) == fnv_np(t.data)


def test_loader_errors(exe, tmp_path):
    t = synth.make_tree(depth=2, basis_dim=1, seed=24)
    p = str(tmp_path / "f32.npz")
    np.savez(p, data_dim=np.int64(t.data_dim), data_format=np.array("SH1"), child=t.child,
             data=t.data.astype(np.float32), invradius3=t.invradius3, offset=t.offset)
    r = subprocess.run([exe, "tree", p], capture_output=True, text=True)
    assert r.returncode == 3 and "half precision" in r.stdout
    r = subprocess.run([exe, "tree", str(tmp_path / "nope.npz")], capture_output=True, text=True)
    assert "does not exist" in r.stdout and "loaded=0" in r.stdout


def test_png_writer(exe, tmp_path):
    from PIL import Image
    p = str(tmp_path / "g.png")
    run(exe, "png", p, "37", "21")
    im = np.asarray(Image.open(p))
    assert im.shape == (21, 37, 4)
    ys, xs = np.mgrid[0:21, 0:37]
    assert np.array_equal(im[..., 0], (xs * 3) % 256) and np.array_equal(im[..., 1], (ys * 5) % 256)
    assert np.array_equal(im[..., 2], (xs ^ ys) % 256) and (im[..., 3] == 255).all()


@pytest.mark.parametrize("w,h", [(37, 21), (800, 800), (1, 1), (5000, 4)])
def test_png_is_stored_not_deflated(exe, tmp_path, w, h):
    """The reference asks libpng for level 0 / no filter (src/imwrite.cpp:29-31): the IDAT is a
    zlib stream of STORED blocks.  Chunk CRCs, block structure, Adler-32 and the exact file size."""
    import struct
    import zlib
    p = str(tmp_path / "s.png")
    run(exe, "png", p, str(w), str(h))
    b = open(p, "rb").read()
    assert b[:8] == b"\x89PNG\r\n\x1a\n"
    at, chunks = 8, []
    while at < len(b):
        n, typ = struct.unpack(">I4s", b[at:at + 8])
        body = b[at + 8:at + 8 + n]
        assert struct.unpack(">I", b[at + 8 + n:at + 12 + n])[0] == zlib.crc32(typ + body)
        chunks.append((typ, body))
        at += 12 + n
    assert [c[0] for c in chunks] == [b"IHDR", b"IDAT", b"IEND"]
    assert struct.unpack(">IIBBBBB", chunks[0][1]) == (w, h, 8, 6, 0, 0, 0)
    z = chunks[1][1]
    raw_len = (w * 4 + 1) * h
    assert z[:2] == b"\x78\x01"
    at, got = 2, bytearray()
    while True:
        final, n, nn = z[at], *struct.unpack("<HH", z[at + 1:at + 5])
        assert final in (0, 1) and n == (~nn & 0xFFFF)      # BTYPE 00: stored
        got += z[at + 5:at + 5 + n]
        at += 5 + n
        if final:
            break
        assert n == 65535
    assert len(got) == raw_len and struct.unpack(">I", z[at:at + 4])[0] == zlib.adler32(bytes(got))
    assert at + 4 == len(z) and zlib.decompress(z) == bytes(got)
    n_blocks = (raw_len + 65534) // 65535
    assert len(b) == 8 + 25 + 12 + (2 + 5 * n_blocks + raw_len + 4) + 12
    rows = np.frombuffer(bytes(got), np.uint8).reshape(h, w * 4 + 1)
    assert (rows[:, 0] == 0).all()                           # filter type none on every row


def test_option_parser(exe):
    out = run(exe, "opts", "tree.npz", "pose/0000.txt", "-w", "400", "--height=300", "--fx", "555.5",
              "-o", "out", "-r", "-s", "1e-3", "--bg", "0.5", "pose/0001.txt", "--unknown", "-e",
              "0.05").splitlines()
    kv = parse_kv(out[0])
    assert kv["file"] == "tree.npz" and kv["w"] == "400" and kv["h"] == "300"
    assert float(kv["fx"]) == pytest.approx(555.5) and float(kv["bg"]) == 0.5
    assert float(kv["step"]) == pytest.approx(1e-3) and float(kv["stop"]) == pytest.approx(0.05)
    assert float(kv["sigma"]) == pytest.approx(1e-2) and kv["out"] == "out" and kv["r"] == "1"
    assert kv["gpu"] == "-1"
    assert out[1:] == ["unmatched=pose/0000.txt", "unmatched=pose/0001.txt", "unmatched=--unknown"]


def test_large_stored_members_are_viewed_zero_copy(exe, tmp_path):
    """np.savez (stored, ZIP64 local headers) with MB-sized members: the loader maps the
    file and hands out views; content must be identical."""
    t = synth.make_tree(depth=5, basis_dim=9, seed=25)
    assert t.data.nbytes > (1 << 16)  # above the loader's zero-copy threshold
    p = str(tmp_path / "big.npz")
    synth.save_npz(t, p, compressed=False)
    out = run(exe, "tree", p).splitlines()
    assert parse_kv(out[0])["capacity"] == str(t.capacity)
    kv = parse_kv(out[3])
    assert int(kv["child_fnv"]) == fnv_np(t.child) and int(kv["data_fnv"]) == fnv_np(t.data)


def test_cli_pose_and_intrinsics_parsing(tmp_path):
    """Pose / intrinsics text formats of the reference CLI (main_headless.cpp:40-75): 4x4 and 3x4
    matrices, stacked 4x4, a file that ends inside a matrix, trailing junk, -r column flips."""
    cli = os.path.join(ROOT, "volrend_amd", "bin", "volrend_headless")
    if not os.path.exists(cli):
        subprocess.check_call(["make", "-C", ROOT, "cli"], stdout=subprocess.DEVNULL)
    rng = np.random.default_rng(3)
    a, b = rng.normal(size=(4, 4)), rng.normal(size=(4, 4))
    f44, f34, fst, fcut, fjunk = (str(tmp_path / n) for n in
                                  ("a44.txt", "a34.txt", "stack.txt", "cut.txt", "junk.txt"))
    np.savetxt(f44, a)
    np.savetxt(f34, a[:3])
    np.savetxt(fst, np.concatenate([a, b]))
    open(fcut, "w").write(" ".join(f"{x:.9g}" for x in a.reshape(-1)[:7]))       # ends in row 2
    open(fjunk, "w").write(" ".join(f"{x:.9g}" for x in a.reshape(-1)[:12]) + " # comment 1 2 3")
    K = np.diag([555.5, 444.25, 1.0, 1.0])
    fk = str(tmp_path / "intrinsics.txt")
    np.savetxt(fk, K)

    def dump(*args):
        r = subprocess.run([cli, "tree.npz", *args, "--dump_poses"], capture_output=True, text=True,
                           timeout=60)
        assert r.returncode == 0, r.stderr
        poses = [l.split() for l in r.stdout.splitlines() if l.startswith("pose ")]
        intr = [l.split() for l in r.stdout.splitlines() if l.startswith("intrin ")]
        return ({p[1]: np.array(p[2:], dtype=np.float64).reshape(4, 3).T for p in poses}, intr)

    poses, intr = dump(f44, f34, fst, fcut, fjunk, "-i", fk)
    want = np.float32(a[:3])
    assert np.allclose(poses["a44"], want, rtol=1e-6) and np.allclose(poses["a34"], want, rtol=1e-6)
    assert np.allclose(poses["stack_000000"], want, rtol=1e-6)
    assert np.allclose(poses["stack_000001"], np.float32(b[:3]), rtol=1e-6)
    cut = np.zeros(12)
    cut[:7] = a.reshape(-1)[:7]
    assert np.allclose(poses["cut"], np.float32(cut.reshape(3, 4)), rtol=1e-6)
    assert np.allclose(poses["junk"], want, rtol=1e-6)
    assert len(poses) == 6
    assert float(intr[0][1]) == pytest.approx(555.5) and float(intr[0][2]) == pytest.approx(444.25)
    flipped, _ = dump(f44, "-r")
    assert np.allclose(flipped["a44"][:, 1], -want[:, 1], rtol=1e-6)
    assert np.allclose(flipped["a44"][:, 2], -want[:, 2], rtol=1e-6)
    assert np.allclose(flipped["a44"][:, [0, 3]], want[:, [0, 3]], rtol=1e-6)


def test_cli_launch_plan(tmp_path):
    """How volrend_headless cuts its pose list into launches (round 5): ceil(P / batch) EQUAL
    launches (sizes differ by at most one pose: no short last launch), batch x N poses per launch
    under --gpus N (at most VR_MAX_BATCH = 512), two render streams for small launches."""
    cli = os.path.join(ROOT, "volrend_amd", "bin", "volrend_headless")
    subprocess.check_call(["make", "-C", ROOT, "cli"], stdout=subprocess.DEVNULL)
    f = str(tmp_path / "poses.txt")
    np.savetxt(f, np.concatenate([np.eye(4)] * 200))

    def plan(*args):
        r = subprocess.run([cli, "tree.npz", f, *args, "--dump_poses"], capture_output=True, text=True,
                           timeout=60)
        assert r.returncode == 0, r.stderr
        w = [l for l in r.stdout.splitlines() if l.startswith("plan ")][0].split()
        return {w[i]: int(w[i + 1]) for i in range(1, len(w), 2)}

    assert plan() == dict(poses=200, launches=4, long=4, batch=50, streams=1)       # default --batch 64
    p = plan("--batch", "32")                                                        # 29 29 29 29 28 28 28
    assert p == dict(poses=200, launches=7, long=4, batch=29, streams=2)
    assert p["long"] * p["batch"] + (p["launches"] - p["long"]) * (p["batch"] - 1) == 200
    assert plan("--batch", "1") == dict(poses=200, launches=200, long=200, batch=1, streams=2)
    assert plan("--batch", "1", "--streams", "1")["streams"] == 1
    assert plan("--batch", "300") == dict(poses=200, launches=1, long=1, batch=200, streams=1)
    assert plan("--max_imgs", "7", "--batch", "4") == dict(poses=7, launches=2, long=1, batch=4, streams=2)
    # --gpus N: the launch grows with N (work per rank and launch as on one GPU), capped at 512
    assert plan("--gpus", "2") == dict(poses=200, launches=2, long=2, batch=100, streams=1)
    assert plan("--gpus", "8") == dict(poses=200, launches=1, long=1, batch=200, streams=1)
    assert plan("--gpus", "8", "--batch", "8") == dict(poses=200, launches=4, long=4, batch=50, streams=1)
    assert plan("--gpus", "8", "--batch", "500")["batch"] == 200


REF = "/root/reference"


def _offsets(tmp_path, header_dir_flags, defines=()):
    """Offsets of every RenderOptions member + sizeof, as a C++ compiler sees the given header."""
    src = tmp_path / "ro.cpp"
    src.write_text(
        '#include <cstddef>\n#include <cstdio>\n#include "volrend/render_options.hpp"\n'
        "int main() { using R = volrend::RenderOptions;\n"
        '  printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", offsetof(R, step_size),'
        " offsetof(R, sigma_thresh), offsetof(R, stop_thresh), offsetof(R, background_brightness),"
        " offsetof(R, render_bbox), offsetof(R, basis_minmax), offsetof(R, rot_dirs), offsetof(R, show_grid),"
        " offsetof(R, grid_max_depth), offsetof(R, render_depth), offsetof(R, enable_probe), offsetof(R, probe),"
        " offsetof(R, probe_disp_size), sizeof(R)); }\n")
    exe = str(tmp_path / "ro")
    subprocess.check_call(["g++", "-std=c++17", *defines, *header_dir_flags, str(src), "-o", exe])
    return [int(x) for x in subprocess.check_output([exe], text=True).split()]


def test_render_options_layout_is_the_references(tmp_path):
    """volrend::RenderOptions: same member order and offsets as the reference's struct in a
    VOLREND_CUDA build (include/volrend/render_options.hpp:11-53) -- aggregate initialisation and
    offset-based bindings carry over.  Against the reference's own header where it is mounted,
    against the numbers it compiles to (pinned in our header's static_assert) everywhere."""
    ours = _offsets(tmp_path, ["-I", os.path.join(ROOT, "include")])
    assert ours == [0, 4, 8, 12, 16, 40, 48, 60, 64, 68, 69, 72, 84, 88]
    if os.path.isdir(os.path.join(REF, "include", "volrend")):
        shim = tmp_path / "shim" / "volrend"
        shim.mkdir(parents=True)
        (shim / "common.hpp").write_text("#pragma once\n")  # (the reference's is generated by CMake)
        ref_dir = tmp_path / "ref" / "volrend"
        ref_dir.mkdir(parents=True)
        (ref_dir / "render_options.hpp").write_text(
            open(os.path.join(REF, "include", "volrend", "render_options.hpp")).read())
        theirs = _offsets(tmp_path, ["-I", str(tmp_path / "shim"), "-I", str(tmp_path / "ref")],
                          defines=["-DVOLREND_CUDA"])
        assert theirs == ours


def test_volume_renderer_interface_compiles_like_upstreams(tmp_path):
    """include/volrend/renderer.hpp offers the members the reference's VolumeRenderer has
    (include/volrend/renderer.hpp:11-42; `meshes` excepted: GL work) -- a caller written against
    upstream compiles.  Compile only: the facade needs a device to run (tests/test_gpu_renderer.py)."""
    src = tmp_path / "use.cpp"
    src.write_text(
        '#include "volrend/renderer.hpp"\n'
        "void drive(volrend::N3Tree& tree) {\n"
        "  volrend::VolumeRenderer rend;\n"
        "  rend.options.step_size = 1e-3f; rend.options.show_grid = false;\n"
        "  rend.camera.center = glm::vec3(0.f, 0.f, 3.f);\n"
        "  rend.resize(800, 800); rend.set(tree); rend.render(); rend.clear();\n"
        '  const char* b = rend.get_backend(); (void)b;\n'
        "  static_assert(!std::is_copy_constructible<volrend::VolumeRenderer>::value, \"\");\n"
        "}\n")
    subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-I", os.path.join(ROOT, "include"),
                           "-include", "type_traits", str(src)])
