"""Host-side logic: synthetic assets, the tree.npz format, DataFormat parsing, the
quantised-codebook decode, pose files, tile sharding math."""
import os

import numpy as np
import pytest

from tests import common

from volrend_amd import api, synth, tiles

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_data_format_parse():
    # DataFormat::parse, src/n3tree.cpp:55-78
    assert api.parse_data_format("SH16") == ("SH", 16)
    assert api.parse_data_format("SG25") == ("SG", 25)
    assert api.parse_data_format("ASG4") == ("ASG", 4)
    assert api.parse_data_format("RGBA") == ("RGBA", -1)
    assert api.parse_data_format("XYZ9")[0] == "RGBA"


def test_tree_topology_invariants():
    t = synth.make_tree(depth=5, basis_dim=4, seed=1)
    child = t.child.reshape(t.capacity, 8)
    n = np.arange(t.capacity)[:, None]
    tgt = np.where(child != 0, n + child, -1)
    linked = tgt[tgt >= 0]
    assert (linked > 0).all() and (linked < t.capacity).all()
    assert len(np.unique(linked)) == len(linked) == t.capacity - 1  # every non-root node once
    # breadth-first numbering: children come after their parent
    assert (child >= 0).all()
    # occupied leaves only at the finest level, all other records are zero
    sig = t.data.reshape(t.capacity, 8, -1)[..., -1]
    assert (sig[child != 0] == 0).all()
    assert (sig > 0).sum() == t.meta["shell_leaves"] > 0


def test_npz_round_trip(tmp_path):
    t = synth.make_tree(depth=4, basis_dim=9, seed=2)
    for compressed in (False, True):
        p = str(tmp_path / f"tree_{int(compressed)}.npz")
        synth.save_npz(t, p, compressed=compressed)
        z = np.load(p)
        # key layout read by N3Tree::load_npz (src/n3tree.cpp:228-277)
        assert z["data_dim"].dtype == np.int64 and z["data_dim"].shape == ()
        assert z["data_format"].dtype.kind == "U" and str(z["data_format"]) == "SH9"
        assert z["child"].dtype == np.int32 and z["child"].shape == (t.capacity, 2, 2, 2)
        assert z["data"].dtype == np.float16 and z["data"].shape[-1] == 28
        assert z["invradius3"].dtype == np.float32 and z["offset"].dtype == np.float32
        n = api.N3Tree()
        n.open(p, upload=False)
        assert n.N == 2 and n.capacity == t.capacity and n.data_dim == 28
        assert n.data_format == ("SH", 9)
        assert np.array_equal(n.child_, t.child) and np.array_equal(n.data_, t.data)
        assert np.allclose(n.scale, t.invradius3) and np.allclose(n.offset, t.offset)
        assert not n.use_ndc


def test_legacy_npz_without_format_and_ndc_sidecar(tmp_path):
    t = synth.make_tree(depth=3, basis_dim=4, seed=3)
    p = str(tmp_path / "old.npz")
    np.savez(p, data_dim=np.int64(t.data_dim), child=t.child, data=t.data,
             invradius=np.float64(0.25), offset=t.offset)
    pb = np.zeros((3, 17))
    pb[:, 4], pb[:, 9], pb[:, 14] = 378.0, 504.0, 410.0  # H, W, focal (n3tree.cpp:26-28)
    np.save(str(tmp_path / "old_poses_bounds.npy"), pb)
    n = api.N3Tree()
    n.open(p, upload=False)
    assert n.data_format == ("SH", 4)          # autodetect, n3tree.cpp:247-253
    assert np.allclose(n.scale, 0.25)          # scalar invradius, n3tree.cpp:260-262
    assert n.use_ndc and (n.ndc_width, n.ndc_height, n.ndc_focal) == (504.0, 378.0, 410.0)
    with pytest.raises(RuntimeError):
        api.N3Tree.from_arrays(t.child, t.data.astype(np.float32), t.offset, t.invradius3, "SH4",
                               upload=False)  # "data must be stored in half precision"


def test_quantised_tree_decodes_to_the_same_data(tmp_path):
    """compress_octree.py layout (scripts/compress_octree.py:106-119) with --retain 1:
    decode per src/n3tree.cpp:279-340 must reproduce the data array."""
    t = synth.make_tree(depth=3, basis_dim=4, seed=4)
    p = str(tmp_path / "q.npz")
    common.write_quantised_npz(t, p, n_retain=1)
    n = api.N3Tree()
    n.open(p, upload=False)
    assert np.array_equal(n.data_.view(np.uint16), t.data.view(np.uint16))


def test_pose_files_round_trip(tmp_path):
    poses = synth.make_poses(4)
    paths = synth.write_pose_dir(str(tmp_path), poses, 800, 1111.111)
    assert [os.path.basename(p) for p in paths] == ["0000.txt", "0001.txt", "0002.txt", "0003.txt"]
    m = np.loadtxt(paths[2])
    assert m.shape == (4, 4) and np.allclose(m, poses[2])
    K = np.loadtxt(str(tmp_path / "intrinsics.txt"))
    assert K[0, 0] == pytest.approx(1111.111) and K[1, 1] == pytest.approx(1111.111)
    tr = synth.c2w_to_transform(m)
    # column-major 4x3: right, up, back, centre (main_headless.cpp:51-56)
    assert np.allclose(tr[9:12], m[:3, 3]) and np.allclose(tr[0:3], m[:3, 0])
    cam = api.Camera(800, 800, 1111.111)
    cam.set_c2w(m)
    assert np.array_equal(cam.transform, tr)
    # camera looks down -z of the c2w and sits on the r=4 orbit
    assert np.linalg.norm(m[:3, 3]) == pytest.approx(4.0)


def test_render_options_defaults_match_reference():
    o = api.RenderOptions()
    c = o.to_c()
    assert (c.step_size, c.sigma_thresh, c.stop_thresh) == pytest.approx((1e-4, 1e-2, 1e-2))
    assert list(c.basis_minmax) == [0, 24] and list(c.render_bbox) == [0, 0, 0, 1, 1, 1]
    assert c.render_depth == 0 and c.enable_probe == 0 and c.probe_disp_size == 100


@pytest.mark.parametrize("w,h,tw,th,world", [(800, 800, 800, 8, 8), (100, 61, 32, 16, 3),
                                             (64, 64, 8, 8, 4), (1920, 1080, 1920, 8, 8),
                                             (50, 50, 0, 0, 1)])
def test_tile_shard_round_trip(w, h, tw, th, world):
    rng = np.random.default_rng(w + h)
    frame = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
    parts = [tiles.frame_to_compact(frame, tw, th, r, world) for r in range(world)]
    n = tiles.compact_pixels(w, h, tw, th, world)
    assert all(p.shape == (n, 4) for p in parts)
    back = tiles.assemble_tiles(np.stack(parts), w, h, tw, th, world)
    assert np.array_equal(back, frame)
    own = tiles.owner_map(w, h, tw, th, world)
    counts = np.bincount(own.reshape(-1), minlength=world)
    assert counts.sum() == w * h and (counts > 0).all()
    # agrees with the C ABI's buffer size
    from volrend_amd import _abi
    assert _abi.lib().vr_compact_bytes(w, h, tw, th, world) == n * 4


def test_bench_reports_traffic_only_for_the_sources_it_was_measured_on(tmp_path):
    """bench.py's roofline.traffic comes from a committed PMC measurement; it must vanish (with the
    reason) the moment the kernel sources no longer hash to what the measurement recorded."""
    import json
    import sys
    sys.path.insert(0, ROOT)
    import bench
    rec = {"config": "C1", "fp_mode": "strict", "frames_per_launch": 64,
           "read_bytes_per_frame": 1.0e9, "write_bytes_per_frame": 5.0e6,
           "kernel_source_sha256": "abc"}
    (tmp_path / "r07_traffic_C1.json").write_text(json.dumps(rec))
    (tmp_path / "r03_traffic_C1.json").write_text(json.dumps(dict(rec, kernel_source_sha256="old",
                                                                    read_bytes_per_frame=2.0e9)))
    got, why, fpl = bench.committed_traffic("C1", "strict", str(tmp_path), have_hash="abc")
    assert got == 1.005e9 and "r07_traffic_C1.json" in why and "verified" in why
    assert fpl == 64  # the launch shape it was profiled at: another shape is flagged as extrapolated
    got, why, _ = bench.committed_traffic("C1", "strict", str(tmp_path), have_hash="old")
    assert got == 2.005e9 and "r03_traffic_C1.json" in why      # an older file that still matches
    got, why, fpl = bench.committed_traffic("C1", "strict", str(tmp_path), have_hash="new")
    assert got is None and "STALE" in why and fpl is None
    got, why, _ = bench.committed_traffic("C1", "fma", str(tmp_path), have_hash="abc")
    assert got is None
    got, why, _ = bench.committed_traffic("C9", "strict", str(tmp_path), have_hash="abc")
    assert got is None and "no profiles" in why
    # a measurement at the run's own launch size wins over one at another size
    (tmp_path / "r07_traffic_C1_20.json").write_text(json.dumps(dict(rec, frames_per_launch=20,
                                                                       read_bytes_per_frame=1.1e9)))
    got, why, fpl = bench.committed_traffic("C1", "strict", str(tmp_path), have_hash="abc", want_fpl=20)
    assert fpl == 20 and got == 1.105e9 and "r07_traffic_C1_20.json" in why
    got, why, fpl = bench.committed_traffic("C1", "strict", str(tmp_path), have_hash="abc", want_fpl=64)
    assert fpl == 64 and got == 1.005e9
    got, why, fpl = bench.committed_traffic("C1", "strict", str(tmp_path), have_hash="abc", want_fpl=7)
    assert fpl in (20, 64)  # no profile at 7 frames per launch: the caller marks it extrapolated
    # the real hash covers the kernel sources and the build flags, and is stable
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from measure_traffic import kernel_source_hash
    assert kernel_source_hash() == kernel_source_hash() and len(kernel_source_hash()) == 64


def test_measure_traffic_parses_rocprofv3_csvs(tmp_path):
    """tools/measure_traffic.collect(): counter values are summed over the XCD instances of a
    dispatch and averaged over the dispatches of the kernel flavour asked for; durations come
    from the kernel trace of the same pass."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import measure_traffic as mt
    k = "void vr::(anonymous namespace)::render_kernel<0, 16, 0>(vr::KParams)"
    other = "void vr::(anonymous namespace)::render_kernel<0, 16, 1>(vr::KParams)"
    d = tmp_path / "box" / "123"
    d.mkdir(parents=True)
    rows = ["Dispatch_Id,Kernel_Name,Counter_Name,Counter_Value"]
    for disp, base in ((7, 100.0), (9, 300.0)):
        for xcd in range(8):                       # one row per XCD instance
            rows.append(f'{disp},"{k}",TCC_EA0_RDREQ_128B_sum,{base}')
            rows.append(f'{disp},"{k}",TCC_EA0_RDREQ_sum,{base}')
    rows.append(f'11,"{other}",TCC_EA0_RDREQ_128B_sum,99999')   # instrumented flavour: ignored
    (d / "rdsize_counter_collection.csv").write_text("\n".join(rows) + "\n")
    (d / "rdsize_kernel_trace.csv").write_text(
        "Kernel_Name,Start_Timestamp,End_Timestamp\n"
        f'"{k}",1000,3000\n"{k}",5000,9000\n"{other}",0,100000\n')
    vals, dur = mt.collect(str(tmp_path), "rdsize", "render_kernel<0, 16, 0>")
    assert sorted(vals["TCC_EA0_RDREQ_128B_sum"]) == [800.0, 2400.0]
    assert sorted(dur) == [2000.0, 4000.0]
    # the group table never offers the counters that abort rocprofv3 on this pool
    assert not any(c.startswith(("TA_", "TD_")) for g in mt.GROUPS.values() for c in g.split())


def test_drain_simulation_conserves_rays():
    """tools/drain_sim.py (the model behind DESIGN.md section 7's bound on re-grouping the rays of
    the drain phase): every policy marches every sample of every ray, a free exchange is never
    slower than no exchange, and phases hand over exactly the rays they did not finish."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import drain_sim as ds
    rng = np.random.default_rng(3)
    rays = np.minimum(rng.geometric(1.0 / 30.0, size=64 * 300), 400).astype(np.int32)
    base, phases = ds.simulate(rays, "none", n_waves=128)
    ideal, _ = ds.simulate(rays, "ideal", n_waves=128)
    assert phases == 1 and ideal <= base
    # a tick costs at least A clocks and no launch can end before its longest ray
    assert base >= ds.A * int(rays.max())
    clocks, left = ds.run_phase(rays, 128, "phased", T=16, R=10 ** 9)
    assert clocks > 0 and 0 < len(left) < len(rays) and left.min() >= 1
    total, n_phases = ds.simulate(rays, "phased", T=16, n_waves=128)
    assert n_phases >= 2 and total > clocks


def test_bench_self_launch_command_and_relay(tmp_path, monkeypatch):
    """`python bench.py --gpus N` without a launcher re-executes itself under
    torch.distributed.run (the driver's documented form): argv, port, environment -- and the
    relay of rank 0's one JSON line + the exit code (the launcher itself is faked here)."""
    import json as _json
    import subprocess
    import sys as _sys
    import bench
    cmd = bench.self_launch_command(["--gpus", "4", "--steps", "20", "--warmup", "5"], 4, port=29999)
    assert cmd[:3] == [_sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29999"
    i = cmd.index(os.path.abspath(bench.__file__))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "20", "--warmup", "5"]
    # a free port is picked when none is given
    p = int(bench.self_launch_command([], 2)[bench.self_launch_command([], 2).index("--master-port") + 1])
    assert 1024 < p < 65536

    seen = {}

    def fake_run(cmd, env=None, stdout=None, **kw):
        seen["cmd"], seen["env"] = cmd, env

        class R:
            returncode = 3
            stdout = b'NCCL version banner\n{"metric": "x", "n_gpus": 2}\n'
        return R()

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setenv("RANK", "7")  # stale launcher variables must not reach the new launcher
    out = tmp_path / "out.txt"
    with open(out, "w") as f:
        monkeypatch.setattr(_sys, "stdout", f)
        rc = bench.self_launch(["--gpus", "2"], 2)
    assert rc == 3
    assert "RANK" not in seen["env"] and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert _json.loads(out.read_text().strip()) == {"metric": "x", "n_gpus": 2}


def test_issue_model_is_hash_checked_and_class_weighted(tmp_path):
    """roofline.model.t_issue prices vector instructions by class (2.5 / 4 / 7.6 cycles) from a static
    histogram of the kernel's ISA (tools/isa_issue_model.py) -- but only while the histogram was made from
    the sources the run was built from; and the classifier itself on a few instructions."""
    import json as _json
    import sys as _sys
    import bench
    _sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_issue_model as im
    assert im.classify("\tv_fma_f32 v1, v2, v3, v4") == "cheap"
    assert im.classify("\tv_mul_f32_e32 v1, 0x4b800000, v2") == "cheap"        # literal operand
    assert im.classify("\tv_add_f32_e32 v1, s4, v2") == "full"                # SGPR operand
    assert im.classify("\tv_med3_f32 v9, v2, 0, v54") == "full"
    assert im.classify("\tv_lshl_or_b32 v10, v26, s34, v10") == "full"
    assert im.classify("\tv_fma_mix_f32 v1, v2, v3, v4 op_sel_hi:[0,1,0]") == "full"
    assert im.classify("\tv_rcp_f32_e32 v1, v2") == "trans"
    assert im.classify("\ts_and_b64 s[0:1], s[2:3], vcc") is None and im.classify("\tds_read_b32 v1, v2") is None
    h = im.histogram(["\tv_fma_f32 v1, v2, v3, v4", "\tv_cmp_lt_f32_e32 vcc, v1, v2", "\tv_exp_f32_e32 v1, v2",
                      "\ts_mov_b32 s1, 0", "\tglobal_load_dword v1, v2, s[0:1]"])
    assert (h["cheap"], h["full"], h["trans"], h["salu"], h["vmem"], h["valu"]) == (1, 1, 1, 1, 1, 3)
    assert h["cycles_per_valu"] == round((2.5 + 4.0 + 7.6) / 3, 4)
    rec = {"kernel_source_sha256": "abc", "kernels": {"strict/SH16/xmajor": {
        "march_round": {"valu": 113, "cycles_per_valu": 3.5}, "rest_of_kernel": {"valu": 1300, "cycles_per_valu": 3.6}}}}
    (tmp_path / "r09_isa_issue_model.json").write_text(_json.dumps(rec))
    got = bench.issue_model("strict", 16, False, have_hash="abc", profiles_dir=str(tmp_path))
    assert got["march_valu"] == 113 and got["march_cpv"] == 3.5 and got["rest_cpv"] == 3.6
    assert bench.issue_model("strict", 16, False, have_hash="other", profiles_dir=str(tmp_path)) is None  # stale
    assert bench.issue_model("strict", 16, True, have_hash="abc", profiles_dir=str(tmp_path)) is None     # other flavour
