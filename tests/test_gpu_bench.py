"""-m gpu: the driver contract of bench.py (one JSON line on stdout with the agreed keys) and a
rehearsal of its N > 1 path with two ranks sharing the one GPU (gloo, host-staged gather)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")


def test_bench_prints_one_json_line_with_the_contract_keys(gpu):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "C0", "--steps",
                        "16", "--warmup", "8", "--batch", "8"], capture_output=True, text=True,
                       timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
              "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline",
              "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 16 and d["warmup"] == 8
    assert d["unit"] == "Mrays/s" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["value"] > 0 and d["ms_per_step"] > 0 and "workload" in d["config"]
    # what the launches really carried, and frames of the timed region against the oracle
    assert d["config"]["frames_per_launch"] == 8 and d["config"]["launches"] == 2
    par = d["parity"]
    assert par["frames_checked"] >= 1 and par["rgba8_equal"] is True and par["psnr_db"] == "inf"
    assert par["max_abs_diff_rgb8"] == 0
    # the line's own noise floor and the scheduler's lane utilisation travel with it
    rep = d["repeats"]
    assert len(rep["ms_per_step"]) == 5 and rep["min"] <= rep["median"] <= rep["max"]
    assert 0 < d["sched"]["march_util"] <= 1 and 0 < d["sched"]["shade_util"] <= 1
    assert "untimed frames" in d["config"]["preroll"]
    roof = d["roofline"]
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert roof["frac"] == pytest.approx(roof["achieved"] / roof["peak"], rel=1e-3)
    # B_unique next to the algorithmic bytes (distinct 128-byte lines)
    assert 0 < roof["unique_bytes_per_frame"] and 0 < roof["unique_bytes_per_launch"]
    assert roof["unique_bytes_per_launch"] <= roof["unique_bytes_per_frame"] * roof["unique_launch_frames"] * 1.5
    assert set(roof["unique_lines_per_frame_by_array"]) == {"leaves", "nodes", "top", "bricks"}
    # no committed traffic measurement for C0: null, with the reason
    assert roof["traffic"] is None and roof["traffic_source"]
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0
    assert cb["sample"] and (cb["kind"] == "port" or cb["port"]["value"] > 0)
    # the reference's GLSL backend on a software rasteriser, same box (or the reason it could not run)
    assert "rtfrag" in cb and ("error" in cb["rtfrag"] or (
        cb["rtfrag"]["value"] > 0 and cb["rtfrag"]["cores"] >= 1 and cb["rtfrag"]["rasteriser"]))


@pytest.mark.parametrize("mode", ["tile", "replicas"])
def test_bench_multirank_rehearsal(gpu, mode):
    env = dict(os.environ, VOLREND_BENCH_SHARE_GPU="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                        "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
                        "29541", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "C0",
                        "--steps", "16", "--warmup", "8", "--batch", "4", "--no-cpu-baseline",
                        "--mode", mode], capture_output=True, text=True, timeout=600, cwd=ROOT,
                       env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout  # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and "REHEARSAL" in d["data"]
    if mode == "tile":
        assert "REHEARSAL" in d["rccl"]["gather"]
        assert d["scaling"] == "strong"
        assert d["config"]["sharded_frame_matches_single_gpu"] is True
    else:
        assert d["scaling"] == "weak" and "replicas" in d["config"]["parallelism"]


def test_bench_one_rank_through_the_product_collective(gpu):
    """VOLREND_FORCE_GATHER=1: the tile-shard path with ONE rank -- the frames travel through
    libvolrend_gather.so (vr_gather_tiles: a grouped self ncclSend / ncclRecv, the collective
    volrend_headless --gpus N ships) and the batch de-interleave, and must still equal the oracle."""
    env = dict(os.environ, VOLREND_FORCE_GATHER="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "C0", "--steps", "16",
                        "--warmup", "8", "--batch", "4", "--no-cpu-baseline"], capture_output=True, text=True,
                       timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.strip().startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["parity"]["rgba8_equal"] is True
    assert d["config"]["sharded_frame_matches_single_gpu"] is True
    assert d["rccl"]["world_size"] == 1 and d["rccl"]["gather"].startswith("vr_gather_tiles"), d["rccl"]


@pytest.mark.parametrize("mode", ["tile", "replicas"])
def test_bench_two_gpus_over_rccl(gpu, mode):
    """The real N > 1 path: one process per GPU, RCCL between two DEVICES (kernel-backed, no
    oracle, no shared-GPU rehearsal).  Runs by itself wherever two GPUs are visible; every box
    this build has seen so far has one, and then the test says so instead of passing silently."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip(f"{torch.cuda.device_count()} GPU visible: the two-device RCCL gather cannot run "
                    "here (covered by the shared-GPU rehearsal above and tests/test_dist_gloo.py)")
    env = {k: v for k, v in os.environ.items() if k != "VOLREND_BENCH_SHARE_GPU"}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                        "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
                        "29543", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "C0",
                        "--steps", "16", "--warmup", "8", "--batch", "4", "--no-cpu-baseline",
                        "--mode", mode], capture_output=True, text=True, timeout=900, cwd=ROOT,
                       env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and "REHEARSAL" not in d["data"]
    assert d["rccl"]["world_size"] == 2 and d["rccl"]["backend"] == "nccl" and d["rccl"]["nccl_version"]
    if mode == "tile":
        assert d["config"]["sharded_frame_matches_single_gpu"] is True
        assert d["parity"]["rgba8_equal"] is True  # the GATHERED frames of the timed region vs the oracle
        assert d["rccl"]["gather"].startswith("vr_gather_tiles"), d["rccl"]  # the product's collective, not a fallback


def test_bench_plain_form_launches_itself(gpu):
    """`python bench.py --gpus 2` WITHOUT a launcher (how a driver may well call it): bench.py
    re-executes itself under torch.distributed.run and relays rank 0's one line.  Shared-GPU
    rehearsal on a one-GPU box; real RCCL between two devices where two are visible."""
    import torch
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    two = torch.cuda.device_count() >= 2
    if not two:
        env["VOLREND_BENCH_SHARE_GPU"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "C0",
                        "--steps", "16", "--warmup", "8", "--batch", "4", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 16
    assert d["config"]["sharded_frame_matches_single_gpu"] is True
    assert ("REHEARSAL" in d["data"]) == (not two)
    assert d["rccl"]["world_size"] == 2
