#!/usr/bin/env python3
"""bench.py -- Mrays/s + FPS of the PlenOctree ray-march hot path on MI355X.

    python bench.py --gpus 1 --steps 200 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one 800x800 frame of BASELINE config C1 (synthetic "lego-like"
depth-9 SH16 tree, ~2 M nodes / ~1.6 GB, pose i of the 200-pose orbit), rendered
by the gfx950 kernel through the C ABI with the tree resident in HBM.
For N > 1 every frame is sharded by interleaved 8-row screen tiles across the N
GPUs (tree replicated) and the RGBA8 tiles are gathered to rank 0 with one RCCL
gather per frame, pipelined one frame deep ("strong" scaling: total work fixed).

Prints ONE JSON line on rank 0 (metric = Mrays/s; FPS rides along), including
  roofline     -- algorithmic bytes per launch (instrumented kernel flavour,
                  counted once per pose OUTSIDE the timed region) / mean kernel
                  duration measured with HIP events on the launch stream
  cpu_baseline -- the CPU oracle (kind "port") timed on a bounded sample of the
                  same frame on the host cores (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
# The kernel's own lower bounds (roofline.model), each from a measurement on this chip:
GATHER_LINES_PER_S = 53.5e9   # random 128-byte line requests the chip retires per second whatever the
                              # record size (tools/gather_bench.hip, profiles/r01_gather_ceiling.txt)
N_SIMD, SIMD_HZ = 1024, 2.4e9  # 256 CUs x 4 SIMDs, peak shader clock (MI355X_MICROARCH.md)
VALU_ISSUE_CYCLES = 4.0       # flat price of a wave64 vector instruction (what SQ_ACTIVE_INST_VALU counts); the chip's
                              # own numbers are 2.5 / 4 / 7.6 cycles by class (profiles/r04_valu_issue_cost.jsonl):
                              # roofline.model.t_issue_ms uses the class histogram of the kernel's ISA
                              # (tools/isa_issue_model.py), t_issue_flat4_ms this flat price
ROUND_FLOOR_US = 0.62         # a dependent march round of a lone wave on an otherwise idle SIMD: 1300-1500
                              # shader clocks (profiles/r04_tail_profile.jsonl, the last buckets of a one-frame launch)
CACHE_DIR = os.environ.get("VOLREND_BENCH_CACHE", "/dev/shm/volrend_amd_cache")


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def load_or_make_tree(synth, name: str, local_rank: int, barrier, seed=None):
    """Rank-local 0 generates the seeded tree once per node; the others mmap it.
    With ``seed`` every rank builds (and caches) its own tree (BASELINE config 4)."""
    os.makedirs(CACHE_DIR, exist_ok=True)
    cfg = dict(synth.CONFIGS[name])
    if seed is not None:
        cfg["seed"] = seed
        local_rank = 0
    tag = f"{name}_s{cfg['seed']}_d{cfg['depth']}_b{cfg['basis_dim']}_sh{cfg['shell_leaves']}"
    fchild = os.path.join(CACHE_DIR, tag + "_child.npy")
    fdata = os.path.join(CACHE_DIR, tag + "_data.npy")
    fdone = os.path.join(CACHE_DIR, tag + ".done")
    if local_rank == 0 and not os.path.exists(fdone):
        t0 = time.time()
        tree = synth.make_config_tree(name, seed=cfg["seed"])
        np.save(fchild, tree.child)
        np.save(fdata, tree.data)
        open(fdone, "w").write("ok")
        log(f"[bench] generated {name}: {tree.capacity} nodes, {tree.nbytes() / 1e9:.2f} GB "
            f"in {time.time() - t0:.1f}s")
    barrier()
    child = np.load(fchild, mmap_mode="r")
    data = np.load(fdata, mmap_mode="r")
    fmt = "RGBA" if cfg["fmt"] == "RGBA" else f"{cfg['fmt']}{cfg['basis_dim']}"
    return synth.SynthTree(child, data, np.full(3, 0.5, np.float32),
                           np.full(3, np.float32(0.5 / 1.5), np.float32), fmt, None, cfg["depth"],
                           dict(config=name))


def _time_cpu(render, label, kind, cores, transforms, width, height, focal, budget_s):
    """Bounded sample of the bench workload on the host: whole frames of the pose orbit, one after
    the other, until ~budget_s of wall time is spent (at least one frame; only a 64-row centre
    band of pose 0 if a whole frame would not fit the budget)."""
    y0 = max(0, (height // 2 - 32) // 8 * 8)
    rows = min(64, height - y0)
    t0 = time.perf_counter()
    render(transforms[0], (0, y0, width, rows))
    t_band = time.perf_counter() - t0
    if t_band * height / rows > 2.5 * budget_s:  # slow host: the band is the sample
        return {"value": round(width * rows / t_band / 1e6, 4), "unit": "Mrays/s", "cores": cores,
                "kind": kind,
                "sample": f"{rows}-row centre band of pose 0 ({width * rows} rays), "
                          f"{t_band:.1f}s, {label}, {cores} threads"}
    rays, t_total, n = 0, 0.0, 0
    while t_total < budget_s and n < len(transforms):
        t0 = time.perf_counter()
        render(transforms[n], None)
        t_total += time.perf_counter() - t0
        rays += width * height
        n += 1
    return {"value": round(rays / t_total / 1e6, 4), "unit": "Mrays/s", "cores": cores, "kind": kind,
            "sample": f"{n} full {width}x{height} frames (poses 0..{n - 1}, {rays} rays), "
                      f"{t_total:.1f}s, {label}, {cores} threads"}


def cpu_baseline(tree, transforms, width, height, focal, budget_s=8.0):
    """The same frames on the host cores.  kind "reference": the reference's own render_kernel
    (src/cuda/volrend.cu + rt_core.cuh compiled for the host, oracle/_ref -- present when
    build() ran where the reference is mounted); kind "port": the oracle's C restatement, always
    timed and reported next to it."""
    from oracle import binding as ob
    th = ob.TreeHandle(tree)
    opt = ob.default_options()
    cores = os.cpu_count() or 1

    def port(tr, region):
        ob.render(th, ob.make_camera(tr, width, height, focal), opt, ob.FP_STRICT, region=region,
                  want_accum=False, nthreads=cores)

    out = _time_cpu(port, "oracle strict mode", "port", cores, transforms, width, height, focal,
                    budget_s)
    if ob.ref_lib(False) is not None:
        def ref(tr, region):
            ob.ref_render(th, ob.make_camera(tr, width, height, focal), opt, region=region,
                          nthreads=cores)

        r = _time_cpu(ref, "reference render_kernel compiled for the host (g++ -O2)", "reference",
                      cores, transforms, width, height, focal, budget_s)
        r["port"] = {"value": out["value"], "sample": out["sample"]}
        out = r
    return out


def committed_traffic(config: str, fp: str, profiles_dir: str | None = None, have_hash: str | None = None,
                      want_fpl: int | None = None):
    """(L2<->fabric bytes per frame, provenance, frames per launch it was profiled at) from the
    newest profiles/r*_traffic_<config>[_<frames>].json whose recorded kernel-source hash equals
    the sources this run was built from -- a measurement taken at THIS run's launch size
    (``want_fpl``) first, else one at another size (the caller marks it extrapolated);
    (None, reason, None) when there is no measurement or it is stale."""
    import glob
    if have_hash is None:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from measure_traffic import kernel_source_hash
        have_hash = kernel_source_hash()
    profiles_dir = profiles_dir or os.path.join(ROOT, "profiles")
    reason = f"no profiles/r*_traffic_{config}.json"
    found = []
    for tpath in sorted(glob.glob(os.path.join(profiles_dir, f"r*_traffic_{config}*.json")), reverse=True):
        with open(tpath) as f:
            tj = json.load(f)
        rel = os.path.join("profiles", os.path.basename(tpath))
        if tj.get("config") != config or tj.get("fp_mode") != fp or "read_bytes_per_frame" not in tj:
            continue
        if tj.get("kernel_source_sha256") != have_hash:
            reason = f"{rel} is STALE (kernel sources changed since it was measured)"
            continue
        found.append((tj, rel))
    found.sort(key=lambda x: 0 if want_fpl is not None and int(x[0]["frames_per_launch"]) == want_fpl else 1)
    for tj, rel in found[:1]:
        committed_traffic.last_profile = tj  # (the same file's instruction counts: see main)
        return (tj["read_bytes_per_frame"] + tj.get("write_bytes_per_frame", 0.0),
                f"{rel}: rocprofv3 --pmc passes at {tj['frames_per_launch']} frames per launch, "
                f"TCC_EA0_RDREQ_128B*128 + _64B*64 + _32B*32 + WRITE_SIZE, kernel source hash verified",
                int(tj["frames_per_launch"]))
    return None, reason, None


committed_traffic.last_profile = None


def issue_model(fp: str, basis_dim: int, blocked: bool, have_hash: str | None = None, profiles_dir: str | None = None):
    """Vector-ALU cycles per instruction of the production flavour, by code region, from the newest
    profiles/r*_isa_issue_model.json (tools/isa_issue_model.py: static class histogram of the ISA x the
    measured per-class costs) whose kernel-source hash equals the sources this run was built from."""
    import glob
    if have_hash is None:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from measure_traffic import kernel_source_hash
        have_hash = kernel_source_hash()
    for path in sorted(glob.glob(os.path.join(profiles_dir or os.path.join(ROOT, "profiles"),
                                              "r*_isa_issue_model.json")), reverse=True):
        with open(path) as f:
            d = json.load(f)
        k = d.get("kernels", {}).get(f"{fp}/SH{basis_dim}/{'blocked' if blocked else 'xmajor'}")
        if d.get("kernel_source_sha256") == have_hash and k:
            return {"march_valu": k["march_round"]["valu"], "march_cpv": k["march_round"]["cycles_per_valu"],
                    "rest_cpv": k["rest_of_kernel"]["cycles_per_valu"],
                    "source": os.path.join("profiles", os.path.basename(path))}
    return None


def live_traffic(config: str, fp: str, frames_per_launch: int, tune: str):
    """L2<->fabric bytes per frame of the render kernel at THIS launch shape, measured now: two
    rocprofv3 --pmc passes (read-request sizes, WRITE_SIZE) of a child bench process
    (tools/measure_traffic.py).  None where rocprofv3 is missing; {"error": ...} when it fails."""
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None
    out = tempfile.NamedTemporaryFile(suffix=".json", delete=False).name
    cmd = [sys.executable, os.path.join(ROOT, "tools", "measure_traffic.py"), "--config", config,
           "--fp", fp, "--batch", str(frames_per_launch), "--groups", "rdsize", "write", "--out", out,
           "--timeout", "45"]  # (a pass takes ~5 s; a profiler that hangs must not hold the bench up)
    if tune:
        cmd += ["--bench-args", f"--tune {tune}"]
    try:
        p = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, timeout=100)
        with open(out) as f:
            d = json.load(f)
        if "read_bytes_per_frame" not in d:
            return {"error": "; ".join(d.get("failed_groups") or []) or p.stderr.decode(errors="replace")[-300:]}
        return d
    except Exception as e:  # noqa: BLE001 -- the committed measurement stays the fallback
        return {"error": repr(e)}
    finally:
        try:
            os.unlink(out)
        except OSError:
            pass


def parity_check(stree, checks, width, height, focal, fp):
    """BASELINE's metric ends in "PSNR vs ref": frames the TIMED region produced (read back after
    it, nothing re-rendered) against the CPU oracle -- the checker, pinned bit for bit to the
    reference's own render_kernel compiled for the host (tests/test_oracle_vs_ref.py).
    ``checks``: [(step index, 12-float pose, HxWx4 uint8 numpy frame)]."""
    from oracle import binding as ob
    th = ob.TreeHandle(stree)
    opt = ob.default_options()
    mode = ob.FP_FMA if fp == "fma" else ob.FP_STRICT
    equal, worst_mse, max_diff, steps = True, 0.0, 0, []
    for step, tr, got in checks:
        want, _, _ = ob.render(th, ob.make_camera(tr, width, height, focal), opt, mode,
                               want_accum=False, nthreads=os.cpu_count() or 1)
        d = want[..., :3].astype(np.int32) - got[..., :3].astype(np.int32)
        equal = equal and bool(np.array_equal(want, got))
        worst_mse = max(worst_mse, float(np.mean(d.astype(np.float64) ** 2)))
        max_diff = max(max_diff, int(np.abs(d).max()))
        steps.append(int(step))
    psnr = None if worst_mse == 0.0 else round(10.0 * np.log10(255.0 ** 2 / worst_mse), 2)
    # the longest dependent chain of the launch (roofline.model.t_chain): samples of the longest ray
    # of the checked frames -- the launch cannot end before that ray has marched them one by one
    longest = 0
    for _, tr, _ in checks:
        samples, _, _ = ob.render_maps(th, ob.make_camera(tr, width, height, focal), opt, mode,
                                       nthreads=os.cpu_count() or 1)
        longest = max(longest, int(samples.max()))
    return {"frames_checked": len(checks), "steps": steps, "rgba8_equal": equal,
            "longest_ray_samples": longest,
            "psnr_db": "inf" if psnr is None else psnr, "max_abs_diff_rgb8": max_diff,
            "against": f"oracle ({fp} mode; == the reference's render_kernel compiled for the host, "
                       "tests/test_oracle_vs_ref.py), frames taken from the timed region"}


def rtfrag_baseline(timeout_s=240):
    """BASELINE config C0 next to the GPU number (north_star): the reference's GLSL backend
    (shaders/rt.frag, its no-CUDA shader_renderer path) on a software GL rasteriser on THIS
    box's host cores -- Mesa llvmpipe when the box has Mesa EGL, else SwiftShader, named in the
    result.  One 400x400 C0 frame after a warm-up frame, in a child process
    (oracle/ref_build/rtfrag_baseline.py; baseline infrastructure, never the product)."""
    import subprocess
    cmd = [sys.executable, os.path.join(ROOT, "oracle", "ref_build", "rtfrag_baseline.py"),
           "--config", "C0", "--frames", "2"]
    try:
        out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout_s,
                             cwd=ROOT)
        if out.returncode != 0:
            return {"error": out.stderr.decode(errors="replace").strip().splitlines()[-1][:200]}
        r = json.loads(out.stdout.decode())
    except Exception as e:  # baseline only: never fail the bench over it
        return {"error": f"{type(e).__name__}: {e}"[:200]}
    return {"value": r["rt_frag_mrays_per_s"], "unit": "Mrays/s", "cores": r["cores"],
            "ms_per_frame": r["rt_frag_ms_per_frame"], "rasteriser": r["rasteriser"],
            "sample": f"one {r['image'][0]}x{r['image'][1]} frame of C0 ({r['nodes']} nodes, SH16), "
                      f"pose {r['pose']}, mean of 2 frames after a warm-up frame",
            "psnr_vs_oracle_db": r["psnr_rt_frag_vs_oracle_db"]}


def self_launch_command(argv, gpus: int, port: int | None = None):
    """The command `python bench.py --gpus N ...` turns itself into when it is started WITHOUT a
    launcher (no WORLD_SIZE in the environment): the documented torch.distributed.run form, one
    process per GPU, same argv, a free local port."""
    if port is None:
        import socket
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]


def self_launch(argv, gpus: int) -> int:
    """Runs self_launch_command and relays rank 0's ONE JSON line (the children's stdout carries
    nothing else; their stderr passes through).  Returns the launcher's exit code."""
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = self_launch_command(argv, gpus)
    log("[bench] no launcher in the environment: " + " ".join(cmd))
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE)
    lines = [l for l in p.stdout.decode(errors="replace").splitlines() if l.startswith("{")]
    if lines:
        sys.stdout.write(lines[-1] + "\n")
        sys.stdout.flush()
    elif p.returncode == 0:
        log("[bench] the ranks ended without a result line")
        return 1
    return p.returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--config", default="C1", choices=["C0", "C1", "C1r", "C1t", "C2", "C3", "X1", "X3"])
    ap.add_argument("--fp", default="strict", choices=["strict", "fma"])
    ap.add_argument("--tile-rows", type=int, default=8, help="rows per interleaved screen tile")
    ap.add_argument("--batch", type=int, default=64,
                    help="poses per launch (vr_render_batch); steps must be a multiple")
    ap.add_argument("--tune", default="", help="k=v,... scheduling knobs (march_max, refill_min, waves_per_cu)")
    ap.add_argument("--mode", default="tile", choices=["tile", "replicas"],
                    help="N > 1: 'tile' = every frame sharded by screen tiles + RCCL gather "
                         "(default, the north-star path); 'replicas' = BASELINE config 4, one "
                         "tree (seed 1010+rank) and one pose stream per GPU, no communication")
    ap.add_argument("--streams", type=int, default=0,
                    help="launch pipelines (0 = auto: 1 on a single GPU, 2 when tile-sharded so "
                         "that one launch's ramp-up / tail overlaps its neighbour)")
    ap.add_argument("--readback", action="store_true",
                    help="also copy every frame to pinned host memory inside the timed region "
                         "(the PCIe-inclusive rate; never the headline value)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true",
                    help="skip the post-run comparison of timed frames with the CPU oracle")
    ap.add_argument("--cpu-budget", type=float, default=12.0)
    ap.add_argument("--preroll", type=int, default=128,
                    help="untimed frames rendered directly in front of the W warm-up steps, so that the "
                         "timed region finds the GPU in the power state of a running render loop "
                         "(after ~5 ms of idling an MI355X needs ~30 ms of work to be back at full "
                         "clocks: profiles/r04_lone_launch_probe.jsonl); 0 = none")
    ap.add_argument("--live-traffic", type=int, default=0,
                    help="0 (default): roofline.traffic is the committed, hash-verified rocprofv3 measurement "
                         "of these kernel sources (profiles/r*_traffic_*.json); 1: measure the L2<->fabric "
                         "traffic of THIS run's launch shape with two rocprofv3 --pmc passes of a child "
                         "process after the timed region (tools/job_final.sh opts in)")
    ap.add_argument("--cold", type=int, default=1,
                    help="after the repeats, time the identical K-step region once more behind 50 ms of "
                         "idling (ms_per_step_cold: a launch out of an idle GPU, the condition rounds 1-3 "
                         "measured -- both conditions in one line); 0 = skip")
    ap.add_argument("--repeats", type=int, default=5,
                    help="after the timed region, the identical K-step region is run this many more "
                         "times (not part of value / ms_per_step): the line's own noise floor")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: become the launcher (one process per GPU), relay the line
        raise SystemExit(self_launch(sys.argv[1:], args.gpus))

    # stdout carries exactly ONE JSON line: everything else that writes to fd 1 (RCCL prints
    # its version banner there) is sent to stderr; the JSON goes to the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    from volrend_amd import _abi, api, synth
    from volrend_amd.dist import GatherPipeline

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and rank == 0:  # the launcher decides; n_gpus in the line is what really ran
        log(f"[bench] --gpus {args.gpus} but the launcher started {world} rank(s) (WORLD_SIZE): "
            f"running with {world}")
    # VOLREND_BENCH_SHARE_GPU=1 (rehearsal only, never a measurement): all ranks use cuda:0 and
    # the collectives go through gloo with host staging -- lets the N > 1 code path (shards,
    # pipeline, assembly, self-check, rank-0 JSON) run on a one-GPU box, where RCCL refuses
    # two ranks on one device.
    share_gpu = os.environ.get("VOLREND_BENCH_SHARE_GPU", "0") == "1"
    dev_index = 0 if share_gpu else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # VOLREND_FORCE_GATHER=1: run the tile-shard + RCCL gather path even with one rank
    force_gather = os.environ.get("VOLREND_FORCE_GATHER", "0") == "1"
    use_dist = world > 1 or force_gather
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # RCCL writes its version banner to stdout; stdout carries the ONE JSON line
        os.environ.setdefault("NCCL_DEBUG_FILE", "/tmp/volrend_bench_rccl.%h.%p.log")
        if share_gpu:
            dist.init_process_group("gloo")
        elif world == 1:
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        else:
            dist.init_process_group("nccl", device_id=dev)

    class _Done:
        def wait(self):
            return True

    class HostStagedDist:
        """gloo stand-in for the rehearsal mode: same gather() signature, host staging."""
        @staticmethod
        def gather(tensor, gather_list, dst=0, async_op=False):
            torch.cuda.synchronize()
            src = tensor.cpu()
            outs = [torch.empty_like(src) for _ in range(world)] if rank == dst else None
            dist.gather(src, outs, dst=dst)
            if rank == dst:
                for g, o in zip(gather_list, outs):
                    g.copy_(o)
            return _Done()

    # The tile gather itself goes through the PRODUCT's collective (libvolrend_gather.so: one grouped
    # ncclSend / ncclRecv to the root per launch -- what volrend_headless --gpus N ships);
    # torch.distributed keeps the rendezvous, the barriers and the reduction of the timings.
    gather_dist, gather_path = dist, None
    if share_gpu:
        gather_dist = HostStagedDist
        gather_path = "REHEARSAL: gloo with host staging (RCCL refuses two ranks on one device)"
    elif use_dist and not (args.mode == "replicas" and world > 1):
        ok, err = 1, ""
        try:
            from volrend_amd import gather as vg
            ids = [vg.unique_id() if rank == 0 else None]
            if world > 1:
                dist.broadcast_object_list(ids, src=0)  # the launcher's rendezvous carries the 128 bytes
            gather_dist = vg.TileGather(ids[0], rank, world, dev_index)
            gather_path = ("vr_gather_tiles (libvolrend_gather.so: grouped ncclSend / ncclRecv to the root, the "
                           f"collective of volrend_headless --gpus N), RCCL {vg.version()}")
        except Exception as e:  # noqa: BLE001 -- a SCALE run must not die here: fall back, and SAY so
            ok, err = 0, repr(e)[:200]
        agree = torch.tensor([ok], dtype=torch.int32, device=dev)
        if world > 1:
            dist.all_reduce(agree, op=dist.ReduceOp.MIN)
        if int(agree.item()) == 0:  # some rank could not join: every rank takes the same detour
            gather_dist = dist
            gather_path = f"torch.distributed.gather (FALLBACK: vr_gather did not initialise on every rank: {err})"
            log(f"[bench r{rank}] {gather_path}")

    def barrier():
        if use_dist:
            dist.barrier()

    cfg = synth.CONFIGS[args.config]
    W, H, focal = cfg["width"], cfg["height"], cfg["focal"]
    replicas = args.mode == "replicas" and world > 1
    stree = load_or_make_tree(synth, args.config, local_rank, barrier,
                              seed=(1010 + rank) if replicas else None)
    t0 = time.time()
    tree = api.N3Tree.from_synth(stree)
    info = tree.info()
    log(f"[bench r{rank}] uploaded {info['capacity']} nodes, {info['device_bytes'] / 1e9:.2f} GB "
        f"in {time.time() - t0:.1f}s, max_depth {info['max_depth']}")

    poses = synth.make_poses(200)
    transforms = [synth.c2w_to_transform(p) for p in poses]
    cam = api.Camera(W, H, focal, focal)
    opts = api.RenderOptions()
    fp_mode = _abi.FP_FMA if args.fp == "fma" else _abi.FP_STRICT
    stream = torch.cuda.current_stream()
    if args.tune:  # the scheduling knobs of THIS tree (results never depend on them)
        tree.set_tuning(**{k: int(v) for k, v in (kv.split("=") for kv in args.tune.split(","))})

    # per-GPU work per launch shrinks with the tile shard: keep it up with more poses per launch
    B = max(1, min(args.batch * (1 if args.mode == "replicas" else world), _abi.MAX_BATCH))
    tile_h = max(8, (args.tile_rows // 8) * 8)
    tile_w = (W + 7) // 8 * 8
    sharded = use_dist and not replicas
    s_rank, s_world = (rank, world) if sharded else (0, 1)
    shard = api.TileShard(tile_w, tile_h, s_rank, s_world, compact=True)
    # double-buffered outputs: launch j writes set j % 2
    frame_sets = [torch.zeros((B, H, W, 4), dtype=torch.uint8, device=dev) for _ in range(2)]
    frames = [[fs[i] for i in range(B)] for fs in frame_sets]
    n_streams = args.streams if args.streams > 0 else (2 if sharded else 1)
    streams = [stream] + [torch.cuda.Stream(device=dev) for _ in range(n_streams - 1)]
    nbytes = api.compact_bytes(W, H, shard) if sharded else 0
    # rank 0 receives into ONE [world, B, nbytes] tensor per set (the gather list are its rows),
    # so a single launch de-interleaves a whole batch
    gather_sets = []

    def make_gather_list():
        g = torch.zeros((world, B, max(nbytes, 1)), dtype=torch.uint8, device=dev)
        gather_sets.append(g)
        return [g[r] for r in range(world)]

    # replicas: every rank is its own root and nothing is gathered
    pipe = GatherPipeline(
        gather_dist, rank if sharded else 0, world if sharded else 1,
        lambda: torch.zeros((B, max(nbytes, 1)), dtype=torch.uint8, device=dev),
        make_gather_list, force_collective=force_gather,
        stream_ctx=lambda j: torch.cuda.stream(streams[j % n_streams]))

    def pose_of(step):
        return transforms[step % len(transforms)]

    timing = {"events": None}

    prepared = {}

    def prepare(first, n_steps):
        """Marshals the launches of run(n_steps, first) ahead of time (the poses are known up
        front, as in volrend_headless): inside the timed region a launch is one C call."""
        j, done = 0, 0
        while done < n_steps:
            n = min(B, n_steps - done)
            tr = [pose_of(first + done + i) for i in range(n)]
            if not sharded:
                imgs, sh = frames[j % 2][:n], None
            else:
                imgs, sh = [pipe.buffer(j)[i] for i in range(n)], shard
            prepared[(j, first + done, n)] = api.PreparedBatch(tree, cam, tr, opts, imgs, True,
                                                               shard=sh, fp_mode=fp_mode)
            done += n
            j += 1

    def render(j, first_step, n, buf):
        """Launch j renders steps [first_step, first_step+n) in one batch."""
        ev = timing["events"][j] if timing["events"] else None
        stream = torch.cuda.current_stream()
        if ev is not None:
            ev[0].record(stream)
        pb = prepared.get((j, first_step, n))
        if pb is None:
            tr = [pose_of(first_step + i) for i in range(n)]
            imgs, sh = (frames[j % 2][:n], None) if not sharded else ([buf[i] for i in range(n)], shard)
            pb = api.PreparedBatch(tree, cam, tr, opts, imgs, True, shard=sh, fp_mode=fp_mode)
        pb.launch(stream)
        if ev is not None:
            ev[1].record(stream)

    host_sets = ([torch.empty((B, H, W, 4), dtype=torch.uint8).pin_memory() for _ in range(2)]
                 if args.readback else None)

    def assemble(j, glist, n):
        if not sharded:
            if host_sets is not None:  # D2H of the batch, async on the render stream
                host_sets[j % 2][:n].copy_(frame_sets[j % 2][:n], non_blocking=True)
            return  # frames were rendered in place
        api.assemble_tiles_batch(frame_sets[j % 2], gather_sets[j % 2], n, W, H, shard,
                                 torch.cuda.current_stream())

    def run(n_steps, first, events=None):
        timing["events"] = events
        return pipe.run(n_steps, B, render, assemble, first)

    # ---- slot sizing (untimed) -------------------------------------------------
    # one full-size launch per launch stream, so that every launch slot the timed region will use
    # owns its ray buffer (the library grows it on first use -- a blocking multi-GB hipMalloc that
    # must not land inside the timed region)
    run(n_streams * B, 0)
    torch.cuda.synchronize()

    # ---- algorithmic bytes: instrumented flavour, outside the timed region -----
    K = args.steps
    n_distinct = min(K, len(transforms))
    counters = torch.zeros((B, 7), dtype=torch.int64, device=dev)
    scratch = [torch.zeros((H, W, 4), dtype=torch.uint8, device=dev) for _ in range(B)]
    full = api.TileShard(tile_w, tile_h, s_rank, s_world, compact=False)
    # B_unique: distinct 128-byte lines of the tree arrays that the accesses of one
    # launch / one frame touch -- the compulsory-traffic lower bound next to the algorithmic bytes
    # (distinct-line bitmaps of the instrumented flavour, vr_touch_enable / vr_touch_count)
    tree.touch_enable(True)
    unique_launch = None
    jd = 0
    while jd < n_distinct:  # same batching as the timed region
        n = min(B, n_distinct - jd)
        tr = [pose_of(args.warmup + jd + i) for i in range(n)]
        api.launch_renderer_batch(tree, cam, tr, opts, scratch[:n], stream, True, shard=full,
                                  fp_mode=fp_mode, counters=[counters[i] for i in range(n)])
        if jd == 0:
            unique_launch = (n, tree.touch_count(reset=True))
        jd += n
    tree.touch_count(reset=True)
    torch.cuda.synchronize()
    sched = tree.sched_stats()  # of the launches above only: the timed region's poses and batching
    unique_frames = []
    side = torch.zeros((1, 7), dtype=torch.int64, device=dev)
    for i in range(4):  # four single frames spread over the timed poses
        tr = [pose_of(args.warmup + (i * n_distinct) // 4)]
        api.launch_renderer_batch(tree, cam, tr, opts, scratch[:1], stream, True, shard=full,
                                  fp_mode=fp_mode, counters=[side[0]])
        unique_frames.append(tree.touch_count(reset=True))
    tree.touch_enable(False)
    del side
    torch.cuda.synchronize()
    cnt = dict(zip(_abi.COUNTER_FIELDS, [int(v) for v in counters.sum(dim=0).cpu().tolist()]))
    alg_bytes_per_frame = cnt["alg_bytes"] / n_distinct  # this rank's share of a frame
    tree.sched_stats()  # (drop the four single-frame launches)
    log(f"[bench r{rank}] sched per frame: " + ", ".join(
        f"{k}={v / n_distinct:.0f}" for k, v in sched.items()) +
        f"; march util {sched['march_lanes'] / max(64 * sched['march_rounds'], 1):.2f}"
        f", shade util {sched['shade_lanes'] / max(64 * sched['shade_rounds'], 1):.2f}")
    del scratch

    # ---- timed region ----------------------------------------------------------
    n_launch = (K + B - 1) // B
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
              for _ in range(n_launch)]
    prepare(args.warmup, K)
    # ---- W untimed warm-up steps, directly in front of the timed ones (everything that is
    # neither warm-up nor timed -- counters, B_unique -- has run before: the chip enters the timed
    # region the way a render loop in progress would find it)
    # Power state: the host-side work above (counters, marshalling) leaves the GPU idle for tens of
    # milliseconds, and an MI355X that has idled for >= 5 ms runs its next ~30 ms of work at reduced
    # clocks -- a lone 20-frame launch takes 5.55 ms instead of 5.04 (tools/lone_launch_probe.py).
    # A render loop in progress never sees that state, so the warm-up is preceded by `preroll`
    # untimed frames of other poses, enqueued back to back with it (disclosed in config.preroll).
    if args.preroll > 0:
        run(args.preroll, 100)
    if args.warmup > 0:
        run(args.warmup, 0)
    barrier()
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    run(K, args.warmup, events)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t_start

    # frames of the LAST timed launch stay in their buffer set: first and last of them are
    # compared with the oracle after the clock has stopped
    parity_frames = []
    if rank == 0 and not args.no_parity and not replicas:
        j_last = n_launch - 1
        first_last = args.warmup + j_last * B
        n_last = K - j_last * B
        for i in sorted({0, n_last - 1}):
            parity_frames.append((first_last + i, pose_of(first_last + i),
                                  frame_sets[j_last % 2][i].cpu().numpy()))

    # ---- the line's own noise floor: the identical K-step region, R more times (same poses, same
    # prepared launches, same barriers; none of it enters value / ms_per_step) ----
    repeat_ms = []
    for _ in range(max(args.repeats, 0)):
        barrier()
        torch.cuda.synchronize()
        t_rep = time.perf_counter()
        run(K, args.warmup)
        torch.cuda.synchronize()
        barrier()
        dt = time.perf_counter() - t_rep
        if use_dist:
            tr_ = torch.tensor([dt], dtype=torch.float64, device="cpu" if share_gpu else dev)
            dist.all_reduce(tr_, op=dist.ReduceOp.MAX)
            dt = float(tr_.item())
        repeat_ms.append(dt / K * 1e3)

    # ---- the same region out of an idle GPU: 50 ms without work, then the K steps (no pre-roll, no
    # warm-up in front: what a lone launch costs a caller that renders now and then) ----
    cold_ms = None
    if args.cold:
        barrier()
        torch.cuda.synchronize()
        time.sleep(0.05)
        t_cold = time.perf_counter()
        run(K, args.warmup)
        torch.cuda.synchronize()
        barrier()
        dt = time.perf_counter() - t_cold
        if use_dist:
            tr_ = torch.tensor([dt], dtype=torch.float64, device="cpu" if share_gpu else dev)
            dist.all_reduce(tr_, op=dist.ReduceOp.MAX)
            dt = float(tr_.item())
        cold_ms = dt / K * 1e3

    # every launch of the timed region has run: rays cut by the sample guard would mean wrong frames
    # behind a good-looking number (vr_render* only enqueues and cannot say so itself)
    st = tree.status()
    if st != 0:
        raise SystemExit(f"[bench r{rank}] render status 0x{st:x}: rays hit the sample guard -- the frames "
                         f"of this run are wrong, no result line")

    if os.environ.get("VR_TIMELINE"):  # profiling build (-DVR_TIMELINE=1): per-phase cycle sums
        tl = tree.sched_stats()
        tot = max(sum(list(tl.values())[:5]), 1)
        log("[bench] timeline (shader-clock cycles summed over waves): " + ", ".join(
            f"{n}={v / tot:.3f}" for n, v in zip(
                ("refill", "march", "shade_load", "shade_math", "shade_acc"), list(tl.values())[:5]))
            + f"; wave-lifetime cycles/wave {list(tl.values())[5] / max(list(tl.values())[6], 1):.0f}"
            + f" over {list(tl.values())[6]} waves; mean tail (queue empty -> wave done) "
            + f"{list(tl.values())[7] / max(list(tl.values())[6], 1):.0f} cycles")
    if use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if share_gpu else dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # self-check of the sharded path (untimed): the gathered + de-interleaved frame must equal
    # the frame rendered by one GPU alone, byte for byte
    shard_ok = None
    if sharded:
        run(1, 0)
        torch.cuda.synchronize()
        if rank == 0:
            direct = torch.zeros((H, W, 4), dtype=torch.uint8, device=dev)
            cam.transform = pose_of(0)
            api.launch_renderer(tree, cam, opts, direct, None, stream, True, fp_mode=fp_mode)
            torch.cuda.synchronize()
            shard_ok = bool(torch.equal(direct, frames[0][0]))
            log(f"[bench] sharded frame == single-GPU frame: {shard_ok}")

    kern_ms = [a.elapsed_time(b) for a, b in events]
    kern_total_s = float(np.sum(kern_ms)) / 1e3
    log(f"[bench r{rank}] launch durations (HIP events, ms): " + " ".join(f"{x:.3f}" for x in kern_ms[:16]))
    kern_mean_s = kern_total_s / n_launch
    alg_bytes_per_launch = alg_bytes_per_frame * K / n_launch

    # HBM-side traffic: PMC counters need rocprofv3 runs of their own (tools/measure_traffic.py, one
    # counter group per pass), so the bench reports the committed measurement of this config,
    # scaled to this launch size -- but only while the SHA-256 of the kernel sources recorded in
    # it still matches the sources this run was built from; otherwise null, and the reason.
    traffic, traffic_src, traffic_extrapolated = None, None, None
    launch_sizes = [min(B, K - j * B) for j in range(n_launch)]
    live, committed_note = None, None
    want_live = args.live_traffic == 1
    if world == 1 and rank == 0 and want_live and not args.readback and n_streams == 1:
        # the child processes of the live passes upload the tree themselves: this process is done
        # with the GPU, so its copy (and the frame sets) go first
        tree.free_device()
        tree = None
        live = live_traffic(args.config, args.fp, launch_sizes[0], args.tune)
    if world == 1:
        per_frame, traffic_src, profiled_fpl = committed_traffic(args.config, args.fp,
                                                                 want_fpl=launch_sizes[0])
        committed_note = traffic_src
        if live is not None and "read_bytes_per_frame" in live:
            # measured by THIS run, at this run's launch shape, on this box
            per_frame = live["read_bytes_per_frame"] + live.get("write_bytes_per_frame", 0.0)
            profiled_fpl = launch_sizes[0]
            traffic_src = (f"measured by this run: two rocprofv3 --kernel-trace --pmc passes of the same "
                           f"launch shape ({launch_sizes[0]} frames per launch) after the timed region "
                           f"(tools/measure_traffic.py: TCC_EA0_RDREQ_128B*128 + _64B*64 + _32B*32 + "
                           f"WRITE_SIZE; kernel under the profiler "
                           f"{live['kernel_ms_under_pmc'].get('rdsize', 0):.3f} ms per launch); committed "
                           f"profile: {committed_note}")
        elif live is not None:
            traffic_src = f"{traffic_src}; live measurement failed: {live.get('error', '?')[:160]}"
        if per_frame is not None:
            traffic = int(per_frame * K / n_launch)
            traffic_extrapolated = any(n != profiled_fpl for n in launch_sizes)
            if traffic_extrapolated:
                traffic_src += (f"; EXTRAPOLATED: profiled at {profiled_fpl} frames per launch, this "
                                f"run launched {launch_sizes[0]} -- bytes per frame scaled to the "
                                f"launch size, not measured at it")

    kernel_name = "render_kernel"
    parity = None
    if rank == 0 and parity_frames:
        parity = parity_check(stree, parity_frames, W, H, focal, args.fp)
        log(f"[bench] parity vs oracle: {parity}")
    if rank == 0:
        # replicas: every rank rendered its own K frames; tile mode: the K frames were shared
        rays_total = W * H * K * (world if replicas else 1)
        mrays = rays_total / elapsed / 1e6
        achieved = alg_bytes_per_launch / kern_mean_s / 1e9
        result = {
            "metric": f"Mrays/s at {W}x{H}, NeRF-synthetic-like {cfg['fmt']}{cfg['basis_dim']} "
                      f"(synthetic {args.config})",
            "value": round(mrays, 3),
            "unit": "Mrays/s",
            "fps": round(K * (world if replicas else 1) / elapsed, 3),
            "n_gpus": world,
            "steps": K,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / K * 1e3, 5),
            "higher_is_better": True,
            "scaling": "weak" if replicas else "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic" + (" (REHEARSAL: ranks share one GPU, gloo)" if share_gpu else ""),
            "config": {
                "workload": f"{args.config}: synthetic lego-like PlenOctree, depth {cfg['depth']}, "
                            f"{cfg['fmt']}{cfg['basis_dim']}, {info['capacity']} nodes, "
                            f"{info['device_bytes'] / 1e9:.2f} GB in HBM, {W}x{H}, "
                            f"fx=fy={focal}, 200-pose orbit, default RenderOptions",
                "fp_mode": args.fp,
                "frames_per_launch": launch_sizes[0] if len(set(launch_sizes)) == 1 else launch_sizes,
                "launches": n_launch,
                "pcie_inclusive": bool(args.readback),
                "launch_streams": n_streams,
                "preroll": (f"{args.preroll} untimed frames of other poses directly in front of the "
                            f"{args.warmup} warm-up steps (GPU clocks of a running render loop; "
                            f"--preroll 0 = a launch out of an idle GPU)") if args.preroll > 0 else 0,
                "sharded_frame_matches_single_gpu": shard_ok,
                "parallelism": "single GPU" if world == 1 else (
                    f"{world} replicas: one tree (seeds 1010..) and pose stream per GPU, no "
                    f"communication" if replicas else
                    f"screen tiles {tile_w}x{tile_h} round-robin over {world} GPUs, "
                    f"tree replicated, one RCCL gather of RGBA8 to rank 0 per launch"),
            },
            "roofline": {
                "bound": "hbm",
                "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5),
                "achieved_is": "ALGORITHMIC bytes (SURVEY 8d formula, counted by the instrumented "
                               "flavour) / launch duration -- mostly served by L2 / Infinity Cache; "
                               "the HBM-side rate is traffic_gbps",
                "traffic": traffic,
                "traffic_unit": "bytes per launch (L2<->fabric reads+writes)",
                "traffic_extrapolated": traffic_extrapolated,
                "traffic_gbps": None if traffic is None else round(traffic / kern_mean_s / 1e9, 1),
                "traffic_frac": None if traffic is None else round(
                    traffic / kern_mean_s / 1e9 / HBM_PEAK_GBS, 4),
                "traffic_source": traffic_src,
                "kernel": f"vr::{kernel_name}<{args.fp}, {cfg['fmt']}{cfg['basis_dim']}> "
                          f"(persistent march/shade, {launch_sizes[0]} frames per launch; the launch "
                          f"also runs prepare_launch_kernel + raygen_kernel, ~5 % of its time)",
                "kernel_ms_mean": round(kern_mean_s * 1e3, 5),
                "kernel_ms_per_frame": round(kern_total_s / K * 1e3, 5),
                "alg_bytes_per_launch": int(alg_bytes_per_launch),
                "samples_per_ray": round(cnt["samples"] / max(cnt["rays"], 1), 2),
                "hit_samples_per_ray": round(cnt["hit_samples"] / max(cnt["rays"], 1), 2),
                "child_words_per_sample": round(cnt["child_reads"] / max(cnt["samples"], 1), 2),
                # B_unique: distinct 128-byte lines touched in the device arrays (x 128 bytes)
                "unique_bytes_per_launch": 128 * sum(unique_launch[1].values()),
                "unique_launch_frames": unique_launch[0],
                "unique_bytes_per_frame": int(128 * np.mean([sum(u.values()) for u in unique_frames])),
                "unique_lines_per_frame_by_array": {
                    k: int(np.mean([u[k] for u in unique_frames])) for k in unique_frames[0]},
            },
        }
        if cold_ms is not None:
            result["ms_per_step_cold"] = round(cold_ms, 5)
            result["cold_is"] = ("the identical K-step region once more behind 50 ms of idling (no pre-roll, "
                                 "no warm-up in front): a launch out of an idle GPU, the condition of rounds 1-3")
        if repeat_ms:
            allms = sorted([elapsed / K * 1e3] + repeat_ms)
            result["repeats"] = {
                "what": "the identical K-step region repeated after the timed one (same launches, "
                        "same poses); min / median / max include the timed region itself",
                "ms_per_step": [round(x, 5) for x in repeat_ms],
                "min": round(allms[0], 5), "median": round(allms[len(allms) // 2], 5),
                "max": round(allms[-1], 5)}
        # lane utilisation of the march and shade rounds (instrumented flavour, same poses, outside
        # the timed region) and -- from the committed PMC passes of these very sources -- the
        # instructions a frame executes: a later change is checkable against this line alone
        prof = committed_traffic.last_profile or {}
        result["sched"] = {
            "march_util": round(sched["march_lanes"] / max(64 * sched["march_rounds"], 1), 4),
            "shade_util": round(sched["shade_lanes"] / max(64 * sched["shade_rounds"], 1), 4),
            "march_rounds_per_frame": int(sched["march_rounds"] / n_distinct),
            "shade_rounds_per_frame": int(sched["shade_rounds"] / n_distinct),
            "valu_insts_per_frame": prof.get("valu_insts_per_frame"),
            "salu_insts_per_frame": prof.get("salu_insts_per_frame"),
            "valu_source": committed_note if prof else "no hash-verified PMC profile of these sources",
        }
        if parity is not None:
            result["parity"] = parity
        # roofline.model: the kernel's OWN lower bounds per launch, each from a measurement -- the
        # algorithmic `frac` above saturates (its bytes are the reference's root-to-leaf reads, which
        # this kernel mostly serves from two cached lookups), this one keeps its headroom visible
        fpl = launch_sizes[0]
        bounds = {}
        if traffic is not None:
            bounds["t_fabric_ms"] = traffic / 128.0 / GATHER_LINES_PER_S * 1e3
        flat4, im = None, None
        if prof.get("valu_insts_per_frame"):
            flat4 = prof["valu_insts_per_frame"] * fpl * VALU_ISSUE_CYCLES / (N_SIMD * SIMD_HZ) * 1e3
            im = issue_model(args.fp, cfg["basis_dim"], bool(info.get("brick_blocked")))
            if im:  # march rounds (instrumented flavour) x the round's static mix + the rest at the rest's mix
                v_march = min(sched["march_rounds"] / n_distinct * im["march_valu"], prof["valu_insts_per_frame"])
                cyc = v_march * im["march_cpv"] + (prof["valu_insts_per_frame"] - v_march) * im["rest_cpv"]
                bounds["t_issue_ms"] = cyc * fpl / (N_SIMD * SIMD_HZ) * 1e3
            else:
                bounds["t_issue_ms"] = flat4
        if parity is not None and parity.get("longest_ray_samples"):
            bounds["t_chain_ms"] = parity["longest_ray_samples"] * ROUND_FLOOR_US * 1e-3
        if bounds:
            binding = max(bounds, key=bounds.get)
            result["roofline"]["model"] = {
                **{k: round(v, 5) for k, v in bounds.items()},
                "binding": binding,
                "kernel_ms": round(kern_mean_s * 1e3, 5),
                "frac_of_model": round(bounds[binding] / (kern_mean_s * 1e3), 4),
                "t_issue_flat4_ms": None if flat4 is None else round(flat4, 5),
                "frac_of_model_flat4": None if flat4 is None else round(
                    max(flat4, *[v for k, v in bounds.items() if k != "t_issue_ms"]) / (kern_mean_s * 1e3), 4),
                "t_issue_is": (f"class-weighted: {im['march_valu']} vector instructions per march round at "
                               f"{im['march_cpv']} cycles, the rest at {im['rest_cpv']} ({im['source']}: ISA class "
                               "histogram x 2.5 / 4 / 7.6 cycles, profiles/r04_valu_issue_cost.jsonl)") if im else
                              "flat 4 cycles per vector instruction (no hash-verified ISA histogram of these sources)",
                "what": "lower bounds of ONE launch: t_fabric = measured L2<->fabric lines / the chip's "
                        f"random-line rate ({GATHER_LINES_PER_S / 1e9:.1f} G lines/s, profiles/r01_gather_ceiling.txt); "
                        f"t_issue = vector instructions (hash-verified PMC profile) x their class-weighted cycles "
                        f"(t_issue_is; flat {VALU_ISSUE_CYCLES:.0f} cycles: t_issue_flat4_ms) / "
                        f"({N_SIMD} SIMDs x {SIMD_HZ / 1e9:.1f} GHz); t_chain = samples of the longest ray of the "
                        f"checked frames x {ROUND_FLOOR_US} us (a dependent march round of a lone wave, "
                        "profiles/r05_tail_profile.jsonl); frac_of_model = the largest / the launch's duration",
            }
        if use_dist:
            result["rccl"] = {"world_size": dist.get_world_size(), "backend": dist.get_backend(),
                              "gather": gather_path,
                              "nccl_version": ".".join(map(str, torch.cuda.nccl.version()))
                              if not share_gpu else None,
                              "devices": torch.cuda.device_count()}
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(stree, transforms, W, H, focal, args.cpu_budget)
            result["cpu_baseline"]["rtfrag"] = rtfrag_baseline()
        else:
            result["cpu_baseline"] = None
        os.write(json_fd, (json.dumps(result) + "\n").encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if tree is not None:
        tree.free_device()


if __name__ == "__main__":
    main()
