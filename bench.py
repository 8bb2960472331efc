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
CACHE_DIR = os.environ.get("VOLREND_BENCH_CACHE", "/dev/shm/volrend_amd_cache")


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def load_or_make_tree(synth, name: str, local_rank: int, barrier):
    """Rank-local 0 generates the seeded tree once per node; the others mmap it."""
    os.makedirs(CACHE_DIR, exist_ok=True)
    cfg = synth.CONFIGS[name]
    tag = f"{name}_s{cfg['seed']}_d{cfg['depth']}_b{cfg['basis_dim']}_sh{cfg['shell_leaves']}"
    fchild = os.path.join(CACHE_DIR, tag + "_child.npy")
    fdata = os.path.join(CACHE_DIR, tag + "_data.npy")
    fdone = os.path.join(CACHE_DIR, tag + ".done")
    if local_rank == 0 and not os.path.exists(fdone):
        t0 = time.time()
        tree = synth.make_config_tree(name)
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


def cpu_baseline(tree, transform, width, height, focal, budget_s=12.0):
    """Oracle (strict) on all host cores over a bounded sample: centre rows of the
    frame, 8 rows at a time, until ~budget_s of CPU time has been spent."""
    from oracle import binding as ob
    th = ob.TreeHandle(tree)
    cam = ob.make_camera(transform, width, height, focal)
    opt = ob.default_options()
    cores = os.cpu_count() or 1
    rays = 0
    t_total = 0.0
    y = (height // 2) // 8 * 8
    bands = 0
    while t_total < budget_s and y + 8 <= height:
        t0 = time.perf_counter()
        ob.render(th, cam, opt, ob.FP_STRICT, region=(0, y, width, 8), want_accum=False,
                  nthreads=cores)
        t_total += time.perf_counter() - t0
        rays += width * 8
        y += 8
        bands += 1
    mrays = rays / t_total / 1e6
    return {"value": round(mrays, 4), "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": f"{bands} 8-row bands ({rays} rays) from the image centre of pose 0, "
                      f"{t_total:.1f}s, oracle strict mode, {cores} threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="C1", choices=["C0", "C1", "C2", "C3"])
    ap.add_argument("--fp", default="strict", choices=["strict", "fma"])
    ap.add_argument("--tile-rows", type=int, default=8, help="rows per interleaved screen tile")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=12.0)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from volrend_amd import _abi, api, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run for --gpus > 1")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    def barrier():
        if world > 1:
            dist.barrier()

    cfg = synth.CONFIGS[args.config]
    W, H, focal = cfg["width"], cfg["height"], cfg["focal"]
    stree = load_or_make_tree(synth, args.config, local_rank, barrier)
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

    tile_h = max(8, (args.tile_rows // 8) * 8)
    tile_w = (W + 7) // 8 * 8
    shard = api.TileShard(tile_w, tile_h, rank, world, compact=True)
    frame = torch.zeros((H, W, 4), dtype=torch.uint8, device=dev)
    if world > 1:
        nbytes = api.compact_bytes(W, H, shard)
        bufs = [torch.zeros(nbytes, dtype=torch.uint8, device=dev) for _ in range(2)]
        gathered = [torch.zeros((world, nbytes), dtype=torch.uint8, device=dev)
                    for _ in range(2)] if rank == 0 else [None, None]

    def render_step(i, ev=None):
        cam.transform = transforms[i % len(transforms)]
        if ev is not None:
            ev[0].record(stream)
        if world == 1:
            api.launch_renderer(tree, cam, opts, frame, None, stream, True, fp_mode=fp_mode)
        else:
            api.launch_renderer(tree, cam, opts, bufs[i % 2], None, stream, True, shard=shard,
                                fp_mode=fp_mode)
        if ev is not None:
            ev[1].record(stream)

    works = {}

    def gather_step(i):
        if world == 1:
            return
        if rank == 0:
            glist = [gathered[i % 2][r] for r in range(world)]
            works[i] = dist.gather(bufs[i % 2], glist, dst=0, async_op=True)
        else:
            works[i] = dist.gather(bufs[i % 2], None, dst=0, async_op=True)

    def retire(i):
        """Frame i's gather must be complete before its buffers are reused."""
        if world == 1 or i not in works:
            return
        works.pop(i).wait()
        if rank == 0:
            api.assemble_tiles(frame, gathered[i % 2], W, H, shard, stream)

    def run(n_steps, first, events=None):
        for s in range(n_steps):
            i = first + s
            retire(i - 2)
            render_step(i, events[s] if events else None)
            gather_step(i)
        retire(first + n_steps - 2)
        retire(first + n_steps - 1)

    # ---- warm-up (untimed) ---------------------------------------------------
    run(args.warmup, 0)
    torch.cuda.synchronize()

    # ---- algorithmic bytes per launch: instrumented flavour, outside the timed region
    K = args.steps
    n_distinct = min(K, len(transforms))
    counters = torch.zeros(7, dtype=torch.int64, device=dev)
    scratch = torch.zeros((H, W, 4), dtype=torch.uint8, device=dev)
    full = api.TileShard(tile_w, tile_h, rank, world, compact=False)
    for j in range(n_distinct):
        cam.transform = transforms[(args.warmup + j) % len(transforms)]
        api.launch_renderer(tree, cam, opts, scratch, None, stream, True, shard=full,
                            fp_mode=fp_mode, counters=counters)
    torch.cuda.synchronize()
    cnt = dict(zip(_abi.COUNTER_FIELDS, [int(v) for v in counters.cpu().tolist()]))
    reps = K / n_distinct
    alg_bytes_per_launch = cnt["alg_bytes"] / n_distinct  # this rank's launches
    del scratch

    # ---- timed region ----------------------------------------------------------
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
              for _ in range(K)]
    barrier()
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    run(K, args.warmup, events)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t_start

    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    kern_ms = [a.elapsed_time(b) for a, b in events]
    kern_mean_s = float(np.mean(kern_ms)) / 1e3

    result = None
    if rank == 0:
        rays_total = W * H * K
        mrays = rays_total / elapsed / 1e6
        achieved = alg_bytes_per_launch / kern_mean_s / 1e9
        result = {
            "metric": "Mrays/s at 800x800, NeRF-synthetic-like lego SH16 (synthetic C1)",
            "value": round(mrays, 3),
            "unit": "Mrays/s",
            "fps": round(K / elapsed, 3),
            "n_gpus": world,
            "steps": K,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / K * 1e3, 5),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{args.config}: synthetic lego-like PlenOctree, depth {cfg['depth']}, "
                            f"{cfg['fmt']}{cfg['basis_dim']}, {info['capacity']} nodes, "
                            f"{info['device_bytes'] / 1e9:.2f} GB in HBM, {W}x{H}, "
                            f"fx=fy={focal}, 200-pose orbit, default RenderOptions",
                "fp_mode": args.fp,
                "parallelism": "single GPU" if world == 1 else
                               f"screen tiles {tile_w}x{tile_h} round-robin over {world} GPUs, "
                               f"tree replicated, RCCL gather of RGBA8 to rank 0 per frame",
            },
            "roofline": {
                "bound": "hbm",
                "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5),
                "traffic": None,
                "kernel": "vr::render_kernel<strict|fma, SH16, FAST>",
                "kernel_ms_mean": round(kern_mean_s * 1e3, 5),
                "alg_bytes_per_launch": int(alg_bytes_per_launch),
                "samples_per_ray": round(cnt["samples"] / max(cnt["rays"], 1), 2),
                "hit_samples_per_ray": round(cnt["hit_samples"] / max(cnt["rays"], 1), 2),
                "child_words_per_sample": round(cnt["child_reads"] / max(cnt["samples"], 1), 2),
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(stree, transforms[0], W, H, focal,
                                                  args.cpu_budget)
        else:
            result["cpu_baseline"] = None
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    tree.free_device()


if __name__ == "__main__":
    main()
