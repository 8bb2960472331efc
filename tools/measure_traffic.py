#!/usr/bin/env python3
"""HBM-side traffic and issue counters of the render kernel, measured with rocprofv3 PMC passes.

    python tools/measure_traffic.py --config C1 [--fp strict] [--batch 64] [--out profiles/r02_traffic_C1.json]
                                    [--groups rdsize write fetch tcc sq1 sq2] [--keep-csv DIR]

Run ON the GPU box (through gpurun) from the repo root.  One counter group per rocprofv3 run
(`--kernel-trace --pmc ...` only: never combined with sys / hip / hsa tracing), each run being
`bench.py --config C --steps 2*batch --warmup batch --no-cpu-baseline` under the profiler.  Counter
values are summed over the dispatch's XCD instances and averaged over the profiled dispatches of
the production kernel flavour (render_kernel<fp, basis, FAST>).

Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section):
  * read bytes come from the request-size histogram, TCC_EA0_RDREQ_128B*128 + _64B*64 + _32B*32;
    FETCH_SIZE (KB) = RDREQ*64 B under-counts a kernel whose requests are all 128 B by 2x -- the
    ratio is recorded as `fetch_size_undercount`;
  * write bytes = WRITE_SIZE (KB).

The result records a SHA-256 of the kernel sources + build flags; bench.py reports `roofline.traffic`
from this file only while that hash still matches what it runs, so the figure cannot go stale
silently.
"""
from __future__ import annotations

import argparse
import csv
import glob
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

GROUPS = {
    "rdsize": "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum",
    "write": "WRITE_SIZE",
    "fetch": "FETCH_SIZE GRBM_GUI_ACTIVE",
    "tcc": "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum",
    "sq1": "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU "
           "SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY",
    "sq2": "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SALU "
           "SQ_INST_LEVEL_VMEM SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_LDS",
    "tcp1": "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum",
    "tcp2": "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum "
            "TCP_TCC_READ_REQ_LATENCY_sum",
    "tcp3": "TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum "
            "TCP_TCR_TCP_STALL_CYCLES_sum",
    "tcp4": "TCP_TCP_LATENCY_sum TCP_RFIFO_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_TCR_RDRET_STALL_sum",
    "sq3": "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH",
    "sq4": "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_WAIT_INST_LDS",
    # TA_* and TD_* counters abort rocprofv3 on this pool (measured twice in round 1): not offered
}
SOURCES = ["volrend_amd/csrc/vr_kernels.hip", "volrend_amd/csrc/vr_device_math.h",
           "volrend_amd/csrc/vr_internal.h", "volrend_amd/csrc/vr_api.cpp", "include/volrend_hip.h"]


def kernel_source_hash() -> str:
    """SHA-256 over the sources the render kernel is built from + the build flags."""
    from volrend_amd import build as vb
    h = hashlib.sha256()
    for rel in SOURCES:
        h.update(open(os.path.join(ROOT, rel), "rb").read())
    h.update(" ".join(vb.FLAGS).encode())
    return h.hexdigest()


def run_group(name: str, counters: str, bench_args: list[str], out_dir: str, timeout: int):
    """One rocprofv3 pass; returns ({counter: [per-dispatch sums]}, [durations ns]) of the kernel."""
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = ["rocprofv3", "--kernel-trace", "--pmc", *counters.split(), "-d", out_dir, "-o", name,
           "--output-format", "csv", "--", sys.executable, os.path.join(ROOT, "bench.py"), *bench_args]
    p = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=timeout)
    if p.returncode != 0:
        raise RuntimeError(f"pass {name} failed: " + p.stderr.decode(errors="replace")[-3000:])
    return p.stdout.decode()


def collect(out_dir: str, group: str, kernel_substr: str):
    per = defaultdict(float)
    for f in glob.glob(os.path.join(out_dir, "**", f"{group}_counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if kernel_substr in row["Kernel_Name"]:
                per[(row["Dispatch_Id"], row["Counter_Name"])] += float(row["Counter_Value"])
    vals = defaultdict(list)
    for (_, cname), v in per.items():
        vals[cname].append(v)
    dur = []
    for f in glob.glob(os.path.join(out_dir, "**", f"{group}_kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if kernel_substr in row["Kernel_Name"]:
                dur.append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
    return vals, dur


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C1")
    ap.add_argument("--fp", default="strict", choices=["strict", "fma"])
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--groups", nargs="+", default=["rdsize", "write", "fetch", "tcc", "sq1"])
    ap.add_argument("--out", default="")
    ap.add_argument("--keep-csv", default="", help="copy the raw rocprofv3 CSVs here")
    ap.add_argument("--timeout", type=int, default=300)
    ap.add_argument("--bench-args", default="", help="extra bench.py arguments (quoted)")
    args = ap.parse_args()

    from volrend_amd import synth
    cfg = synth.CONFIGS[args.config]
    basis = cfg["basis_dim"] if cfg["fmt"] != "RGBA" else -1
    fpi = 1 if args.fp == "fma" else 0
    kernel = f"render_kernel<{fpi}, {basis}, 0"  # (+ the brick-order flavour: ", false>" / ", true>")
    bench_args = ["--config", args.config, "--fp", args.fp, "--batch", str(args.batch), "--steps",
                  str(2 * args.batch), "--warmup", str(args.batch), "--no-cpu-baseline", "--no-parity",
                  "--live-traffic", "0", "--repeats", "0", "--preroll", "0",
                  *args.bench_args.split()]
    tmp = tempfile.mkdtemp(prefix="vr_pmc_")
    c, durations, failed = {}, {}, []
    for g in args.groups:
        try:
            run_group(g, GROUPS[g], bench_args, tmp, args.timeout)
            vals, dur = collect(tmp, g, kernel)
            for k, v in vals.items():
                c[k] = sum(v) / len(v)
            durations[g] = sum(dur) / max(len(dur), 1)
            print(f"[pmc] {g}: {len(dur)} dispatches of {kernel}, mean {durations[g] / 1e6:.3f} ms",
                  file=sys.stderr, flush=True)
        except Exception as e:  # keep the passes that worked
            failed.append(f"{g}: {e}"[-600:])
            print(f"[pmc] {g} FAILED: {e}", file=sys.stderr, flush=True)
    if args.keep_csv:
        os.makedirs(args.keep_csv, exist_ok=True)
        for f in glob.glob(os.path.join(tmp, "**", "*.csv"), recursive=True):
            if "counter_collection" in f or "kernel_trace" in f:
                shutil.copy(f, args.keep_csv)
    shutil.rmtree(tmp, ignore_errors=True)

    n = args.batch
    out = {"config": args.config, "fp_mode": args.fp, "frames_per_launch": n,
           "kernel": f"vr::{kernel}", "kernel_source_sha256": kernel_source_hash(),
           "method": "rocprofv3 --kernel-trace --pmc, one counter group per run (tools/measure_traffic.py); "
                     "read bytes = TCC_EA0_RDREQ_128B*128 + _64B*64 + _32B*32 (FETCH_SIZE = RDREQ*64 B "
                     "under-counts 128-byte requests, MI355X_MICROARCH.md); write bytes = WRITE_SIZE KB",
           "groups": args.groups, "failed_groups": failed,
           "kernel_ms_under_pmc": {g: round(d / 1e6, 4) for g, d in durations.items()}}
    if "TCC_EA0_RDREQ_128B_sum" in c:
        rd = c["TCC_EA0_RDREQ_128B_sum"] * 128 + c["TCC_EA0_RDREQ_64B_sum"] * 64 + \
            c["TCC_EA0_RDREQ_32B_sum"] * 32
        out.update(read_bytes_per_launch=rd, read_bytes_per_frame=rd / n,
                   read_requests_per_launch=c["TCC_EA0_RDREQ_sum"],
                   frac_requests_128B=c["TCC_EA0_RDREQ_128B_sum"] / max(c["TCC_EA0_RDREQ_sum"], 1))
        if "FETCH_SIZE" in c:
            out["fetch_size_undercount"] = c["FETCH_SIZE"] * 1024 / rd
        if "rdsize" in durations:
            out["read_GBps_under_pmc"] = rd / durations["rdsize"]
    if "WRITE_SIZE" in c:
        out.update(write_bytes_per_launch=c["WRITE_SIZE"] * 1024,
                   write_bytes_per_frame=c["WRITE_SIZE"] * 1024 / n)
    if "TCC_REQ_sum" in c:
        out["l2_hit_rate"] = c["TCC_HIT_sum"] / max(c["TCC_REQ_sum"], 1)
        out["l2_requests_per_frame"] = c["TCC_REQ_sum"] / n
    if "SQ_INSTS_VALU" in c:
        out.update(valu_insts_per_frame=c["SQ_INSTS_VALU"] / n,
                   valu_lane_utilisation=c["SQ_THREAD_CYCLES_VALU"] / max(64 * c["SQ_ACTIVE_INST_VALU"], 1),
                   wave_wait_fraction=c["SQ_WAIT_ANY"] / max(c["SQ_WAVE_CYCLES"], 1),
                   vmem_read_insts_per_frame=c["SQ_INSTS_VMEM_RD"] / n)
        # SQ_ACTIVE_INST_VALU counts cycles (x4 quad-cycles on gfx9) in which a SIMD issues VALU work,
        # SQ_BUSY_CYCLES the cycles an SQ (one per XCD/SE instance) had waves: the ratio is only a
        # relative figure; the absolute one below prices every VALU instruction at 4 cycles on the
        # 1024 SIMDs over the kernel's duration.
        if "sq1" in durations:
            out["valu_issue_cycles_per_simd_over_kernel_cycles_at_2p4GHz"] = (
                c["SQ_INSTS_VALU"] * 4 / 1024) / (durations["sq1"] * 2.4)
    if "SQ_INSTS_SALU" in c:
        out.update(salu_insts_per_frame=c["SQ_INSTS_SALU"] / n, lds_insts_per_frame=c["SQ_INSTS_LDS"] / n,
                   valu_trans_insts_per_frame=c["SQ_INSTS_VALU_TRANS_F32"] / n)
    out["raw_counters_per_launch"] = {k: v for k, v in sorted(c.items())}
    js = json.dumps(out, indent=1)
    print(js)
    if args.out:
        with open(args.out, "w") as f:
            f.write(js + "\n")


if __name__ == "__main__":
    main()
