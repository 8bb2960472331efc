#!/usr/bin/env python3
"""Static instruction-class histogram of the production render kernel, for roofline.model.t_issue.

bench.py priced every vector instruction at 4 SIMD cycles.  The chip's own numbers
(tools/ubench/valu_rate*.hip, profiles/r04_valu_issue_cost.jsonl; 5 waves per SIMD) say:
    2.5  fp32 add / sub / mul / fma / fmac and the simple integer ops (add, sub, and, or, xor, right
         shift, mov) with register, literal or inline operands
    4    ANY vector instruction with an SGPR operand, compares, min / max / med3, left shifts,
         bfe / lshl_or / add_lshl / or3, conversions, fract, ldexp, rndne, mbcnt, alignbit, cndmask,
         v_fma_mix_f32, v_pk_*_f32, fp64, div_scale / fmas / fixup, ...
    7.6  v_rcp_f32 / v_exp_f32 (transcendental unit)
This tool compiles vr_kernels.hip to ISA (hipcc -S, no GPU needed), takes the FAST flavour of a basis
size (default SH16, strict), and counts the three classes (a) in the march round -- the loop from its
header to the first shade round, as tools/march_loop_isa.sh cuts it -- and (b) in the rest of the
kernel (shade rounds, retire / refill).  bench.py weights (a) with the march rounds the instrumented
flavour counts and (b) with what remains of the PMC total (SQ_INSTS_VALU); the transcendental count
is also measured (SQ_INSTS_VALU_TRANS_F32) and reported next to the static estimate.

    python tools/isa_issue_model.py [--basis 16] [--out profiles/r06_isa_issue_model.json]
"""
from __future__ import annotations

import argparse
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

CYCLES = {"cheap": 2.5, "full": 4.0, "trans": 7.6}
CHEAP_OPS = {"v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_add_u32",
             "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_mov_b32"}
TRANS_OPS = ("v_rcp_", "v_exp_", "v_log_", "v_sqrt_", "v_rsq_", "v_sin_", "v_cos_")


def classify(line: str):
    """'cheap' / 'full' / 'trans' for a vector ALU instruction, None for anything else."""
    m = re.match(r"\s+(v_[a-z0-9_]+)\s*(.*)", line.split(";")[0])
    if not m:
        return None
    op, args = m.group(1), m.group(2)
    if op.startswith(TRANS_OPS):
        return "trans"
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    sgpr_operand = re.search(r"(?<![a-z0-9_])(s\d+|s\[\d+:\d+\]|vcc|exec|m0)(?![a-z0-9_])", args) is not None
    if base in CHEAP_OPS and not sgpr_operand and not op.endswith(("_sdwa", "_dpp")) and \
            (not op.endswith("_e64") or base == "v_fma_f32"):
        return "cheap"
    return "full"


def histogram(lines):
    h = {"cheap": 0, "full": 0, "trans": 0, "salu": 0, "vmem": 0, "lds": 0}
    for l in lines:
        c = classify(l)
        if c:
            h[c] += 1
            continue
        m = re.match(r"\s+([a-z_0-9]+)", l.split(";")[0])
        if not m:
            continue
        op = m.group(1)
        if op.startswith(("s_waitcnt", "s_nop", "s_cbranch", "s_branch", "s_endpgm", "s_barrier")):
            continue
        if op.startswith("s_"):
            h["salu"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            h["vmem"] += 1
        elif op.startswith("ds_"):
            h["lds"] += 1
    h["valu"] = h["cheap"] + h["full"] + h["trans"]
    h["cycles_per_valu"] = round(sum(h[k] * CYCLES[k] for k in CYCLES) / max(h["valu"], 1), 4)
    return h


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--basis", default="16,9,25", help="basis sizes (one entry each)")
    ap.add_argument("--fp", default="strict,fma")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    from measure_traffic import kernel_source_hash
    from volrend_amd import build as vb
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "k.s")
        flags = [f for f in vb.FLAGS if f not in ("-shared", "-fPIC")]
        subprocess.check_call([vb.HIPCC, *flags, "-I", os.path.join(ROOT, "include"), "-I", vb.CSRC, "-S",
                               "--cuda-device-only", os.path.join(vb.CSRC, "vr_kernels.hip"), "-o", asm],
                              stderr=subprocess.DEVNULL)
        text = open(asm).read().split("\n")
    kernels = {}
    for fp in args.fp.split(","):
        for basis in [int(b) for b in args.basis.split(",")]:
            for blk in (0, 1):  # brick entry order (x-major / line blocks): a template parameter of the FAST flavours
                fma = 1 if fp == "fma" else 0
                sym = f"_ZN2vr12_GLOBAL__N_113render_kernelILi{fma}ELi{basis}ELi0ELb{blk}EEEvNS_7KParamsE"
                start = next(i for i, l in enumerate(text) if l.startswith(sym + ":"))
                end = next(i for i in range(start, len(text)) if "s_endpgm" in text[i])
                body = text[start:end + 1]
                hdr = [i for i, l in enumerate(body) if "Loop Header: Depth=2" in l and "Inner" not in l][0]
                d3 = [i for i, l in enumerate(body) if "Depth=3" in l and i > hdr]
                march = body[hdr - 8: d3[1]]
                rest = body[:hdr - 8] + body[d3[1]:]
                kernels[f"{fp}/SH{basis}/{'blocked' if blk else 'xmajor'}"] = {
                    "march_round": histogram(march), "rest_of_kernel": histogram(rest), "whole_kernel": histogram(body)}
    rec = {"kernel": "render_kernel<fp, basis, FAST, brick order>", "kernel_source_sha256": kernel_source_hash(),
           "cycles_per_class": CYCLES,
           "class_rule": "cheap = fp32 add/sub/mul/fma/fmac and integer add/sub/and/or/xor/lshr/mov without an SGPR operand "
                         "(VOP3 forms of two-operand ops count as full); trans = rcp/exp/log/sqrt/rsq; full = every other "
                         "vector ALU instruction (profiles/r04_valu_issue_cost.jsonl)",
           "kernels": kernels}
    s = json.dumps(rec, indent=1)
    for k, v in kernels.items():
        print(k, "march", v["march_round"]["valu"], v["march_round"]["cycles_per_valu"], "rest", v["rest_of_kernel"]["valu"],
              v["rest_of_kernel"]["cycles_per_valu"], file=sys.stderr)
    if args.out:
        open(args.out, "w").write(s + "\n")


if __name__ == "__main__":
    main()
