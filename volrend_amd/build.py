"""Builds libvolrend_hip.so (gfx950) in-tree with hipcc.

    python -m volrend_amd.build [--force]
    python -m volrend_amd.build --variant abl4 -DVR_ABLATE=4     (measurement builds, below)

-ffp-contract=off is part of the numerical contract (see csrc/vr_device_math.h).

The product sources carry no experiment hooks.  A measurement build (timing ablations
-DVR_ABLATE=n, shader-clock timelines -DVR_TIMELINE=n, register / round-size knobs -DVR_SH16_WAVES=...)
is generated: the sources are copied to a scratch directory, tools/experiments/kernel_hooks.patch
puts the hook sites into the copy (VR_EXP_* / TL_* macros of tools/experiments/vr_experiment_hooks.h)
and the copy is compiled into libvolrend_hip_<variant>.so, which a measurement tool selects with
VOLREND_HIP_LIB.  tests/test_abi.py keeps the patch applicable.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libvolrend_hip.so")
SOURCES = ["vr_kernels.hip", "vr_api.cpp"]
HEADERS = ["vr_internal.h", "vr_device_math.h", os.path.join(ROOT, "include", "volrend_hip.h")]
EXPERIMENTS = os.path.join(ROOT, "tools", "experiments")
HOOK_FLAGS = ("-DVR_ABLATE", "-DVR_TIMELINE", "-DVR_ROLE_DEBUG", "-DVR_MIN_WAVES_PER_EU", "-DVR_SH16_ROWS",
              "-DVR_SHADE_SCHED_BARRIER", "-DVR_SH16_WAVES", "-DVR_SH25_WAVES", "-DVR_SH9_WAVES", "-DVR_PACKED_EXP",
              "-DVR_STEAL_MIN", "-DVR_EXP_SH25_STRIDE", "-DVR_TOUCH_LEAF_SHIFT")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    "-ffp-contract=off",                                # explicit fmaf only
    "-fhip-fp32-correctly-rounded-divide-sqrt",         # IEEE / and sqrt, as nvcc's default
    "-fno-gpu-flush-denormals-to-zero",
    "-fno-fast-math",
    "-x", "hip",
]


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [
        h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, extra_flags=(), variant: str = "") -> str:
    """variant: experiment / profiling build next to the product library
    (libvolrend_hip_<variant>.so, selected at run time with VOLREND_HIP_LIB)."""
    hooked = any(f.startswith(HOOK_FLAGS) for f in extra_flags) or "--hooks" in extra_flags
    extra_flags = [f for f in extra_flags if f != "--hooks"]
    if not variant and hooked:
        raise ValueError("experiment hooks (VR_ABLATE / VR_TIMELINE / knob overrides) never go into the product "
                         "library: build them with --variant NAME")
    out = LIB if not variant else os.path.join(HERE, f"libvolrend_hip_{variant}.so")
    if not variant and not force and not needs_build():
        return LIB
    src_dir, scratch = CSRC, None
    if hooked:
        scratch = tempfile.mkdtemp(prefix="vr_hooks_")
        src_dir = hooked_sources(scratch)
    try:
        cmd = [HIPCC, *FLAGS, *extra_flags, "-I", os.path.join(ROOT, "include"), "-I", src_dir,
               *[os.path.join(src_dir, s) for s in SOURCES], "-o", out]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    finally:
        if scratch:
            shutil.rmtree(scratch, ignore_errors=True)
    return out


def hooked_sources(dst: str, dry_run: bool = False) -> str:
    """Copy of csrc/ with the experiment hook sites patched in (tools/experiments/)."""
    for f in os.listdir(CSRC):
        if f.endswith((".hip", ".h", ".cpp")):
            shutil.copy(os.path.join(CSRC, f), os.path.join(dst, f))
    shutil.copy(os.path.join(EXPERIMENTS, "vr_experiment_hooks.h"), os.path.join(dst, "vr_experiment_hooks.h"))
    cmd = ["patch", "-p1", "--no-backup-if-mismatch", "-s", "-i", os.path.join(EXPERIMENTS, "kernel_hooks.patch")]
    if dry_run:
        cmd.insert(1, "--dry-run")
    subprocess.check_call(cmd, cwd=dst)
    return dst


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a != "--force"]
    variant = ""
    if "--variant" in args:
        i = args.index("--variant")
        variant = args[i + 1]
        del args[i:i + 2]
    print(build(force="--force" in sys.argv, verbose=True, extra_flags=args, variant=variant))
