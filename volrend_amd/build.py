"""Builds libvolrend_hip.so (gfx950) in-tree with hipcc.

    python -m volrend_amd.build [--force]

-ffp-contract=off is part of the numerical contract (see csrc/vr_device_math.h).
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libvolrend_hip.so")
SOURCES = ["vr_kernels.hip", "vr_api.cpp"]
HEADERS = ["vr_internal.h", "vr_device_math.h", "vr_experiment_hooks.h", os.path.join(ROOT, "include", "volrend_hip.h")]
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
    if not variant and any(f.startswith(("-DVR_ABLATE", "-DVR_TIMELINE", "-DVR_ROLE_DEBUG"))
                           for f in extra_flags):
        raise ValueError("experiment hooks (VR_ABLATE / VR_TIMELINE) never go into the product library: "
                         "build them with --variant NAME")
    out = LIB if not variant else os.path.join(HERE, f"libvolrend_hip_{variant}.so")
    if not variant and not force and not needs_build():
        return LIB
    cmd = [HIPCC, *FLAGS, *extra_flags, "-I", os.path.join(ROOT, "include"), "-I", CSRC,
           *[os.path.join(CSRC, s) for s in SOURCES], "-o", out]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a != "--force"]
    variant = ""
    if "--variant" in args:
        i = args.index("--variant")
        variant = args[i + 1]
        del args[i:i + 2]
    print(build(force="--force" in sys.argv, verbose=True, extra_flags=args, variant=variant))
