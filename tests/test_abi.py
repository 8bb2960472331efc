"""The C-ABI boundary: include/volrend_hip.h <-> libvolrend_hip.so <-> volrend_amd/_abi.py.
No GPU needed: only symbol tables, struct layouts and host-only entry points."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import pytest

from volrend_amd import _abi, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "volrend_hip.h")


def header_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"^\s*(?:const\s+)?(?:int64_t|int|void|const char\*|char\*)\s*\*?\s*(vr_\w+)\s*\(",
                       src, flags=re.M)
    return sorted(set(names))


@pytest.fixture(scope="module")
def lib_path():
    if not os.path.exists(_abi.LIB_PATH):
        build.build()
    return _abi.LIB_PATH


def test_every_header_symbol_is_exported(lib_path):
    funcs = header_functions()
    assert len(funcs) >= 18
    out = subprocess.check_output(["nm", "-D", "--defined-only", lib_path], text=True)
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    missing = [f for f in funcs if f not in exported]
    assert not missing, f"declared in volrend_hip.h but not exported: {missing}"
    # nothing of the C++ implementation leaks through unmangled
    stray = [s for s in exported if s.startswith("vr_") and s not in funcs]
    assert not stray, f"exported but undeclared: {stray}"


def test_ctypes_prototypes_cover_the_header(lib_path):
    assert sorted(_abi.PROTOTYPES) == header_functions()
    L = _abi.lib()  # resolves every symbol; raises AttributeError otherwise
    assert L.vr_abi_version() == _abi.ABI_VERSION == 3


def test_struct_layouts_match_the_c_compiler():
    structs = {"VrTreeDesc": _abi.VrTreeDesc, "VrQuantDesc": _abi.VrQuantDesc,
               "VrTreeInfo": _abi.VrTreeInfo,
               "VrCamera": _abi.VrCamera, "VrRenderOptions": _abi.VrRenderOptions,
               "VrFrame": _abi.VrFrame}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "volrend_hip.h"',
             'int main(void){']
    for name, st in structs.items():
        lines.append(f'printf("{name} %zu\\n", sizeof({name}));')
        for fname, _ in st._fields_:
            lines.append(f'printf("{name}.{fname} %zu\\n", offsetof({name}, {fname}));')
    lines.append('printf("VrCounters %zu\\n", sizeof(VrCounters)); return 0;}')
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "layout.c")
        open(src, "w").write("\n".join(lines))
        exe = os.path.join(td, "layout")
        subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), src, "-o", exe])
        got = dict(l.split() for l in subprocess.check_output([exe], text=True).splitlines())
    for name, st in structs.items():
        assert int(got[name]) == C.sizeof(st), name
        for fname, _ in st._fields_:
            assert int(got[f"{name}.{fname}"]) == getattr(st, fname).offset, f"{name}.{fname}"
    assert int(got["VrCounters"]) == 8 * len(_abi.COUNTER_FIELDS)


def test_host_only_entry_points(lib_path):
    L = _abi.lib()
    o = _abi.VrRenderOptions()
    L.vr_default_options(C.byref(o))
    # render_options.hpp:11-53 defaults
    assert abs(o.step_size - 1e-4) < 1e-10 and abs(o.sigma_thresh - 1e-2) < 1e-9
    assert abs(o.stop_thresh - 1e-2) < 1e-9 and o.background_brightness == 1.0
    assert list(o.render_bbox) == [0, 0, 0, 1, 1, 1] and list(o.basis_minmax) == [0, 24]
    assert o.probe_disp_size == 100 and o.grid_max_depth == 4 and list(o.probe) == [0, 0, 1]
    f = _abi.VrFrame()
    L.vr_default_frame(C.byref(f))
    assert f.offscreen == 1 and f.world == 1 and f.layout == _abi.LAYOUT_FRAME
    # compact buffer size of the 8-row-band shard used by bench.py at 8 GPUs
    assert L.vr_compact_bytes(800, 800, 800, 8, 8) == 13 * 800 * 8 * 4
    assert L.vr_compact_bytes(800, 800, 0, 0, 1) == 800 * 800 * 4
    assert L.vr_compact_bytes(800, 800, 12, 8, 2) == -1  # not a multiple of 8
    assert b"multiples of 8" in L.vr_last_error()
    assert L.vr_set_tuning(b"no_such_knob", 1) != 0
    assert L.vr_set_tuning(b"march_max", 2) == 0
    assert L.vr_set_tuning(b"march_max", 12) == 0   # (the default again: trees uploaded later copy it)
    assert L.vr_tree_set_tuning(None, b"march_max", 2) != 0 and b"NULL" in L.vr_last_error()
    assert L.vr_reserve_tiles(None, 800, 800, 1, 0, 0, 1, 2) != 0
    # argument validation happens before any device call
    assert L.vr_tree_upload(None, None) == 1
    d = _abi.VrTreeDesc()
    L.vr_default_tree_desc(C.byref(d))
    h = C.c_void_p()
    assert L.vr_tree_upload(C.byref(d), C.byref(h)) == 1 and not h.value


def test_bad_trees_are_rejected_on_the_host(lib_path):
    """Topology validation runs before any device work: cycles / out-of-range links
    would otherwise hang the descent loop on the GPU."""
    import numpy as np
    L = _abi.lib()

    def try_upload(child):
        child = np.ascontiguousarray(child, dtype=np.int32)
        cap = child.shape[0]
        data = np.zeros((cap, 8, 4), dtype=np.float16)
        d = _abi.VrTreeDesc()
        L.vr_default_tree_desc(C.byref(d))
        d.child, d.data = child.ctypes.data, data.ctypes.data
        d.capacity, d.data_dim, d.N = cap, 4, 2
        for i in range(3):
            d.scale[i], d.offset[i] = 1.0, 0.0
        h = C.c_void_p()
        rc = L.vr_tree_upload(C.byref(d), C.byref(h))
        return rc, L.vr_last_error().decode()

    cyc = np.zeros((2, 8), np.int32)
    cyc[0, 0] = 1
    cyc[1, 0] = -1  # child points back at the root
    rc, msg = try_upload(cyc)
    assert rc == 4 and "bad tree" in msg
    oob = np.zeros((2, 8), np.int32)
    oob[0, 3] = 5
    rc, msg = try_upload(oob)
    assert rc == 4 and "outside" in msg
    dag = np.zeros((3, 8), np.int32)
    dag[0, 0] = 1
    dag[0, 1] = 1  # two slots share one child
    rc, msg = try_upload(dag)
    assert rc == 4


def test_product_never_touches_the_oracle():
    """oracle/ is test infrastructure: nothing under volrend_amd/ or include/ may import, link
    or even name it, and the library must not fall back to a CPU path (it raises instead)."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    offenders = []
    for top in ("volrend_amd", "include"):
        for d, _, files in os.walk(os.path.join(root, top)):
            for f in files:
                if not f.endswith((".py", ".cpp", ".hip", ".h", ".hpp")):
                    continue
                text = open(os.path.join(d, f), errors="ignore").read()
                if re.search(r"\bimport\s+oracle\b|\bfrom\s+oracle\b|oracle/|vr_oracle|libvr_oracle", text):
                    offenders.append(os.path.join(d, f))
    assert not offenders, offenders
    makefile = open(os.path.join(root, "Makefile")).read()
    lib_rules = makefile.split("oracle:")[0]  # everything before the oracle target
    assert "vr_oracle" not in lib_rules


def test_query_mode_rule_is_arithmetic():
    """Which trees take the integer lookup and which the reference's literal float descent
    (include/volrend_hip.h VR_QUERY_*): N == 2, leaves within 24 levels, fewer than 2^27 nodes
    (32-bit byte offsets into the node words).  Pure host arithmetic -- the 2^27-node boundary is
    checked here without allocating such a tree (a valid tree.npz can reach it:
    n3tree_query.hpp:22-47 descends anything)."""
    L = _abi.lib()
    look, desc = _abi.QUERY_LOOKUP, _abi.QUERY_DESCENT
    assert L.vr_query_mode_for(2, 8, 2_000_000) == look          # lego-class
    assert L.vr_query_mode_for(2, 23, 100) == look               # deepest leaf reads 24 child words
    assert L.vr_query_mode_for(2, 24, 100) == desc
    assert L.vr_query_mode_for(2, 29, 59) == desc
    assert L.vr_query_mode_for(2, 9, (1 << 27) - 1) == look
    assert L.vr_query_mode_for(2, 9, 1 << 27) == desc            # node * 8 + slot words * 4 bytes >= 2^32
    assert L.vr_query_mode_for(2, 9, 1 << 40) == desc
    assert L.vr_query_mode_for(3, 3, 100) == desc and L.vr_query_mode_for(4, 2, 10) == desc


def test_gather_library_exports_its_header():
    """include/volrend_gather.h <-> libvolrend_gather.so <-> volrend_amd/gather.py: the tile
    shard's RCCL collective behind a C ABI of its own (no compute calls here: symbols only)."""
    from volrend_amd import gather
    hdr = open(os.path.join(ROOT, "include", "volrend_gather.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    funcs = sorted(set(re.findall(r"^\s*(?:const char\*|int)\s+(vr_gather_\w+)\s*\(", hdr, flags=re.M)))
    assert len(funcs) >= 11
    if not os.path.exists(gather.LIB_PATH):
        subprocess.check_call(["make", "-C", ROOT, "gather"], stdout=subprocess.DEVNULL)
    out = subprocess.check_output(["nm", "-D", "--defined-only", gather.LIB_PATH], text=True)
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    assert sorted(s for s in exported if s.startswith("vr_")) == funcs
    assert sorted(gather.PROTOTYPES) == funcs
    L = gather.lib()  # resolves every symbol
    assert L.vr_gather_version() > 20000 and gather.ID_BYTES == 128
    # the library that renders stays free of RCCL (single-GPU users need none)
    needed = subprocess.check_output(["readelf", "-d", _abi.LIB_PATH], text=True)
    assert "rccl" not in needed
    assert "rccl" in subprocess.check_output(["readelf", "-d", gather.LIB_PATH], text=True)


def test_product_sources_carry_no_experiment_hooks_and_the_patch_applies(tmp_path):
    """Measurement builds (timing ablations, shader-clock timelines, knob overrides) are GENERATED:
    volrend_amd/build.py copies csrc/ and applies tools/experiments/kernel_hooks.patch.  The product
    sources themselves must not name a hook, the patch must apply to them as they are, and the
    product build must refuse the hooks' flags."""
    csrc = os.path.join(ROOT, "volrend_amd", "csrc")
    for f in ("vr_kernels.hip", "vr_api.cpp", "vr_internal.h", "vr_device_math.h"):
        text = open(os.path.join(csrc, f)).read()
        assert not re.search(r"VR_EXP_|\bTL3?_[A-Z]|VR_ABLATE|VR_TIMELINE|vr_experiment_hooks", text), f
    assert not os.path.exists(os.path.join(csrc, "vr_experiment_hooks.h"))
    build.hooked_sources(str(tmp_path), dry_run=True)      # raises if a hunk does not apply
    build.hooked_sources(str(tmp_path))
    patched = open(os.path.join(str(tmp_path), "vr_kernels.hip")).read()
    assert "VR_EXP_BRICK_WORD(" in patched and "TL3_ROUND(go)" in patched and "vr_experiment_hooks.h" in patched
    with pytest.raises(ValueError):
        build.build(extra_flags=["-DVR_ABLATE=4"])          # never into libvolrend_hip.so
