"""ctypes bindings for the CPU ORACLE (``liboracle.so``) and, when present, the
host build of the reference's own device code (``_ref/libvolrend_ref.so``).

TEST INFRASTRUCTURE.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module; the product
(``volrend_amd``) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "liboracle.so")
REF_SO = os.path.join(HERE, "_ref", "libvolrend_ref.so")
REF_LIBM_SO = os.path.join(HERE, "_ref", "libvolrend_ref_libm.so")
RTFRAG_GLSL = os.path.join(HERE, "_ref", "rt_frag_es300.glsl")
REFERENCE_ROOT = "/root/reference"

FP_STRICT, FP_FMA = 0, 1
FORMATS = {"RGBA": 0, "SH": 1, "SG": 2, "ASG": 3}


class OrTree(C.Structure):
    _fields_ = [("child", C.c_void_p), ("data", C.c_void_p), ("extra", C.c_void_p),
                ("offset", C.c_float * 3), ("scale", C.c_float * 3), ("N", C.c_int32),
                ("data_dim", C.c_int32), ("format", C.c_int32), ("basis_dim", C.c_int32),
                ("ndc_width", C.c_float), ("ndc_height", C.c_float), ("ndc_focal", C.c_float)]


class OrCamera(C.Structure):
    _fields_ = [("transform", C.c_float * 12), ("width", C.c_int32), ("height", C.c_int32),
                ("fx", C.c_float), ("fy", C.c_float)]


class OrOptions(C.Structure):
    _fields_ = [("step_size", C.c_float), ("sigma_thresh", C.c_float), ("stop_thresh", C.c_float),
                ("background_brightness", C.c_float), ("render_bbox", C.c_float * 6),
                ("basis_minmax", C.c_int32 * 2), ("rot_dirs", C.c_float * 3),
                ("show_grid", C.c_int32), ("grid_max_depth", C.c_int32),
                ("render_depth", C.c_int32), ("enable_probe", C.c_int32),
                ("probe", C.c_float * 3), ("probe_disp_size", C.c_int32)]


class OrCounters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("rays", "rays_hit_box", "samples", "child_reads",
                                          "hit_samples", "alg_bytes", "early_stops")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


def build_oracle(force: bool = False) -> str:
    """Compile the C restatement (gcc; a few hundred ms)."""
    srcs = [os.path.join(HERE, f) for f in ("vr_oracle.c", "vr_oracle_core.inc", "vr_oracle.h",
                                            "vr_detmath.h")]
    if force or not os.path.exists(ORACLE_SO) or any(
            os.path.getmtime(s) > os.path.getmtime(ORACLE_SO) for s in srcs):
        subprocess.check_call(["make", "-C", HERE, "-B", "liboracle.so"],
                              stdout=subprocess.DEVNULL)
    return ORACLE_SO


def build_ref(force: bool = False) -> str | None:
    """Compile the reference's device code for the host -- only where the
    read-only reference mount exists (never on the GPU box)."""
    if not os.path.isdir(REFERENCE_ROOT):
        return REF_SO if os.path.exists(REF_SO) else None
    if force or not os.path.exists(REF_SO) or not os.path.exists(REF_LIBM_SO) or \
            not os.path.exists(RTFRAG_GLSL):
        subprocess.check_call(["make", "-C", os.path.join(HERE, "ref_build"), "-B"],
                              stdout=subprocess.DEVNULL)
    return REF_SO


_oracle = None


def lib():
    global _oracle
    if _oracle is None:
        build_oracle()
        L = C.CDLL(ORACLE_SO)
        L.or_render.restype = C.c_int
        L.or_render.argtypes = [C.POINTER(OrTree), C.POINTER(OrCamera), C.POINTER(OrOptions),
                                C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.POINTER(OrCounters), C.c_int]
        L.or_default_options.argtypes = [C.POINTER(OrOptions)]
        L.or_probe_coeffs.argtypes = [C.POINTER(OrTree), C.POINTER(OrOptions), C.c_void_p]
        L.or_query.restype = C.c_int64
        L.or_query.argtypes = [C.POINTER(OrTree), C.POINTER(C.c_float * 3), C.POINTER(C.c_float),
                               C.POINTER(C.c_int)]
        L.or_expf.restype = C.c_float
        L.or_expf.argtypes = [C.c_float]
        L.or_half2float.restype = C.c_float
        L.or_half2float.argtypes = [C.c_uint16]
        L.or_basis.argtypes = [C.POINTER(OrTree), C.POINTER(C.c_float * 3), C.c_int,
                               C.POINTER(C.c_float * 25)]
        _oracle = L
    return _oracle


_refs = {}


def ref_lib(libm_expf: bool = False):
    """The reference's own code built for the host, or None if unavailable."""
    key = bool(libm_expf)
    if key not in _refs:
        build_ref()
        path = REF_LIBM_SO if libm_expf else REF_SO
        if not os.path.exists(path):
            _refs[key] = None
        else:
            L = C.CDLL(path)
            L.ref_render.restype = C.c_int
            L.ref_render.argtypes = [C.POINTER(OrTree), C.POINTER(OrCamera), C.POINTER(OrOptions),
                                     C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_int]
            L.ref_trace.restype = C.c_int
            L.ref_trace.argtypes = [C.POINTER(OrTree), C.POINTER(OrCamera), C.POINTER(OrOptions),
                                    C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
            L.ref_probe_coeffs.argtypes = [C.POINTER(OrTree), C.POINTER(OrOptions), C.c_void_p]
            L.ref_query.restype = C.c_int64
            L.ref_query.argtypes = [C.POINTER(OrTree), C.POINTER(C.c_float * 3),
                                    C.POINTER(C.c_float)]
            L.ref_basis.argtypes = [C.POINTER(OrTree), C.POINTER(C.c_float * 3),
                                    C.POINTER(C.c_float * 25)]
            _refs[key] = L
    return _refs[key]


class TreeHandle:
    """Keeps the numpy buffers alive next to the C struct that points at them."""

    def __init__(self, tree, ndc=None):
        self.child = np.ascontiguousarray(tree.child, dtype=np.int32)
        self.data = np.ascontiguousarray(tree.data, dtype=np.float16)
        self.extra = None if tree.extra is None else np.ascontiguousarray(tree.extra, np.float32)
        s = OrTree()
        s.child = self.child.ctypes.data
        s.data = self.data.ctypes.data
        s.extra = self.extra.ctypes.data if self.extra is not None else None
        for i in range(3):
            s.offset[i] = float(tree.offset[i])
            s.scale[i] = float(tree.invradius3[i])
        s.N = tree.N
        s.data_dim = tree.data_dim
        s.format = FORMATS[tree.format_name]
        s.basis_dim = tree.basis_dim
        if ndc:
            s.ndc_width, s.ndc_height, s.ndc_focal = ndc
        else:
            s.ndc_width, s.ndc_height, s.ndc_focal = -1.0, 0.0, 0.0
        self.struct = s
        self.tree = tree


def make_camera(transform12, width, height, fx, fy=None) -> OrCamera:
    c = OrCamera()
    for i in range(12):
        c.transform[i] = float(transform12[i])
    c.width, c.height = int(width), int(height)
    c.fx = float(fx)
    c.fy = float(fx if fy is None else fy)
    return c


def default_options(**kw) -> OrOptions:
    o = OrOptions()
    lib().or_default_options(C.byref(o))
    for k, v in kw.items():
        cur = getattr(o, k)
        if hasattr(cur, "__len__"):
            for i, x in enumerate(v):
                cur[i] = x
        else:
            setattr(o, k, v)
    return o


def render(th: TreeHandle, cam: OrCamera, opt: OrOptions, fp_mode=FP_STRICT, region=None,
           offscreen=True, rgba_init=None, depth_init=None, want_accum=True, nthreads=None):
    """-> (rgba uint8 [H,W,4], accum float32 [H,W,4] or None, counters dict)."""
    W, H = cam.width, cam.height
    x0, y0, w, h = region if region else (0, 0, W, H)
    rgba = np.zeros((H, W, 4), dtype=np.uint8)
    accum = np.zeros((H, W, 4), dtype=np.float32) if want_accum else None
    cnt = OrCounters()
    probe = None
    if opt.enable_probe:
        probe = np.zeros(max(th.struct.data_dim - 1, 1), dtype=np.float32)
        lib().or_probe_coeffs(C.byref(th.struct), C.byref(opt), probe.ctypes.data)
    if nthreads is None:
        nthreads = os.cpu_count() or 1
    rc = lib().or_render(C.byref(th.struct), C.byref(cam), C.byref(opt), fp_mode,
                         1 if offscreen else 0, x0, y0, w, h, rgba.ctypes.data,
                         accum.ctypes.data if accum is not None else None,
                         rgba_init.ctypes.data if rgba_init is not None else None,
                         depth_init.ctypes.data if depth_init is not None else None,
                         probe.ctypes.data if probe is not None else None, C.byref(cnt),
                         int(nthreads))
    if rc != 0:
        raise RuntimeError(f"or_render failed rc={rc}")
    return rgba, accum, cnt.as_dict()


def render_maps(th: TreeHandle, cam: OrCamera, opt: OrOptions, fp_mode=FP_STRICT, nthreads=None):
    """Per-pixel work of one frame: (samples uint32 [H,W], hit samples uint32 [H,W], counters)."""
    W, H = cam.width, cam.height
    samples = np.zeros((H, W), dtype=np.uint32)
    hits = np.zeros((H, W), dtype=np.uint32)
    cnt = OrCounters()
    L = lib()
    L.or_render_maps.restype = C.c_int
    L.or_render_maps.argtypes = [C.POINTER(OrTree), C.POINTER(OrCamera), C.POINTER(OrOptions),
                                 C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.POINTER(OrCounters), C.c_int, C.c_void_p, C.c_void_p]
    rc = L.or_render_maps(C.byref(th.struct), C.byref(cam), C.byref(opt), fp_mode, 1, 0, 0, W, H,
                          None, None, None, None, None, C.byref(cnt),
                          int(nthreads or os.cpu_count() or 1), samples.ctypes.data,
                          hits.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"or_render_maps failed rc={rc}")
    return samples, hits, cnt.as_dict()


def ref_render(th: TreeHandle, cam: OrCamera, opt: OrOptions, region=None, offscreen=True,
               rgba_init=None, depth_init=None, libm_expf=False, nthreads=None):
    """Run the reference's render_kernel on the host. -> rgba uint8 [H,W,4]."""
    L = ref_lib(libm_expf)
    if L is None:
        raise RuntimeError("reference host build unavailable")
    W, H = cam.width, cam.height
    x0, y0, w, h = region if region else (0, 0, W, H)
    rgba = np.zeros((H, W, 4), dtype=np.uint8) if rgba_init is None else rgba_init.copy()
    probe = None
    if opt.enable_probe:
        probe = np.zeros(max(th.struct.data_dim - 1, 1), dtype=np.float32)
        L.ref_probe_coeffs(C.byref(th.struct), C.byref(opt), probe.ctypes.data)
    if nthreads is None:
        nthreads = os.cpu_count() or 1
    L.ref_render(C.byref(th.struct), C.byref(cam), C.byref(opt), 1 if offscreen else 0, x0, y0,
                 w, h, rgba.ctypes.data, depth_init.ctypes.data if depth_init is not None else None,
                 probe.ctypes.data if probe is not None else None, int(nthreads))
    return rgba


def ref_trace(th: TreeHandle, cam: OrCamera, opt: OrOptions, region=None, libm_expf=False):
    """fp32 accumulators of the reference's trace_ray. -> float32 [H,W,4]."""
    L = ref_lib(libm_expf)
    if L is None:
        raise RuntimeError("reference host build unavailable")
    W, H = cam.width, cam.height
    x0, y0, w, h = region if region else (0, 0, W, H)
    accum = np.zeros((H, W, 4), dtype=np.float32)
    L.ref_trace(C.byref(th.struct), C.byref(cam), C.byref(opt), x0, y0, w, h, accum.ctypes.data)
    return accum
