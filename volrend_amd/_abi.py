"""ctypes view of the C ABI declared in ``include/volrend_hip.h``.

The product path has no CPU fallback: if ``libvolrend_hip.so`` is missing this
module raises at load time (run ``python -m volrend_amd.build`` or
``__graft_entry__.build()``).
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# VOLREND_HIP_LIB: a profiling / experiment build of the SAME library (python -m volrend_amd.build
# --variant NAME -D...), never something else -- there is no fallback either way.
LIB_PATH = os.environ.get("VOLREND_HIP_LIB") or os.path.join(HERE, "libvolrend_hip.so")

VR_OK = 0
FORMAT_RGBA, FORMAT_SH, FORMAT_SG, FORMAT_ASG = 0, 1, 2, 3
FORMATS = {"RGBA": FORMAT_RGBA, "SH": FORMAT_SH, "SG": FORMAT_SG, "ASG": FORMAT_ASG}
FP_STRICT, FP_FMA = 0, 1
QUERY_LOOKUP, QUERY_DESCENT = 0, 1  # VrTreeInfo.query_mode
LAYOUT_FRAME, LAYOUT_COMPACT = 0, 1
MAX_BASIS = 25


class VrTreeDesc(C.Structure):
    _fields_ = [("child", C.c_void_p), ("data", C.c_void_p), ("extra", C.c_void_p),
                ("extra_count", C.c_uint64), ("offset", C.c_float * 3), ("scale", C.c_float * 3),
                ("N", C.c_int32), ("capacity", C.c_int64), ("data_dim", C.c_int32),
                ("format", C.c_int32), ("basis_dim", C.c_int32), ("ndc_width", C.c_float),
                ("ndc_height", C.c_float), ("ndc_focal", C.c_float), ("memory", C.c_int32)]


class VrQuantDesc(C.Structure):
    _fields_ = [("quant_colors", C.c_void_p), ("quant_map", C.c_void_p), ("sigma", C.c_void_p),
                ("data_retained", C.c_void_p), ("n_quant", C.c_int32), ("n_retained", C.c_int32)]


class VrTreeInfo(C.Structure):
    _fields_ = [("capacity", C.c_int64), ("N", C.c_int32), ("data_dim", C.c_int32),
                ("format", C.c_int32), ("basis_dim", C.c_int32), ("max_depth", C.c_int32),
                ("device", C.c_int32), ("device_bytes", C.c_uint64), ("leaf_stride", C.c_uint64),
                ("query_mode", C.c_int32), ("top_levels", C.c_int32), ("brick_levels", C.c_int32),
                ("brick_blocked", C.c_int32)]


class VrCamera(C.Structure):
    _fields_ = [("transform", C.c_float * 12), ("width", C.c_int32), ("height", C.c_int32),
                ("fx", C.c_float), ("fy", C.c_float)]


class VrRenderOptions(C.Structure):
    _fields_ = [("step_size", C.c_float), ("sigma_thresh", C.c_float), ("stop_thresh", C.c_float),
                ("background_brightness", C.c_float), ("render_bbox", C.c_float * 6),
                ("basis_minmax", C.c_int32 * 2), ("rot_dirs", C.c_float * 3),
                ("show_grid", C.c_int32), ("grid_max_depth", C.c_int32),
                ("render_depth", C.c_int32), ("enable_probe", C.c_int32),
                ("probe", C.c_float * 3), ("probe_disp_size", C.c_int32)]


class VrFrame(C.Structure):
    _fields_ = [("rgba", C.c_void_p), ("pitch", C.c_int64), ("depth", C.c_void_p),
                ("accum", C.c_void_p), ("offscreen", C.c_int32), ("layout", C.c_int32),
                ("tile_w", C.c_int32), ("tile_h", C.c_int32), ("rank", C.c_int32),
                ("world", C.c_int32), ("fp_mode", C.c_int32), ("reserved", C.c_int32),
                ("counters", C.c_void_p)]


ABI_VERSION = 3  # VR_ABI_VERSION of include/volrend_hip.h
MAX_BATCH = 512  # VR_MAX_BATCH

COUNTER_FIELDS = ("rays", "rays_hit_box", "samples", "child_reads", "hit_samples", "alg_bytes",
                  "early_stops")


# name -> (restype, argtypes); also the list tests check against the header
PROTOTYPES = {
    "vr_abi_version": (C.c_int, []),
    "vr_last_error": (C.c_char_p, []),
    "vr_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "vr_set_device": (C.c_int, [C.c_int]),
    "vr_device_name": (C.c_int, [C.c_int, C.c_char_p, C.c_size_t]),
    "vr_default_tree_desc": (None, [C.POINTER(VrTreeDesc)]),
    "vr_tree_upload": (C.c_int, [C.POINTER(VrTreeDesc), C.POINTER(C.c_void_p)]),
    "vr_tree_upload_quantized": (C.c_int, [C.POINTER(VrTreeDesc), C.POINTER(VrQuantDesc),
                                           C.POINTER(C.c_void_p)]),
    "vr_decode_quantized": (C.c_int, [C.POINTER(VrTreeDesc), C.POINTER(VrQuantDesc), C.c_void_p]),
    "vr_tree_clone": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "vr_tree_free": (C.c_int, [C.c_void_p]),
    "vr_tree_info": (C.c_int, [C.c_void_p, C.POINTER(VrTreeInfo)]),
    "vr_query_mode_for": (C.c_int, [C.c_int, C.c_int, C.c_int64]),
    "vr_default_options": (None, [C.POINTER(VrRenderOptions)]),
    "vr_default_frame": (None, [C.POINTER(VrFrame)]),
    "vr_compact_bytes": (C.c_int64, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "vr_render": (C.c_int, [C.c_void_p, C.POINTER(VrCamera), C.POINTER(VrRenderOptions),
                            C.POINTER(VrFrame), C.c_void_p]),
    "vr_render_batch": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(VrCamera),
                                  C.POINTER(VrRenderOptions), C.POINTER(VrFrame), C.c_void_p]),
    "vr_reserve": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "vr_reserve_tiles": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_int]),
    "vr_tree_status": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.c_int]),
    "vr_tree_status_on": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.c_int, C.c_void_p]),
    "vr_set_tuning": (C.c_int, [C.c_char_p, C.c_int]),
    "vr_tree_set_tuning": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "vr_sched_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64 * 8), C.c_int]),
    "vr_touch_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "vr_touch_count": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64 * 4), C.c_int]),
    "vr_touch_read": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64),
                                C.POINTER(C.c_uint64)]),
    "vr_assemble_tiles": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                    C.c_int, C.c_int, C.c_void_p]),
    "vr_assemble_tiles_batch": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64,
                                          C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_int, C.c_void_p]),
    "vr_probe_coeffs": (C.c_int, [C.c_void_p, C.POINTER(VrRenderOptions), C.c_void_p, C.c_void_p]),
    "vr_read_back": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    "vr_stream_sync": (C.c_int, [C.c_void_p]),
}

_lib = None


class VolrendError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"volrend_hip error {code}: {msg}")
        self.code = code


def lib():
    """The loaded library.  Raises if it has not been built -- never falls back."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP library is mandatory (no CPU fallback). "
                "Build it with `python -m volrend_amd.build`.")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(L, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if L.vr_abi_version() != ABI_VERSION:
            raise RuntimeError("libvolrend_hip.so ABI version mismatch")
        _lib = L
    return _lib


def check(rc: int) -> None:
    if rc != VR_OK:
        raise VolrendError(rc, (lib().vr_last_error() or b"").decode("utf-8", "replace"))
