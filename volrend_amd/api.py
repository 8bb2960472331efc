"""Host-side mirror of the reference's renderer interface, over the C ABI.

Same names, argument meaning and defaults as the reference for this path:

* ``RenderOptions``   -- ``volrend::RenderOptions`` (include/volrend/render_options.hpp:11-53)
* ``Camera``          -- ``volrend::Camera`` pose/intrinsics part (include/volrend/camera.hpp:14-70)
* ``N3Tree``          -- ``volrend::N3Tree`` (include/volrend/n3tree.hpp:24-105; loader
  semantics of src/n3tree.cpp:111-362 incl. the quantised variant)
* ``launch_renderer`` -- ``volrend::launch_renderer`` (include/volrend/cuda/renderer_kernel.hpp:9-12)

Device memory and streams are the caller's (torch tensors / ``torch.cuda`` streams
are fine: anything with ``data_ptr()`` / ``cuda_stream``); the rendering itself is
always the HIP library -- there is no eager / CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field

import numpy as np

from . import _abi

CAMERA_DEFAULT_FOCAL_LENGTH = 1111.11  # camera.hpp:12
VOLREND_GLOBAL_BASIS_MAX = 25  # render_options.hpp:6


@dataclass
class RenderOptions:
    step_size: float = 1e-4
    sigma_thresh: float = 1e-2
    stop_thresh: float = 1e-2
    background_brightness: float = 1.0
    render_bbox: tuple = (0.0, 0.0, 0.0, 1.0, 1.0, 1.0)
    basis_minmax: tuple = (0, VOLREND_GLOBAL_BASIS_MAX - 1)
    rot_dirs: tuple = (0.0, 0.0, 0.0)
    show_grid: bool = False
    grid_max_depth: int = 4
    render_depth: bool = False
    enable_probe: bool = False
    probe: tuple = (0.0, 0.0, 1.0)
    probe_disp_size: int = 100

    def to_c(self) -> _abi.VrRenderOptions:
        o = _abi.VrRenderOptions()
        o.step_size = self.step_size
        o.sigma_thresh = self.sigma_thresh
        o.stop_thresh = self.stop_thresh
        o.background_brightness = self.background_brightness
        for i in range(6):
            o.render_bbox[i] = self.render_bbox[i]
        o.basis_minmax[0], o.basis_minmax[1] = self.basis_minmax
        for i in range(3):
            o.rot_dirs[i] = self.rot_dirs[i]
            o.probe[i] = self.probe[i]
        o.show_grid = int(self.show_grid)
        o.grid_max_depth = self.grid_max_depth
        o.render_depth = int(self.render_depth)
        o.enable_probe = int(self.enable_probe)
        o.probe_disp_size = self.probe_disp_size
        return o


class Camera:
    """Pose + intrinsics.  ``transform`` is the 4x3 column-major camera-to-world
    (right, up, back, centre) exactly as ``CameraSpec::transform``."""

    def __init__(self, width: int = 256, height: int = 256,
                 fx: float = CAMERA_DEFAULT_FOCAL_LENGTH, fy: float = -1.0):
        self.width, self.height = int(width), int(height)
        self.fx = float(CAMERA_DEFAULT_FOCAL_LENGTH if fx < 0 else fx)
        self.fy = float(self.fx if fy < 0 else fy)
        self.transform = np.zeros(12, dtype=np.float32)
        # default pose of camera.cpp:32-36
        self.v_back = np.array([-0.7071068, 0.0, 0.7071068], dtype=np.float32)
        self.v_world_up = np.array([0.0, 0.0, 1.0], dtype=np.float32)
        self.center = np.array([-3.55, 0.0, 3.55], dtype=np.float32)
        self._update()

    def _update(self, transform_from_vecs: bool = True) -> None:
        """camera.cpp:47-58 (the K / w2c matrices serve the mesh rasteriser only)."""
        if transform_from_vecs:
            b = self.v_back / np.linalg.norm(self.v_back)
            r = np.cross(self.v_world_up, b)
            r = r / np.linalg.norm(r)
            u = np.cross(b, r)
            self.transform = np.concatenate([r, u, b, self.center]).astype(np.float32)

    def set_c2w(self, c2w) -> None:
        """4x4 / 3x4 row-major camera-to-world, as read from a pose .txt
        (main_headless.cpp:40-63)."""
        m = np.asarray(c2w, dtype=np.float32)[:3, :4]
        self.transform = np.ascontiguousarray(m.T).reshape(12)

    def to_c(self) -> _abi.VrCamera:
        c = _abi.VrCamera()
        for i in range(12):
            c.transform[i] = float(self.transform[i])
        c.width, c.height, c.fx, c.fy = self.width, self.height, self.fx, self.fy
        return c


def parse_data_format(s: str):
    """``DataFormat::parse`` (src/n3tree.cpp:55-78) -> (format name, basis_dim)."""
    idx = next((i for i, ch in enumerate(s) if not ch.isalpha()), -1)
    if idx < 0:
        return "RGBA", -1
    head = s[:idx]
    try:
        dim = int("".join(ch for ch in s[idx:] if ch.isdigit() or ch == "-") or "0")
    except ValueError:
        dim = 0
    return (head if head in ("ASG", "SG", "SH") else "RGBA"), dim


class N3Tree:
    """Read-only N^3 tree: host arrays + the device copy behind an opaque handle."""

    def __init__(self, path: str | None = None, upload: bool = True):
        self.N = 0
        self.data_dim = 0
        self.data_format = ("RGBA", -1)
        self.capacity = 0
        self.scale = np.zeros(3, np.float32)
        self.offset = np.zeros(3, np.float32)
        self.use_ndc = False
        self.ndc_width = self.ndc_height = self.ndc_focal = 0.0
        self.child_ = None
        self.data_ = None
        self.extra_ = None
        self.quant_ = None  # codebook arrays of a quantised file awaiting the device decode
        self._handle = C.c_void_p()
        self._loaded = False
        if path is not None:
            self.open(path, upload=upload)

    # ---- construction ----------------------------------------------------
    @classmethod
    def from_arrays(cls, child, data, offset, invradius3, data_format: str, extra=None,
                    ndc=None, upload: bool = True) -> "N3Tree":
        t = cls()
        t._set_arrays(child, data, offset, invradius3, data_format, extra)
        if ndc:
            t.use_ndc = True
            t.ndc_width, t.ndc_height, t.ndc_focal = map(float, ndc)
        if upload:
            t.load_device()
        return t

    @classmethod
    def from_synth(cls, tree, upload: bool = True, ndc=None) -> "N3Tree":
        return cls.from_arrays(tree.child, tree.data, tree.offset, tree.invradius3,
                               tree.data_format, tree.extra, ndc=ndc, upload=upload)

    def _set_arrays(self, child, data, offset, invradius3, data_format, extra, data_dim=None):
        child = np.ascontiguousarray(child, dtype=np.int32)
        if child.ndim != 4 or not (child.shape[1] == child.shape[2] == child.shape[3]):
            raise RuntimeError("child must be int32 [capacity, N, N, N]")
        self.N = int(child.shape[1])
        self.capacity = int(child.shape[0])
        self.child_ = child.reshape(self.capacity, self.N, self.N, self.N)
        self.quant_ = None
        if data is None:  # quantised file, decoded on the device at upload
            self.data_ = None
            self.data_dim = int(data_dim)
        else:
            data = np.asarray(data)
            if data.dtype != np.float16:
                raise RuntimeError("data must be stored in half precision")  # n3tree.cpp:344-346
            self.data_ = np.ascontiguousarray(data)
            self.data_dim = int(self.data_.shape[-1])
            if self.data_.size != self.capacity * self.N ** 3 * self.data_dim:
                raise RuntimeError("data does not have capacity * N^3 * data_dim values")
        self.data_format = parse_data_format(data_format)
        self.scale = np.asarray(invradius3, dtype=np.float32).reshape(3).copy()
        self.offset = np.asarray(offset, dtype=np.float32).reshape(3).copy()
        self.extra_ = None if extra is None else np.ascontiguousarray(extra, dtype=np.float32)

    def open(self, path: str, upload: bool = True, device_decode: bool = True) -> None:
        """``N3Tree::open`` + ``load_npz`` (src/n3tree.cpp:111-154, 228-362).
        ``upload=False`` stops before ``load_cuda`` (host-only use, e.g. format tests).
        A quantised file is decoded on the device during the upload (``data_`` stays None
        until ``decode_host()``) unless ``device_decode=False`` or ``upload=False``."""
        if not path.endswith(".npz"):
            raise ValueError("tree file must end in .npz")  # assert at n3tree.cpp:119
        if not os.path.exists(path):
            raise FileNotFoundError(f"Can't load because file does not exist: {path}")
        z = np.load(path)
        data_dim = int(z["data_dim"])
        if "data_format" in z.files:
            fmt = str(z["data_format"])
        else:  # legacy files, n3tree.cpp:240-254
            fmt = "RGBA" if data_dim == 4 else f"SH{(data_dim - 1) // 3}"
        if "invradius3" in z.files:
            scale = z["invradius3"].astype(np.float32)
        else:
            scale = np.full(3, float(z["invradius"]), dtype=np.float32)
        child = z["child"]
        extra = z["extra_data"] if "extra_data" in z.files else None
        if "quant_colors" in z.files and upload and device_decode:
            self._set_arrays(child, None, z["offset"], scale, fmt, extra, data_dim=data_dim)
            self.quant_ = _quant_arrays(z, child)
        else:
            if "quant_colors" in z.files:
                data = _decode_quantised(z, child, data_dim)
            else:
                data = z["data"]
            self._set_arrays(child, data, z["offset"], scale, fmt, extra)
            if self.data_dim != data_dim:
                raise RuntimeError("data_dim does not match the data array")
        # LLFF NDC sidecar, n3tree.cpp:121,131-148
        pb = path[:-4] + "_poses_bounds.npy"
        if os.path.exists(pb):
            arr = np.load(pb).reshape(-1)
            self.use_ndc = True
            self.ndc_height, self.ndc_width, self.ndc_focal = float(arr[4]), float(arr[9]), float(
                arr[14])
        if upload:
            self.load_device()

    # ---- device ----------------------------------------------------------
    def load_device(self) -> None:
        """``N3Tree::load_cuda`` (src/cuda/n3tree.cu:9-41)."""
        L = _abi.lib()
        self.free_device()
        d = _abi.VrTreeDesc()
        L.vr_default_tree_desc(C.byref(d))
        d.child = self.child_.ctypes.data
        if self.data_ is not None:
            d.data = self.data_.ctypes.data
        elif self.quant_ is None:
            raise RuntimeError("tree has no data (clear_cpu_memory was called)")
        if self.extra_ is not None:
            d.extra = self.extra_.ctypes.data
            d.extra_count = self.extra_.size
        for i in range(3):
            d.offset[i] = float(self.offset[i])
            d.scale[i] = float(self.scale[i])
        d.N = self.N
        d.capacity = self.capacity
        d.data_dim = self.data_dim
        d.format = _abi.FORMATS[self.data_format[0]]
        d.basis_dim = self.data_format[1]
        d.ndc_width = self.ndc_width if self.use_ndc else -1.0
        d.ndc_height = self.ndc_height
        d.ndc_focal = self.ndc_focal
        d.memory = 0
        h = C.c_void_p()
        if self.data_ is None:
            _abi.check(L.vr_tree_upload_quantized(C.byref(d), C.byref(self._quant_desc()),
                                                  C.byref(h)))
        else:
            _abi.check(L.vr_tree_upload(C.byref(d), C.byref(h)))
        self._handle = h
        self._loaded = True

    def _quant_desc(self) -> "_abi.VrQuantDesc":
        q = _abi.VrQuantDesc()
        a = self.quant_
        q.n_quant = a["quant_map"].shape[0]
        q.quant_colors = a["quant_colors"].ctypes.data
        q.quant_map = a["quant_map"].ctypes.data
        q.sigma = a["sigma"].ctypes.data
        if a["data_retained"] is not None:
            q.n_retained = a["data_retained"].shape[0]
            q.data_retained = a["data_retained"].ctypes.data
        return q

    def decode_host(self, on_device: bool = False) -> np.ndarray:
        """Materialises ``data_`` of a quantised tree: numpy restatement of the reference loop
        (src/n3tree.cpp:310-339), or the library's device decode copied back."""
        if self.data_ is None and self.quant_ is not None:
            shape = (self.capacity, self.N, self.N, self.N, self.data_dim)
            if on_device:
                d = _abi.VrTreeDesc()
                _abi.lib().vr_default_tree_desc(C.byref(d))
                d.N, d.capacity, d.data_dim, d.memory = self.N, self.capacity, self.data_dim, 0
                out = np.empty(shape, np.float16)
                _abi.check(_abi.lib().vr_decode_quantized(C.byref(d), C.byref(self._quant_desc()),
                                                          out.ctypes.data))
                self.data_ = out
            else:
                a = self.quant_
                self.data_ = _decode_arrays(a["quant_colors"], a["quant_map"], a["sigma"],
                                            a["data_retained"], self.data_dim).reshape(shape)
        return self.data_

    def clone_to(self, device: int) -> "N3Tree":
        """A replica of the DEVICE copy on another (or the same) device of this process
        (vr_tree_clone: device-to-device, no second upload / re-layout).  The replica shares the
        host-side metadata and owns its device copy."""
        t = N3Tree()
        for k in ("N", "data_dim", "data_format", "capacity", "scale", "offset", "use_ndc",
                  "ndc_width", "ndc_height", "ndc_focal", "child_"):
            setattr(t, k, getattr(self, k))
        h = C.c_void_p()
        _abi.check(_abi.lib().vr_tree_clone(self.handle, int(device), C.byref(h)))
        t._handle = h
        t._loaded = True
        return t

    def free_device(self) -> None:
        if self._handle:
            _abi.lib().vr_tree_free(self._handle)
            self._handle = C.c_void_p()
        self._loaded = False

    def is_device_loaded(self) -> bool:  # is_cuda_loaded
        return self._loaded

    def clear_cpu_memory(self) -> None:
        """n3tree.cpp:441-447: keeps ``child_`` (wireframes), drops ``data_``."""
        self.data_ = None
        self.quant_ = None

    def sched_stats(self, reset: bool = True) -> dict:
        """Scheduling tallies of instrumented launches (see vr_sched_stats)."""
        out = (C.c_uint64 * 8)()
        _abi.check(_abi.lib().vr_sched_stats(self._handle, C.byref(out), 1 if reset else 0))
        names = ("march_rounds", "march_lanes", "shade_rounds", "shade_lanes", "distinct_leaves",
                 "retire_rounds", "retired", "iterations")
        return dict(zip(names, [int(v) for v in out]))

    def touch_enable(self, enable: bool = True) -> None:
        """Distinct-line meter of instrumented launches on / off (vr_touch_enable)."""
        _abi.check(_abi.lib().vr_touch_enable(self.handle, 1 if enable else 0))

    def touch_count(self, reset: bool = True) -> dict:
        """Distinct 128-byte lines touched since the last reset (vr_touch_count), per array."""
        out = (C.c_uint64 * 4)()
        _abi.check(_abi.lib().vr_touch_count(self.handle, C.byref(out), 1 if reset else 0))
        return dict(zip(("leaves", "nodes", "top", "bricks"), [int(v) for v in out]))

    def touch_read(self, which: int):
        """The distinct-line bitmap of array ``which`` (0 records, 1 child words, 2 top grid,
        3 bricks) as (uint32 numpy array, bytes per bit) -- vr_touch_read."""
        words, gran = C.c_uint64(0), C.c_uint64(0)
        _abi.check(_abi.lib().vr_touch_read(self.handle, int(which), None, 0, C.byref(words), C.byref(gran)))
        out = np.zeros(int(words.value), dtype=np.uint32)
        if out.size:
            _abi.check(_abi.lib().vr_touch_read(self.handle, int(which), out.ctypes.data, out.size, None, None))
        return out, int(gran.value)

    def reserve(self, width: int, height: int, n_frames: int, shard: "TileShard | None" = None,
                n_slots: int = 2) -> None:
        """Pre-allocate the per-launch ray buffers (vr_reserve / vr_reserve_tiles): no later
        launch of that size blocks or allocates."""
        if shard is None and n_slots == 2:
            _abi.check(_abi.lib().vr_reserve(self.handle, int(width), int(height), int(n_frames)))
        else:
            tw, th, world = (shard.tile_w, shard.tile_h, shard.world) if shard else (0, 0, 1)
            _abi.check(_abi.lib().vr_reserve_tiles(self.handle, int(width), int(height),
                                                   int(n_frames), tw, th, world, int(n_slots)))

    def set_tuning(self, **kw) -> None:
        """Scheduling knobs of THIS tree (vr_tree_set_tuning): march_max, refill_min,
        waves_per_cu, records_nt, ...  Results never depend on them."""
        for k, v in kw.items():
            _abi.check(_abi.lib().vr_tree_set_tuning(self.handle, k.encode(), int(v)))

    def status(self, reset: bool = False, stream=None) -> int:
        """Sticky device status word (vr_tree_status): bit 0 = a ray hit the sample guard.
        With ``stream``: read ON that stream and wait for it alone (vr_tree_status_on) -- other
        streams that render this tree are not waited for."""
        out = C.c_uint32(0)
        if stream is not None:
            _abi.check(_abi.lib().vr_tree_status_on(self.handle, C.byref(out), 1 if reset else 0,
                                                    _stream_ptr(stream)))
        else:
            _abi.check(_abi.lib().vr_tree_status(self.handle, C.byref(out), 1 if reset else 0))
        return int(out.value)

    def info(self) -> dict:
        i = _abi.VrTreeInfo()
        _abi.check(_abi.lib().vr_tree_info(self._handle, C.byref(i)))
        return {n: int(getattr(i, n)) for n, _ in i._fields_}

    @property
    def handle(self):
        if not self._loaded:
            raise RuntimeError("tree is not on the device (call load_device)")
        return self._handle

    def __del__(self):
        try:
            self.free_device()
        except Exception:
            pass


def _quant_arrays(z, child) -> dict:
    """The codebook members of a quantised tree.npz (scripts/compress_octree.py:106-119),
    checked as src/n3tree.cpp:279-293 does and flattened to [.., n_slots, ..]."""
    qc = z["quant_colors"]
    if qc.dtype != np.float16:
        raise RuntimeError("codebook must be stored in half precision")
    qm = z["quant_map"]
    n_q = qm.shape[0]
    if qc.shape[0] != n_q:
        raise RuntimeError("codebook and map basis numbers does not match")
    cap, N = qm.shape[1], child.shape[1]
    if cap != child.shape[0] or qc.shape[1:] != (65536, 3):
        raise RuntimeError("quantised arrays do not match the tree")
    n_slots = cap * N * N * N
    retained = z["data_retained"] if "data_retained" in z.files else None
    return dict(
        quant_colors=np.ascontiguousarray(qc),
        quant_map=np.ascontiguousarray(qm.reshape(n_q, n_slots), dtype=np.uint16),
        sigma=np.ascontiguousarray(z["sigma"].reshape(n_slots), dtype=np.float16),
        data_retained=None if retained is None else np.ascontiguousarray(
            retained.reshape(retained.shape[0], n_slots, 3), dtype=np.float16))


def _decode_arrays(qc, qm, sigma, retained, data_dim: int) -> np.ndarray:
    """Median-cut codebook decode, src/n3tree.cpp:279-340.

    data[slot, j + n_retain + c*n_basis] = quant_colors[j, quant_map[j, slot], c]
    data[slot, j + c*n_basis]            = data_retained[j, slot, c]
    data[slot, data_dim-1]               = sigma[slot]
    """
    n_q, n_slots = qm.shape
    n_ret = 0 if retained is None else retained.shape[0]
    n_basis = n_q + n_ret
    data = np.zeros((n_slots, data_dim), dtype=np.float16)
    for j in range(n_q):
        cols = qc[j][qm[j].astype(np.int64)]  # [n_slots, 3]
        for c in range(3):
            data[:, j + n_ret + c * n_basis] = cols[:, c]
    data[:, data_dim - 1] = sigma
    for j in range(n_ret):
        for c in range(3):
            data[:, j + c * n_basis] = retained[j, :, c]
    return data


def _decode_quantised(z, child, data_dim: int) -> np.ndarray:
    a = _quant_arrays(z, child)
    cap, N = child.shape[0], child.shape[1]
    return _decode_arrays(a["quant_colors"], a["quant_map"], a["sigma"], a["data_retained"],
                          data_dim).reshape(cap, N, N, N, data_dim)


def _ptr(x) -> int | None:
    if x is None:
        return None
    if isinstance(x, int):
        return x
    if hasattr(x, "data_ptr"):
        return int(x.data_ptr())
    raise TypeError(f"expected a device tensor or raw pointer, got {type(x)}")


def _stream_ptr(stream) -> int | None:
    if stream is None:
        return None
    if isinstance(stream, int):
        return stream
    return int(stream.cuda_stream)  # torch.cuda.Stream


@dataclass
class TileShard:
    """Screen-tile shard of one rank (multi-GPU): tiles ``t % world == rank``."""
    tile_w: int = 0
    tile_h: int = 0
    rank: int = 0
    world: int = 1
    compact: bool = False


def launch_renderer(tree: N3Tree, cam: Camera, options: RenderOptions, image, depth=None,
                    stream=None, offscreen: bool = False, *, accum=None, pitch: int = 0,
                    shard: TileShard | None = None, fp_mode: int = _abi.FP_STRICT,
                    counters=None) -> None:
    """Enqueue one frame on ``stream`` (asynchronous, like the reference).

    ``image``: device RGBA8 buffer (``torch.uint8`` [H,W,4] or a raw pointer);
    ``depth``: device R32F mesh depth or None; ``offscreen=True`` is the
    ``volrend_headless`` mode (background_brightness composite).
    """
    f = _abi.VrFrame()
    L = _abi.lib()
    L.vr_default_frame(C.byref(f))
    f.rgba = _ptr(image)
    f.pitch = pitch
    f.depth = _ptr(depth)
    f.accum = _ptr(accum)
    f.offscreen = 1 if offscreen else 0
    f.fp_mode = fp_mode
    f.counters = _ptr(counters)  # device int64[7], zeroed by the caller (instrumentation)
    if shard is not None:
        f.tile_w, f.tile_h, f.rank, f.world = shard.tile_w, shard.tile_h, shard.rank, shard.world
        f.layout = _abi.LAYOUT_COMPACT if shard.compact else _abi.LAYOUT_FRAME
    c = cam.to_c()
    o = options.to_c()
    _abi.check(L.vr_render(tree.handle, C.byref(c), C.byref(o), C.byref(f), _stream_ptr(stream)))


class PreparedBatch:
    """The marshalled arguments of one ``vr_render_batch`` call.  Building them costs ~20 us of
    Python per pose; a render loop that knows its poses up front (``volrend_headless``,
    main_headless.cpp:207-225) builds them once and each ``launch`` is a single C call, so the
    host never leaves the GPU idle between launches."""

    def __init__(self, tree: N3Tree, cam: Camera, transforms, options: RenderOptions, images,
                 offscreen: bool = True, *, accums=None, depths=None, pitch: int = 0,
                 shard: TileShard | None = None, fp_mode: int = _abi.FP_STRICT, counters=None):
        n = len(transforms)
        if n != len(images):
            raise ValueError("one image per pose")
        L = _abi.lib()
        self.tree, self.n = tree, n
        self.cams = (_abi.VrCamera * n)()
        self.frames = (_abi.VrFrame * n)()
        self._keep = (images, accums, depths, counters)  # the buffers must outlive the launch
        for i in range(n):
            cam.transform = np.asarray(transforms[i], dtype=np.float32)
            self.cams[i] = cam.to_c()
            f = self.frames[i]
            L.vr_default_frame(C.byref(f))
            f.rgba = _ptr(images[i])
            f.pitch = pitch
            f.depth = _ptr(depths[i]) if depths else None
            f.accum = _ptr(accums[i]) if accums else None
            f.offscreen = 1 if offscreen else 0
            f.fp_mode = fp_mode
            f.counters = _ptr(counters[i]) if counters else None
            if shard is not None:
                f.tile_w, f.tile_h, f.rank, f.world = (shard.tile_w, shard.tile_h, shard.rank,
                                                       shard.world)
                f.layout = _abi.LAYOUT_COMPACT if shard.compact else _abi.LAYOUT_FRAME
        self.opts = options.to_c()

    def launch(self, stream=None) -> None:
        _abi.check(_abi.lib().vr_render_batch(self.tree.handle, self.n, self.cams,
                                              C.byref(self.opts), self.frames, _stream_ptr(stream)))


def launch_renderer_batch(tree: N3Tree, cam: Camera, transforms, options: RenderOptions, images,
                          stream=None, offscreen: bool = True, *, accums=None, depths=None,
                          pitch: int = 0, shard: TileShard | None = None,
                          fp_mode: int = _abi.FP_STRICT, counters=None) -> None:
    """Several poses in ONE launch: ``transforms[i]`` (12-float c2w) -> ``images[i]``.

    The pose loop of ``volrend_headless`` (main_headless.cpp:207-225) with the
    poses known up front; intrinsics / options / sharding are shared.  ``counters``:
    optional list of device int64[7] tensors (instrumented flavour)."""
    PreparedBatch(tree, cam, transforms, options, images, offscreen, accums=accums, depths=depths,
                  pitch=pitch, shard=shard, fp_mode=fp_mode, counters=counters).launch(stream)


def set_tuning(**kw) -> None:
    """DEFAULT knobs of trees uploaded from now on (vr_set_tuning), incl. the upload-time
    top_levels / brick_levels; an existing tree is changed with ``N3Tree.set_tuning``."""
    for k, v in kw.items():
        _abi.check(_abi.lib().vr_set_tuning(k.encode(), int(v)))


def compact_bytes(width: int, height: int, shard: TileShard) -> int:
    n = _abi.lib().vr_compact_bytes(width, height, shard.tile_w, shard.tile_h, shard.world)
    if n < 0:
        raise _abi.VolrendError(1, (_abi.lib().vr_last_error() or b"").decode())
    return int(n)


def assemble_tiles(frame, gathered, width: int, height: int, shard: TileShard, stream=None,
                   pitch: int = 0) -> None:
    _abi.check(_abi.lib().vr_assemble_tiles(_ptr(frame), pitch, _ptr(gathered), width, height,
                                            shard.tile_w, shard.tile_h, shard.world,
                                            _stream_ptr(stream)))


def assemble_tiles_batch(frames, gathered, n_frames: int, width: int, height: int,
                         shard: TileShard, stream=None) -> None:
    """``frames``: contiguous device [n, H, W, 4] uint8; ``gathered``: contiguous device
    [world, n_alloc, compact_bytes] (rank-major gather result, n_alloc >= n_frames).
    One launch de-interleaves all frames."""
    cb = compact_bytes(width, height, shard)
    n_alloc = int(gathered.shape[1]) if hasattr(gathered, "shape") else n_frames
    _abi.check(_abi.lib().vr_assemble_tiles_batch(
        _ptr(frames), width * height * 4, 0, _ptr(gathered), n_alloc * cb, cb, n_frames, width,
        height, shard.tile_w, shard.tile_h, shard.world, _stream_ptr(stream)))
