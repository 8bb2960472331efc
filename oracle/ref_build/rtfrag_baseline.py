#!/usr/bin/env python3
"""BASELINE config C0: the reference's GLSL backend (shaders/rt.frag, the no-CUDA
`shader_renderer` path) on a SOFTWARE GL rasteriser, timed on host cores.

    python oracle/ref_build/rtfrag_baseline.py [--size 400] [--config C0] [--out profiles/...json]

TEST / BASELINE INFRASTRUCTURE.  The fragment shader is the reference's own: read from
/root/reference/shaders/rt.frag where that mount exists, otherwise from the git-ignored build
artefact oracle/_ref/rt_frag_es300.glsl that oracle/ref_build/Makefile generates from it (so the
baseline can also be timed on the GPU box's host cores); never part of this repo's history.
It is compiled with the "#version 300 es" prefix the reference itself uses for its WebGL build
(include/volrend/internal/shader.hpp:42-46).  The harness mirrors what
src/shader_renderer.cpp does around it: tree packed into an R16F and an R32I 2-D texture
(:263-342), uniforms (:344-368, :178-190), one full-screen triangle strip (:206-207).

Rasteriser: Mesa llvmpipe through the system libEGL (surfaceless platform) when that creates a
GLES 3 context; otherwise SwiftShader (GLES 3.0 over EGL pbuffers) as bundled with the
`kaleido` wheel -- the only software GL that creates a context in these containers (no X
server / Mesa EGL / OSMesa).  The result names the rasteriser that ran.  Reports ms per frame, Mrays/s,
the host core count, and PSNR of the frame against the CPU oracle (the CUDA-path
semantics): a cross-check of the oracle against the reference's OTHER backend.
"""
from __future__ import annotations

import argparse
import ctypes as C
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = os.environ.get("VOLREND_REFERENCE", "/root/reference")

EGL_NONE, EGL_SURFACE_TYPE, EGL_PBUFFER_BIT = 0x3038, 0x3033, 0x0001
EGL_RENDERABLE_TYPE, EGL_OPENGL_ES3_BIT = 0x3040, 0x0040
EGL_RED_SIZE, EGL_GREEN_SIZE, EGL_BLUE_SIZE, EGL_ALPHA_SIZE = 0x3024, 0x3023, 0x3022, 0x3021
EGL_WIDTH, EGL_HEIGHT, EGL_OPENGL_ES_API, EGL_CONTEXT_CLIENT_VERSION = 0x3057, 0x3056, 0x30A0, 0x3098
GL_VERTEX_SHADER, GL_FRAGMENT_SHADER, GL_COMPILE_STATUS, GL_LINK_STATUS = 0x8B31, 0x8B30, 0x8B81, 0x8B82
GL_TEXTURE_2D, GL_R16F, GL_R32F, GL_R32I, GL_RED, GL_RED_INTEGER = 0x0DE1, 0x822D, 0x822E, 0x8235, 0x1903, 0x8D94
GL_INT, GL_FLOAT, GL_RGBA8, GL_RGBA, GL_UNSIGNED_BYTE = 0x1404, 0x1406, 0x8058, 0x1908, 0x1401
GL_TEXTURE_MIN_FILTER, GL_TEXTURE_MAG_FILTER, GL_NEAREST, GL_TEXTURE0 = 0x2801, 0x2800, 0x2600, 0x84C0
GL_ARRAY_BUFFER, GL_STATIC_DRAW, GL_TRIANGLE_STRIP, GL_COLOR_BUFFER_BIT = 0x8892, 0x88E4, 0x0005, 0x4000
GL_MAX_TEXTURE_SIZE, GL_UNPACK_ALIGNMENT, GL_PACK_ALIGNMENT = 0x0D33, 0x0CF5, 0x0D05

VERT_SRC = b"""#version 300 es
in vec3 aPos;
void main() { gl_Position = vec4(aPos.x, aPos.y, aPos.z, 1.0); }
"""


def find_swiftshader():
    for pat in ("/usr/local/lib/python3*/dist-packages/kaleido/executable/bin/swiftshader",
                "/opt/conda/lib/python3*/site-packages/kaleido/executable/bin/swiftshader"):
        for d in glob.glob(pat):
            if os.path.exists(os.path.join(d, "libEGL.so")):
                return d
    return None


def find_mesa():
    """System Mesa EGL + GLES (llvmpipe), if installed: (libEGL path, libGLESv2 path) or None."""
    import ctypes.util
    egl, gles = ctypes.util.find_library("EGL"), ctypes.util.find_library("GLESv2")
    return (egl, gles) if egl and gles else None


class GL:
    def __init__(self, width, height, prefer="auto"):
        mesa = find_mesa() if prefer in ("auto", "mesa") else None
        if mesa is not None:
            try:
                os.environ.setdefault("EGL_PLATFORM", "surfaceless")
                os.environ.setdefault("LIBGL_ALWAYS_SOFTWARE", "1")
                os.environ.setdefault("LP_NUM_THREADS", str(os.cpu_count() or 1))
                self._open(mesa[1], mesa[0], width, height)
                self.stack = "Mesa EGL (surfaceless)"
                return
            except (OSError, RuntimeError):
                if prefer == "mesa":
                    raise
        d = find_swiftshader()
        if d is None:
            raise RuntimeError("no software GL (Mesa EGL or SwiftShader) found")
        self._open(os.path.join(d, "libGLESv2.so"), os.path.join(d, "libEGL.so"), width, height)
        self.stack = "SwiftShader EGL pbuffer"

    def _open(self, gles_path, egl_path, width, height):
        self.gles = C.CDLL(gles_path, mode=C.RTLD_GLOBAL)
        self.egl = C.CDLL(egl_path, mode=C.RTLD_GLOBAL)
        e = self.egl
        e.eglGetDisplay.restype = C.c_void_p
        e.eglGetDisplay.argtypes = [C.c_void_p]
        e.eglCreatePbufferSurface.restype = C.c_void_p
        e.eglCreatePbufferSurface.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        e.eglCreateContext.restype = C.c_void_p
        e.eglCreateContext.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        e.eglMakeCurrent.argtypes = [C.c_void_p] * 4
        e.eglInitialize.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        e.eglChooseConfig.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_void_p),
                                      C.c_int, C.POINTER(C.c_int)]
        dpy = e.eglGetDisplay(None)
        ma, mi = C.c_int(), C.c_int()
        if not e.eglInitialize(dpy, C.byref(ma), C.byref(mi)):
            raise RuntimeError("eglInitialize failed")
        attrs = (C.c_int * 13)(EGL_SURFACE_TYPE, EGL_PBUFFER_BIT, EGL_RENDERABLE_TYPE,
                               EGL_OPENGL_ES3_BIT, EGL_RED_SIZE, 8, EGL_GREEN_SIZE, 8,
                               EGL_BLUE_SIZE, 8, EGL_ALPHA_SIZE, 8, EGL_NONE)
        cfg, n = C.c_void_p(), C.c_int()
        if not e.eglChooseConfig(dpy, attrs, C.byref(cfg), 1, C.byref(n)) or n.value < 1:
            raise RuntimeError("eglChooseConfig failed")
        pb = (C.c_int * 5)(EGL_WIDTH, width, EGL_HEIGHT, height, EGL_NONE)
        surf = e.eglCreatePbufferSurface(dpy, cfg, pb)
        e.eglBindAPI(EGL_OPENGL_ES_API)
        ca = (C.c_int * 3)(EGL_CONTEXT_CLIENT_VERSION, 3, EGL_NONE)
        ctx = e.eglCreateContext(dpy, cfg, None, ca)
        if not surf or not ctx or not e.eglMakeCurrent(dpy, surf, surf, ctx):
            raise RuntimeError("EGL context creation failed")
        g = self.gles
        g.glGetString.restype = C.c_char_p
        g.glGetUniformLocation.argtypes = [C.c_uint, C.c_char_p]
        g.glUniform1f.argtypes = [C.c_int, C.c_float]
        g.glUniform2f.argtypes = [C.c_int, C.c_float, C.c_float]
        g.glUniform3f.argtypes = [C.c_int, C.c_float, C.c_float, C.c_float]
        g.glTexImage2D.argtypes = [C.c_uint, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint,
                                   C.c_uint, C.c_void_p]
        g.glReadPixels.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint, C.c_uint, C.c_void_p]
        g.glBufferData.argtypes = [C.c_uint, C.c_ssize_t, C.c_void_p, C.c_uint]
        g.glVertexAttribPointer.argtypes = [C.c_uint, C.c_int, C.c_uint, C.c_ubyte, C.c_int, C.c_void_p]
        self.renderer = (g.glGetString(0x1F01) or b"?").decode()
        self.version = (g.glGetString(0x1F02) or b"?").decode()

    def shader(self, kind, src: bytes):
        g = self.gles
        s = g.glCreateShader(kind)
        p = C.c_char_p(src)
        g.glShaderSource(s, 1, C.byref(p), None)
        g.glCompileShader(s)
        ok = C.c_int()
        g.glGetShaderiv(s, GL_COMPILE_STATUS, C.byref(ok))
        if not ok.value:
            log = C.create_string_buffer(8192)
            g.glGetShaderInfoLog(s, 8192, None, log)
            raise RuntimeError("shader compile failed:\n" + log.value.decode())
        return s

    def program(self, vs: bytes, fs: bytes):
        g = self.gles
        prog = g.glCreateProgram()
        g.glAttachShader(prog, self.shader(GL_VERTEX_SHADER, vs))
        g.glAttachShader(prog, self.shader(GL_FRAGMENT_SHADER, fs))
        g.glBindAttribLocation(prog, 0, b"aPos")
        g.glLinkProgram(prog)
        ok = C.c_int()
        g.glGetProgramiv(prog, GL_LINK_STATUS, C.byref(ok))
        if not ok.value:
            log = C.create_string_buffer(8192)
            g.glGetProgramInfoLog(prog, 8192, None, log)
            raise RuntimeError("program link failed:\n" + log.value.decode())
        g.glUseProgram(prog)
        return prog

    def texture(self, unit, internal, w, h, fmt, typ, data):
        g = self.gles
        t = C.c_uint()
        g.glGenTextures(1, C.byref(t))
        g.glActiveTexture(GL_TEXTURE0 + unit)
        g.glBindTexture(GL_TEXTURE_2D, t)
        g.glPixelStorei(GL_UNPACK_ALIGNMENT, 1)
        g.glTexImage2D(GL_TEXTURE_2D, 0, internal, w, h, 0, fmt, typ, data.ctypes.data)
        g.glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_NEAREST)
        g.glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_NEAREST)
        return t


def auto_size_2d(size, base_dim=1):
    """shader_renderer.cpp:263-279"""
    width = int(np.sqrt(size))
    if width % base_dim:
        width += base_dim - width % base_dim
    height = (size - 1) // width + 1
    return width, height


def psnr(a, b):
    mse = float(((a.astype(np.float64) - b.astype(np.float64)) ** 2).mean())
    return float("inf") if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C0")
    ap.add_argument("--size", type=int, default=0, help="override the square image size")
    ap.add_argument("--pose", type=int, default=0)
    ap.add_argument("--out", default="")
    ap.add_argument("--rasteriser", default="auto", choices=["auto", "mesa", "swiftshader"])
    ap.add_argument("--frames", type=int, default=1, help="timed frames after the warm-up frame")
    args = ap.parse_args()
    frag_path = os.path.join(REF, "shaders", "rt.frag")
    built = os.path.join(ROOT, "oracle", "_ref", "rt_frag_es300.glsl")
    if os.path.exists(frag_path):
        frag = b"#version 300 es\n" + open(frag_path, "rb").read()
    elif os.path.exists(built):
        frag = open(built, "rb").read()
    else:
        raise SystemExit("needs the reference's rt.frag: the reference mount, or oracle/_ref/"
                         "rt_frag_es300.glsl generated by build() where the mount exists")
    from oracle import binding as ob
    from volrend_amd import synth

    cfg = synth.CONFIGS[args.config]
    W = H = args.size or cfg["width"]
    focal = cfg["focal"] * W / cfg["width"]
    tree = synth.make_config_tree(args.config)
    n_slots = tree.capacity * 8
    dd = tree.data_dim
    gl = GL(W, H, args.rasteriser)
    g = gl.gles
    mx = C.c_int()
    g.glGetIntegerv(GL_MAX_TEXTURE_SIZE, C.byref(mx))
    dw, dh = auto_size_2d(n_slots * dd, dd)
    cw, ch = auto_size_2d(n_slots)
    if max(dw, dh, cw, ch) > mx.value:
        raise SystemExit(f"tree exceeds GL_MAX_TEXTURE_SIZE={mx.value}")

    prog = gl.program(VERT_SRC, frag)
    loc = lambda n: g.glGetUniformLocation(prog, n.encode())  # noqa: E731

    data = np.zeros(dw * dh, dtype=np.float32)
    data[:n_slots * dd] = tree.data.reshape(-1).astype(np.float32)
    child = np.zeros(cw * ch, dtype=np.int32)
    child[:n_slots] = tree.child.reshape(-1)
    gl.texture(0, GL_R32I, cw, ch, GL_RED_INTEGER, GL_INT, child)
    gl.texture(1, GL_R16F, dw, dh, GL_RED, GL_FLOAT, data)
    depth = np.full(W * H, 1e9, dtype=np.float32)
    gl.texture(2, GL_R32F, W, H, GL_RED, GL_FLOAT, depth)
    bgc = np.full((H, W, 4), 255, dtype=np.uint8)  # background_brightness = 1
    gl.texture(3, GL_RGBA8, W, H, GL_RGBA, GL_UNSIGNED_BYTE, bgc)
    for name, unit in (("tree_child_tex", 0), ("tree_data_tex", 1), ("mesh_depth_tex", 2),
                       ("mesh_color_tex", 3)):
        g.glUniform1i(loc(name), unit)
    g.glUniform1i(loc("tree_data_dim"), dw)
    g.glUniform1i(loc("tree_child_dim"), cw)
    g.glUniform1i(loc("tree.N"), 2)
    g.glUniform1i(loc("tree.data_dim"), dd)
    g.glUniform1i(loc("tree.format"), ob.FORMATS[tree.format_name])
    g.glUniform1i(loc("tree.basis_dim"), tree.basis_dim if tree.basis_dim > 0 else 1)
    g.glUniform3f(loc("tree.center"), *[float(x) for x in tree.offset])
    g.glUniform3f(loc("tree.scale"), *[float(x) for x in tree.invradius3])
    g.glUniform1f(loc("tree.ndc_width"), -1.0)
    poses = synth.make_poses(200)
    tr = synth.c2w_to_transform(poses[args.pose])
    g.glUniformMatrix4x3fv(loc("cam.transform"), 1, 0, (C.c_float * 12)(*[float(x) for x in tr]))
    g.glUniform2f(loc("cam.focal"), focal, focal)
    g.glUniform2f(loc("cam.reso"), float(W), float(H))
    g.glUniform1f(loc("opt.step_size"), 1e-4)
    g.glUniform1f(loc("opt.background_brightness"), 1.0)
    g.glUniform1f(loc("opt.stop_thresh"), 1e-2)
    g.glUniform1f(loc("opt.sigma_thresh"), 1e-2)
    g.glUniform1fv(loc("opt.render_bbox"), 6, (C.c_float * 6)(0, 0, 0, 1, 1, 1))
    # VolumeRenderer::set narrows basis_minmax to the tree's basis (shader_renderer.cpp:219-220)
    g.glUniform1iv(loc("opt.basis_minmax"), 2, (C.c_int * 2)(0, max(tree.basis_dim - 1, 0)))
    g.glUniform3f(loc("opt.rot_dirs"), 0.0, 0.0, 0.0)

    quad = np.array([-1, -1, .5, 1, -1, .5, -1, 1, .5, 1, 1, .5], dtype=np.float32)
    vbo, vao = C.c_uint(), C.c_uint()
    g.glGenBuffers(1, C.byref(vbo))
    g.glGenVertexArrays(1, C.byref(vao))
    g.glBindVertexArray(vao)
    g.glBindBuffer(GL_ARRAY_BUFFER, vbo)
    g.glBufferData(GL_ARRAY_BUFFER, quad.nbytes, quad.ctypes.data, GL_STATIC_DRAW)
    g.glVertexAttribPointer(0, 3, GL_FLOAT, 0, 12, None)
    g.glEnableVertexAttribArray(0)
    g.glViewport(0, 0, W, H)

    def draw():
        g.glClear(GL_COLOR_BUFFER_BIT)
        g.glDrawArrays(GL_TRIANGLE_STRIP, 0, 4)
        g.glFinish()

    t0 = time.perf_counter()
    draw()  # warm-up: includes the rasteriser's shader JIT
    t_warm = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(max(1, args.frames)):
        draw()
    t_frame = (time.perf_counter() - t0) / max(1, args.frames)
    img = np.zeros((H, W, 4), dtype=np.uint8)
    g.glPixelStorei(GL_PACK_ALIGNMENT, 1)
    g.glReadPixels(0, 0, W, H, GL_RGBA, GL_UNSIGNED_BYTE, img.ctypes.data)
    img = img[::-1].copy()  # GL rows are bottom-up
    err = g.glGetError()

    # the CUDA-path semantics on the same inputs (CPU oracle)
    th = ob.TreeHandle(tree)
    cam = ob.make_camera(tr, W, H, focal)
    t0 = time.perf_counter()
    ref, _, cnt = ob.render(th, cam, ob.default_options(), want_accum=False)
    t_oracle = time.perf_counter() - t0
    d = np.abs(img[..., :3].astype(int) - ref[..., :3].astype(int))
    result = {
        "config": args.config, "image": [W, H], "pose": args.pose, "nodes": tree.capacity,
        "rasteriser": f"{gl.renderer}; {gl.version}; {gl.stack}" + (
            "" if "llvmpipe" in gl.renderer.lower() else " -- not Mesa llvmpipe"),
        "cores": os.cpu_count(),
        "rt_frag_ms_per_frame": round(t_frame * 1e3, 2),
        "rt_frag_mrays_per_s": round(W * H / t_frame / 1e6, 4),
        "rt_frag_warmup_ms_incl_shader_jit": round(t_warm * 1e3, 1),
        "oracle_ms_per_frame_same_cores": round(t_oracle * 1e3, 2),
        "psnr_rt_frag_vs_oracle_db": round(psnr(img[..., :3], ref[..., :3]), 2),
        "max_abs_diff": int(d.max()), "pixels_differing": int((d > 0).any(-1).sum()),
        "pixels_differing_by_more_than_2": int((d > 2).any(-1).sum()),
        "gl_error": int(err), "samples": cnt["samples"],
    }
    print(json.dumps(result, indent=1))
    if args.out:
        json.dump(result, open(args.out, "w"), indent=1)
        from PIL import Image
        Image.fromarray(img).save(os.path.splitext(args.out)[0] + "_rtfrag.png")
        Image.fromarray(ref).save(os.path.splitext(args.out)[0] + "_oracle.png")


if __name__ == "__main__":
    main()
