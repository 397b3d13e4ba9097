"""ctypes binding of the CPU twin (tests/cpu_twin/librt_twin.so) -- test scaffolding only.

The twin runs the renderer's per-ray code compiled for the host, so its results tell whether the
LOGIC the kernels execute agrees with the oracle; the GPU tests then check the kernels themselves.
"""
import ctypes as C
import os
import subprocess

import numpy as np

import ray_tracer_amd as rt

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
_lib = None


def build():
    src = os.path.join(HERE, "cpu_twin", "twin.cpp")
    out = os.path.join(HERE, "cpu_twin", "librt_twin.so")
    deps = [src] + [os.path.join(ROOT, "ray_tracer_amd", "csrc", f) for f in ("rt_core.h", "rt_lbvh.h", "rt_prepare.h")]
    deps.append(os.path.join(ROOT, "include", "rt_amd.h"))
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-fopenmp", "-ffp-contract=off", "-fno-fast-math",
                               "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "ray_tracer_amd", "csrc"),
                               "-shared", "-Wl,-Bsymbolic",  # its own copies of the headers' inline functions, not the loaded library's
                               "-o", out, src])
    return out


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        vp = C.c_void_p
        L.twin_create.restype = vp
        L.twin_create.argtypes = [C.POINTER(rt.SceneDesc), C.c_char_p, C.c_int]
        L.twin_destroy.argtypes = [vp]
        L.twin_build_accel.argtypes = [vp]
        L.twin_trace_rays.argtypes = [vp, C.c_int64, vp, vp, C.c_int, C.c_int, vp, vp, C.POINTER(C.c_int)]
        L.twin_intersect_object.argtypes = [vp, C.c_int, C.c_int64, vp, vp, C.c_int, vp]
        L.twin_render.argtypes = [vp, C.POINTER(rt.RenderParams), C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp]
        L.twin_copy_bvh.argtypes = [vp, C.c_int, vp, vp]
        L.twin_ref_count.argtypes = [vp, C.c_int]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Twin:
    def __init__(self, scene, build=True):
        self.L = lib()
        self.scene = scene  # keeps the description's arrays alive
        err = C.create_string_buffer(256)
        self.h = self.L.twin_create(scene.flatten(), err, 256)
        if not self.h:
            raise ValueError(err.value.decode())
        self.height = self.L.twin_build_accel(self.h) if build else 0

    def __del__(self):
        try:
            if self.h:
                self.L.twin_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def trace_rays(self, o, d, max_depth=8, accel=rt.ACCEL_LBVH):
        o = np.ascontiguousarray(o, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
        hits = np.zeros(len(o), rt.HIT_DTYPE)
        rad = np.zeros((len(o), 3), np.float32)
        ms = C.c_int()
        self.L.twin_trace_rays(self.h, len(o), _p(o), _p(d), max_depth, accel, _p(hits), _p(rad), C.byref(ms))
        self.max_stack = ms.value
        return hits, rad

    def intersect_object(self, obj, o, d, accel=rt.ACCEL_LBVH):
        o = np.ascontiguousarray(o, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
        hits = np.zeros(len(o), rt.HIT_DTYPE)
        self.L.twin_intersect_object(self.h, obj, len(o), _p(o), _p(d), accel, _p(hits))
        return hits

    def render(self, width, height, rect=None, **kw):
        p = rt.Context.params(width, height, **kw)
        x0, y0, x1, y1 = rect if rect else (0, 0, width, height)
        rgb = np.zeros((height, width, 3), np.float32)
        rgb8 = np.zeros((height, width, 3), np.uint8)
        counters = np.zeros(10, np.uint64)
        self.L.twin_render(self.h, C.byref(p), x0, y0, x1, y1, _p(rgb), _p(rgb8), _p(counters))
        return rgb, rgb8, counters

    def copy_bvh(self, mesh, ntri=None):
        """nodes and the triangle behind every leaf (a triangle cut into pieces appears once per piece)"""
        ntri = self.L.twin_ref_count(self.h, mesh)
        nodes = np.zeros((max(ntri - 1, 1), 16), np.float32)
        order = np.zeros(max(ntri, 1), np.int32)
        nn = self.L.twin_copy_bvh(self.h, mesh, _p(nodes), _p(order))
        return nodes[:nn], order[:ntri]
