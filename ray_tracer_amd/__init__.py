"""ray_tracer_amd -- Python binding of the MI355X renderer.

Two shared libraries built by `ray_tracer_amd/csrc/Makefile` sit next to this file:

* ``librt_amd.so``  HIP kernels + the C ABI declared in ``include/rt_amd.h`` (the product);
* ``librt_host.so`` the C++ object API (Sphere / Plane / Cone / Mesh / Box, OBJ and PNG loaders,
  presets, PPM writer) that lowers scenes to the arrays the C ABI takes.

This module only forwards to them through ctypes.  There is no Python or CPU rendering path: if a
library is missing, or no GPU is usable, calls raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
ASSETS = os.path.join(ROOT, "assets")

MODE_REF_1SPP, MODE_REF_AA9, MODE_GRID = 0, 1, 2
ACCEL_LBVH, ACCEL_BRUTE = 0, 1
TILE = 16
MAX_DEPTH = 16
OBJ_SPHERE, OBJ_PLANE, OBJ_CONE, OBJ_MESH = 0, 1, 2, 3
KERNEL_AUTO, KERNEL_PIXEL, KERNEL_SPLIT, KERNEL_SPLIT2, KERNEL_PER_TILE, KERNEL_QUEUE = range(6)
KERNEL_NAMES = ("k_render", "k_render_split<1>", "k_render_split<2>", "k_render_hybrid", "k_render_queue")  # by Counters.render_variant


class RtError(RuntimeError):
    pass


class Material(C.Structure):
    _fields_ = [("ambient", C.c_float * 3), ("diffuse", C.c_float * 3), ("specular", C.c_float * 3),
                ("shininess", C.c_float), ("reflectiveness", C.c_float),
                ("refractiveness", C.c_float), ("refractive_ratio", C.c_float),
                ("texture", C.c_int32)]

    def __init__(self, ambient=(0, 0, 0), diffuse=(1, 1, 1), specular=(0, 0, 0), shininess=0.0,
                 reflectiveness=0.0, refractiveness=0.0, refractive_ratio=0.0, texture=-1):
        super().__init__()
        self.ambient[:] = ambient
        self.diffuse[:] = diffuse
        self.specular[:] = specular if hasattr(specular, "__len__") else (specular,) * 3
        self.shininess, self.reflectiveness = shininess, reflectiveness
        self.refractiveness, self.refractive_ratio = refractiveness, refractive_ratio
        self.texture = texture


class Texture(C.Structure):
    _fields_ = [("kind", C.c_int32), ("width", C.c_uint32), ("height", C.c_uint32),
                ("base_color", C.c_void_p), ("normal", C.c_void_p), ("occlusion", C.c_void_p),
                ("roughness", C.c_void_p)]


class Plane(C.Structure):
    _fields_ = [("point", C.c_float * 3), ("normal", C.c_float * 3), ("b1", C.c_float * 3),
                ("b2", C.c_float * 3), ("material", C.c_int32)]


class Sphere(C.Structure):
    _fields_ = [("to_local", C.c_float * 12), ("to_world", C.c_float * 12),
                ("normal_mat", C.c_float * 12), ("material", C.c_int32)]


class Cone(C.Structure):
    _fields_ = [("to_local", C.c_float * 12), ("to_world", C.c_float * 12),
                ("normal_mat", C.c_float * 12), ("cap", Plane), ("material", C.c_int32)]


class Mesh(C.Structure):
    _fields_ = [("to_local", C.c_float * 12), ("to_world", C.c_float * 12),
                ("normal_mat", C.c_float * 12), ("tri_offset", C.c_int32), ("tri_count", C.c_int32),
                ("boxed", C.c_int32), ("box_to_local", C.c_float * 12), ("box_min", C.c_float * 3),
                ("box_max", C.c_float * 3)]


class Object(C.Structure):
    _fields_ = [("kind", C.c_int32), ("index", C.c_int32)]


class SceneDesc(C.Structure):
    _fields_ = [("n_objects", C.c_int32), ("objects", C.POINTER(Object)),
                ("n_spheres", C.c_int32), ("spheres", C.POINTER(Sphere)),
                ("n_planes", C.c_int32), ("planes", C.POINTER(Plane)),
                ("n_cones", C.c_int32), ("cones", C.POINTER(Cone)),
                ("n_meshes", C.c_int32), ("meshes", C.POINTER(Mesh)),
                ("n_triangles", C.c_int32), ("tri_pos", C.POINTER(C.c_float)),
                ("tri_nrm", C.POINTER(C.c_float)), ("tri_uv", C.POINTER(C.c_float)),
                ("tri_material", C.POINTER(C.c_int32)), ("tri_frag", C.POINTER(C.c_int32)),
                ("tri_sub", C.POINTER(C.c_int32)),
                ("n_materials", C.c_int32), ("materials", C.POINTER(Material)),
                ("n_textures", C.c_int32), ("textures", C.POINTER(Texture)),
                ("n_lights", C.c_int32), ("light_pos", C.POINTER(C.c_float)),
                ("light_color", C.POINTER(C.c_float)), ("ambient", C.c_float * 3)]


class RenderParams(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("fov", C.c_float), ("mode", C.c_int32),
                ("grid_m", C.c_int32), ("max_depth", C.c_int32), ("accel", C.c_int32),
                ("rank", C.c_int32), ("world_size", C.c_int32), ("count_work", C.c_int32), ("kernel", C.c_int32)]


class Counters(C.Structure):
    _fields_ = [("rays_primary", C.c_uint64), ("rays_shadow", C.c_uint64),
                ("rays_secondary", C.c_uint64), ("shadow_queries", C.c_uint64),
                ("tri_tests", C.c_uint64), ("box_tests", C.c_uint64),
                ("gate_tests", C.c_uint64), ("gate_tri_tests", C.c_uint64), ("quadric_tests", C.c_uint64),
                ("plane_tests", C.c_uint64), ("shade_evals", C.c_uint64), ("tex_fetches", C.c_uint64),
                ("node_steps", C.c_uint64), ("node_lane_visits", C.c_uint64),
                ("last_trace_ms", C.c_float), ("last_build_ms", C.c_float),
                ("bvh_max_depth", C.c_int32), ("render_variant", C.c_int32)]

    @property
    def rays(self):
        return self.rays_primary + self.rays_shadow + self.rays_secondary

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ }


HIT_DTYPE = np.dtype([("hit", "i4"), ("object", "i4"), ("frag", "i4"), ("sub", "i4"),
                      ("from_outside", "i4"), ("distance", "f4"), ("normal", "f4", 3),
                      ("point", "f4", 3), ("uv", "f4", 2)])

# every symbol include/rt_amd.h declares
ABI_SYMBOLS = ["rt_context_create", "rt_context_destroy", "rt_last_error", "rt_scene_upload", "rt_scene_update",
               "rt_build_accel", "rt_render", "rt_shard_tiles", "rt_render_shard_device",
               "rt_untile_device", "rt_sync", "rt_stream", "rt_set_stream", "rt_trace_rays", "rt_intersect_object",
               "rt_get_counters", "rt_reset_counters", "rt_debug_copy_bvh", "rt_untile_device_on",
               "rt_render_shard_device_noalloc"]

_amd = None
_host = None


def _load(name):
    path = os.path.join(_HERE, name)
    if name == "librt_amd.so" and os.environ.get("RT_AMD_LIB"):
        path = os.environ["RT_AMD_LIB"]  # development: A/B another build of the same library
    if not os.path.exists(path):
        raise RtError(f"{path} is missing: build it with `make -C ray_tracer_amd/csrc` "
                      "(or __graft_entry__.build())")
    return C.CDLL(path, mode=C.RTLD_GLOBAL)


def amd():
    """librt_amd.so with argument types set."""
    global _amd
    if _amd is None:
        L = _load("librt_amd.so")
        vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
        L.rt_context_create.argtypes = [C.c_int, C.POINTER(vp)]
        L.rt_context_destroy.argtypes = [vp]
        L.rt_last_error.argtypes = [vp]
        L.rt_last_error.restype = C.c_char_p
        L.rt_scene_upload.argtypes = [vp, C.POINTER(SceneDesc)]
        L.rt_scene_update.argtypes = [vp, C.POINTER(SceneDesc)]
        L.rt_build_accel.argtypes = [vp]
        L.rt_render.argtypes = [vp, C.POINTER(RenderParams), vp, vp]
        L.rt_shard_tiles.argtypes = [C.POINTER(RenderParams), C.POINTER(i32), C.POINTER(i32)]
        L.rt_render_shard_device.argtypes = [vp, C.POINTER(RenderParams), vp, vp]
        L.rt_render_shard_device_noalloc.argtypes = [vp, C.POINTER(RenderParams), vp, vp]
        L.rt_untile_device.argtypes = [vp, C.POINTER(RenderParams), vp, vp, vp, vp]
        L.rt_untile_device_on.argtypes = [vp, vp, C.POINTER(RenderParams), vp, vp, vp, vp]
        L.rt_sync.argtypes = [vp]
        L.rt_stream.argtypes = [vp]
        L.rt_stream.restype = vp
        L.rt_set_stream.argtypes = [vp, vp]
        L.rt_trace_rays.argtypes = [vp, i64, vp, vp, i32, i32, vp, vp]
        L.rt_intersect_object.argtypes = [vp, i32, i64, vp, vp, i32, vp]
        L.rt_get_counters.argtypes = [vp, C.POINTER(Counters)]
        L.rt_reset_counters.argtypes = [vp]
        L.rt_debug_copy_bvh.argtypes = [vp, i32, vp, C.POINTER(i32), vp, C.POINTER(i32)]
        _amd = L
    return _amd


def host():
    """librt_host.so with argument types set."""
    global _host
    if _host is None:
        amd()  # librt_host links against librt_amd
        L = _load("librt_host.so")
        vp = C.c_void_p
        L.rth_scene_new.restype = vp
        L.rth_scene_free.argtypes = [vp]
        L.rth_last_error.argtypes = [vp]
        L.rth_last_error.restype = C.c_char_p
        L.rth_build_preset.argtypes = [vp, C.c_char_p, C.c_char_p]
        L.rth_add_checker_texture.argtypes = [vp]
        L.rth_add_rainbow_texture.argtypes = [vp]
        L.rth_add_image_texture.argtypes = [vp, C.c_uint, C.c_uint, C.POINTER(vp)]
        L.rth_add_image_texture_sized.argtypes = [vp, C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.POINTER(vp)]
        L.rth_add_image_texture_files.argtypes = [vp, C.POINTER(C.c_char_p)]
        L.rth_add_sphere.argtypes = [vp, vp, C.POINTER(Material)]
        L.rth_add_plane.argtypes = [vp, vp, vp, C.POINTER(Material)]
        L.rth_add_cone.argtypes = [vp, vp, C.POINTER(Material)]
        L.rth_add_mesh_obj.argtypes = [vp, C.c_char_p, vp, C.POINTER(Material), C.c_int, C.c_int, C.c_int]
        L.rth_add_mesh_tris.argtypes = [vp, C.c_int, vp, vp, vp, vp, C.POINTER(Material), C.c_int]
        L.rth_add_light.argtypes = [vp, vp, vp]
        L.rth_flatten.argtypes = [vp]
        L.rth_flatten.restype = C.POINTER(SceneDesc)
        L.rth_mat_identity.argtypes = [vp]
        L.rth_mat_translate.argtypes = [vp, vp, vp]
        L.rth_mat_rotate.argtypes = [vp, C.c_float, vp, vp]
        L.rth_mat_scale.argtypes = [vp, vp, vp]
        L.rth_mat_mul.argtypes = [vp, vp, vp]
        L.rth_decode_png.argtypes = [C.c_char_p, C.POINTER(C.c_uint), C.POINTER(C.c_uint), vp, C.c_size_t]
        L.rth_write_ppm.argtypes = [C.c_char_p, C.c_int, C.c_int, vp, C.c_int]
        _host = L
    return _host


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


# ---- glm-compatible matrices (column-major 4x4 as a flat float32[16]) --------------------------

def mat_identity():
    m = np.zeros(16, np.float32)
    host().rth_mat_identity(_ptr(m))
    return m


def translate(v, m=None):
    m = mat_identity() if m is None else _f32(m)
    out, v = np.zeros(16, np.float32), _f32(v)
    host().rth_mat_translate(_ptr(m), _ptr(v), _ptr(out))
    return out


def rotate(angle, axis, m=None):
    m = mat_identity() if m is None else _f32(m)
    out, axis = np.zeros(16, np.float32), _f32(axis)
    host().rth_mat_rotate(_ptr(m), float(np.float32(angle)), _ptr(axis), _ptr(out))
    return out


def scale(v, m=None):
    m = mat_identity() if m is None else _f32(m)
    v = _f32(v if hasattr(v, "__len__") else (v, v, v))
    out = np.zeros(16, np.float32)
    host().rth_mat_scale(_ptr(m), _ptr(v), _ptr(out))
    return out


def matmul(a, b):
    a, b, out = _f32(a), _f32(b), np.zeros(16, np.float32)
    host().rth_mat_mul(_ptr(a), _ptr(b), _ptr(out))
    return out


def decode_png(path):
    """PNG -> RGBA8 array [h, w, 4] with the renderer's own decoder."""
    w, h = C.c_uint(), C.c_uint()
    if host().rth_decode_png(path.encode(), C.byref(w), C.byref(h), None, 0) != 0:
        raise RtError(f"cannot decode {path}")
    out = np.zeros((h.value, w.value, 4), np.uint8)
    if host().rth_decode_png(path.encode(), None, None, _ptr(out), out.nbytes) != 0:
        raise RtError(f"cannot decode {path}")
    return out


def write_ppm(path, rgb8, binary=False):
    rgb8 = np.ascontiguousarray(rgb8, dtype=np.uint8)
    h, w = rgb8.shape[:2]
    host().rth_write_ppm(path.encode(), w, h, _ptr(rgb8), 1 if binary else 0)


def write_png(path, rgb8):
    rgb8 = np.ascontiguousarray(rgb8, dtype=np.uint8)
    h, w = rgb8.shape[:2]
    if host().rth_write_ppm(path.encode(), w, h, _ptr(rgb8), 2) != 0:
        raise RtError(f"cannot write {path}")


class HostScene:
    """A scene built through the C++ object API (host side only, no GPU needed)."""

    def __init__(self, preset=None, assets=ASSETS):
        self.L = host()
        self.h = self.L.rth_scene_new()
        self._keep = []
        if preset is not None:
            if self.L.rth_build_preset(self.h, preset.encode(), assets.encode()) != 0:
                raise RtError(f"preset {preset!r}: {self.L.rth_last_error(self.h).decode()}")

    def __del__(self):
        try:
            if self.h:
                self.L.rth_scene_free(self.h)
                self.h = None
        except Exception:
            pass

    def add_checker_texture(self):
        return self.L.rth_add_checker_texture(self.h)

    def add_rainbow_texture(self):
        return self.L.rth_add_rainbow_texture(self.h)

    def add_image_texture(self, maps):
        """maps: five RGBA8 arrays [h, w, 4]: base, height, normal, occlusion, roughness."""
        maps = [np.ascontiguousarray(m, dtype=np.uint8) for m in maps]
        arr = (C.c_void_p * 5)(*[m.ctypes.data for m in maps])
        ws = (C.c_uint * 5)(*[m.shape[1] for m in maps])
        hs = (C.c_uint * 5)(*[m.shape[0] for m in maps])
        return self.L.rth_add_image_texture_sized(self.h, ws, hs, arr)

    def add_image_texture_files(self, paths):
        """paths: five PNG files: base, height, normal, occlusion, roughness."""
        arr = (C.c_char_p * 5)(*[p.encode() for p in paths])
        t = self.L.rth_add_image_texture_files(self.h, arr)
        if t < 0:
            raise RtError("image texture: " + self.L.rth_last_error(self.h).decode())
        return t

    def add_sphere(self, M, mat):
        M = _f32(M)
        return self.L.rth_add_sphere(self.h, _ptr(M), C.byref(mat))

    def add_plane(self, point, normal, mat):
        p, n = _f32(point), _f32(normal)
        return self.L.rth_add_plane(self.h, _ptr(p), _ptr(n), C.byref(mat))

    def add_cone(self, M, mat):
        M = _f32(M)
        return self.L.rth_add_cone(self.h, _ptr(M), C.byref(mat))

    def add_mesh_obj(self, path, M, mat, boxed=True, flip_normals=False, compat=True):
        M = _f32(M)
        return self.L.rth_add_mesh_obj(self.h, path.encode(), _ptr(M), C.byref(mat), int(boxed),
                                       int(flip_normals), int(compat))

    def add_mesh_tris(self, pos, nrm, M, mat, mats=None, boxed=False):
        pos, nrm, M = _f32(pos).reshape(-1, 9), _f32(nrm).reshape(-1, 9), _f32(M)
        marr = None
        if mats is not None:
            marr = (Material * len(mats))(*mats)
        return self.L.rth_add_mesh_tris(self.h, len(pos), _ptr(pos), _ptr(nrm),
                                        C.cast(marr, C.c_void_p) if marr is not None else None,
                                        _ptr(M), C.byref(mat), int(boxed))

    def add_light(self, pos, color):
        p = _f32(pos)
        c = _f32(color if hasattr(color, "__len__") else (color,) * 3)
        return self.L.rth_add_light(self.h, _ptr(p), _ptr(c))

    def flatten(self):
        """Pointer to the rt_scene_desc (valid while this object lives)."""
        d = self.L.rth_flatten(self.h)
        if not d:
            raise RtError("flatten: " + self.L.rth_last_error(self.h).decode())
        return d


class Context:
    """One GPU context of librt_amd.so."""

    def __init__(self, device=0):
        self.L = amd()
        h = C.c_void_p()
        rc = self.L.rt_context_create(device, C.byref(h))
        if rc != 0:
            raise RtError(f"rt_context_create({device}) failed with {rc}: no usable GPU")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.rt_context_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise RtError(f"{what}: rc={rc}: {self.L.rt_last_error(self.h).decode()}")

    def upload(self, scene, build=True):
        desc = scene.flatten() if isinstance(scene, HostScene) else scene
        self._check(self.L.rt_scene_upload(self.h, desc), "rt_scene_upload")
        if build:
            self.build_accel()

    def update(self, scene):
        """New placement / materials / lights of the uploaded geometry; keeps triangles, textures and LBVHs."""
        desc = scene.flatten() if isinstance(scene, HostScene) else scene
        self._check(self.L.rt_scene_update(self.h, desc), "rt_scene_update")

    def build_accel(self):
        self._check(self.L.rt_build_accel(self.h), "rt_build_accel")

    @staticmethod
    def params(width, height, fov=90.0, mode=MODE_REF_1SPP, grid_m=1, max_depth=8, accel=ACCEL_LBVH,
               rank=0, world_size=1, count_work=False, kernel=KERNEL_AUTO):
        return RenderParams(width, height, fov, mode, grid_m, max_depth, accel, rank, world_size,
                            int(count_work), kernel)

    def render(self, width, height, want_rgb=True, want_rgb8=True, into=None, **kw):
        """into: (rgb, rgb8) arrays to render into instead of fresh ones (several ranks of one frame in turn)."""
        p = self.params(width, height, **kw)
        if into is not None:
            rgb, rgb8 = into
            assert rgb is None or (rgb.dtype == np.float32 and rgb.shape == (height, width, 3) and rgb.flags.c_contiguous)
            assert rgb8 is None or (rgb8.dtype == np.uint8 and rgb8.shape == (height, width, 3) and rgb8.flags.c_contiguous)
        else:
            rgb = np.zeros((height, width, 3), np.float32) if want_rgb else None
            rgb8 = np.zeros((height, width, 3), np.uint8) if want_rgb8 else None
        self._check(self.L.rt_render(self.h, C.byref(p), _ptr(rgb), _ptr(rgb8)), "rt_render")
        return rgb, rgb8

    def shard_tiles(self, p):
        per, tot = C.c_int32(), C.c_int32()
        self.L.rt_shard_tiles(C.byref(p), C.byref(per), C.byref(tot))
        return per.value, tot.value

    def render_shard_device(self, p, d_rgb_tiles, d_rgb8_tiles):
        self._check(self.L.rt_render_shard_device(self.h, C.byref(p), d_rgb_tiles, d_rgb8_tiles),
                    "rt_render_shard_device")

    def render_shard_device_noalloc(self, p, d_rgb_tiles, d_rgb8_tiles):
        self._check(self.L.rt_render_shard_device_noalloc(self.h, C.byref(p), d_rgb_tiles, d_rgb8_tiles),
                    "rt_render_shard_device_noalloc")

    def untile_device(self, p, d_rgb_shards, d_rgb_frame, d_rgb8_shards, d_rgb8_frame):
        self._check(self.L.rt_untile_device(self.h, C.byref(p), d_rgb_shards, d_rgb_frame,
                                            d_rgb8_shards, d_rgb8_frame), "rt_untile_device")

    def untile_device_on(self, hip_stream, p, d_rgb_shards, d_rgb_frame, d_rgb8_shards, d_rgb8_frame):
        self._check(self.L.rt_untile_device_on(self.h, hip_stream, C.byref(p), d_rgb_shards, d_rgb_frame,
                                               d_rgb8_shards, d_rgb8_frame), "rt_untile_device_on")

    def sync(self):
        self._check(self.L.rt_sync(self.h), "rt_sync")

    def stream(self):
        return self.L.rt_stream(self.h)

    def set_stream(self, hip_stream):
        self._check(self.L.rt_set_stream(self.h, hip_stream), "rt_set_stream")

    def trace_rays(self, o, d, max_depth=8, accel=ACCEL_LBVH, want_hits=True, want_radiance=True):
        o, d = _f32(o).reshape(-1, 3), _f32(d).reshape(-1, 3)
        hits = np.zeros(len(o), HIT_DTYPE) if want_hits else None
        rad = np.zeros((len(o), 3), np.float32) if want_radiance else None
        self._check(self.L.rt_trace_rays(self.h, len(o), _ptr(o), _ptr(d), max_depth, accel,
                                         _ptr(hits), _ptr(rad)), "rt_trace_rays")
        return hits, rad

    def intersect_object(self, obj, o, d, accel=ACCEL_LBVH):
        o, d = _f32(o).reshape(-1, 3), _f32(d).reshape(-1, 3)
        hits = np.zeros(len(o), HIT_DTYPE)
        self._check(self.L.rt_intersect_object(self.h, obj, len(o), _ptr(o), _ptr(d), accel, _ptr(hits)),
                    "rt_intersect_object")
        return hits

    def counters(self):
        c = Counters()
        self._check(self.L.rt_get_counters(self.h, C.byref(c)), "rt_get_counters")
        return c

    def reset_counters(self):
        self._check(self.L.rt_reset_counters(self.h), "rt_reset_counters")

    def copy_bvh(self, mesh):
        nn, nt = C.c_int32(), C.c_int32()
        self._check(self.L.rt_debug_copy_bvh(self.h, mesh, None, C.byref(nn), None, C.byref(nt)),
                    "rt_debug_copy_bvh")
        nodes = np.zeros((max(nn.value, 1), 16), np.float32)
        order = np.zeros(max(nt.value, 1), np.int32)
        self._check(self.L.rt_debug_copy_bvh(self.h, mesh, _ptr(nodes), C.byref(nn), _ptr(order),
                                             C.byref(nt)), "rt_debug_copy_bvh")
        return nodes[:nn.value], order[:nt.value]
