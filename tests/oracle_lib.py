"""ctypes binding of the CPU oracle (oracle/librt_oracle.so) -- test infrastructure only.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
import os
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASSETS = os.path.join(ROOT, "assets")
TEXSETS = {  # order expected by orc_build_preset: sand, water, stone, waffle
    "sand": ["Sand_005_baseColor", "Sand_005_height", "Sand_005_normal",
             "Sand_005_ambientOcclusion", "Sand_005_roughness"],
    "water": ["Water_002_COLOR", "Water_002_DISP", "Water_002_NORM", "Water_002_OCC",
              "Water_002_ROUGH"],
    "stone": ["Stylized_Stone_Floor_005_basecolor", "Stylized_Stone_Floor_005_height",
              "Stylized_Stone_Floor_005_normal", "Stylized_Stone_Floor_005_ambientOcclusion",
              "Stylized_Stone_Floor_005_roughness"],
    "waffle": ["Waffle_001_basecolor", "Waffle_001_height", "Waffle_001_normal",
               "Waffle_001_ambientOcclusion", "Waffle_001_roughness"],
}
TEXTURED_PRESETS = ("shells", "textured")


class OrcMaterial(C.Structure):
    _fields_ = [("ambient", C.c_float * 3), ("diffuse", C.c_float * 3), ("specular", C.c_float * 3),
                ("shininess", C.c_float), ("reflectiveness", C.c_float),
                ("refractiveness", C.c_float), ("refractive_ratio", C.c_float),
                ("texture", C.c_int)]


class OrcTexset(C.Structure):
    _fields_ = [("w", C.c_uint), ("h", C.c_uint), ("maps", C.c_void_p * 5)]


HIT_DTYPE = np.dtype([("hit", "i4"), ("object", "i4"), ("frag", "i4"), ("sub", "i4"),
                      ("from_outside", "i4"), ("distance", "f4"), ("normal", "f4", 3),
                      ("point", "f4", 3), ("uv", "f4", 2)])

_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(ROOT, "oracle", "librt_oracle.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle/librt_oracle.so missing: run `make -C oracle` "
                               "(or __graft_entry__.build())")
        L = C.CDLL(path)
        L.orc_scene_new.restype = C.c_void_p
        L.orc_scene_free.argtypes = [C.c_void_p]
        L.orc_build_preset.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_void_p]
        L.orc_render.restype = C.c_uint64
        L.orc_render.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int,
                                 C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_find_closest_hit.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_intersect_object.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                           C.c_void_p]
        L.orc_trace_rays.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                     C.c_void_p, C.c_void_p]
        L.orc_camera_rays.argtypes = [C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_mesh_info.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_mesh_copy.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p]
        L.orc_plane_basis.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_sample_texture.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_add_image_texture.argtypes = [C.c_void_p, C.c_uint, C.c_uint] + [C.c_void_p] * 5
        L.orc_add_checker_texture.argtypes = [C.c_void_p]
        L.orc_add_rainbow_texture.argtypes = [C.c_void_p]
        L.orc_add_sphere.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_add_plane.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_add_cone.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_add_mesh_obj.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.orc_add_mesh_tris.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_int]
        L.orc_add_mesh_frags.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_int]
        L.orc_add_light.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_num_objects.argtypes = [C.c_void_p]
        L.orc_num_lights.argtypes = [C.c_void_p]
        L.orc_tonemap_quantise.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
        L.orc_write_ppm.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_material_default.argtypes = [C.c_void_p]
        for f in ("orc_mat_identity",):
            getattr(L, f).argtypes = [C.c_void_p]
        L.orc_mat_translate.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_mat_scale.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_mat_rotate.argtypes = [C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]
        L.orc_mat_mul.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def decode_png_rgba(path):
    """RGBA8 expansion with Pillow (what lodepng::decode delivers by default)."""
    from PIL import Image
    im = Image.open(path).convert("RGBA")
    return np.ascontiguousarray(np.asarray(im, dtype=np.uint8))


_tex_cache = {}


def load_texsets(decoder=decode_png_rgba):
    key = id(decoder)
    if key not in _tex_cache:
        sets = []
        for name in ("sand", "water", "stone", "waffle"):
            maps = [decoder(os.path.join(ASSETS, "textures", "png", f + ".png")) for f in TEXSETS[name]]
            sets.append(maps)
        _tex_cache[key] = sets
    return _tex_cache[key]


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class Scene:
    """A scene owned by the oracle."""

    def __init__(self, preset=None, texsets=None):
        self.L = lib()
        self.h = self.L.orc_scene_new()
        self._keep = []
        if preset is not None:
            tex = None
            if preset in TEXTURED_PRESETS:
                sets = texsets if texsets is not None else load_texsets()
                arr = (OrcTexset * 4)()
                for i, maps in enumerate(sets):
                    arr[i].h, arr[i].w = maps[0].shape[0], maps[0].shape[1]
                    for k in range(5):
                        arr[i].maps[k] = maps[k].ctypes.data
                self._keep.append((sets, arr))
                tex = C.cast(arr, C.c_void_p)
            rc = self.L.orc_build_preset(self.h, preset.encode(), ASSETS.encode(), tex)
            if rc != 0:
                raise ValueError(f"oracle preset {preset!r}: rc={rc}")

    def __del__(self):
        try:
            if self.h:
                self.L.orc_scene_free(self.h)
                self.h = None
        except Exception:
            pass

    @property
    def num_objects(self):
        return self.L.orc_num_objects(self.h)

    def render(self, w, h, fov=90.0, mode=0, grid_m=1, max_depth=-1, rect=None, threads=0):
        x0, y0, x1, y1 = rect if rect else (0, 0, w, h)
        rgb = np.zeros((h, w, 3), np.float32)
        rgb8 = np.zeros((h, w, 3), np.uint8)
        rays = self.L.orc_render(self.h, w, h, fov, mode, grid_m, max_depth, x0, y0, x1, y1,
                                 _p(rgb), _p(rgb8), threads)
        return rgb, rgb8, int(rays)

    def find_closest_hit(self, o, d):
        o, d = f32(o).reshape(-1, 3), f32(d).reshape(-1, 3)
        out = np.zeros(len(o), HIT_DTYPE)
        self.L.orc_find_closest_hit(self.h, len(o), _p(o), _p(d), _p(out))
        return out

    def intersect_object(self, obj, o, d):
        o, d = f32(o).reshape(-1, 3), f32(d).reshape(-1, 3)
        out = np.zeros(len(o), HIT_DTYPE)
        self.L.orc_intersect_object(self.h, obj, len(o), _p(o), _p(d), _p(out))
        return out

    def trace_rays(self, o, d, max_depth=-1):
        o, d = f32(o).reshape(-1, 3), f32(d).reshape(-1, 3)
        rad = np.zeros((len(o), 3), np.float32)
        rays = np.zeros(len(o), np.uint64)
        self.L.orc_trace_rays(self.h, len(o), _p(o), _p(d), max_depth, _p(rad), _p(rays))
        return rad, rays

    def mesh(self, obj):
        lo, hi = np.zeros(3, np.float32), np.zeros(3, np.float32)
        n = self.L.orc_mesh_info(self.h, obj, _p(lo), _p(hi))
        pos, nrm = np.zeros((n, 9), np.float32), np.zeros((n, 9), np.float32)
        uv, fs = np.zeros((n, 6), np.float32), np.zeros((n, 2), np.int32)
        self.L.orc_mesh_copy(self.h, obj, _p(pos), _p(nrm), _p(uv), _p(fs))
        return dict(n=n, lo=lo, hi=hi, pos=pos, nrm=nrm, uv=uv, fragsub=fs)

    def plane_basis(self, obj):
        b1, b2 = np.zeros(3, np.float32), np.zeros(3, np.float32)
        self.L.orc_plane_basis(self.h, obj, _p(b1), _p(b2))
        return b1, b2

    def sample_texture(self, tex, which, uv):
        uv = f32(uv).reshape(-1, 2)
        out = np.zeros((len(uv), 3), np.float32)
        self.L.orc_sample_texture(self.h, tex, which, len(uv), _p(uv), _p(out))
        return out


def camera_rays(w, h, fov=90.0, mode=0, grid_m=1, rect=None):
    x0, y0, x1, y1 = rect if rect else (0, 0, w, h)
    taps = 1 if mode == 0 else 9 if mode == 1 else grid_m * grid_m
    n = (x1 - x0) * (y1 - y0) * taps
    o, d, wts = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32), np.zeros(taps, np.float32)
    lib().orc_camera_rays(w, h, fov, mode, grid_m, x0, y0, x1, y1, _p(o), _p(d), _p(wts))
    return o, d, wts


def tonemap_quantise(rgb):
    rgb = f32(rgb).reshape(-1, 3)
    out = np.zeros((len(rgb), 3), np.uint8)
    lib().orc_tonemap_quantise(len(rgb), _p(rgb), _p(out))
    return out


def write_ppm(path, rgb8):
    h, w = rgb8.shape[:2]
    rgb8 = np.ascontiguousarray(rgb8)
    return lib().orc_write_ppm(path.encode(), w, h, _p(rgb8))
