"""Oracle texture samplers, quantisation and PPM writer against the reference's OWN Textures.cpp and
Image.cpp, compiled unmodified into oracle/_ref/libref_parts.so (built by `make -C oracle` when the
reference checkout is present; the prebuilt library travels to machines without it)."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as O

REF = os.path.join(O.ROOT, "oracle", "_ref", "libref_parts.so")
pytestmark = pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref not built (reference checkout absent)")


@pytest.fixture(scope="module")
def ref():
    L = C.CDLL(REF)
    L.ref_sample.argtypes = [C.c_int, C.c_uint, C.c_uint, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.ref_write_image.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_void_p]
    return L


def _ref_sample(ref, which, img, uv):
    uv = np.ascontiguousarray(uv, np.float32)
    out = np.zeros((len(uv), 3), np.float32)
    if img is None:
        ref.ref_sample(which, 1, 1, None, len(uv), uv.ctypes.data, out.ctypes.data)
    else:
        img = np.ascontiguousarray(img)
        ref.ref_sample(which, img.shape[1], img.shape[0], img.ctypes.data, len(uv), uv.ctypes.data, out.ctypes.data)
    return out


def test_image_texture_samplers_bit_exact(ref):
    rng = np.random.default_rng(7)
    img = rng.integers(0, 256, size=(64, 32, 4), dtype=np.uint8)  # non-square, so x/y mix-ups show
    s = O.Scene()
    tex = s.L.orc_add_image_texture(s.h, 32, 64, *([img.ctypes.data] * 5))
    uv = np.concatenate([rng.uniform(-4, 4, size=(4000, 2)), [[0, 0], [0.5, 0.5], [-0.0, 1.0], [0.49999, -0.25]]]).astype(np.float32)
    for which in range(4):  # texture(), normal(), occlusion(), roughness()
        got = s.sample_texture(tex, which, uv)
        want = _ref_sample(ref, which, img, uv)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"sampler {which}"


def test_procedural_textures_bit_exact(ref):
    rng = np.random.default_rng(8)
    uv = rng.uniform(-3, 3, size=(4000, 2)).astype(np.float32)
    s = O.Scene()
    ck, rb = s.L.orc_add_checker_texture(s.h), s.L.orc_add_rainbow_texture(s.h)
    assert np.array_equal(s.sample_texture(ck, 0, uv), _ref_sample(ref, 4, None, uv))
    assert np.array_equal(s.sample_texture(rb, 0, uv), _ref_sample(ref, 5, None, uv))
    assert (s.sample_texture(ck, 0, uv) == -1).any()  # negative sums give -1 (C remainder)


def test_quantisation_and_ppm_writer_byte_exact(ref, tmp_path):
    rng = np.random.default_rng(9)
    h, w = 37, 53
    rgb01 = rng.uniform(0, 1, size=(h, w, 3)).astype(np.float32)
    rgb01[0, 0] = [0, 1, 0.999999]
    a, b = str(tmp_path / "ref.ppm"), str(tmp_path / "orc.ppm")
    ref.ref_write_image(a.encode(), w, h, rgb01.ctypes.data)
    # the oracle quantises with Image::setPixel's rule: int = (float)(255 * c)
    q = (np.float32(255) * rgb01).astype(np.int32).astype(np.uint8)
    O.write_ppm(b, q)
    assert open(a, "rb").read() == open(b, "rb").read()
