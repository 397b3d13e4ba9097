"""The CPU oracle against values recorded from the unmodified reference (tests/golden/reference_anchors.json).

The reference ships no tests and its main.cpp cannot be compiled in this image (un-vendored lodepng
header), so these recorded outputs -- above all the sha256 of the whole 1024x768 frame -- are what
pins the oracle to the reference.
"""
import hashlib
import os

import numpy as np
import pytest

import oracle_lib as O


def test_camera(anchors):
    cam = anchors["camera"]
    o, d, w = O.camera_rays(cam["width"], cam["height"])
    d = d.reshape(cam["height"], cam["width"], 3)
    s = cam["s"]
    # main.cpp:1262-1263: dx = X + i*s + s/2, dy = Y - j*s + s/2 (half a pixel UP), un-normalised
    assert d[0, 0, 0] == np.float32(cam["X"] + s / 2) and d[0, 0, 1] == np.float32(cam["Y"] + s / 2)
    assert d[384, 512, 0] == np.float32(s / 2) and d[384, 512, 2] == 1.0
    assert np.all(o == 0)


def test_scene_layout(anchors, oracle_scene):
    s = oracle_scene("shells")
    assert s.num_objects == len(anchors["objects"])
    assert s.L.orc_num_lights(s.h) == anchors["n_lights"]
    for obj, key in ((0, "seashell_bbox"), (1, "shell_bbox")):
        m = s.mesh(obj)
        assert m["n"] == anchors[key]["n_triangles"]
        np.testing.assert_allclose(m["lo"], anchors[key]["min"], rtol=2e-6, atol=2e-6)
        np.testing.assert_allclose(m["hi"], anchors[key]["max"], rtol=2e-6, atol=2e-6)
    for obj, key in ((2, "water_plane"), (3, "sand_plane")):
        b1, b2 = s.plane_basis(obj)
        np.testing.assert_allclose(b1, anchors[key]["b1"], atol=1e-7)
        np.testing.assert_allclose(b2, anchors[key]["b2"], atol=1e-7)


def test_pixel_anchors(anchors, oracle_scene):
    s = oracle_scene("shells")
    cam = anchors["camera"]
    _, d, _ = O.camera_rays(cam["width"], cam["height"])
    d = d.reshape(cam["height"], cam["width"], 3)
    for px in anchors["pixels"]:
        x, y = px["xy"]
        h = s.find_closest_hit(np.zeros(3), d[y, x])[0]
        rad, _ = s.trace_rays(np.zeros(3), d[y, x])
        assert h["object"] == px["object"]
        assert list(O.tonemap_quantise(rad)[0]) == px["rgb8"]
        if "distance" in px:
            assert abs(h["distance"] - px["distance"]) < 1e-6
            np.testing.assert_allclose(h["normal"], px["normal"], atol=1e-5)
            np.testing.assert_allclose(h["uv"], px["uv"], atol=1e-5)
            np.testing.assert_allclose(rad[0], px["radiance"], rtol=1e-7)


def test_teapot_anchor(anchors, oracle_scene):
    a = anchors["teapot"]
    for preset in ("teapot", "teapot_head"):
        s = oracle_scene(preset)
        m = s.mesh(0)
        assert m["n"] == a["n_triangles"]
        np.testing.assert_allclose(m["lo"], a["bbox_min"], atol=1e-6)
        np.testing.assert_allclose(m["hi"], a["bbox_max"], atol=1e-6)
        h = s.intersect_object(0, np.zeros(3), a["ray_d"])[0]
        assert h["hit"] == 1 and abs(h["distance"] - a["distance"]) < 1e-6
        np.testing.assert_allclose(h["point"], a["point"], atol=1e-4)
        np.testing.assert_allclose(np.abs(h["normal"]), a["abs_normal"], atol=1e-4)
        # flat normals as loaded face away from the ray: never from_outside, so the closest-hit
        # filter drops the mesh; negated normals make it visible
        assert h["from_outside"] == (1 if preset == "teapot" else 0)
        c = s.find_closest_hit(np.zeros(3), a["ray_d"])[0]
        assert (c["object"] == 0) == (preset == "teapot")


def test_box_gate_is_one_sided(anchors, oracle_scene):
    s = oracle_scene("teapot")
    a = anchors["teapot"]
    h = s.intersect_object(0, np.zeros(3), a["ray_d"])[0]
    assert h["hit"] == 1
    # a ray that starts on the surface (inside the box) never passes the gate, whatever its direction
    rng = np.random.default_rng(0)
    dirs = rng.normal(size=(256, 3)).astype(np.float32)
    start = h["point"] + 1e-4 * h["normal"]
    inside = s.intersect_object(0, np.tile(start, (256, 1)), dirs)
    assert inside["hit"].sum() == 0


def test_texture_anchors(anchors, oracle_scene):
    s = oracle_scene("shells")  # texture 0 = sand, 1 = water
    for a in anchors["sand_basecolor_rgba"]:
        lin = s.sample_texture(0, 0, [a["uv"]])[0]  # pow(rgb/255, 2.2)
        back = np.round(255.0 * lin.astype(np.float64) ** (1 / 2.2))
        assert list(back.astype(int)) == a["rgba"][:3]
    uv = np.random.default_rng(1).uniform(-3, 3, size=(64, 2))
    lin = s.sample_texture(1, 0, uv)
    want = (np.array(anchors["water_color_constant_rgba"][:3], np.float32) / np.float32(255)) ** np.float32(2.2)
    np.testing.assert_allclose(lin, np.tile(want, (64, 1)), rtol=1e-6)


def test_procedural_texture_anchors(anchors):
    s = O.Scene()
    ck = s.L.orc_add_checker_texture(s.h)
    rb = s.L.orc_add_rainbow_texture(s.h)
    assert s.sample_texture(ck, 0, [anchors["checkerboard"]["uv"]])[0, 0] == anchors["checkerboard"]["value"]
    assert list(s.sample_texture(rb, 0, [anchors["rainbow"]["uv"]])[0]) == anchors["rainbow"]["value"]


@pytest.mark.slow
def test_head_frame_sha256(anchors, oracle_scene, tmp_path):
    """Every pixel of the frame the reference writes at HEAD, through the oracle's own P3 writer."""
    s = oracle_scene("shells")
    _, rgb8, rays = s.render(1024, 768, 90.0, mode=0, threads=os.cpu_count())
    path = str(tmp_path / "head.ppm")
    O.write_ppm(path, rgb8)
    assert hashlib.sha256(open(path, "rb").read()).hexdigest() == anchors["head_frame_sha256"]
    assert rays > 3 * 1024 * 768  # primary + shadow + refraction at least


def test_head_frame_fixture_is_the_reference_frame(anchors, tmp_path):
    """tests/golden/head_frame.npz (the frame the GPU test compares with) written through the oracle's P3 writer
    hashes to the sha256 recorded from the reference binary's output file."""
    import numpy as np
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "head_frame.npz"))
    assert g["rgb8"].shape == (768, 1024, 3) and g["rgb_lattice"].shape == (192, 256, 3)
    path = str(tmp_path / "head.ppm")
    O.write_ppm(path, g["rgb8"])
    assert hashlib.sha256(open(path, "rb").read()).hexdigest() == anchors["head_frame_sha256"]
    assert bytes(g["sha256"]).decode() == anchors["head_frame_sha256"]


def test_refract_anchors(anchors):
    import ctypes as C
    L = O.lib()
    L.orc_refract.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
    for a in anchors["refract"]:
        i = np.array(a["i_unnormalised"], np.float32)
        i = i / np.float32(np.sqrt(np.float32((i * i).sum())))
        n = np.array(a["n"], np.float32)
        out = np.zeros(3, np.float32)
        L.orc_refract(i.ctypes.data, n.ctypes.data, a["index"], out.ctypes.data)
        if a.get("nan"):
            assert np.isnan(out).any()  # no total-internal-reflection branch: grazing rays give NaN
        else:
            np.testing.assert_allclose(out, a["result"], atol=2e-6)
