"""Host side: the C++ object API, its loaders and the flattening to the C ABI's arrays, checked
against the oracle's independent construction of the same scenes.  No GPU needed."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as O
import ray_tracer_amd as rt

PRESETS = ["analytic", "teapot", "teapot_head", "shells", "textured"]


def _arr(ptr, n, dtype=np.float32):
    return np.ctypeslib.as_array(ptr, shape=(n,)).view(dtype).copy() if n else np.zeros(0, dtype)


def test_abi_library_loads_and_exports_every_symbol():
    L = rt.amd()
    header = open(os.path.join(rt.ROOT, "include", "rt_amd.h")).read()
    import re
    declared = sorted(set(re.findall(r"\b(rt_[a-z_]+)\s*\(", header)))
    assert declared == sorted(rt.ABI_SYMBOLS)
    for sym in declared:
        assert hasattr(L, sym), sym


def test_no_gpu_means_error_not_fallback():
    """Without a usable GPU the context cannot be created: there is no CPU path to fall back to."""
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    assert rt.amd().rt_context_create(0, C.byref(h)) == -2  # RT_ERR_NO_DEVICE
    with pytest.raises(rt.RtError):
        rt.Context(0)


@pytest.mark.parametrize("preset", PRESETS)
def test_flatten_matches_oracle_scene(preset, host_scene, oracle_scene):
    d = host_scene(preset).flatten().contents
    o = oracle_scene(preset)
    assert d.n_objects == o.num_objects
    assert d.n_lights == o.L.orc_num_lights(o.h)
    mesh_objs = [k for k in range(d.n_objects) if d.objects[k].kind == rt.OBJ_MESH]
    for k in mesh_objs:
        m = d.meshes[d.objects[k].index]
        om = o.mesh(k)
        assert m.tri_count == om["n"]
        sl = slice(m.tri_offset, m.tri_offset + m.tri_count)
        pos = _arr(d.tri_pos, 9 * d.n_triangles).reshape(-1, 9)[sl]
        nrm = _arr(d.tri_nrm, 9 * d.n_triangles).reshape(-1, 9)[sl]
        uv = _arr(d.tri_uv, 6 * d.n_triangles).reshape(-1, 6)[sl]
        frag = _arr(d.tri_frag, d.n_triangles, np.int32)[sl]
        sub = _arr(d.tri_sub, d.n_triangles, np.int32)[sl]
        assert np.array_equal(pos.view(np.uint32), om["pos"].view(np.uint32))
        assert np.array_equal(nrm.view(np.uint32), om["nrm"].view(np.uint32))
        assert np.array_equal(uv.view(np.uint32), om["uv"].view(np.uint32))
        assert np.array_equal(np.stack([frag, sub], 1), om["fragsub"])
        assert m.boxed == 1
        np.testing.assert_array_equal(np.array(m.box_min[:]), om["lo"])
        np.testing.assert_array_equal(np.array(m.box_max[:]), om["hi"])
    for k in range(d.n_objects):
        if d.objects[k].kind == rt.OBJ_PLANE:
            pl = d.planes[d.objects[k].index]
            b1, b2 = o.plane_basis(k)
            np.testing.assert_array_equal(np.array(pl.b1[:], np.float32), b1)
            np.testing.assert_array_equal(np.array(pl.b2[:], np.float32), b2)


def test_obj_loader_pollution_quirk(host_scene, anchors):
    """seashell_obj.obj interleaves v/vt/vn blocks: in reference-compatible mode later faces index vt/vn
    rows as positions, which is what gives the mesh its recorded (huge) bounds."""
    d = host_scene("shells").flatten().contents
    m = d.meshes[0]
    np.testing.assert_allclose(m.box_min[:], anchors["seashell_bbox"]["min"], rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(m.box_max[:], anchors["seashell_bbox"]["max"], rtol=2e-6, atol=2e-6)
    clean = rt.HostScene()
    clean.add_mesh_obj(os.path.join(rt.ASSETS, "seashell_obj.obj"), rt.mat_identity(), rt.Material(), compat=False)
    dm = clean.flatten().contents.meshes[0]
    assert dm.tri_count == m.tri_count
    # the clean parse resolves the later blocks to their real vertices (a second shell and a ground quad)
    assert list(dm.box_max[:]) != list(m.box_max[:])


def test_missing_obj_gives_empty_mesh(tmp_path):
    s = rt.HostScene()
    s.add_mesh_obj(str(tmp_path / "nope.obj"), rt.mat_identity(), rt.Material())
    d = s.flatten().contents
    assert d.n_meshes == 1 and d.meshes[0].tri_count == 0
    assert d.meshes[0].box_min[0] == np.inf and d.meshes[0].box_max[0] == -np.inf  # main.cpp:495-523


def test_matrix_helpers_match_oracle():
    L = O.lib()
    rng = np.random.default_rng(3)
    for _ in range(20):
        v, ax, ang = rng.normal(size=3).astype(np.float32), rng.normal(size=3).astype(np.float32), np.float32(rng.uniform(-3, 3))
        m = rt.matmul(rt.translate(v), rt.matmul(rt.rotate(ang, ax), rt.scale(np.abs(v) + 0.1)))
        I = np.zeros(16, np.float32); L.orc_mat_identity(I.ctypes.data)
        a, b, c, ab, abc = (np.zeros(16, np.float32) for _ in range(5))
        s = (np.abs(v) + np.float32(0.1)).astype(np.float32)
        L.orc_mat_translate(I.ctypes.data, v.ctypes.data, a.ctypes.data)
        L.orc_mat_rotate(I.ctypes.data, float(ang), ax.ctypes.data, b.ctypes.data)
        L.orc_mat_scale(I.ctypes.data, s.ctypes.data, c.ctypes.data)
        L.orc_mat_mul(b.ctypes.data, c.ctypes.data, ab.ctypes.data)
        L.orc_mat_mul(a.ctypes.data, ab.ctypes.data, abc.ctypes.data)
        assert np.array_equal(m.view(np.uint32), abc.view(np.uint32))


def test_png_decoder_matches_pillow_on_every_asset():
    d = os.path.join(rt.ASSETS, "textures", "png")
    files = sorted(f for f in os.listdir(d) if f.endswith(".png"))
    assert len(files) == 20
    kinds = set()
    for f in files:
        from PIL import Image
        im = Image.open(os.path.join(d, f))
        kinds.add((im.mode, im.info.get("interlace", 0)))
        want = np.asarray(im.convert("RGBA"), dtype=np.uint8)
        got = rt.decode_png(os.path.join(d, f))
        assert got.shape == want.shape and np.array_equal(got, want), f
    assert any(i for _, i in kinds), "the Adam7 file should be among the assets"
    assert any(m == "P" for m, _ in kinds), "the palette file should be among the assets"


def test_png_decoder_synthetic_formats(tmp_path):
    from PIL import Image
    rng = np.random.default_rng(5)
    a = rng.integers(0, 256, size=(19, 23, 4), dtype=np.uint8)
    cases = {"rgba": Image.fromarray(a, "RGBA"), "rgb": Image.fromarray(a[..., :3], "RGB"),
             "grey": Image.fromarray(a[..., 0], "L"), "la": Image.fromarray(a[..., :2], "LA"),
             "pal": Image.fromarray(a[..., :3], "RGB").quantize(16), "bit": Image.fromarray(a[..., 0] > 127)}
    for name, im in cases.items():
        for interlace in (False,):
            p = str(tmp_path / f"{name}.png")
            im.save(p)
            want = np.asarray(Image.open(p).convert("RGBA"), dtype=np.uint8)
            assert np.array_equal(rt.decode_png(p), want), name
    with pytest.raises(rt.RtError):
        rt.decode_png(str(tmp_path / "missing.png"))


def test_ppm_writer_matches_oracle_writer(tmp_path):
    rng = np.random.default_rng(6)
    img = rng.integers(0, 256, size=(21, 34, 3), dtype=np.uint8)
    a, b = str(tmp_path / "a.ppm"), str(tmp_path / "b.ppm")
    rt.write_ppm(a, img)
    O.write_ppm(b, img)
    assert open(a, "rb").read() == open(b, "rb").read()
    rt.write_ppm(a, img, binary=True)
    raw = open(a, "rb").read()
    assert raw.startswith(b"P6\n34 21\n255\n") and raw[-img.size:] == img.tobytes()


def test_tiling_roundtrip():
    from ray_tracer_amd import tiling
    rng = np.random.default_rng(2)
    for (w, h) in ((64, 48), (50, 37), (16, 16), (1, 1)):
        frame = rng.uniform(size=(h, w, 3)).astype(np.float32)
        for world in (1, 2, 3, 8):
            shards = np.stack([tiling.shard_from_frame(frame, r, world) for r in range(world)])
            assert np.array_equal(tiling.frame_from_shards(shards, w, h), frame)


def test_png_writer_round_trips_through_the_decoder(tmp_path):
    """The PNG sink writes the pixels it was given: read back with the renderer's own PNG decoder and, as
    an independent check of the container, with zlib by hand."""
    import struct
    import zlib
    rng = np.random.default_rng(5)
    for h, w in [(1, 1), (7, 13), (64, 48)]:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        path = str(tmp_path / f"f_{h}x{w}.png")
        rt.write_png(path, img)
        back = rt.decode_png(path)
        assert back.shape == (h, w, 4) and np.array_equal(back[..., :3], img) and (back[..., 3] == 255).all()
        blob = open(path, "rb").read()
        assert blob[:8] == b"\x89PNG\r\n\x1a\n"
        pos, idat, kinds = 8, b"", []
        while pos < len(blob):
            n, kind = struct.unpack(">I4s", blob[pos:pos + 8])
            body = blob[pos + 8:pos + 8 + n]
            assert struct.unpack(">I", blob[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(kind + body)
            kinds.append(kind)
            if kind == b"IHDR":
                assert struct.unpack(">IIBBBBB", body) == (w, h, 8, 2, 0, 0, 0)
            if kind == b"IDAT":
                idat += body
            pos += 12 + n
        assert kinds[0] == b"IHDR" and kinds[-1] == b"IEND"
        raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + 3 * w)
        assert (raw[:, 0] == 0).all() and np.array_equal(raw[:, 1:].reshape(h, w, 3), img)


def test_image_texture_maps_of_unequal_size_are_refused(tmp_path):
    """The reference samples every map of an ImageTexture with that map's own size (Textures.cpp:27-30); the scene
    description carries one size for the four sampled maps, so unequal ones are an error, not an over-read."""
    big = np.full((8, 8, 4), 200, np.uint8)
    small = np.full((4, 4, 4), 100, np.uint8)
    sc = rt.HostScene()
    tex = sc.add_image_texture([big, big, big, small, big])  # the occlusion map is smaller
    sc.add_plane((0, -1, 0), (0, 1, 0), rt.Material(texture=tex))
    sc.add_light((0, 5, 0), 1.0)
    with pytest.raises(rt.RtError, match="occlusion map is 4x4"):
        sc.flatten()
    # the height map (never sampled, not uploaded) may have any size
    ok = rt.HostScene()
    tex = ok.add_image_texture([big, small, big, big, big])
    ok.add_plane((0, -1, 0), (0, 1, 0), rt.Material(texture=tex))
    d = ok.flatten().contents
    assert d.n_textures == 1 and d.textures[0].width == 8 and d.textures[0].height == 8
    # the same check on files
    paths = []
    for name, arr in (("base", big), ("height", big), ("normal", small), ("occ", big), ("rough", big)):
        path = str(tmp_path / f"{name}.png")
        rt.write_png(path, np.ascontiguousarray(arr[:, :, :3]))
        paths.append(path)
    with pytest.raises(rt.RtError, match="must have one size"):
        rt.HostScene().add_image_texture_files(paths)
