"""Logic of the renderer (the per-ray code the HIP kernels run, compiled for the host by
tests/cpu_twin) against the CPU oracle.  Runs without a GPU; the GPU tests repeat the same checks
through the C ABI on the real kernels."""
import numpy as np
import pytest

import oracle_lib as O
import parity
import ray_tracer_amd as rt
import random_scenes
import twin_lib as T

PRESETS = ["analytic", "teapot", "teapot_head", "shells", "textured"]
_twins = {}


@pytest.fixture(scope="module")
def twin(host_scene):
    def get(preset):
        if preset not in _twins:
            _twins[preset] = T.Twin(host_scene(preset))
        return _twins[preset]
    return get


@pytest.mark.parametrize("preset", PRESETS)
@pytest.mark.parametrize("accel", [rt.ACCEL_LBVH, rt.ACCEL_BRUTE])
def test_rays_match_oracle(preset, accel, twin, oracle_scene, host_scene):
    osc, tw = oracle_scene(preset), twin(preset)
    o, d = parity.ray_set(osc, host_scene(preset).flatten().contents)
    ref_hits = osc.find_closest_hit(o, d)
    ref_rad, _ = osc.trace_rays(o, d, 8)
    hits, rad = tw.trace_rays(o, d, 8, accel)
    ties = parity.check_hits(ref_hits, hits, f"{preset}/accel{accel}")
    assert ties == 0  # same arithmetic on the same CPU: not even grazing ties may differ
    err = parity.check_radiance(ref_rad, rad, preset, frac=1.0)
    assert err.max() <= 1e-5  # shading sums are re-associated; decisions are not


@pytest.mark.parametrize("preset", ["analytic", "teapot", "shells"])
def test_per_object_intersect_matches_oracle(preset, twin, oracle_scene, host_scene):
    osc, tw = oracle_scene(preset), twin(preset)
    o, d = parity.ray_set(osc, host_scene(preset).flatten().contents, w=48, h=36, n_random=500)
    for obj in range(osc.num_objects):
        ref = osc.intersect_object(obj, o, d)
        got = tw.intersect_object(obj, o, d)
        far = np.array([parity.is_spurious_far_hit(r) for r in ref])
        assert far.sum() <= 2e-3 * len(ref)
        assert np.array_equal(ref["hit"][~far], got["hit"][~far]), (preset, obj)
        m = (ref["hit"] == 1) & ~far
        for f in ("frag", "sub", "from_outside"):
            assert np.array_equal(ref[f][m], got[f][m]), (preset, obj, f)
        for f in ("distance", "normal", "point", "uv"):
            assert np.array_equal(ref[f][m], got[f][m]), (preset, obj, f)


@pytest.mark.parametrize("preset,mode,grid", [("analytic", rt.MODE_REF_AA9, 1), ("teapot", rt.MODE_GRID, 2),
                                              ("shells", rt.MODE_REF_1SPP, 1), ("textured", rt.MODE_REF_AA9, 1)])
def test_frames_match_oracle(preset, mode, grid, twin, oracle_scene):
    w, h = 64, 48
    ref_rgb, ref_rgb8, ref_rays = oracle_scene(preset).render(w, h, mode=mode, grid_m=grid, max_depth=8)
    rgb, rgb8, counters = twin(preset).render(w, h, mode=mode, grid_m=grid, max_depth=8)
    parity.check_radiance(ref_rgb, rgb, preset, frac=1.0)
    diff = parity.check_rgb8(ref_rgb8, rgb8, preset)
    assert diff.max() <= 1
    assert int(counters[:3].sum()) == ref_rays  # same ray accounting: one ray per closest-hit query


def test_lbvh_structure(twin, host_scene):
    """Every triangle is behind a leaf (a large one behind several, one per piece), the leaves of a triangle
    cover it, child boxes enclose their subtrees, the height fits the stack; cutting shortens the shells' tree."""
    for preset in ("teapot", "shells"):
        d = host_scene(preset).flatten().contents
        tw = twin(preset)
        pos_all = np.ctypeslib.as_array(d.tri_pos, shape=(d.n_triangles * 9,)).reshape(-1, 3, 3)
        for mi in range(d.n_meshes):
            m = d.meshes[mi]
            nodes, order = tw.copy_bvh(mi)
            height = parity.check_lbvh(nodes, order, pos_all, m.tri_offset, m.tri_count)
            assert len(order) <= 4 * m.tri_count + 64
            if preset == "teapot":
                assert len(order) == m.tri_count  # no triangle of it is large enough to cut
        assert tw.height <= 64


def test_degenerate_meshes(oracle_scene):
    """Zero, one and two triangles; a NaN ray; rays parallel to the triangle plane."""
    tri = np.array([[0, 0, 5, 0, 1, 5, 1, 0, 5], [1, 0, 5, 0, 1, 5, 1, 1, 5]], np.float32)
    nrm = np.tile(np.array([0, 0, -1], np.float32), (2, 3))
    rays_o = np.array([[0.2, 0.2, 0], [0.8, 0.8, 0], [0.5, 0.5, 0], [0.2, 0.2, 0], [0.2, 0.2, 5], [5, 5, 0]], np.float32)
    rays_d = np.array([[0, 0, 1], [0, 0, 1], [0, 0, 1], [np.nan, 0, 1], [1, 0, 0], [0, 0, 1]], np.float32)
    for n in (0, 1, 2):
        hs = rt.HostScene()
        hs.add_mesh_tris(tri[:n], nrm[:n], rt.mat_identity(), rt.Material(diffuse=(1, 0, 0)), boxed=False)
        hs.add_light((0, 5, 0), 1.0)
        osc = O.Scene()
        mat = O.OrcMaterial(); osc.L.orc_material_default(C_byref(mat)); mat.diffuse[:] = (1, 0, 0)
        I = rt.mat_identity()
        osc.L.orc_add_mesh_tris(osc.h, n, tri[:n].ctypes.data if n else None, nrm[:n].ctypes.data if n else None, None,
                                I.ctypes.data, C_byref(mat), 0)
        lp = np.array([0, 5, 0], np.float32); lc = np.ones(3, np.float32)
        osc.L.orc_add_light(osc.h, lp.ctypes.data, lc.ctypes.data)
        tw = T.Twin(hs)
        for accel in (rt.ACCEL_LBVH, rt.ACCEL_BRUTE):
            hits, rad = tw.trace_rays(rays_o, rays_d, 4, accel)
            ref = osc.find_closest_hit(rays_o, rays_d)
            ref_rad, _ = osc.trace_rays(rays_o, rays_d, 4)
            parity.check_hits(ref, hits, f"n={n}")
            np.testing.assert_allclose(rad, ref_rad, rtol=1e-5, atol=1e-9)
        if n == 2:
            assert list(ref["hit"]) == [1, 1, 1, 0, 0, 0]
            assert ref["frag"][2] in (0, 1)  # the shared diagonal: decided by the tie rule, equal on both sides


@pytest.mark.parametrize("builder", ["radix", "sah"])
def test_tall_tree_equals_brute_force(builder, monkeypatch):
    """A mesh built to give a tall radix tree (Morton chain + equal codes): tree and linear scan agree on every
    ray, and the traversal stack (one entry per level at most) holds.  The surface-area build of the same
    leaves stays low."""
    monkeypatch.setenv("RT_BUILD", builder)
    hs, o, d = parity.deep_tree_scene()
    tw = T.Twin(hs)
    assert (30 if builder == "radix" else 1) <= tw.height <= 64
    a, ra = tw.trace_rays(o, d, 4, rt.ACCEL_LBVH)
    deepest, height = tw.max_stack, tw.height
    b, rb = tw.trace_rays(o, d, 4, rt.ACCEL_BRUTE)
    assert a.tobytes() == b.tobytes() and np.array_equal(ra, rb, equal_nan=True)
    assert (a["hit"] == 1).all() and (a["object"] == 0).sum() > 1000  # most rays end on the mesh
    assert 1 <= deepest <= height


@pytest.mark.parametrize("seed", range(8))
def test_random_scene_equals_oracle(seed):
    """Random mixes of every object kind, transformation and material: the twin (tree and linear scan) against
    the oracle on camera rays, rays through mesh vertices and edge midpoints, and random rays."""
    sp, what = random_scenes.build(seed)
    o, d = random_scenes.rays(sp, seed)
    ref = sp.orc.find_closest_hit(o, d)
    ref_rad, _ = sp.orc.trace_rays(o, d, 6)
    tw = T.Twin(sp.host)
    for accel in (rt.ACCEL_LBVH, rt.ACCEL_BRUTE):
        hits, rad = tw.trace_rays(o, d, 6, accel)
        parity.check_hits(ref, hits, what)
        parity.check_radiance(ref_rad, rad, what)
    assert (ref["hit"] == 1).mean() > 0.5


@pytest.mark.parametrize("n_lights,with_sphere", [(11, False), (9, True)])
def test_more_lights_than_the_scene_record_holds(n_lights, with_sphere):
    """Lights beyond the scene record's own table live right behind the record: a frame under 9 and 11 lights (with
    and without the wave-wide skipping of shadow queries) against the oracle's."""
    sp, what = random_scenes.many_lights(n_lights, with_sphere)
    ref_rgb, ref_rgb8, _ = sp.orc.render(64, 48, mode=rt.MODE_GRID, grid_m=2, max_depth=4)
    tw = T.Twin(sp.host)
    rgb, rgb8, _ = tw.render(64, 48, mode=rt.MODE_GRID, grid_m=2, max_depth=4)
    parity.check_radiance(ref_rgb, rgb, what)
    parity.check_rgb8(ref_rgb8, rgb8, what)
    assert (np.asarray(ref_rgb).reshape(-1, 3).sum(axis=1) > 0).mean() > 0.9


@pytest.mark.parametrize("seed", range(3))
def test_sphere_ball_test_drops_no_hit(seed):
    """Spheres and cones under hard transformations, rays that graze them: the ball tests in front of the exact
    tests (`misses_ball`, `misses_cone`) never drop a hit the oracle, which has no such test, finds."""
    sp, o, d = random_scenes.ellipsoids(seed)
    ref = sp.orc.find_closest_hit(o, d)
    ref_rad, _ = sp.orc.trace_rays(o, d, 6)
    tw = T.Twin(sp.host)
    hits, rad = tw.trace_rays(o, d, 6, rt.ACCEL_LBVH)
    parity.check_hits(ref, hits, "ellipsoids")
    parity.check_radiance(ref_rad, rad, "ellipsoids")
    on_sphere = (ref["hit"] == 1) & (ref["object"] >= 1) & (ref["object"] <= 7)
    on_cone = (ref["hit"] == 1) & (ref["object"] >= 8)
    assert 0.03 < on_sphere.mean() < 0.9 and 0.03 < on_cone.mean() < 0.9  # grazing rays fall on both sides


def C_byref(x):
    import ctypes
    return ctypes.byref(x)


def test_invalid_scene_rejected(host_scene):
    import copy
    import ctypes as C
    d = host_scene("teapot").flatten().contents
    bad = rt.SceneDesc.from_buffer_copy(d)
    bad.n_materials = 0
    err = C.create_string_buffer(256)
    assert not T.lib().twin_create(C.byref(bad), err, 256) and b"material" in err.value
    bad = rt.SceneDesc.from_buffer_copy(d)
    meshes = (rt.Mesh * 1)(d.meshes[0])
    meshes[0].tri_count = d.n_triangles + 1
    bad.meshes = meshes
    assert not T.lib().twin_create(C.byref(bad), err, 256) and b"range" in err.value
