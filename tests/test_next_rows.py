"""Parity of the two widened rows either side of the hot path:

* the OBJ loader's non-polluting mode (the reference's parser, main.cpp:796-913, also appends `vt` / `vn`
  lines to the vertex list; `compat=False` does not): the faces this mode produces are handed to the oracle
  as explicit fragments, so the oracle applies the reference's Triangle / Diamond / Mesh / Box rules to the
  same triangles, and hits, radiance and a frame are compared;
* a sequence of frames of moving objects (the reference's frames/ + *.mp4 workflow, .gitignore:3-4, re-runs
  main() per frame): after `rt_scene_update` the library must render what the oracle renders for a scene
  built from scratch in the new pose.

Each comparison runs on the CPU twin (no GPU needed) and, marked gpu, through the C ABI on the device.
"""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as O
import parity
import ray_tracer_amd as rt

SEASHELL = os.path.join(rt.ASSETS, "seashell_obj.obj")


def _omat(m):
    om = O.OrcMaterial()
    om.ambient[:] = m.ambient[:]; om.diffuse[:] = m.diffuse[:]; om.specular[:] = m.specular[:]
    om.shininess, om.reflectiveness = m.shininess, m.reflectiveness
    om.refractiveness, om.refractive_ratio, om.texture = m.refractiveness, m.refractive_ratio, m.texture
    return om


def _mesh_arrays(desc, mesh):
    """corners, normals, uv and (fragment, sub) of one mesh of a flattened scene description"""
    n = desc.n_triangles
    m = desc.meshes[mesh]
    sl = slice(m.tri_offset, m.tri_offset + m.tri_count)
    pos = np.ctypeslib.as_array(desc.tri_pos, shape=(n * 9,)).reshape(n, 9)[sl].copy()
    nrm = np.ctypeslib.as_array(desc.tri_nrm, shape=(n * 9,)).reshape(n, 9)[sl].copy()
    uv = np.ctypeslib.as_array(desc.tri_uv, shape=(n * 6,)).reshape(n, 6)[sl].copy()
    frag = np.ctypeslib.as_array(desc.tri_frag, shape=(n,))[sl]
    sub = np.ctypeslib.as_array(desc.tri_sub, shape=(n,))[sl]
    return pos, nrm, uv, np.ascontiguousarray(np.stack([frag, sub], 1), dtype=np.int32)


def _room(add_plane, add_light):
    add_plane((0, -6, 0), (0, 1, 0), rt.Material(ambient=(0.02, 0.02, 0.02), diffuse=(0.7, 0.7, 0.6)))
    add_plane((0, 0, 40), (0, 0, -1), rt.Material(ambient=(0.02, 0.03, 0.02), diffuse=(0.3, 0.8, 0.4)))
    add_light((10, 16, -7), (0.6, 0.6, 0.6))
    add_light((-6, 9, 2), (0.3, 0.3, 0.3))


def clean_seashell_pair():
    """-> (host scene, oracle scene): seashell_obj.obj parsed WITHOUT the vertex pollution, behind its Box gate, in a
    small room; the oracle gets the very triangles the host's parser produced."""
    mat = rt.Material(ambient=(0.01, 0.02, 0.02), diffuse=(0.8, 0.2, 0.2), specular=0.5, shininess=10.0)
    M = rt.scale(0.15, rt.rotate(-1.2490457, (1, 0, 0), rt.translate((-2, 1, 22))))
    host = rt.HostScene()
    host.add_mesh_obj(SEASHELL, M, mat, boxed=True, compat=False)
    _room(host.add_plane, host.add_light)
    desc = host.flatten().contents
    pos, nrm, uv, fragsub = _mesh_arrays(desc, 0)
    orc = O.Scene()
    om = _omat(mat)
    Mf = O.f32(M)
    assert orc.L.orc_add_mesh_frags(orc.h, len(pos), pos.ctypes.data, nrm.ctypes.data, uv.ctypes.data, fragsub.ctypes.data,
                                    Mf.ctypes.data, C.byref(om), 1) == 0

    def oplane(p, n, m):
        p, n, mm = O.f32(p), O.f32(n), _omat(m)
        orc.L.orc_add_plane(orc.h, p.ctypes.data, n.ctypes.data, C.byref(mm))

    def olight(p, c):
        p, c = O.f32(p), O.f32(c)
        orc.L.orc_add_light(orc.h, p.ctypes.data, c.ctypes.data)
    _room(oplane, olight)
    return host, orc, desc


def test_clean_parse_differs_from_the_reference_parse():
    host, _, desc = clean_seashell_pair()
    pos, _, uv, fragsub = _mesh_arrays(desc, 0)
    assert len(pos) == 2 * 7536 and fragsub[:, 0].max() == 7535  # 7536 quad faces, two triangles each
    assert np.abs(uv).max() > 0  # texture coordinates come from the vt lines, not from polluted vertices
    polluted = rt.HostScene()
    polluted.add_mesh_obj(SEASHELL, rt.mat_identity(), rt.Material(), boxed=True, compat=True)
    ppos = _mesh_arrays(polluted.flatten().contents, 0)[0]
    assert ppos.shape == pos.shape and not np.array_equal(ppos, pos)
    # a clean parse never reads a vt / vn row as a position: every corner is one of the file's 7542 v rows
    vs = np.array([[float(x) for x in l.split()[1:4]] for l in open(SEASHELL) if l.startswith("v ")], np.float32)
    assert len(vs) == 7542
    corners = {tuple(c) for c in pos.reshape(-1, 3)}
    assert corners <= {tuple(v) for v in vs}


def _rays(orc, desc, w, h, seed, n_random):
    """parity.ray_set plus the same rays with unit directions: a plane reports the ray parameter and a mesh the
    Euclidean distance (main.cpp:235, 545), so rays aimed at a mesh with long directions mostly end on the walls"""
    o, d = parity.ray_set(orc, desc, w=w, h=h, seed=seed, n_random=n_random)
    n = np.linalg.norm(d, axis=1, keepdims=True)
    return np.concatenate([o, o]), np.concatenate([d, d / np.maximum(n, 1e-20)]).astype(np.float32)


def _compare(backend_trace, backend_frame, orc, desc, what, w=96, h=72, max_depth=4):
    o, d = _rays(orc, desc, 48, 36, 3, 600)
    ref_hits = orc.find_closest_hit(o, d)
    ref_rad, _ = orc.trace_rays(o, d, max_depth)
    assert (ref_hits["object"] == 0).sum() > 200, "the ray set must reach the mesh"
    hits, rad = backend_trace(o, d, max_depth)
    parity.check_hits(ref_hits, hits, what)
    parity.check_radiance(ref_rad, rad, what)
    ref_rgb, ref_rgb8, _ = orc.render(w, h, mode=0, max_depth=max_depth)
    rgb, rgb8 = backend_frame(w, h, max_depth)
    parity.check_radiance(ref_rgb, rgb, what + " frame")
    parity.check_rgb8(ref_rgb8, rgb8, what + " frame")


def test_clean_obj_parse_matches_oracle_on_twin():
    import twin_lib as T
    host, orc, desc = clean_seashell_pair()
    tw = T.Twin(host)
    _compare(lambda o, d, md: tw.trace_rays(o, d, md), lambda w, h, md: tw.render(w, h, mode=rt.MODE_REF_1SPP, max_depth=md)[:2],
             orc, desc, "clean seashell, twin")


@pytest.mark.gpu
@pytest.mark.parametrize("accel", [rt.ACCEL_LBVH, rt.ACCEL_BRUTE])
def test_clean_obj_parse_matches_oracle_on_gpu(accel):
    host, orc, desc = clean_seashell_pair()
    ctx = rt.Context(0)
    try:
        ctx.upload(host)
        _compare(lambda o, d, md: ctx.trace_rays(o, d, md, accel=accel),
                 lambda w, h, md: ctx.render(w, h, mode=rt.MODE_REF_1SPP, max_depth=md, accel=accel),
                 orc, desc, f"clean seashell, gpu accel {accel}")
    finally:
        ctx.close()


# ---- frames of moving objects -----------------------------------------------------------------------

def _pose(k):
    """placement of the moving objects in frame k"""
    a = 0.5 * k
    mesh_M = rt.scale((1.0 + 0.1 * k, 1.0, 1.0), rt.rotate(a, (0.2, 1, 0.1), rt.translate((-2 + 0.8 * k, 0.5 * k - 1, 12 + k))))
    ball_M = rt.scale(1.0 + 0.25 * k, rt.translate((3 - 0.5 * k, -1 + 0.3 * k, 9)))
    cone_M = rt.scale((1, 2, 1), rt.rotate(0.3 * k, (0, 0, 1), rt.translate((0.5 * k, 3, 14))))
    light = (6 - 2 * k, 12, -3 + k)
    wall_z = 30 - 2 * k
    return mesh_M, ball_M, cone_M, light, wall_z


def moving_pair(k):
    """-> ScenePair of frame k: a closed smooth mesh behind a Box gate, a mirror sphere, a cone, two walls, one light"""
    import random_scenes as R
    rng = np.random.default_rng(77)
    pos, nrm = R._shell(rng, 9, 12)
    mesh_M, ball_M, cone_M, light, wall_z = _pose(k)
    sp = R.ScenePair()
    sp.mesh(pos, nrm, mesh_M, rt.Material(ambient=(0.02, 0.02, 0.04), diffuse=(0.3, 0.4, 0.9), specular=0.4, shininess=30.0), True)
    sp.sphere(ball_M, rt.Material(ambient=(0.03, 0.03, 0.03), diffuse=(0.8, 0.8, 0.8), specular=0.5, shininess=40.0, reflectiveness=0.4))
    sp.cone(cone_M, rt.Material(ambient=(0.03, 0.05, 0.02), diffuse=(0.4, 0.9, 0.3)))
    sp.plane((0, -3, 0), (0, 1, 0), rt.Material(ambient=(0.02, 0.02, 0.02), diffuse=(0.7, 0.6, 0.5)))
    sp.plane((0, 0, wall_z), (0.1 * k, 0, -1), rt.Material(ambient=(0.03, 0.02, 0.02), diffuse=(0.9, 0.5, 0.4)))
    sp.light(light, (0.8, 0.8, 0.7))
    return sp


def _frames_match_oracle(upload, update, trace, frame, what):
    w, h, md = 80, 60, 5
    for k in range(3):
        sp = moving_pair(k)
        if k == 0:
            upload(sp.host)
        else:
            update(sp.host)  # only the small tables change; triangles and the tree are kept
        desc = sp.host.flatten().contents
        o, d = _rays(sp.orc, desc, 40, 30, k, 400)
        ref_hits = sp.orc.find_closest_hit(o, d)
        ref_rad, _ = sp.orc.trace_rays(o, d, md)
        assert (ref_hits["object"] == 0).sum() > 100
        hits, rad = trace(o, d, md)
        parity.check_hits(ref_hits, hits, f"{what} frame {k}")
        parity.check_radiance(ref_rad, rad, f"{what} frame {k}")
        ref_rgb, ref_rgb8, _ = sp.orc.render(w, h, mode=1, max_depth=md)
        rgb, rgb8 = frame(w, h, md)
        parity.check_radiance(ref_rgb, rgb, f"{what} frame {k} image")
        parity.check_rgb8(ref_rgb8, rgb8, f"{what} frame {k} image")


@pytest.mark.gpu
def test_updated_scene_matches_oracle_of_the_same_pose_on_gpu():
    ctx = rt.Context(0)
    try:
        _frames_match_oracle(ctx.upload, ctx.update, lambda o, d, md: ctx.trace_rays(o, d, md),
                             lambda w, h, md: ctx.render(w, h, mode=rt.MODE_REF_AA9, max_depth=md), "gpu")
    finally:
        ctx.close()


def test_every_pose_matches_oracle_on_twin():
    """The twin has no update call (it is rebuilt per pose): this checks the poses themselves on the CPU, so that a
    failure of the GPU test above points at rt_scene_update and not at the scene."""
    import twin_lib as T
    state = {}

    def fresh(host):
        state["tw"] = T.Twin(host)
    _frames_match_oracle(fresh, fresh, lambda o, d, md: state["tw"].trace_rays(o, d, md),
                         lambda w, h, md: state["tw"].render(w, h, mode=rt.MODE_REF_AA9, max_depth=md)[:2], "twin")
