"""Seeded random scenes, built twice from one description: as a `ray_tracer_amd.HostScene` (what the GPU
library and the CPU twin take) and as an oracle scene.  Test scaffolding.

Every scene mixes the object kinds of the reference (planes with axis-aligned and general normals, spheres,
cones, triangle meshes with and without a Box gate) under random affine transformations, with diffuse,
specular, reflective and refractive materials, procedural textures and one to three lights.
"""
import ctypes as C

import numpy as np

import oracle_lib as O
import ray_tracer_amd as rt


class ScenePair:
    def __init__(self):
        self.host = rt.HostScene()
        self.orc = O.Scene()
        self._keep = []

    def _omat(self, m):
        om = O.OrcMaterial()
        om.ambient[:] = m.ambient[:]; om.diffuse[:] = m.diffuse[:]; om.specular[:] = m.specular[:]
        om.shininess, om.reflectiveness = m.shininess, m.reflectiveness
        om.refractiveness, om.refractive_ratio, om.texture = m.refractiveness, m.refractive_ratio, m.texture
        return om

    def checker(self):
        a, b = self.host.add_checker_texture(), self.orc.L.orc_add_checker_texture(self.orc.h)
        assert a == b
        return a

    def rainbow(self):
        a, b = self.host.add_rainbow_texture(), self.orc.L.orc_add_rainbow_texture(self.orc.h)
        assert a == b
        return a

    def plane(self, p, n, m):
        p, n = O.f32(p), O.f32(n)
        om = self._omat(m)
        a = self.host.add_plane(p, n, m)
        b = self.orc.L.orc_add_plane(self.orc.h, p.ctypes.data, n.ctypes.data, C.byref(om))
        assert a == b
        return a

    def sphere(self, M, m):
        M = O.f32(M); om = self._omat(m)
        a = self.host.add_sphere(M, m)
        b = self.orc.L.orc_add_sphere(self.orc.h, M.ctypes.data, C.byref(om))
        assert a == b
        return a

    def cone(self, M, m):
        M = O.f32(M); om = self._omat(m)
        a = self.host.add_cone(M, m)
        b = self.orc.L.orc_add_cone(self.orc.h, M.ctypes.data, C.byref(om))
        assert a == b
        return a

    def mesh(self, pos, nrm, M, m, boxed):
        pos, nrm, M = O.f32(pos).reshape(-1, 9), O.f32(nrm).reshape(-1, 9), O.f32(M)
        om = self._omat(m)
        a = self.host.add_mesh_tris(pos, nrm, M, m, boxed=boxed)
        b = self.orc.L.orc_add_mesh_tris(self.orc.h, len(pos), pos.ctypes.data, nrm.ctypes.data, None, M.ctypes.data,
                                         C.byref(om), int(boxed))
        assert a == b
        return a

    def light(self, p, c):
        p, c = O.f32(p), O.f32(c)
        self.host.add_light(p, c)
        self.orc.L.orc_add_light(self.orc.h, p.ctypes.data, c.ctypes.data)


def _placement(rng, spread=1.0):
    M = rt.translate((rng.uniform(-4, 4) * spread, rng.uniform(-2, 3), rng.uniform(6, 14)))
    M = rt.rotate(rng.uniform(0, 6.28), rng.normal(size=3), M)
    return rt.scale(rng.uniform(0.6, 1.8, 3), M)


def _material(rng, kind):
    col = rng.uniform(0.1, 1.0, 3)
    if kind == "mirror":
        return rt.Material(ambient=0.05 * col, diffuse=col, specular=0.5, shininess=20.0, reflectiveness=float(rng.uniform(0.2, 0.7)))
    if kind == "glass":
        return rt.Material(ambient=0.02 * col, diffuse=col, specular=0.3, shininess=50.0,
                           refractiveness=float(rng.choice([1.5, 0.825, 1.2])), refractive_ratio=float(rng.uniform(0.1, 0.5)))
    if kind == "shiny":
        return rt.Material(ambient=0.05 * col, diffuse=col, specular=float(rng.uniform(0.1, 1.0)), shininess=float(rng.uniform(1, 80)))
    return rt.Material(ambient=0.05 * col, diffuse=col)


def _soup(rng, n):
    """n triangles scattered in a cube of side 3; shading normals = the face normal, turned towards the
    camera side for two thirds of them (the others are back faces as seen from the camera)."""
    c = rng.uniform(-1.5, 1.5, (n, 1, 3))
    p = c + rng.uniform(-0.5, 0.5, (n, 3, 3))
    fn = np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0])
    fn /= np.maximum(np.linalg.norm(fn, axis=1, keepdims=True), 1e-20)
    flip = (fn[:, 2] > 0) ^ (rng.uniform(size=n) < 1 / 3)
    fn[flip] *= -1
    return p.reshape(n, 9).astype(np.float32), np.repeat(fn[:, None, :], 3, 1).reshape(n, 9).astype(np.float32)


def _shell(rng, nu, nv):
    """A closed, smooth, outward-oriented surface (a bumpy ellipsoid) as quads split into triangles, with
    smooth vertex normals: the from-outside rule, the Box gate and refraction all see a proper solid."""
    u = np.linspace(0, np.pi, nu + 1)[:, None]
    v = np.linspace(0, 2 * np.pi, nv + 1)[None, :]
    r = 1.0 + 0.15 * np.sin(3 * u) * np.cos(2 * v)
    P = np.stack([r * np.sin(u) * np.cos(v), r * np.cos(u) + 0 * v, r * np.sin(u) * np.sin(v)], -1)
    Nn = P / np.maximum(np.linalg.norm(P, axis=-1, keepdims=True), 1e-20)
    pos, nrm = [], []
    for i in range(nu):
        for j in range(nv):
            q = [(i, j), (i + 1, j), (i + 1, j + 1), (i, j + 1)]
            for a, b, c in ((0, 2, 1), (0, 3, 2)):  # outward winding
                pos.append(np.concatenate([P[q[a]], P[q[b]], P[q[c]]]))
                nrm.append(np.concatenate([Nn[q[a]], Nn[q[b]], Nn[q[c]]]))
    return np.asarray(pos, np.float32), np.asarray(nrm, np.float32)


def build(seed):
    """-> (ScenePair, description string)"""
    rng = np.random.default_rng(1000 + seed)
    sp = ScenePair()
    what = []
    checker = sp.checker() if seed % 2 == 0 else -1
    rainbow = sp.rainbow() if seed % 3 == 0 else -1
    floor = _material(rng, "mirror" if seed % 4 == 1 else "diffuse")
    if checker >= 0:
        floor.texture = checker
    sp.plane((0, -3, 0), (0, 1, 0), floor)
    n = rng.normal(size=3); n[2] = -abs(n[2]) - 0.5; n /= np.linalg.norm(n)
    back = _material(rng, "diffuse")
    if rainbow >= 0:
        back.texture = rainbow
    sp.plane((0, 0, 22), n, back)
    if seed % 2:
        sp.plane((-9, 0, 0), (1, 0, 0), _material(rng, "shiny"))
    kinds = ["diffuse", "shiny", "mirror", "glass"]
    for k in range(int(rng.integers(0, 4))):
        sp.sphere(_placement(rng), _material(rng, kinds[(seed + k) % 4]))
        what.append("sphere")
    for k in range(int(rng.integers(0, 3))):
        sp.cone(_placement(rng), _material(rng, kinds[(seed + k + 1) % 4]))
        what.append("cone")
    for k in range(int(rng.integers(1, 3))):
        boxed = bool(rng.integers(0, 2))
        if (seed + k) % 2:
            pos, nrm = _soup(rng, int(rng.integers(40, 250)))
            m = _material(rng, kinds[(seed + k) % 2])
        else:
            pos, nrm = _shell(rng, int(rng.integers(5, 12)), int(rng.integers(6, 16)))
            m = _material(rng, kinds[(seed + k) % 4])
        sp.mesh(pos, nrm, _placement(rng, 0.7), m, boxed)
        what.append(("boxed " if boxed else "") + "mesh")
    for k in range(int(rng.integers(1, 4))):
        sp.light((rng.uniform(-8, 8), rng.uniform(4, 14), rng.uniform(-4, 12)), rng.uniform(0.2, 1.0, 3))
    return sp, f"seed {seed}: " + ", ".join(what)


def many_lights(n_lights, with_sphere):
    """A room of axis planes with a smooth closed mesh (and, with_sphere, a sphere, which keeps every shadow query)
    under n_lights lights: more than a scene record's own light table holds."""
    rng = np.random.default_rng(77 + n_lights + (1 if with_sphere else 0))
    sp = ScenePair()
    grey = _material(rng, "diffuse"); shiny = _material(rng, "shiny")
    sp.plane((0, -3, 0), (0, 1, 0), grey); sp.plane((0, 0, 24), (0, 0, -1), grey); sp.plane((-12, 0, 0), (1, 0, 0), shiny)
    sp.plane((12, 0, 0), (-1, 0, 0), grey); sp.plane((0, 18, 0), (0, -1, 0), grey)
    pos, nrm = _shell(rng, 10, 14)
    M = rt.matmul(rt.translate((0.5, 0.5, 11.0)), rt.scale((2.5, 2.0, 2.5)))
    sp.mesh(pos, nrm, M, shiny, False)
    if with_sphere:
        sp.sphere(rt.translate((-4.0, -1.0, 9.0)), _material(rng, "diffuse"))
    for k in range(n_lights):
        a = 2 * np.pi * k / n_lights
        sp.light((7 * np.cos(a), 9 + 4 * np.sin(3 * a), 9 + 7 * np.sin(a)), rng.uniform(0.05, 0.25, 3))
    return sp, f"{n_lights} lights" + (", sphere" if with_sphere else "")


def rays(sp, seed, w=72, h=54, n_random=1200):
    """Camera rays of a small frame, rays aimed at mesh vertices and edge midpoints, and random rays from random origins."""
    import parity
    rng = np.random.default_rng(2000 + seed)
    o, d, _ = O.camera_rays(w, h)
    t = parity.mesh_world_targets(sp.host.flatten().contents, rng, per_mesh=150)
    ro = rng.uniform(-6, 6, (n_random, 3)) * np.array([1, 0.5, 1]) + np.array([0, 2, 8])
    rd = rng.normal(size=(n_random, 3))
    return (np.concatenate([o, np.zeros_like(t), ro]).astype(np.float32),
            np.concatenate([d, t, rd]).astype(np.float32))


def ellipsoids(seed):
    """-> (ScenePair, origins, directions): spheres and cones under hard transformations (axis ratios up to 60:1, turned,
    tiny and far away, huge and near, far from the world's origin) in front of a back wall, and rays that graze
    them: through points at (1 + e) times the tangent points seen from each ray's origin, e from -1e-2 to 1e-2,
    plus rays from inside the spheres and rays that leave them behind."""
    rng = np.random.default_rng(3000 + seed)
    sp = ScenePair()
    sp.plane((0, 0, 4000), (0, 0, -1), _material(rng, "diffuse"))
    mats = []
    for k in range(7):
        centre = [(0, 0, 12), (300, -80, 900), (-2, 1, 6), (1500, 1500, 3000), (0.5, -0.5, 3), (-40, 10, 60), (3, 2, 20)][k]
        scl = [(1, 1, 1), (0.05, 0.05, 0.05), (3, 0.05, 1), (60, 1, 20), (0.02, 0.5, 0.1), (8, 8, 8), (1, 2.5, 0.7)][k]
        M = rt.translate(centre)
        M = rt.rotate(rng.uniform(0, 6.28), rng.normal(size=3), M)
        M = rt.scale(scl, M)
        sp.sphere(M, _material(rng, ["diffuse", "shiny", "mirror", "glass"][k % 4]))
        mats.append(np.asarray(M, np.float64).reshape(4, 4).T)  # (stored by columns)
    cones = []
    for k in range(4):
        centre = [(5, -2, 15), (-300, 40, 800), (-1, 3, 5), (20, 20, 70)][k]
        scl = [(1, 3, 1), (2, 30, 2), (0.3, 0.1, 2), (6, 6, 6)][k]
        M = rt.translate(centre)
        M = rt.rotate(rng.uniform(0, 6.28), rng.normal(size=3), M)
        M = rt.scale(scl, M)
        sp.cone(M, _material(rng, ["shiny", "diffuse", "mirror", "glass"][k]))
        cones.append(np.asarray(M, np.float64).reshape(4, 4).T)
    sp.light((0, 50, -10), (1, 1, 1))
    os_, ds_ = [], []
    for M in cones:  # rays that graze the cone's surface at heights -3 .. 3 (apex 0, cap 1), shallow and steep
        Mi = np.linalg.inv(M)
        for _ in range(60):
            y = rng.choice([-3, -1, -0.3, -0.01, 0.01, 0.3, 0.7, 0.99, 1.01, 1.5, 3])
            phi = rng.uniform(0, 6.28)
            surf = np.array([abs(y) * np.cos(phi), y, abs(y) * np.sin(phi)])
            nrm = np.array([np.cos(phi), -np.sign(y), np.sin(phi)]) / np.sqrt(2)  # outward normal of the nappe
            tang = np.cross(nrm, rng.normal(size=3)); tang /= np.linalg.norm(tang)
            dist = rng.choice([3.0, 30.0, 300.0, 3000.0, 30000.0])
            for e in (-1e-2, -1e-3, -1e-4, -1e-5, 0.0, 1e-5, 1e-4, 1e-3, 1e-2):
                Pl = surf + nrm * e * max(abs(y), 0.1)
                Ol = Pl - tang * dist
                O = M[:3, :3] @ Ol + M[:3, 3]; P = M[:3, :3] @ Pl + M[:3, 3]
                os_.append(O); ds_.append(P - O)
                os_.append(O); ds_.append((P - O) / np.linalg.norm(P - O))
        for _ in range(40):  # and rays at the ball around apex..cap from outside
            O = rng.uniform(-30, 30, 3) + np.array([0, 0, -5])
            Pl = np.array([0, 0.5, 0]) + rng.normal(size=3) * 0.9
            P = M[:3, :3] @ Pl + M[:3, 3]
            os_.append(O); ds_.append(P - O)
    for M in mats:
        Mi = np.linalg.inv(M)
        for _ in range(60):
            O = rng.uniform(-30, 30, 3) * np.array([1, 1, 0.3]) + np.array([0, 0, -5])
            ol = Mi[:3, :3] @ O + Mi[:3, 3]
            n2 = ol @ ol
            if n2 <= 1.0:
                continue
            w = np.cross(ol, rng.normal(size=3)); w /= np.linalg.norm(w)
            u = ol / n2 + np.sqrt(1 - 1 / n2) * w  # |u| = 1 and u . ol = 1: a tangent point seen from O
            for e in (-1e-2, -1e-3, -1e-4, -1e-5, -1e-6, 0.0, 1e-6, 1e-5, 1e-4, 1e-3, 1e-2):
                P = M[:3, :3] @ (u * (1 + e)) + M[:3, 3]
                os_.append(O); ds_.append(P - O)
                os_.append(O); ds_.append((P - O) / np.linalg.norm(P - O))
            os_.append(O); ds_.append(O - (M[:3, 3]))  # away from it
        for _ in range(20):  # from inside
            u = rng.normal(size=3); u *= rng.uniform(0, 0.99) / np.linalg.norm(u)
            os_.append(M[:3, :3] @ u + M[:3, 3]); ds_.append(rng.normal(size=3))
    return sp, np.asarray(os_, np.float32), np.asarray(ds_, np.float32)
