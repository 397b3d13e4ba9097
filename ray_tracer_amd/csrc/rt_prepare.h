// rt_prepare.h -- host-side preparation of a scene description for the kernels: validation of
// everything a kernel will index with, per-mesh bounds and padding, the Box gate triangles
// (Box::Box, main.cpp:568-595) and the node ranges of the per-mesh LBVHs.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "rt_core.h"
#include "rt_lbvh.h"

namespace rtc {

struct MeshBuild {
    int node_offset = 0;
    float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};  // bounds of the finite vertices (Morton normalisation)
    float pad = 0;                                // leaf box padding
    int height = 0;                               // LBVH height after the build
};

// Returns an empty string when the description is consistent, else what is wrong.
inline std::string validate_scene(const rt_scene_desc *sc) {
    if (sc->n_objects < 0 || sc->n_spheres < 0 || sc->n_planes < 0 || sc->n_cones < 0 || sc->n_meshes < 0 ||
        sc->n_triangles < 0 || sc->n_materials <= 0 || sc->n_textures < 0 || sc->n_lights < 0)
        return "negative count (or no materials) in scene description";
    auto mat_ok = [&](int m) { return m >= 0 && m < sc->n_materials; };
    for (int i = 0; i < sc->n_objects; i++) {
        rt_object ob = sc->objects[i];
        int lim = ob.kind == RT_OBJ_SPHERE ? sc->n_spheres : ob.kind == RT_OBJ_PLANE ? sc->n_planes
                  : ob.kind == RT_OBJ_CONE ? sc->n_cones : ob.kind == RT_OBJ_MESH ? sc->n_meshes : -1;
        if (ob.index < 0 || ob.index >= lim) return "object " + std::to_string(i) + ": bad kind/index";
    }
    for (int i = 0; i < sc->n_spheres; i++) if (!mat_ok(sc->spheres[i].material)) return "sphere material out of range";
    for (int i = 0; i < sc->n_planes; i++) if (!mat_ok(sc->planes[i].material)) return "plane material out of range";
    for (int i = 0; i < sc->n_cones; i++)
        if (!mat_ok(sc->cones[i].material) || !mat_ok(sc->cones[i].cap.material)) return "cone material out of range";
    for (int i = 0; i < sc->n_triangles; i++) {
        if (!mat_ok(sc->tri_material[i])) return "triangle material out of range";
        if (sc->tri_sub[i] < 0 || sc->tri_sub[i] > 1 || sc->tri_frag[i] < 0 || sc->tri_frag[i] > 0x3fffffff)
            return "triangle frag/sub out of range";
    }
    for (int i = 0; i < sc->n_meshes; i++) {
        const rt_mesh &m = sc->meshes[i];
        if (m.tri_offset < 0 || m.tri_count < 0 || (long long)m.tri_offset + m.tri_count > sc->n_triangles)
            return "mesh triangle range out of bounds";
    }
    for (int i = 0; i < sc->n_materials; i++)
        if (sc->materials[i].texture < -1 || sc->materials[i].texture >= sc->n_textures) return "material texture out of range";
    for (int i = 0; i < sc->n_textures; i++) {
        const rt_texture &t = sc->textures[i];
        if (t.kind < RT_TEX_CHECKER || t.kind > RT_TEX_IMAGE) return "bad texture kind";
        if (t.kind == RT_TEX_IMAGE && (!t.base_color || !t.normal || !t.occlusion || !t.roughness || t.width == 0 || t.height == 0 ||
                                       (uint64_t)t.width * t.height > (1ull << 28)))
            return "image texture needs four maps and a sane size";
    }
    return "";
}

inline void make_gate_tris(const float *v1, const float *v2, f4 *out) {
    float A[3] = {v1[0], v1[1], v1[2]}, B[3] = {v2[0], v1[1], v1[2]}, C[3] = {v2[0], v2[1], v1[2]},
          D[3] = {v1[0], v2[1], v1[2]}, G[3] = {v2[0], v2[1], v2[2]}, E[3] = {v1[0], v1[1], v2[2]},
          F[3] = {v2[0], v1[1], v2[2]}, H[3] = {v1[0], v2[1], v2[2]};
    const float *q[12][3] = {{B, A, D}, {C, B, D}, {F, B, C}, {G, F, C}, {E, F, G}, {H, E, G},
                             {A, E, H}, {D, A, H}, {C, D, H}, {G, C, H}, {F, E, A}, {B, F, A}};
    for (int i = 0; i < 12; i++) {
        float p[9];
        for (int k = 0; k < 3; k++)
            for (int a = 0; a < 3; a++) p[3 * k + a] = q[i][k][a];
        make_tri_test(p, out + 3 * i);
    }
}

// which kernel features (F_*) a scene needs
inline int scene_features(const rt_scene_desc *sc) {
    int f = 0;
    for (int i = 0; i < sc->n_objects; i++) {
        int k = sc->objects[i].kind;
        if (k == RT_OBJ_MESH) f |= F_MESH;
        if (k == RT_OBJ_SPHERE || k == RT_OBJ_CONE) f |= F_QUADRIC;
    }
    for (int i = 0; i < sc->n_materials; i++) {
        const rt_material &m = sc->materials[i];
        if (m.texture >= 0) f |= F_TEXTURE;
        if (m.reflectiveness > 0 || m.refractiveness > 0) f |= F_SECONDARY;
    }
    return f;
}

// object index of every plane / sphere / cone / mesh, in the order of their own arrays
inline void kind_to_object(const rt_scene_desc *sc, std::vector<int32_t> &plane_obj, std::vector<int32_t> &sphere_obj,
                           std::vector<int32_t> &cone_obj, std::vector<int32_t> &mesh_obj) {
    plane_obj.assign(sc->n_planes, -1); sphere_obj.assign(sc->n_spheres, -1);
    cone_obj.assign(sc->n_cones, -1); mesh_obj.assign(sc->n_meshes, -1);
    for (int i = 0; i < sc->n_objects; i++) {
        rt_object ob = sc->objects[i];
        std::vector<int32_t> &v = ob.kind == RT_OBJ_PLANE ? plane_obj : ob.kind == RT_OBJ_SPHERE ? sphere_obj
                                  : ob.kind == RT_OBJ_CONE ? cone_obj : mesh_obj;
        if (v[ob.index] < 0) v[ob.index] = i;
    }
}

// Fills the device-side mesh records (everything but device pointers), the per-mesh build data and
// the gate triangles (36 f4 per mesh).  Returns the total number of LBVH nodes.
inline int prepare_meshes(const rt_scene_desc *sc, std::vector<DevMesh> &meshes, std::vector<MeshBuild> &build,
                          std::vector<f4> &gate) {
    meshes.assign(sc->n_meshes, DevMesh{});
    build.assign(sc->n_meshes, MeshBuild{});
    gate.assign(36 * (size_t)sc->n_meshes, f4{0, 0, 0, 0});
    int node_total = 0;
    for (int i = 0; i < sc->n_meshes; i++) {
        const rt_mesh &m = sc->meshes[i];
        DevMesh &dm = meshes[i];
        MeshBuild &mb = build[i];
        std::memcpy(dm.to_local, m.to_local, sizeof dm.to_local);
        std::memcpy(dm.to_world, m.to_world, sizeof dm.to_world);
        std::memcpy(dm.normal_mat, m.normal_mat, sizeof dm.normal_mat);
        std::memcpy(dm.box_to_local, m.box_to_local, sizeof dm.box_to_local);
        dm.tri_offset = m.tri_offset; dm.tri_count = m.tri_count; dm.boxed = m.boxed;
        dm.node_offset = node_total; dm.gate_tri_offset = 12 * i;
        dm.pad[0] = dm.pad[1] = dm.pad[2] = 0;
        mb.node_offset = node_total;
        node_total += std::max(m.tri_count - 1, 0);
        float maxabs = 0;
        for (int a = 0; a < 3; a++) { mb.lo[a] = INFINITY; mb.hi[a] = -INFINITY; }
        for (int t = 0; t < m.tri_count; t++) {
            const float *p = sc->tri_pos + 9 * (size_t)(m.tri_offset + t);
            for (int v = 0; v < 3; v++) {
                if (!(std::isfinite(p[3 * v]) && std::isfinite(p[3 * v + 1]) && std::isfinite(p[3 * v + 2]))) continue;
                for (int a = 0; a < 3; a++) {
                    mb.lo[a] = std::min(mb.lo[a], p[3 * v + a]);
                    mb.hi[a] = std::max(mb.hi[a], p[3 * v + a]);
                    maxabs = std::max(maxabs, std::fabs(p[3 * v + a]));
                }
            }
        }
        mb.pad = std::max(maxabs * 2e-5f, 1e-30f);
        float gmax = 0;
        for (int a = 0; a < 3; a++) gmax = std::max(gmax, std::max(std::fabs(m.box_min[a]), std::fabs(m.box_max[a])));
        float gpad = std::isfinite(gmax) ? std::max(gmax * 2e-5f, 1e-30f) : 0.0f;
        for (int a = 0; a < 3; a++) {
            dm.gate_lo[a] = m.box_min[a] - gpad; dm.gate_hi[a] = m.box_max[a] + gpad;
            dm.gate_in_lo[a] = m.box_min[a] + gpad; dm.gate_in_hi[a] = m.box_max[a] - gpad;
        }
        make_gate_tris(m.box_min, m.box_max, gate.data() + 36 * (size_t)i);
    }
    return node_total;
}

}  // namespace rtc
