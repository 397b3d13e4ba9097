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
    int ref_offset = 0, ref_count = 0;            // this mesh's leaves in the sorted arrays (>= its triangle count)
    float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};  // bounds of the finite vertices (Morton normalisation)
    float pad = 0;                                // leaf box padding
    int height = 0;                               // LBVH height after the build
    int root = 0;                                 // root node after the build
};

// The leaves of a mesh's LBVH ("references").  A triangle whose box is much larger than the mesh's
// typical triangle is cut into pieces (longest edge halved, recursively) and every piece becomes a
// leaf of its own with the piece's box; all pieces still test the WHOLE triangle, so a hit is found
// by whichever piece's box the ray enters and is the same hit.  Long thin triangles -- the
// reference's OBJ loader makes thousands from vt / vn lines (main.cpp:796-805) -- otherwise have
// boxes that cover most of the mesh and defeat the tree.
// ref_pos: 9 floats per reference (the piece), ref_tri: its triangle (index into the scene's arrays).
inline void split_piece(const float *p, float tau2, int depth, int tri, std::vector<float> &ref_pos, std::vector<int32_t> &ref_tri) {
    float box[6], d2 = 0;
    piece_box(p, box);
    for (int a = 0; a < 3; a++) d2 += (box[3 + a] - box[a]) * (box[3 + a] - box[a]);
    if (depth <= 0 || !(d2 > tau2) || !std::isfinite(d2)) {
        ref_pos.insert(ref_pos.end(), p, p + 9);
        ref_tri.push_back(tri);
        return;
    }
    int e = 0;
    float best = -1;
    for (int k = 0; k < 3; k++) {  // edge k: vertex k -> vertex (k + 1) % 3
        const float *u = p + 3 * k, *v = p + 3 * ((k + 1) % 3);
        float l2 = (u[0] - v[0]) * (u[0] - v[0]) + (u[1] - v[1]) * (u[1] - v[1]) + (u[2] - v[2]) * (u[2] - v[2]);
        if (l2 > best) { best = l2; e = k; }
    }
    const float *u = p + 3 * e, *v = p + 3 * ((e + 1) % 3), *w = p + 3 * ((e + 2) % 3);
    float m[3] = {0.5f * (u[0] + v[0]), 0.5f * (u[1] + v[1]), 0.5f * (u[2] + v[2])};
    float a[9] = {u[0], u[1], u[2], m[0], m[1], m[2], w[0], w[1], w[2]};
    float b[9] = {m[0], m[1], m[2], v[0], v[1], v[2], w[0], w[1], w[2]};
    split_piece(a, tau2, depth - 1, tri, ref_pos, ref_tri);
    split_piece(b, tau2, depth - 1, tri, ref_pos, ref_tri);
}
inline void make_mesh_refs(const rt_scene_desc *sc, const rt_mesh &m, std::vector<float> &ref_pos, std::vector<int32_t> &ref_tri) {
    std::vector<float> diag(std::max(m.tri_count, 0));
    for (int t = 0; t < m.tri_count; t++) {
        float box[6], d2 = 0;
        piece_box(sc->tri_pos + 9 * (size_t)(m.tri_offset + t), box);
        for (int a = 0; a < 3; a++) d2 += (box[3 + a] - box[a]) * (box[3 + a] - box[a]);
        diag[t] = std::isfinite(d2) ? d2 : 0.0f;
    }
    float tau2 = INFINITY;
    if (m.tri_count > 0) {
        std::vector<float> tmp = diag;
        std::nth_element(tmp.begin(), tmp.begin() + tmp.size() / 2, tmp.end());
        float median2 = tmp[tmp.size() / 2];
        if (median2 > 0) tau2 = 16.0f * median2;  // pieces at most 4 x the median triangle's diagonal
    }
    for (int attempt = 0;; attempt++) {
        size_t pos0 = ref_pos.size(), tri0 = ref_tri.size();
        for (int t = 0; t < m.tri_count; t++)
            split_piece(sc->tri_pos + 9 * (size_t)(m.tri_offset + t), tau2, 8, m.tri_offset + t, ref_pos, ref_tri);
        if (ref_tri.size() - tri0 <= 4 * (size_t)m.tri_count + 64 || attempt >= 8) break;
        ref_pos.resize(pos0); ref_tri.resize(tri0);  // too many pieces: coarser
        tau2 *= 4.0f;
    }
}

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
                                       t.width > 65535u || t.height > 65535u))
            return "image texture needs four maps and a size of 1..65535 texels a side";
    }
    return "";
}

// inverse of an affine map stored as four 3-row columns, in double precision; false if singular
inline bool invert_affine(const float *m, double *out) {
    double a[9];
    for (int i = 0; i < 9; i++) a[i] = m[i];  // columns c0 c1 c2
    double c00 = a[4] * a[8] - a[7] * a[5], c01 = a[7] * a[2] - a[1] * a[8], c02 = a[1] * a[5] - a[4] * a[2];
    double det = a[0] * c00 + a[3] * c01 + a[6] * c02;
    if (!(std::fabs(det) > 0) || !std::isfinite(det)) return false;
    double id = 1.0 / det;
    // inverse of the 3x3 block, stored by columns
    out[0] = c00 * id; out[1] = c01 * id; out[2] = c02 * id;
    out[3] = (a[6] * a[5] - a[3] * a[8]) * id; out[4] = (a[0] * a[8] - a[6] * a[2]) * id; out[5] = (a[3] * a[2] - a[0] * a[5]) * id;
    out[6] = (a[3] * a[7] - a[6] * a[4]) * id; out[7] = (a[6] * a[1] - a[0] * a[7]) * id; out[8] = (a[0] * a[4] - a[3] * a[1]) * id;
    for (int r = 0; r < 3; r++) out[9 + r] = -(out[r] * m[9] + out[3 + r] * m[10] + out[6 + r] * m[11]);
    for (int i = 0; i < 12; i++) if (!std::isfinite(out[i])) return false;
    return true;
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

// Largest stretch (spectral norm) of the linear part of a 3x4 map stored by columns: square root of the
// largest eigenvalue of A^T A by the trigonometric formula for symmetric 3x3 matrices, kept between the
// two elementary bounds (longest column <= norm <= Frobenius norm) whatever rounding does to the formula.
inline double largest_stretch(const double A[3][3]) {
    double S[3][3], frob = 0, col = 0;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            S[i][j] = A[0][i] * A[0][j] + A[1][i] * A[1][j] + A[2][i] * A[2][j];
            if (i == j) { frob += S[i][i]; col = std::max(col, S[i][i]); }
        }
    if (!std::isfinite(frob)) return NAN;
    double q = frob / 3, p1 = S[0][1] * S[0][1] + S[0][2] * S[0][2] + S[1][2] * S[1][2];
    double p2 = (S[0][0] - q) * (S[0][0] - q) + (S[1][1] - q) * (S[1][1] - q) + (S[2][2] - q) * (S[2][2] - q) + 2 * p1;
    double top = q;
    if (p2 > 0) {
        double p = std::sqrt(p2 / 6), B[3][3];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) B[i][j] = (S[i][j] - (i == j ? q : 0)) / p;
        double det = B[0][0] * (B[1][1] * B[2][2] - B[1][2] * B[2][1]) - B[0][1] * (B[1][0] * B[2][2] - B[1][2] * B[2][0]) +
                     B[0][2] * (B[1][0] * B[2][1] - B[1][1] * B[2][0]);
        double r = std::min(1.0, std::max(-1.0, det / 2));
        top = q + 2 * p * std::cos(std::acos(r) / 3);
    }
    if (!(top >= col)) top = col;
    if (!(top <= frob)) top = frob;
    return std::sqrt(top);
}

// The ball of every sphere (SphereCull).  A ray is tested in the sphere's own space through `to_local`
// and its hit point comes back through `to_world` (main.cpp:124-125, 163), two float matrices that are
// inverses of each other only up to rounding: the ball is taken around the image of the unit sphere
// under `to_world`, wide enough for the one under the inverse of `to_local` as well, then widened by 1e-3
// of its radius and by 1e-5 of its distance from the origin (the float transformations round coordinates
// of that size to ~1e-7 of it).
// The shell: Sphere::intersect forms b = 2 d.o in float (error <= 1.8e-7 |d||o|), 4ac as a float product of
// floats (2.4e-7 relative) and a, c rounded to float, so delta / (4 d.d) = 1 - (distance line-centre)^2 is
// uncertain by 6e-7 |o|^2, and the roots lie inside the chord of a unit sphere inflated by that much again.
// With |o| <= |oc| * stretch(to_local) and world distances <= stretch(to_world) * local distances, every
// accepted hit lies in the ball of squared radius r^2 + s |oc|^2, s = 8e-6 kappa^2 (kappa = the product of
// the two stretches, the condition number; a factor 2 on top of 2 x 2e-6).  keep = 1 - s - 1e-5 and
// keep_far = (1 - sqrt(s))^2 - 1e-5 apply it to the squared distance and to the distance; at kappa ~ 350
// they reach 0 and the sphere is always tested.  Non-finite matrices: NaN radius, never skipped.
inline void sphere_culls(const rt_scene_desc *sc, const std::vector<int32_t> &sphere_obj, std::vector<SphereCull> &out) {
    out.assign(sc->n_spheres, SphereCull{});
    for (int i = 0; i < sc->n_spheres; i++) {
        const rt_sphere &sp = sc->spheres[i];
        double W[3][3], L[3][3], c[3], tl[3];
        for (int r = 0; r < 3; r++) {
            for (int k = 0; k < 3; k++) { W[r][k] = sp.to_world[3 * k + r]; L[r][k] = sp.to_local[3 * k + r]; }
            c[r] = sp.to_world[9 + r]; tl[r] = sp.to_local[9 + r];
        }
        double radius = largest_stretch(W);
        // the same from to_local: x = L^-1 (u - t) for |u| <= 1
        double det = L[0][0] * (L[1][1] * L[2][2] - L[1][2] * L[2][1]) - L[0][1] * (L[1][0] * L[2][2] - L[1][2] * L[2][0]) +
                     L[0][2] * (L[1][0] * L[2][1] - L[1][1] * L[2][0]);
        double I[3][3];
        I[0][0] = (L[1][1] * L[2][2] - L[1][2] * L[2][1]) / det; I[0][1] = (L[0][2] * L[2][1] - L[0][1] * L[2][2]) / det;
        I[0][2] = (L[0][1] * L[1][2] - L[0][2] * L[1][1]) / det; I[1][0] = (L[1][2] * L[2][0] - L[1][0] * L[2][2]) / det;
        I[1][1] = (L[0][0] * L[2][2] - L[0][2] * L[2][0]) / det; I[1][2] = (L[0][2] * L[1][0] - L[0][0] * L[1][2]) / det;
        I[2][0] = (L[1][0] * L[2][1] - L[1][1] * L[2][0]) / det; I[2][1] = (L[0][1] * L[2][0] - L[0][0] * L[2][1]) / det;
        I[2][2] = (L[0][0] * L[1][1] - L[0][1] * L[1][0]) / det;
        double shift = 0;
        for (int r = 0; r < 3; r++) {
            double c2 = -(I[r][0] * tl[0] + I[r][1] * tl[1] + I[r][2] * tl[2]);
            shift += (c2 - c[r]) * (c2 - c[r]);
        }
        double other = largest_stretch(I) + std::sqrt(shift);
        if (other > radius || other != other) radius = other;  // (NaN from a singular to_local stays NaN)
        double far = std::sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
        radius = radius * 1.001 + 1e-5 * (far + radius);
        SphereCull &k = out[i];
        for (int r = 0; r < 3; r++) k.c[r] = (float)c[r];
        k.r = (float)(radius * 1.000001);  // (rounding to float must not shrink it)
        k.r2 = (float)(radius * radius * 1.000001);
        double kappa = largest_stretch(W) * largest_stretch(L), shell = 8e-6 * kappa * kappa;
        if (!(shell < 1)) shell = 1;  // (also for NaN)
        k.keep = (float)std::max(0.0, 1 - shell - 1e-5);
        k.keep_far = (float)std::max(0.0, (1 - std::sqrt(shell)) * (1 - std::sqrt(shell)) - 1e-5);
        k.object = sphere_obj[i];
    }
}

// Fills the device-side mesh records (everything but device pointers), the per-mesh build data and
// the gate triangles (36 f4 per mesh).  Returns the total number of LBVH nodes.
// refs == true: also cuts the meshes into references (ref_pos, ref_tri); false: the reference ranges in
// `build` are kept (same geometry, new placement).
inline int prepare_meshes(const rt_scene_desc *sc, std::vector<DevMesh> &meshes, std::vector<MeshBuild> &build,
                          std::vector<f4> &gate, bool refs, std::vector<float> &ref_pos, std::vector<int32_t> &ref_tri) {
    std::vector<MeshBuild> previous = build;
    meshes.assign(sc->n_meshes, DevMesh{});
    build.assign(sc->n_meshes, MeshBuild{});
    if (refs) { ref_pos.clear(); ref_tri.clear(); }
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
        if (refs) {
            mb.ref_offset = (int)ref_tri.size();
            make_mesh_refs(sc, m, ref_pos, ref_tri);
            mb.ref_count = (int)ref_tri.size() - mb.ref_offset;
        } else {
            mb.ref_offset = previous[i].ref_offset; mb.ref_count = previous[i].ref_count;
            mb.root = previous[i].root; mb.height = previous[i].height;
        }
        dm.tri_offset = mb.ref_offset; dm.tri_count = mb.ref_count; dm.boxed = m.boxed;  // what the kernels walk: references
        dm.node_offset = node_total; dm.gate_tri_offset = 12 * i; dm.root = mb.root;
        mb.node_offset = node_total;
        node_total += kOctants * std::max(mb.ref_count - 1, 0);  // one copy of the tree per ray-direction octant
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
        dm.same_xform = (!m.boxed || std::memcmp(m.box_to_local, m.to_local, sizeof m.to_local) == 0) ? 1 : 0;
        // World-space bounds for the first rejection test: the padded gate box (boxed meshes) or the
        // padded vertex bounds, mapped to world space in double precision and padded again.
        double lo[3], hi[3], M[12];
        for (int a = 0; a < 3; a++) { lo[a] = m.boxed ? dm.gate_lo[a] : mb.lo[a] - mb.pad; hi[a] = m.boxed ? dm.gate_hi[a] : mb.hi[a] + mb.pad; }
        bool have = m.boxed ? invert_affine(m.box_to_local, M) : true;
        if (!m.boxed) for (int a = 0; a < 12; a++) M[a] = m.to_world[a];
        double wlo[3] = {INFINITY, INFINITY, INFINITY}, whi[3] = {-INFINITY, -INFINITY, -INFINITY}, wmax = 0;
        for (int c = 0; c < 8 && have; c++) {
            double x = (c & 1) ? hi[0] : lo[0], y = (c & 2) ? hi[1] : lo[1], z = (c & 4) ? hi[2] : lo[2];
            for (int a = 0; a < 3; a++) {
                double w = M[a] * x + M[3 + a] * y + M[6 + a] * z + M[9 + a];
                if (!(w == w)) have = false;
                wlo[a] = std::min(wlo[a], w); whi[a] = std::max(whi[a], w); wmax = std::max(wmax, std::fabs(w));
            }
        }
        for (int a = 0; a < 3; a++) {
            bool ok = have && std::isfinite(wlo[a]) && std::isfinite(whi[a]);
            double wp = wmax * 1e-4 + 1e-30;
            dm.world_lo[a] = ok ? (float)(wlo[a] - wp) : -INFINITY;  // unbounded when the bounds cannot be trusted
            dm.world_hi[a] = ok ? (float)(whi[a] + wp) : INFINITY;
        }
    }
    return node_total;
}

// The shadow-query summaries per light (LightCull) and the mask of lights they are valid for.
// Needs the world boxes prepare_meshes computed and the plane records' cannot_shadow bits.
inline uint32_t make_light_cull(const rt_scene_desc *sc, const std::vector<DevMesh> &meshes, uint32_t planes_skip_lights,
                                std::vector<LightCull> &cull) {
    LightCull none;
    for (int k = 0; k < kCullMeshes; k++) none.cone[k] = f4{0, 0, 0, 2.0f};
    cull.assign(std::max(sc->n_lights, 1), none);
    uint32_t cullable = 0;
    // objects of other kinds in the object list, and meshes beyond the table's capacity, forbid culling
    int listed_meshes = 0;
    for (const DevMesh &m : meshes) if (m.object >= 0) listed_meshes++;
    bool quadrics = false;
    for (int i = 0; i < sc->n_objects; i++) quadrics = quadrics || sc->objects[i].kind == RT_OBJ_SPHERE || sc->objects[i].kind == RT_OBJ_CONE;
    if (quadrics || listed_meshes > kCullMeshes) return 0;
    for (int l = 0; l < sc->n_lights && l < 32; l++) {
        bool ok = ((planes_skip_lights >> l) & 1u) != 0;  // no plane can shadow this light
        int slot = 0;
        for (size_t i = 0; i < meshes.size() && ok; i++) {
            const DevMesh &m = meshes[i];
            if (m.object < 0) continue;
            f4 &out = cull[l].cone[slot++];
            out = f4{0, 0, 0, -2.0f};  // unless shown otherwise: always inside the cone
            double c[3], r2 = 0, v[3], dist2 = 0;
            bool bounded = true;
            for (int a = 0; a < 3; a++) {
                bounded = bounded && std::isfinite(m.world_lo[a]) && std::isfinite(m.world_hi[a]);
                c[a] = 0.5 * ((double)m.world_lo[a] + m.world_hi[a]);
                double h = 0.5 * ((double)m.world_hi[a] - m.world_lo[a]);
                r2 += h * h;
                v[a] = c[a] - (double)sc->light_pos[3 * l + a];
                dist2 += v[a] * v[a];
            }
            if (!bounded) continue;
            double dist = std::sqrt(dist2);
            // shadow rays pass the light only up to rounding and their EPS offset: inflate the sphere well beyond that
            double R = std::sqrt(r2) * 1.001 + 1e-3 * dist + 1e-3;
            if (!(dist > R * 1.001) || !std::isfinite(dist)) continue;
            double cosv = std::sqrt(1.0 - (R / dist) * (R / dist));
            out.x = (float)(-v[0] / dist); out.y = (float)(-v[1] / dist); out.z = (float)(-v[2] / dist);
            out.w = (float)(cosv * (1.0 - 1e-5) - 1e-5);  // covers the float rounding of the axis, the direction and the dot product
        }
        if (ok) cullable |= 1u << l;
    }
    return cullable;
}

// The four RGBA8 maps of an image texture as one 16-byte record per texel.
inline void interleave_texture(const rt_texture &t, std::vector<u4> &out) {
    size_t n = (size_t)t.width * t.height;
    out.resize(n);
    const uint8_t *src[4] = {t.base_color, t.normal, t.occlusion, t.roughness};
    for (size_t i = 0; i < n; i++) {
        uint32_t px[4];
        for (int k = 0; k < 4; k++) std::memcpy(&px[k], src[k] + 4 * i, 4);
        out[i] = u4{px[0], px[1], px[2], px[3]};
    }
}
// pow(b / 255, 2.2) per byte value (ImageTexture::texture, Textures.cpp:41-43)
inline void make_srgb_lut(std::vector<float> &lut) {
    lut.resize(256);
    for (int b = 0; b < 256; b++) lut[b] = powf((float)b / 255.f, 2.2f);
}

// The plane records find_closest walks: the planes that occur in the object list, those with a normal +-e_a as
// pairs per axis (pairs 0-2: the first plane of either orientation of x, y, z; pair 3 empty; overflow pairs from
// kFixedAxisPairs on, padded to whole loads with empty pairs), the others as general records (padded to the group
// size).  Returns the lights that no listed plane can shadow.
inline uint32_t make_plane_recs(const rt_scene_desc *sc, const std::vector<int32_t> &plane_obj, std::vector<AxisPair> &axis,
                                std::vector<PlaneRec> &gen) {
    const AxisPair empty = {NAN, -1, 0, NAN, -1, 0, 0, 0};
    axis.assign(kFixedAxisPairs, empty);
    for (int a = 0; a < 3; a++) axis[a].axis = a;
    axis[3].axis = -1;  // (never used)
    gen.clear();
    uint32_t skip = sc->n_lights >= 32 ? 0xffffffffu : (1u << std::max(sc->n_lights, 0)) - 1u;
    for (int i = 0; i < sc->n_planes; i++) {
        if (plane_obj[i] < 0) continue;
        const float *n = sc->planes[i].normal, *p = sc->planes[i].point;
        // lights on the front side of the plane, by a margin of 1e-3 of the scene's scale there
        uint32_t cannot_shadow = 0;
        double nlen = std::sqrt((double)n[0] * n[0] + (double)n[1] * n[1] + (double)n[2] * n[2]);
        for (int l = 0; l < sc->n_lights && l < 32 && nlen > 0 && std::isfinite(nlen); l++) {
            const float *L = sc->light_pos + 3 * l;
            double side = 0, scale = 1;
            for (int a = 0; a < 3; a++) { side += (double)n[a] * ((double)L[a] - p[a]); scale += std::fabs((double)L[a]) + std::fabs((double)p[a]); }
            if (side / nlen > 1e-3 * scale) cannot_shadow |= 1u << l;
        }
        skip &= cannot_shadow;
        int ax = -1;
        bool positive = false;
        for (int a = 0; a < 3; a++)
            if (std::fabs(n[a]) == 1.0f && n[(a + 1) % 3] == 0.0f && n[(a + 2) % 3] == 0.0f) { ax = a; positive = n[a] > 0; }
        if (ax >= 0) {
            size_t at = axis.size();
            for (size_t k = 0; k < axis.size(); k++)
                if (axis[k].axis == ax && (positive ? axis[k].object_pos : axis[k].object_neg) < 0) { at = k; break; }
            if (at == axis.size()) { axis.push_back(empty); axis.back().axis = ax; }
            AxisPair &r = axis[at];
            if (positive) { r.coord_pos = p[ax]; r.object_pos = plane_obj[i]; r.cs_pos = cannot_shadow; }
            else { r.coord_neg = p[ax]; r.object_neg = plane_obj[i]; r.cs_neg = cannot_shadow; }
        } else {
            PlaneRec r;
            for (int a = 0; a < 3; a++) { r.n[a] = n[a]; r.p[a] = p[a]; }
            r.object = plane_obj[i]; r.cannot_shadow = cannot_shadow;
            gen.push_back(r);
        }
    }
    while (axis.size() % kAxisPairGroup) axis.push_back(empty);
    while (gen.size() % kPlaneGroup) gen.push_back(PlaneRec{{0, 0, 0}, {0, 0, 0}, -1, 0});
    return skip;
}
// upper bounds of the two arrays' sizes for a scene with n planes (rt_scene_update re-partitions them in place)
inline size_t axis_pair_capacity(int n_planes) { return ((size_t)std::max(n_planes, 0) + 3 + kFixedAxisPairs + kAxisPairGroup) / kAxisPairGroup * kAxisPairGroup; }
inline size_t gen_plane_capacity(int n_planes) { return ((size_t)std::max(n_planes, 0) + kPlaneGroup) / kPlaneGroup * kPlaneGroup; }

// The materials as shading reads them (MatRec) and the per-object records (ObjRec).  tex_addr[i]: where texture
// i's texel records live for the code that will read these tables (device address, or host address for a host build).
inline void make_mat_recs(const rt_scene_desc *sc, const std::vector<const void *> &tex_addr, std::vector<MatRec> &out) {
    out.assign(sc->n_materials, MatRec{});
    for (int i = 0; i < sc->n_materials; i++) {
        const rt_material &m = sc->materials[i];
        MatRec &r = out[i];
        for (int a = 0; a < 3; a++) {
            r.ambient[a] = sc->ambient[a] * m.ambient[a];  // ambient_light * material.ambient, main.cpp:646
            r.diffuse[a] = m.diffuse[a]; r.specular[a] = m.specular[a];
        }
        r.shininess = m.shininess; r.reflectiveness = m.reflectiveness; r.refractiveness = m.refractiveness;
        r.refractive_ratio = m.refractive_ratio;
        r.tex_wh = 0; r.tex_lo = 0; r.tex_hi = 0;
        if (m.texture >= 0) {
            const rt_texture &t = sc->textures[m.texture];
            uint64_t addr = t.kind == RT_TEX_IMAGE ? (uint64_t)(uintptr_t)tex_addr[m.texture] : 0;
            r.tex_wh = t.kind == RT_TEX_IMAGE ? (t.width | (t.height << 16)) : 0u;
            r.tex_lo = (uint32_t)(addr & 0xfffffff0ull) | (uint32_t)(t.kind + 1);
            r.tex_hi = (uint32_t)(addr >> 32);
        }
    }
}
inline void make_obj_recs(const rt_scene_desc *sc, const std::vector<MatRec> &mats, std::vector<ObjRec> &out) {
    out.assign(std::max(sc->n_objects, 1), ObjRec{});
    for (int i = 0; i < sc->n_objects; i++) {
        const rt_object ob = sc->objects[i];
        ObjRec &r = out[i];
        r.kind = ob.kind; r.index = ob.index; r.material = -1; r.pad0 = 0;
        if (ob.kind == RT_OBJ_PLANE) {
            const rt_plane &pl = sc->planes[ob.index];
            v3 n = ld3(pl.normal), nn = normalize(n);  // the reference's own float expression (glm::normalize)
            r.n[0] = n.x; r.n[1] = n.y; r.n[2] = n.z; r.nn[0] = nn.x; r.nn[1] = nn.y; r.nn[2] = nn.z;
            for (int a = 0; a < 3; a++) { r.b1[a] = pl.b1[a]; r.b2[a] = pl.b2[a]; }
            r.material = pl.material;
        } else if (ob.kind == RT_OBJ_SPHERE) {
            r.material = sc->spheres[ob.index].material;
        } else if (ob.kind == RT_OBJ_CONE) {
            r.material = sc->cones[ob.index].material;
        } else {  // a mesh: its material when all its triangles share one
            const rt_mesh &m = sc->meshes[ob.index];
            for (int t = 0; t < m.tri_count; t++) {
                int tm = sc->tri_material[m.tri_offset + t];
                if (t == 0) r.material = tm;
                else if (tm != r.material) { r.material = -1; break; }
            }
        }
        if (r.material >= 0) r.mat = mats[r.material];
    }
}

// The per-light records: position, colour and the cones of make_light_cull.
// cullable: bit l set when the cones of light l decide alone whether a shadow query is needed (make_light_cull)
inline void make_light_recs(const rt_scene_desc *sc, const std::vector<LightCull> &cull, uint32_t cullable, std::vector<LightRec> &out) {
    out.assign(std::max(sc->n_lights, 1), LightRec{});
    for (int l = 0; l < sc->n_lights; l++) {
        LightRec &r = out[l];
        for (int a = 0; a < 3; a++) { r.pos[a] = sc->light_pos[3 * l + a]; r.color[a] = sc->light_color[3 * l + a]; }
        r.flags = (l < 32 && ((cullable >> l) & 1u)) ? kLightCullable : 0u;
        r.cull = cull[l];
    }
}
// The tables that are part of the scene record itself: the fixed axis pairs and the first entries of the others.
inline void fill_embedded(DevScene &S, const std::vector<AxisPair> &axis, const std::vector<PlaneRec> &gen, const std::vector<DevMesh> &meshes,
                          const std::vector<ObjRec> &objs, const std::vector<LightRec> &lights) {
    for (int i = 0; i < kFixedAxisPairs; i++) S.axis_pairs[i] = axis[i];
    for (int i = 0; i < kEmbGenPlanes; i++) S.gen_first[i] = i < (int)gen.size() ? gen[i] : PlaneRec{{0, 0, 0}, {0, 0, 0}, -1, 0};
    for (int i = 0; i < kEmbMeshes; i++) {
        if (i < (int)meshes.size()) S.mesh_heads[i] = meshes[i];  // (the MeshHead part)
        else { S.mesh_heads[i] = MeshHead{}; S.mesh_heads[i].object = -1; }
    }
    for (int i = 0; i < kEmbObjects; i++) S.obj_first[i] = i < (int)objs.size() ? objs[i] : ObjRec{};
    for (int i = 0; i < kEmbLights; i++) S.light_first[i] = i < (int)lights.size() ? lights[i] : LightRec{};
}

}  // namespace rtc
