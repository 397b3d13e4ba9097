// twin.cpp -- TEST SCAFFOLDING ONLY.  Compiles the renderer's per-ray code (rt_core.h), LBVH build
// steps (rt_lbvh.h) and scene preparation (rt_prepare.h) as host C++ and drives them serially, so
// that the logic the HIP kernels execute can be checked against the oracle on a machine without a
// GPU.  The kernels' thread-per-element loops become plain loops here.  Nothing in the shipped
// libraries links this file; it is built by tests/conftest.py into tests/cpu_twin/.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "rt_amd.h"
#include "rt_core.h"
#include "rt_lbvh.h"
#include "rt_prepare.h"

using namespace rtc;

namespace {

struct VecStack {  // the host build's "wave" is one lane
    int32_t node[kStackDepth];
    uint64_t mask[kStackDepth];
    int max_sp = 0;
    float cold[kColdSlots + kTapSlots];
    void cold_put(int slot, float v) { cold[slot] = v; }
    float cold_get(int slot) const { return cold[slot]; }
    void put(int slot, int32_t n, uint64_t m) { if (slot < kStackDepth) { node[slot] = n; mask[slot] = m; if (slot + 1 > max_sp) max_sp = slot + 1; } }
    void get(int slot, int32_t &n, uint64_t &m) { n = node[slot]; m = mask[slot]; }
};

constexpr int kTwinMaxLights = 64;  // (test scenes have a handful)
struct Twin {
    std::string err;
    // the scene record with the records of the lights beyond its own table right behind it (light_rec_at)
    struct alignas(64) Record { DevScene S{}; LightRec more_lights[kTwinMaxLights - kEmbLights]; } rec;
    DevScene &S = rec.S;
    std::vector<rt_object> objects;
    std::vector<rt_sphere> spheres;
    std::vector<rt_plane> planes;
    std::vector<rt_cone> cones;
    std::vector<DevMesh> meshes;
    std::vector<MeshBuild> build;
    std::vector<f4> tri_test, gate;
    std::vector<uint32_t> tri_key;
    std::vector<int32_t> tri_orig, tri_material, tri_frag, tri_sub;
    std::vector<int32_t> plane_obj, sphere_obj, cone_obj, mesh_obj;
    std::vector<LightCull> light_cull;
    std::vector<LightRec> light_recs;
    std::vector<SphereCull> sphere_cull;
    std::vector<AxisPair> axis_pairs;
    std::vector<PlaneRec> gen_planes;
    std::vector<MatRec> mat_recs;
    std::vector<ObjRec> obj_recs;
    std::vector<float> ref_pos, leaf_box;  // pieces the triangles are cut into; boxes of the sorted pieces
    std::vector<int32_t> ref_tri;
    std::vector<float> tri_pos, tri_nrm, tri_uv, light_pos, light_color;
    std::vector<BvhNode> nodes;
    std::vector<rt_material> materials;
    std::vector<DevTexture> textures;
    std::vector<std::vector<u4>> texels;
    std::vector<float> srgb_lut;
    int bvh_max_depth = 0;
    Work total{};
};

void gather(Twin &t, const DevMesh &m, const std::vector<uint64_t> *keys) {
    for (int k = 0; k < m.tri_count; k++) {
        int local = keys ? (int)(uint32_t)((*keys)[k] & 0xffffffffull) : k;
        int orig = t.ref_tri[m.tri_offset + local];
        make_tri_test(&t.tri_pos[9 * (size_t)orig], &t.tri_test[3 * (size_t)(m.tri_offset + k)]);
        t.tri_key[m.tri_offset + k] = 2u * (uint32_t)t.tri_frag[orig] + (1u - (uint32_t)t.tri_sub[orig]);
        t.tri_orig[m.tri_offset + k] = orig;
        piece_box(&t.ref_pos[9 * (size_t)(m.tri_offset + local)], &t.leaf_box[6 * (size_t)(m.tri_offset + k)]);
    }
}

// The top-down build of k_sah_level (binned surface-area splits, rt_lbvh.h), one job after the other:
// the same bins, costs, tie rule, order-preserving partition and node ids.  Returns the root's id.
int32_t sah_children(int n, const float *leaf_box, std::vector<int32_t> &c0, std::vector<int32_t> &c1) {
    struct Job { int first, last, slot, depth; };
    std::vector<int32_t> idx(n), tmp(n);
    for (int k = 0; k < n; k++) idx[k] = k;
    std::vector<Job> level{{0, n - 1, -1, 0}}, next;
    int32_t root = 0;
    while (!level.empty()) {
        next.clear();
        for (const Job &j : level) {
            const int len = j.last - j.first + 1;
            float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY}, ext[3];
            bool usable[3];
            for (int k = j.first; k <= j.last; k++)
                for (int a = 0; a < 3; a++) { float c = sah_centre(leaf_box, idx[k], a); clo[a] = fminf(clo[a], c); chi[a] = fmaxf(chi[a], c); }
            for (int a = 0; a < 3; a++) { ext[a] = chi[a] - clo[a]; usable[a] = ext[a] > 0 && ext[a] < INFINITY; }
            int best = -1;
            int32_t n_left = 0;
            float best_cost = INFINITY;
            if (j.depth < kSahDepthLimit)
                for (int a = 0; a < 3; a++) {
                    if (!usable[a]) continue;
                    float bins[kSahBins * 6];
                    int32_t counts[kSahBins] = {0};
                    for (int b = 0; b < kSahBins; b++)
                        for (int c = 0; c < 6; c++) bins[6 * b + c] = c < 3 ? INFINITY : -INFINITY;
                    for (int k = j.first; k <= j.last; k++) {
                        int q = idx[k], b = sah_bin(sah_centre(leaf_box, q, a), clo[a], ext[a]);
                        sah_grow(bins + 6 * b, leaf_box + 6 * (size_t)q);
                        counts[b]++;
                    }
                    for (int b = 0; b < kSahBins - 1; b++) {
                        int32_t nl;
                        float cost = sah_split_cost(bins, counts, b, nl);
                        if (cost < best_cost) { best_cost = cost; best = a * (kSahBins - 1) + b; n_left = nl; }
                    }
                }
            const int total_left = best < 0 ? (len + 1) / 2 : n_left;
            if (best >= 0) {
                const int axis = best / (kSahBins - 1), bin = best % (kSahBins - 1);
                int l = j.first, r = j.first + total_left;
                for (int k = j.first; k <= j.last; k++) {
                    int q = idx[k];
                    if (sah_bin(sah_centre(leaf_box, q, axis), clo[axis], ext[axis]) <= bin) tmp[l++] = q; else tmp[r++] = q;
                }
                for (int k = j.first; k <= j.last; k++) idx[k] = tmp[k];
            }
            const int32_t g = j.first + total_left - 1;
            if (j.slot < 0) root = g;
            else if (j.slot & 1) c1[j.slot >> 1] = g;
            else c0[j.slot >> 1] = g;
            if (total_left == 1) c0[g] = ~idx[j.first]; else next.push_back({j.first, g, 2 * g, j.depth + 1});
            if (len - total_left == 1) c1[g] = ~idx[j.last]; else next.push_back({g + 1, j.last, 2 * g + 1, j.depth + 1});
        }
        level.swap(next);
    }
    return root;
}

}  // namespace

extern "C" {

void *twin_create(const rt_scene_desc *sc, char *errbuf, int errcap) {
    std::string bad = validate_scene(sc);
    if (!bad.empty()) {
        if (errbuf && errcap > 0) { std::strncpy(errbuf, bad.c_str(), errcap - 1); errbuf[errcap - 1] = 0; }
        return nullptr;
    }
    if (sc->n_lights > kTwinMaxLights) {
        if (errbuf && errcap > 0) { std::strncpy(errbuf, "the twin holds at most 64 lights", errcap - 1); errbuf[errcap - 1] = 0; }
        return nullptr;
    }
    Twin *t = new Twin();
    static_assert(offsetof(Twin::Record, more_lights) == sizeof(DevScene), "the lights beyond the scene record's table follow it directly");
    t->objects.assign(sc->objects, sc->objects + sc->n_objects);
    t->spheres.assign(sc->spheres, sc->spheres + sc->n_spheres);
    t->planes.assign(sc->planes, sc->planes + sc->n_planes);
    t->cones.assign(sc->cones, sc->cones + sc->n_cones);
    size_t nt = sc->n_triangles;
    t->tri_pos.assign(sc->tri_pos, sc->tri_pos + 9 * nt);
    t->tri_nrm.assign(sc->tri_nrm, sc->tri_nrm + 9 * nt);
    t->tri_uv.assign(sc->tri_uv, sc->tri_uv + 6 * nt);
    t->tri_material.assign(sc->tri_material, sc->tri_material + nt);
    t->tri_frag.assign(sc->tri_frag, sc->tri_frag + nt);
    t->tri_sub.assign(sc->tri_sub, sc->tri_sub + nt);
    t->materials.assign(sc->materials, sc->materials + sc->n_materials);
    t->light_pos.assign(sc->light_pos, sc->light_pos + 3 * (size_t)sc->n_lights);
    t->light_color.assign(sc->light_color, sc->light_color + 3 * (size_t)sc->n_lights);
    t->textures.resize(sc->n_textures);
    for (int i = 0; i < sc->n_textures; i++) {
        const rt_texture &x = sc->textures[i];
        DevTexture &d = t->textures[i];
        d.kind = x.kind; d.w = x.width; d.h = x.height; d.pad = 0;
        d.texels = nullptr;
        for (int k = 0; k < 3; k++) d.pad2[k] = nullptr;
        if (x.kind == RT_TEX_IMAGE) {
            t->texels.emplace_back();
            interleave_texture(x, t->texels.back());
        }
    }
    for (int i = 0, k = 0; i < sc->n_textures; i++)  // pointers after the outer vector stopped growing
        if (sc->textures[i].kind == RT_TEX_IMAGE) t->textures[i].texels = t->texels[k++].data();
    make_srgb_lut(t->srgb_lut);
    int node_total = prepare_meshes(sc, t->meshes, t->build, t->gate, true, t->ref_pos, t->ref_tri);
    size_t nr = t->ref_tri.size();
    t->tri_test.assign(3 * nr + 1, f4{0, 0, 0, 0});
    t->tri_key.assign(nr + 1, 0);
    t->tri_orig.assign(nr + 1, 0);
    t->leaf_box.assign(6 * nr + 1, 0.0f);
    t->ref_pos.push_back(0.0f);
    t->ref_tri.push_back(0);
    t->nodes.assign(std::max(node_total, 1), BvhNode{});
    for (const DevMesh &m : t->meshes) gather(*t, m, nullptr);
    DevScene &S = t->S;
    S.n_objects = sc->n_objects; S.n_lights = sc->n_lights;
    S.n_planes = sc->n_planes; S.n_spheres = sc->n_spheres; S.n_cones = sc->n_cones; S.n_meshes = sc->n_meshes;
    kind_to_object(sc, t->plane_obj, t->sphere_obj, t->cone_obj, t->mesh_obj);
    for (int i = 0; i < sc->n_meshes; i++) t->meshes[i].object = t->mesh_obj[i];
    S.planes_skip_lights = make_plane_recs(sc, t->plane_obj, t->axis_pairs, t->gen_planes);
    S.n_axis_overflow = (int32_t)t->axis_pairs.size() - kFixedAxisPairs; S.n_gen_planes = (int32_t)t->gen_planes.size();
    S.axis_overflow = t->axis_pairs.data() + kFixedAxisPairs; S.gen_planes = t->gen_planes.data();
    std::vector<const void *> tex_addr(sc->n_textures, nullptr);
    for (int i = 0; i < sc->n_textures; i++) tex_addr[i] = t->textures[i].texels;
    make_mat_recs(sc, tex_addr, t->mat_recs);
    make_obj_recs(sc, t->mat_recs, t->obj_recs);
    S.mat_recs = t->mat_recs.data(); S.obj_recs = t->obj_recs.data();
    sphere_culls(sc, t->sphere_obj, t->sphere_cull);
    S.plane_obj = t->plane_obj.data(); S.sphere_cull = t->sphere_cull.data(); S.cone_obj = t->cone_obj.data();
    S.objects = t->objects.data(); S.spheres = t->spheres.data(); S.planes = t->planes.data(); S.cones = t->cones.data();
    S.meshes = t->meshes.data(); S.tri_test = t->tri_test.data(); S.tri_key = t->tri_key.data(); S.tri_orig = t->tri_orig.data();
    S.tri_nrm = t->tri_nrm.data(); S.tri_uv = t->tri_uv.data(); S.tri_material = t->tri_material.data();
    S.tri_frag = t->tri_frag.data(); S.tri_sub = t->tri_sub.data(); S.gate_tri = t->gate.data(); S.nodes = t->nodes.data();
    S.cullable_lights = make_light_cull(sc, t->meshes, S.planes_skip_lights, t->light_cull);
    make_light_recs(sc, t->light_cull, S.cullable_lights, t->light_recs);
    S.unused_lights = nullptr;
    for (int l = kEmbLights; l < sc->n_lights; l++) t->rec.more_lights[l - kEmbLights] = t->light_recs[l];
    S.srgb_lut = t->srgb_lut.data(); S.unused_gate_order = 0;
    S.materials = t->materials.data(); S.textures = t->textures.data();
    fill_embedded(S, t->axis_pairs, t->gen_planes, t->meshes, t->obj_recs, t->light_recs);
    return t;
}

void twin_destroy(void *h) { delete static_cast<Twin *>(h); }

// the LBVH build of rt_build_accel, one mesh after the other; returns the maximum tree height
int twin_build_accel(void *h) {
    Twin *t = static_cast<Twin *>(h);
    t->bvh_max_depth = 0;
    for (size_t mi = 0; mi < t->meshes.size(); mi++) {
        const DevMesh &m = t->meshes[mi];
        int n = m.tri_count;
        if (n <= 0) continue;
        std::vector<uint64_t> keys(n);
        for (int i = 0; i < n; i++) keys[i] = tri_sort_key(&t->ref_pos[9 * (size_t)(m.tri_offset + i)], t->build[mi].lo, t->build[mi].hi, (uint32_t)i);
        std::sort(keys.begin(), keys.end());
        gather(*t, m, &keys);
        if (n < 2) continue;
        std::vector<int32_t> c0(n - 1), c1(n - 1), done(n - 1, 0), height(n - 1, 0);
        std::vector<float> lo(3 * (size_t)(n - 1)), hi(3 * (size_t)(n - 1));
        for (int i = 0; i < n - 1; i++) lbvh_node_children(keys.data(), n, i, c0[i], c1[i]);
        const float *leaf_box = &t->leaf_box[6 * (size_t)m.tri_offset];
        int32_t root = 0;
        const char *builder = std::getenv("RT_BUILD");
        if (!(builder && std::strcmp(builder, "radix") == 0)) root = sah_children(n, leaf_box, c0, c1);
        t->build[mi].root = root;
        t->meshes[mi].root = root;
        if (mi < (size_t)kEmbMeshes) t->S.mesh_heads[mi] = t->meshes[mi];  // (the copy in the scene record)
        for (int pass = 1; pass <= 4096; pass++) {
            int left = 0;
            for (int i = 0; i < n - 1; i++)
                if (lbvh_fit_node(i, pass, c0.data(), c1.data(), leaf_box, t->build[mi].pad, done.data(), lo.data(), hi.data(), height.data(),
                                  &t->nodes[m.node_offset]) == 2)
                    left++;
            if (!left) break;
        }
        for (int o = 1; o < kOctants; o++)
            for (int i = 0; i < n - 1; i++)
                t->nodes[m.node_offset + (size_t)o * (n - 1) + i] = lbvh_octant_variant(t->nodes[m.node_offset + i], o);
        for (int i = 0; i < n - 1; i++) t->nodes[m.node_offset + i] = lbvh_octant_variant(t->nodes[m.node_offset + i], 0);
        t->build[mi].height = height[root];
        t->bvh_max_depth = std::max(t->bvh_max_depth, height[root]);
    }
    return t->bvh_max_depth;
}

void twin_trace_rays(void *h, int64_t n, const float *o, const float *d, int max_depth, int accel, rt_hit_record *hits,
                     float *radiance, int *max_stack) {
    Twin *t = static_cast<Twin *>(h);
    int ms = 0;
#pragma omp parallel for schedule(dynamic, 64) reduction(max : ms)
    for (int64_t r = 0; r < n; r++) {
        VecStack stk;
        Work wk;
        work_zero(wk);
        rt_hit_record hr;
        v3 c = trace<F_ALL, false>(t->S, ld3(o + 3 * r), ld3(d + 3 * r), max_depth, accel, stk, wk, hits ? &hr : nullptr);
        if (hits) hits[r] = hr;
        if (radiance) { radiance[3 * r] = c.x; radiance[3 * r + 1] = c.y; radiance[3 * r + 2] = c.z; }
        ms = std::max(ms, stk.max_sp);
    }
    if (max_stack) *max_stack = ms;
}

void twin_intersect_object(void *h, int object, int64_t n, const float *o, const float *d, int accel, rt_hit_record *hits) {
    Twin *t = static_cast<Twin *>(h);
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t r = 0; r < n; r++) {
        VecStack stk;
        Work wk;
        work_zero(wk);
        intersect_object<false>(t->S, object, ld3(o + 3 * r), ld3(d + 3 * r), accel, hits[r], stk, wk);
    }
}

// the pixel loop of k_render over [x0,x1) x [y0,y1); counters[10] accumulates the Work fields
void twin_render(void *h, const rt_render_params *p, int x0, int y0, int x1, int y1, float *rgb, uint8_t *rgb8,
                 unsigned long long *counters) {
    Twin *t = static_cast<Twin *>(h);
    Camera cam = make_camera(p->width, p->height, p->fov);
    unsigned long long acc[10] = {0};
    float grid_off[kMaxGrid];
    fill_grid_offsets(grid_off, p->grid_m, cam.s);
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : acc[:10])
    for (int j = y0; j < y1; j++)
        for (int i = x0; i < x1; i++) {
            VecStack stk;
            Work wk;
            work_zero(wk);
            v3 c = render_pixel<F_ALL, true, true>(t->S, cam, p->mode, p->grid_m, p->max_depth, p->accel, i, j, true, grid_off, stk, wk);
            size_t q = 3 * ((size_t)j * p->width + i);
            if (rgb) { rgb[q] = c.x; rgb[q + 1] = c.y; rgb[q + 2] = c.z; }
            if (rgb8) { v3 tm = tone_map(c); rgb8[q] = quantise(tm.x); rgb8[q + 1] = quantise(tm.y); rgb8[q + 2] = quantise(tm.z); }
            uint32_t vals[10] = {wk.rays_primary, wk.rays_shadow, wk.rays_secondary, wk.tri, wk.box, wk.gate, wk.quadric, wk.plane, wk.shade, wk.tex};
            for (int k = 0; k < 10; k++) acc[k] += vals[k];
        }
    if (counters) for (int k = 0; k < 10; k++) counters[k] += acc[k];
}

// number of references (leaves) of a mesh's tree
int twin_ref_count(void *h, int mesh) { return static_cast<Twin *>(h)->meshes[mesh].tri_count; }

int twin_copy_bvh(void *h, int mesh, float *nodes, int32_t *order) {
    Twin *t = static_cast<Twin *>(h);
    const DevMesh &m = t->meshes[mesh];
    int nn = std::max(m.tri_count - 1, 0);
    if (nodes && nn) std::memcpy(nodes, &t->nodes[m.node_offset], (size_t)nn * sizeof(BvhNode));
    if (order) std::memcpy(order, &t->tri_orig[m.tri_offset], (size_t)m.tri_count * sizeof(int32_t));
    return nn;
}
}
