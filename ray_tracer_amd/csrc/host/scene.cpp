// scene.cpp -- object API: transformation matrices, bounds, and the lowering of an object graph to
// the plain arrays of the C ABI (Scene::flatten).  Reference citations are into main.cpp.
#include "scene.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <stdexcept>

namespace rta {

// Object::setTransformation, main.cpp:85-93: the normal matrix is the matrix itself when
// |det| is EXACTLY 1 (pure translations / rotations that happen to round to 1), else (M^-1)^T.
void Object::setTransformation(mat4 matrix) {
    transformationMatrix = matrix;
    inverseTransformationMatrix = inverse(matrix);
    normalMatrix = (std::fabs(determinant(matrix)) == 1.0f) ? matrix : transpose(inverseTransformationMatrix);
}

static vec3 xyz(vec4 v) { return {v.x, v.y, v.z}; }
static vec3 vmin(vec3 a, vec3 b) { return {std::min(a.x, b.x), std::min(a.y, b.y), std::min(a.z, b.z)}; }
static vec3 vmax(vec3 a, vec3 b) { return {std::max(a.x, b.x), std::max(a.y, b.y), std::max(a.z, b.z)}; }

vec3 Sphere::getMinCoords() const { return xyz(transformationMatrix * vec4(-1, -1, -1, 1)); }  // main.cpp:118
vec3 Sphere::getMaxCoords() const { return xyz(transformationMatrix * vec4(1, 1, 1, 1)); }
vec3 Cone::getMinCoords() const { return xyz(transformationMatrix * vec4(-1, 0, -1, 1)); }     // main.cpp:267
vec3 Cone::getMaxCoords() const { return xyz(transformationMatrix * vec4(1, 1, 1, 1)); }
vec3 Box::getMinCoords() const { return xyz(transformationMatrix * vec4(v1, 1)); }             // main.cpp:597
vec3 Box::getMaxCoords() const { return xyz(transformationMatrix * vec4(v2, 1)); }

// Plane::computeB1 / computeBaseVectors, main.cpp:198-210
void Plane::computeBaseVectors() {
    vec3 a = cross(normal, vec3(1, 0, 0));
    vec3 b = cross(normal, vec3(0, 1, 0));
    vec3 max_ab = dot(a, a) > dot(b, b) ? a : b;
    vec3 c = cross(normal, vec3(0, 0, 1));
    vec3 first = normalize(dot(max_ab, max_ab) < dot(c, c) ? c : max_ab);
    b1 = .05f * first;
    b2 = .1f * cross(normal, b1);
}

// main.cpp:338-340: the flat normal is cross(p3-p1, p2-p1), i.e. it points AGAINST the
// geometric normal cross(p2-p1, p3-p1) that the hit test uses.
Triangle::Triangle(vec3 a, vec3 b, vec3 c) : p1(a), p2(b), p3(c) { n1 = n2 = n3 = normalize(cross(p3 - p1, p2 - p1)); }
vec3 Triangle::getMinCoords() const { return vmin(p1, vmin(p2, p3)); }
vec3 Triangle::getMaxCoords() const { return vmax(p1, vmax(p2, p3)); }

Diamond::Diamond(vec3 p1, vec3 p2, vec3 p3, vec3 p4) : t1(new Triangle(p1, p2, p4)), t2(new Triangle(p2, p3, p4)) {}
Diamond::Diamond(vec3 p1, vec3 p2, vec3 p3, vec3 p4, vec3 n1, vec3 n2, vec3 n3, vec3 n4)
    : t1(new Triangle(p1, p2, p4, n1, n2, n4)), t2(new Triangle(p2, p3, p4, n2, n3, n4)) {}
Diamond::Diamond(vec3 p1, vec3 p2, vec3 p3, vec3 p4, vec3 n1, vec3 n2, vec3 n3, vec3 n4, vec2 uv1, vec2 uv2, vec2 uv3, vec2 uv4)
    : t1(new Triangle(p1, p2, p4, n1, n2, n4, uv1, uv2, uv4)), t2(new Triangle(p2, p3, p4, n2, n3, n4, uv2, uv3, uv4)) {}
vec3 Diamond::getMinCoords() const { return vmin(t1->getMinCoords(), t2->getMinCoords()); }
vec3 Diamond::getMaxCoords() const { return vmax(t1->getMaxCoords(), t2->getMaxCoords()); }

Mesh::~Mesh() {
    if (owns_fragments)
        for (auto f : fragments) delete f;
}
// main.cpp:495-523: bounds over the fragments in LOCAL space; an empty mesh gives (+inf, -inf)
vec3 Mesh::getMinCoords() const {
    vec3 lo(INFINITY);
    for (auto f : fragments) lo = vmin(lo, f->getMinCoords());
    return lo;
}
vec3 Mesh::getMaxCoords() const {
    vec3 hi(-INFINITY);
    for (auto f : fragments) hi = vmax(hi, f->getMaxCoords());
    return hi;
}
void Mesh::triangles(std::vector<const Triangle *> &out, std::vector<int> &frag, std::vector<int> &sub) const {
    for (size_t i = 0; i < fragments.size(); i++) {
        if (auto d = dynamic_cast<const Diamond *>(fragments[i])) {
            out.push_back(d->t1.get()); frag.push_back((int)i); sub.push_back(0);
            out.push_back(d->t2.get()); frag.push_back((int)i); sub.push_back(1);
        } else if (auto t = dynamic_cast<const Triangle *>(fragments[i])) {
            out.push_back(t); frag.push_back((int)i); sub.push_back(0);
        }
    }
}

// ---- flattening ----------------------------------------------------------------------------------

static void put_affine(const mat4 &m, float *out) {  // four columns, last row dropped
    for (int j = 0; j < 4; j++)
        for (int r = 0; r < 3; r++) out[3 * j + r] = m[j][r];
}
static void put3(vec3 v, float *out) { out[0] = v.x; out[1] = v.y; out[2] = v.z; }

void FlatScene::finish() {
    desc.n_objects = (int32_t)objects.size(); desc.objects = objects.data();
    desc.n_spheres = (int32_t)spheres.size(); desc.spheres = spheres.data();
    desc.n_planes = (int32_t)planes.size(); desc.planes = planes.data();
    desc.n_cones = (int32_t)cones.size(); desc.cones = cones.data();
    desc.n_meshes = (int32_t)meshes.size(); desc.meshes = meshes.data();
    desc.n_triangles = (int32_t)tri_material.size();
    desc.tri_pos = tri_pos.data(); desc.tri_nrm = tri_nrm.data(); desc.tri_uv = tri_uv.data();
    desc.tri_material = tri_material.data(); desc.tri_frag = tri_frag.data(); desc.tri_sub = tri_sub.data();
    desc.n_materials = (int32_t)materials.size(); desc.materials = materials.data();
    desc.n_textures = (int32_t)textures.size(); desc.textures = textures.data();
    desc.n_lights = (int32_t)(light_pos.size() / 3);
    desc.light_pos = light_pos.data(); desc.light_color = light_color.data();
}

namespace {
struct Flattener {
    FlatScene &fs;
    std::string err;  // first problem found while lowering (checked by Scene::flatten)
    explicit Flattener(FlatScene &f) : fs(f) {}

    int texture(const Texture *t) {
        if (!t) return -1;
        for (size_t i = 0; i < fs.texture_keys.size(); i++)
            if (fs.texture_keys[i] == t) return (int)i;
        rt_texture rt{};
        rt.kind = t->kind();
        if (auto it = dynamic_cast<const ImageTexture *>(t)) {
            // The description carries one width x height for the four sampled maps, and the upload reads
            // width * height texels from each.  The reference samples every map with its own size
            // (Textures.cpp:27-30); maps of unequal size are refused instead of being read out of bounds.
            const PNG_Image_t *maps[4] = {&it->base_color, &it->normals_map, &it->ambient_occlusion_map, &it->roughness_map};
            const char *names[4] = {"base colour", "normal", "occlusion", "roughness"};
            for (int k = 0; k < 4; k++) {
                const PNG_Image_t &m = *maps[k];
                if (m.width != maps[0]->width || m.height != maps[0]->height || m.data.size() != (size_t)m.width * m.height * 4) {
                    if (err.empty())
                        err = std::string("image texture: the ") + names[k] + " map is " + std::to_string(m.width) + "x" + std::to_string(m.height) +
                              " (" + std::to_string(m.data.size()) + " bytes) but the base colour map is " + std::to_string(maps[0]->width) + "x" +
                              std::to_string(maps[0]->height) + "; the four sampled maps must have one size";
                    return -1;
                }
            }
            rt.width = it->base_color.width;
            rt.height = it->base_color.height;
            rt.base_color = it->base_color.data.data();
            rt.normal = it->normals_map.data.data();
            rt.occlusion = it->ambient_occlusion_map.data.data();
            rt.roughness = it->roughness_map.data.data();
        }
        fs.texture_keys.push_back(t);
        fs.textures.push_back(rt);
        return (int)fs.textures.size() - 1;
    }
    int material(const Material &m) {
        rt_material rm{};
        put3(m.ambient, rm.ambient); put3(m.diffuse, rm.diffuse); put3(m.specular, rm.specular);
        rm.shininess = m.shininess; rm.reflectiveness = m.reflectiveness;
        rm.refractiveness = m.refractiveness; rm.refractive_ratio = m.refractive_ratio;
        rm.texture = texture(m.texture);
        for (size_t i = 0; i < fs.materials.size(); i++)
            if (!std::memcmp(&fs.materials[i], &rm, sizeof rm)) return (int)i;
        fs.materials.push_back(rm);
        return (int)fs.materials.size() - 1;
    }
    rt_plane plane(const Plane &p) {
        rt_plane r{};
        put3(p.point, r.point); put3(p.normal, r.normal); put3(p.b1, r.b1); put3(p.b2, r.b2);
        r.material = material(p.material);
        return r;
    }
    void mesh(const Mesh &m, const Box *box) {
        rt_mesh r{};
        put_affine(m.inverseTransformation(), r.to_local);
        put_affine(m.transformation(), r.to_world);
        put_affine(m.normalTransformation(), r.normal_mat);
        std::vector<const Triangle *> tris;
        std::vector<int> frag, sub;
        m.triangles(tris, frag, sub);
        r.tri_offset = (int32_t)fs.tri_material.size();
        r.tri_count = (int32_t)tris.size();
        int last_mat = -1;
        const Triangle *last_tri = nullptr;
        for (size_t i = 0; i < tris.size(); i++) {
            const Triangle &t = *tris[i];
            const vec3 *pv[6] = {&t.p1, &t.p2, &t.p3, &t.n1, &t.n2, &t.n3};
            for (int k = 0; k < 3; k++) { fs.tri_pos.push_back((*pv[k]).x); fs.tri_pos.push_back((*pv[k]).y); fs.tri_pos.push_back((*pv[k]).z); }
            for (int k = 3; k < 6; k++) { fs.tri_nrm.push_back((*pv[k]).x); fs.tri_nrm.push_back((*pv[k]).y); fs.tri_nrm.push_back((*pv[k]).z); }
            const vec2 *uv[3] = {&t.uv1, &t.uv2, &t.uv3};
            for (int k = 0; k < 3; k++) { fs.tri_uv.push_back(uv[k]->x); fs.tri_uv.push_back(uv[k]->y); }
            // consecutive triangles nearly always share one material: skip the table search then
            if (!last_tri || std::memcmp(&last_tri->material, &t.material, sizeof(Material)) != 0) last_mat = material(t.material);
            last_tri = &t;
            fs.tri_material.push_back(last_mat);
            fs.tri_frag.push_back(frag[i]);
            fs.tri_sub.push_back(sub[i]);
        }
        if (box) {
            r.boxed = 1;
            put_affine(box->inverseTransformation(), r.box_to_local);
            put3(box->v1, r.box_min);
            put3(box->v2, r.box_max);
        } else {
            r.boxed = 0;
            put_affine(mat4(1.0f), r.box_to_local);
        }
        fs.meshes.push_back(r);
    }
};
}  // namespace

bool Scene::flatten(FlatScene &out, std::string &err) const {
    out = FlatScene();
    Flattener f(out);
    for (size_t k = 0; k < objects.size(); k++) {
        const Object *o = objects[k];
        rt_object ro{};
        if (auto s = dynamic_cast<const Sphere *>(o)) {
            rt_sphere r{};
            put_affine(s->inverseTransformation(), r.to_local);
            put_affine(s->transformation(), r.to_world);
            put_affine(s->normalTransformation(), r.normal_mat);
            r.material = f.material(s->material);
            ro.kind = RT_OBJ_SPHERE; ro.index = (int32_t)out.spheres.size();
            out.spheres.push_back(r);
        } else if (auto p = dynamic_cast<const Plane *>(o)) {
            ro.kind = RT_OBJ_PLANE; ro.index = (int32_t)out.planes.size();
            out.planes.push_back(f.plane(*p));
        } else if (auto c = dynamic_cast<const Cone *>(o)) {
            rt_cone r{};
            put_affine(c->inverseTransformation(), r.to_local);
            put_affine(c->transformation(), r.to_world);
            put_affine(c->normalTransformation(), r.normal_mat);
            r.cap = f.plane(c->base);
            r.material = f.material(c->material);
            ro.kind = RT_OBJ_CONE; ro.index = (int32_t)out.cones.size();
            out.cones.push_back(r);
        } else if (auto m = dynamic_cast<const Mesh *>(o)) {
            ro.kind = RT_OBJ_MESH; ro.index = (int32_t)out.meshes.size();
            f.mesh(*m, nullptr);
        } else if (auto b = dynamic_cast<const Box *>(o)) {
            auto inner = dynamic_cast<const Mesh *>(b->obj);
            if (!inner) { err = "object " + std::to_string(k) + ": a Box must wrap a Mesh"; return false; }
            ro.kind = RT_OBJ_MESH; ro.index = (int32_t)out.meshes.size();
            f.mesh(*inner, b);
        } else {
            err = "object " + std::to_string(k) + ": unsupported top-level object (wrap triangles in a Mesh)";
            return false;
        }
        out.objects.push_back(ro);
    }
    if (out.materials.empty()) f.material(Material());
    if (!f.err.empty()) { err = f.err; return false; }
    for (auto l : lights) {
        out.light_pos.push_back(l->position.x); out.light_pos.push_back(l->position.y); out.light_pos.push_back(l->position.z);
        out.light_color.push_back(l->color.x); out.light_color.push_back(l->color.y); out.light_color.push_back(l->color.z);
    }
    put3(ambient_light, out.desc.ambient);
    out.finish();
    return true;
}

// ---- GPU-backed Object::intersect ------------------------------------------------------------------

static Hit to_hit(const rt_hit_record &r, Object *obj) {
    Hit h;
    h.hit = r.hit != 0;
    h.normal = vec3(r.normal[0], r.normal[1], r.normal[2]);
    h.intersection = vec3(r.point[0], r.point[1], r.point[2]);
    h.distance = r.distance;
    h.object = h.hit ? obj : nullptr;
    h.fragment = r.frag; h.sub = r.sub;
    h.uv = vec2(r.uv[0], r.uv[1]);
    h.from_outside = r.from_outside != 0;
    return h;
}

void Object::intersect(const std::vector<Ray> &rays, std::vector<Hit> &hits) {
    Scene sc;
    sc.objects.push_back(this);
    Renderer rd(0);
    rd.upload(sc);
    std::vector<float> o(3 * rays.size()), d(3 * rays.size());
    for (size_t i = 0; i < rays.size(); i++) { put3(rays[i].origin, &o[3 * i]); put3(rays[i].direction, &d[3 * i]); }
    std::vector<rt_hit_record> rec(rays.size());
    int rc = rt_intersect_object(rd.handle(), 0, (int64_t)rays.size(), o.data(), d.data(), RT_ACCEL_LBVH, rec.data());
    if (rc != RT_OK) throw std::runtime_error(std::string("rt_intersect_object: ") + rt_last_error(rd.handle()));
    hits.resize(rays.size());
    for (size_t i = 0; i < rays.size(); i++) hits[i] = to_hit(rec[i], this);
}
Hit Object::intersect(Ray ray) {
    std::vector<Hit> h;
    intersect(std::vector<Ray>{ray}, h);
    return h[0];
}

// ---- Renderer --------------------------------------------------------------------------------------

Renderer::Renderer(int device) {
    int rc = rt_context_create(device, &ctx);
    if (rc != RT_OK) throw std::runtime_error("rt_context_create failed (" + std::to_string(rc) + "): no usable GPU?");
}
Renderer::~Renderer() { if (ctx) rt_context_destroy(ctx); }
void Renderer::upload(const Scene &scene) {
    FlatScene fs;
    std::string err;
    if (!scene.flatten(fs, err)) throw std::runtime_error("flatten: " + err);
    if (rt_scene_upload(ctx, &fs.desc) != RT_OK) throw std::runtime_error(std::string("rt_scene_upload: ") + rt_last_error(ctx));
    if (rt_build_accel(ctx) != RT_OK) throw std::runtime_error(std::string("rt_build_accel: ") + rt_last_error(ctx));
}
void Renderer::update(const Scene &scene) {
    FlatScene fs;
    std::string err;
    if (!scene.flatten(fs, err)) throw std::runtime_error("flatten: " + err);
    if (rt_scene_update(ctx, &fs.desc) != RT_OK) throw std::runtime_error(std::string("rt_scene_update: ") + rt_last_error(ctx));
}
void Renderer::render(const rt_render_params &p, float *rgb_linear, uint8_t *rgb8) {
    if (rt_render(ctx, &p, rgb_linear, rgb8) != RT_OK) throw std::runtime_error(std::string("rt_render: ") + rt_last_error(ctx));
}
void Renderer::trace(const std::vector<Ray> &rays, int max_depth, std::vector<Hit> *hits, std::vector<vec3> *radiance, const Scene &scene) {
    std::vector<float> o(3 * rays.size()), d(3 * rays.size()), rad(radiance ? 3 * rays.size() : 0);
    for (size_t i = 0; i < rays.size(); i++) { put3(rays[i].origin, &o[3 * i]); put3(rays[i].direction, &d[3 * i]); }
    std::vector<rt_hit_record> rec(hits ? rays.size() : 0);
    int rc = rt_trace_rays(ctx, (int64_t)rays.size(), o.data(), d.data(), max_depth, RT_ACCEL_LBVH, hits ? rec.data() : nullptr,
                           radiance ? rad.data() : nullptr);
    if (rc != RT_OK) throw std::runtime_error(std::string("rt_trace_rays: ") + rt_last_error(ctx));
    if (hits) {
        hits->resize(rays.size());
        for (size_t i = 0; i < rays.size(); i++)
            (*hits)[i] = to_hit(rec[i], rec[i].object >= 0 && rec[i].object < (int)scene.objects.size() ? scene.objects[rec[i].object] : nullptr);
    }
    if (radiance) {
        radiance->resize(rays.size());
        for (size_t i = 0; i < rays.size(); i++) (*radiance)[i] = vec3(rad[3 * i], rad[3 * i + 1], rad[3 * i + 2]);
    }
}
rt_counters Renderer::counters() {
    rt_counters c{};
    rt_get_counters(ctx, &c);
    return c;
}

}  // namespace rta
