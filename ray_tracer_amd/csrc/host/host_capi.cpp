// host_capi.cpp -- C entry points of librt_host.so, so that non-C++ callers (the Python tests and
// bench) can build scenes through the object API and obtain the flattened description that
// rt_scene_upload takes.  No ray is traced here.
#include <cstring>

#include "scene.hpp"

using namespace rta;

namespace {
struct HostScene {
    PresetHolder holder;
    FlatScene flat;
    std::vector<Material> materials;  // handles for the builder API
    std::string err;
    bool flattened = false;
};
mat4 to_mat4(const float *m) {
    mat4 r;
    for (int j = 0; j < 4; j++)
        for (int i = 0; i < 4; i++) r[j][i] = m[4 * j + i];
    return r;
}
Material to_material(HostScene *s, const rt_material *m, int texture_handle) {
    Material r;
    r.ambient = vec3(m->ambient[0], m->ambient[1], m->ambient[2]);
    r.diffuse = vec3(m->diffuse[0], m->diffuse[1], m->diffuse[2]);
    r.specular = vec3(m->specular[0], m->specular[1], m->specular[2]);
    r.shininess = m->shininess; r.reflectiveness = m->reflectiveness;
    r.refractiveness = m->refractiveness; r.refractive_ratio = m->refractive_ratio;
    r.texture = (texture_handle >= 0 && texture_handle < (int)s->holder.owned_textures.size()) ? s->holder.owned_textures[texture_handle].get() : nullptr;
    return r;
}
}  // namespace

extern "C" {

void *rth_scene_new() { return new HostScene(); }
void rth_scene_free(void *h) { delete static_cast<HostScene *>(h); }
const char *rth_last_error(void *h) { return static_cast<HostScene *>(h)->err.c_str(); }

int rth_build_preset(void *h, const char *name, const char *asset_dir) {
    auto s = static_cast<HostScene *>(h);
    return build_preset(name, asset_dir, s->holder, s->err) ? 0 : -1;
}

// textures: return a handle usable as rt_material.texture in the calls below
int rth_add_checker_texture(void *h) {
    auto s = static_cast<HostScene *>(h);
    s->holder.owned_textures.emplace_back(new CheckerBoardTexture());
    return (int)s->holder.owned_textures.size() - 1;
}
int rth_add_rainbow_texture(void *h) {
    auto s = static_cast<HostScene *>(h);
    s->holder.owned_textures.emplace_back(new RainbowTexture());
    return (int)s->holder.owned_textures.size() - 1;
}
// the same with a size per map (w[i] x hgt[i]); maps of unequal size are refused when the scene is flattened
int rth_add_image_texture_sized(void *h, const unsigned w[5], const unsigned hgt[5], const uint8_t *const maps[5]) {
    auto s = static_cast<HostScene *>(h);
    PNG_Image_t *img[5];
    for (int i = 0; i < 5; i++) {
        auto im = std::make_unique<PNG_Image_t>();
        im->width = w[i]; im->height = hgt[i];
        if (maps[i]) im->data.assign(maps[i], maps[i] + (size_t)w[i] * hgt[i] * 4);
        else im->data.assign((size_t)w[i] * hgt[i] * 4, 0);
        img[i] = im.get();
        s->holder.owned_images.push_back(std::move(im));
    }
    s->holder.owned_textures.emplace_back(new ImageTexture(*img[0], *img[1], *img[2], *img[3], *img[4]));
    return (int)s->holder.owned_textures.size() - 1;
}
// five RGBA8 maps (base, height, normal, occlusion, roughness) of w*h texels each, copied
int rth_add_image_texture(void *h, unsigned w, unsigned hgt, const uint8_t *const maps[5]) {
    auto s = static_cast<HostScene *>(h);
    PNG_Image_t *img[5];
    for (int i = 0; i < 5; i++) {
        auto im = std::make_unique<PNG_Image_t>();
        im->width = w; im->height = hgt;
        if (maps[i]) im->data.assign(maps[i], maps[i] + (size_t)w * hgt * 4);
        else im->data.assign((size_t)w * hgt * 4, 0);
        img[i] = im.get();
        s->holder.owned_images.push_back(std::move(im));
    }
    s->holder.owned_textures.emplace_back(new ImageTexture(*img[0], *img[1], *img[2], *img[3], *img[4]));
    return (int)s->holder.owned_textures.size() - 1;
}
int rth_add_image_texture_files(void *h, const char *const paths[5]) {
    auto s = static_cast<HostScene *>(h);
    PNG_Image_t *img[5];
    for (int i = 0; i < 5; i++) {
        auto im = std::make_unique<PNG_Image_t>();
        std::string e;
        if (!decode_png_rgba8(paths[i], *im, e)) { s->err = std::string(paths[i]) + ": " + e; return -1; }
        img[i] = im.get();
        s->holder.owned_images.push_back(std::move(im));
    }
    for (int i = 2; i < 5; i++)  // the sampled maps (base colour, normal, occlusion, roughness) share one size; the height map is never read
        if (img[i]->width != img[0]->width || img[i]->height != img[0]->height) {
            s->err = std::string(paths[i]) + " is " + std::to_string(img[i]->width) + "x" + std::to_string(img[i]->height) + " but " + paths[0] +
                     " is " + std::to_string(img[0]->width) + "x" + std::to_string(img[0]->height) + ": the sampled maps of an image texture must have one size";
            return -1;
        }
    s->holder.owned_textures.emplace_back(new ImageTexture(*img[0], *img[1], *img[2], *img[3], *img[4]));
    return (int)s->holder.owned_textures.size() - 1;
}

// objects: M is a column-major 4x4 (glm::mat4 layout); mat->texture is a handle from above or -1
int rth_add_sphere(void *h, const float *M, const rt_material *mat) {
    auto s = static_cast<HostScene *>(h);
    auto o = new Sphere(to_material(s, mat, mat->texture));
    o->setTransformation(to_mat4(M));
    s->holder.owned_objects.emplace_back(o);
    s->holder.scene.objects.push_back(o);
    return (int)s->holder.scene.objects.size() - 1;
}
int rth_add_plane(void *h, const float *point, const float *normal, const rt_material *mat) {
    auto s = static_cast<HostScene *>(h);
    auto o = new Plane(vec3(point[0], point[1], point[2]), vec3(normal[0], normal[1], normal[2]), to_material(s, mat, mat->texture));
    s->holder.owned_objects.emplace_back(o);
    s->holder.scene.objects.push_back(o);
    return (int)s->holder.scene.objects.size() - 1;
}
int rth_add_cone(void *h, const float *M, const rt_material *mat) {
    auto s = static_cast<HostScene *>(h);
    auto o = new Cone(to_material(s, mat, mat->texture));
    o->setTransformation(to_mat4(M));
    s->holder.owned_objects.emplace_back(o);
    s->holder.scene.objects.push_back(o);
    return (int)s->holder.scene.objects.size() - 1;
}
static int finish_mesh(HostScene *s, Mesh *mesh, const float *M, int boxed) {
    s->holder.owned_objects.emplace_back(mesh);
    mesh->setTransformation(to_mat4(M));
    Object *top = mesh;
    if (boxed) {
        auto box = new Box(mesh->getMinCoords(), mesh->getMaxCoords(), mesh);
        box->setTransformation(to_mat4(M));
        s->holder.owned_objects.emplace_back(box);
        top = box;
    }
    s->holder.scene.objects.push_back(top);
    return (int)s->holder.scene.objects.size() - 1;
}
int rth_add_mesh_obj(void *h, const char *path, const float *M, const rt_material *mat, int boxed, int flip_normals, int compat) {
    auto s = static_cast<HostScene *>(h);
    Mesh *mesh = getMeshFromOBJ(path, compat != 0);
    mesh->setMaterial(to_material(s, mat, mat->texture));
    if (flip_normals)
        for (auto f : mesh->fragments) {
            if (auto t = dynamic_cast<Triangle *>(f)) t->set_normals(-t->n1, -t->n2, -t->n3);
            else if (auto d = dynamic_cast<Diamond *>(f)) {
                d->t1->set_normals(-d->t1->n1, -d->t1->n2, -d->t1->n3);
                d->t2->set_normals(-d->t2->n1, -d->t2->n2, -d->t2->n3);
            }
        }
    return finish_mesh(s, mesh, M, boxed);
}
// explicit triangles: pos/nrm [n][9]; mats (optional) one material per triangle, else `mat` for all
int rth_add_mesh_tris(void *h, int n, const float *pos, const float *nrm, const rt_material *mats, const float *M,
                      const rt_material *mat, int boxed) {
    auto s = static_cast<HostScene *>(h);
    std::vector<Fragment *> frags;
    for (int i = 0; i < n; i++) {
        const float *p = pos + 9 * i, *q = nrm + 9 * i;
        const rt_material *m = mats ? &mats[i] : mat;
        frags.push_back(new Triangle(vec3(p[0], p[1], p[2]), vec3(p[3], p[4], p[5]), vec3(p[6], p[7], p[8]), vec3(q[0], q[1], q[2]),
                                     vec3(q[3], q[4], q[5]), vec3(q[6], q[7], q[8]), to_material(s, m, m->texture)));
    }
    Mesh *mesh = new Mesh(frags);
    mesh->owns_fragments = true;
    return finish_mesh(s, mesh, M, boxed);
}
int rth_add_light(void *h, const float *pos, const float *color) {
    auto s = static_cast<HostScene *>(h);
    s->holder.owned_lights.emplace_back(new Light(vec3(pos[0], pos[1], pos[2]), vec3(color[0], color[1], color[2])));
    s->holder.scene.lights.push_back(s->holder.owned_lights.back().get());
    return (int)s->holder.scene.lights.size() - 1;
}

// Flattens the scene; the description stays valid until the scene is freed or flattened again.
const rt_scene_desc *rth_flatten(void *h) {
    auto s = static_cast<HostScene *>(h);
    if (!s->holder.scene.flatten(s->flat, s->err)) return nullptr;
    s->flattened = true;
    return &s->flat.desc;
}

// glm-compatible matrix helpers (column-major float[16])
void rth_mat_identity(float *out) { mat4 m(1.0f); std::memcpy(out, &m, sizeof m); }
void rth_mat_translate(const float *m, const float *v, float *out) { mat4 r = translate(to_mat4(m), vec3(v[0], v[1], v[2])); std::memcpy(out, &r, sizeof r); }
void rth_mat_rotate(const float *m, float angle, const float *axis, float *out) { mat4 r = rotate(to_mat4(m), angle, vec3(axis[0], axis[1], axis[2])); std::memcpy(out, &r, sizeof r); }
void rth_mat_scale(const float *m, const float *v, float *out) { mat4 r = scale(to_mat4(m), vec3(v[0], v[1], v[2])); std::memcpy(out, &r, sizeof r); }
void rth_mat_mul(const float *a, const float *b, float *out) { mat4 r = to_mat4(a) * to_mat4(b); std::memcpy(out, &r, sizeof r); }

// PNG -> RGBA8 into a caller buffer of w*h*4 bytes (query the size with out == NULL)
int rth_decode_png(const char *path, unsigned *w, unsigned *h, uint8_t *out, size_t cap) {
    PNG_Image_t im;
    std::string e;
    if (!decode_png_rgba8(path, im, e)) return -1;
    if (w) *w = im.width;
    if (h) *h = im.height;
    if (out) {
        if (cap < im.data.size()) return -2;
        std::memcpy(out, im.data.data(), im.data.size());
    }
    return 0;
}

// Image sink: quantised frame -> ASCII P3 (binary = 0), P6 (binary = 1) or PNG (binary = 2)
int rth_write_ppm(const char *path, int w, int h, const uint8_t *rgb8, int binary) {
    Image im(w, h);
    im.setPixels8(rgb8);
    if (binary == 2) return im.writeImagePNG(path) ? 0 : -1;
    if (binary) im.writeImageBinary(path); else im.writeImage(path);
    return 0;
}
}
