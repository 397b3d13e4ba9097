// scene.hpp -- host-side object API of the renderer.
//
// Mirrors the scene-construction interface of the reference (main.cpp:28-98 Ray/Hit/Object,
// 103-326 Sphere/Plane/Cone, 330-548 Triangle/Diamond/Mesh, 553-619 Light/Box, Material.h,
// Textures.h, Image.h) so that a scene written against the reference's classes can be written the
// same way here.  The objects only DESCRIBE the scene: every intersection and all shading run on
// the GPU through the C ABI in rt_amd.h (Scene::flatten lowers the objects to its plain arrays).
#pragma once

#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "glmini.hpp"
#include "rt_amd.h"

namespace rta {

// ---- textures (Textures.h:18-50) ----------------------------------------------------------------
struct PNG_Image_t {
    unsigned width = 0, height = 0;
    std::vector<unsigned char> data;  // RGBA8, 4 bytes per texel, row-major
};

class Texture {
  public:
    virtual ~Texture() = default;
    virtual int kind() const = 0;  // RT_TEX_*
};
class CheckerBoardTexture : public Texture {
  public:
    int kind() const override { return RT_TEX_CHECKER; }
};
class RainbowTexture : public Texture {
  public:
    int kind() const override { return RT_TEX_RAINBOW; }
};
class ImageTexture : public Texture {
  public:
    PNG_Image_t &base_color, &height_map, &normals_map, &ambient_occlusion_map, &roughness_map;
    ImageTexture(PNG_Image_t &base_color, PNG_Image_t &height_map, PNG_Image_t &normals, PNG_Image_t &ambient_occlusion,
                 PNG_Image_t &roughness)
        : base_color(base_color), height_map(height_map), normals_map(normals), ambient_occlusion_map(ambient_occlusion),
          roughness_map(roughness) {}
    int kind() const override { return RT_TEX_IMAGE; }
};

// decodeOneStep (main.cpp:764-794): any PNG -> RGBA8.  Returns nullptr (after printing the reason)
// when the file cannot be decoded.
PNG_Image_t *decodeOneStep(const char *filename);
// decoder proper; err receives a message on failure
bool decode_png_rgba8(const std::string &path, PNG_Image_t &out, std::string &err);

// ---- material (Material.h:17-26) ----------------------------------------------------------------
struct Material {
    vec3 ambient = vec3(0.0f);
    vec3 diffuse = vec3(1.0f);
    vec3 specular = vec3(0.0f);
    float shininess = 0.0f;
    float reflectiveness = 0.0f;
    float refractiveness = 0.0f;
    float refractive_ratio = 0.0f;
    Texture *texture = nullptr;
};

// ---- rays and hits (main.cpp:28-53) -------------------------------------------------------------
struct Ray {
    vec3 origin, direction;
    Ray(vec3 o, vec3 d) : origin(o), direction(d) {}
};
class Object;
struct Hit {
    bool hit = false;
    vec3 normal, intersection;
    float distance = 0;
    Object *object = nullptr;
    int fragment = 0, sub = 0;  // where inside a mesh the reference's `object` pointer would point
    vec2 uv;
    bool from_outside = true;
};

// ---- objects ------------------------------------------------------------------------------------
class Object {
  protected:
    mat4 transformationMatrix, inverseTransformationMatrix, normalMatrix;

  public:
    Material material;
    virtual ~Object() = default;
    Material getMaterial() const { return material; }
    virtual void setMaterial(Material m) { material = m; }
    // main.cpp:85-93
    void setTransformation(mat4 matrix);
    const mat4 &transformation() const { return transformationMatrix; }
    const mat4 &inverseTransformation() const { return inverseTransformationMatrix; }
    const mat4 &normalTransformation() const { return normalMatrix; }
    virtual vec3 getMinCoords() const = 0;
    virtual vec3 getMaxCoords() const = 0;
    // Object::intersect (main.cpp:72), evaluated on the GPU: the object is uploaded on its own and
    // the ray goes through rt_intersect_object.  Meant for probing; render whole scenes instead.
    Hit intersect(Ray ray);
    void intersect(const std::vector<Ray> &rays, std::vector<Hit> &hits);
};

class Sphere : public Object {  // main.cpp:103-190
  public:
    explicit Sphere(Material m) { material = m; }
    vec3 getMinCoords() const override;
    vec3 getMaxCoords() const override;
};

class Plane : public Object {  // main.cpp:192-258
  public:
    vec3 point, normal, b1, b2;
    Plane(vec3 point, vec3 normal) : point(point), normal(normal) { computeBaseVectors(); }
    Plane(vec3 point, vec3 normal, Material m) : point(point), normal(normal) { computeBaseVectors(); material = m; }
    void computeBaseVectors();  // main.cpp:198-210
    vec3 getMinCoords() const override { return point; }
    vec3 getMaxCoords() const override { return point; }
};

class Cone : public Object {  // main.cpp:260-326
  public:
    Plane base;
    explicit Cone(Material m) : base(vec3(0, 1, 0), vec3(0, 1, 0), m) { material = m; }
    vec3 getMinCoords() const override;
    vec3 getMaxCoords() const override;
};

class Fragment : public Object {};

class Triangle : public Fragment {  // main.cpp:330-423
  public:
    vec3 p1, p2, p3, n1, n2, n3;
    vec2 uv1, uv2, uv3;  // the reference leaves these uninitialised for the constructors without uv
    Triangle(vec3 p1, vec3 p2, vec3 p3);
    Triangle(vec3 p1, vec3 p2, vec3 p3, vec3 n1, vec3 n2, vec3 n3) : p1(p1), p2(p2), p3(p3), n1(n1), n2(n2), n3(n3) {}
    Triangle(vec3 p1, vec3 p2, vec3 p3, vec3 n1, vec3 n2, vec3 n3, vec2 uv1, vec2 uv2, vec2 uv3)
        : p1(p1), p2(p2), p3(p3), n1(n1), n2(n2), n3(n3), uv1(uv1), uv2(uv2), uv3(uv3) {}
    Triangle(vec3 p1, vec3 p2, vec3 p3, vec3 n1, vec3 n2, vec3 n3, Material m) : p1(p1), p2(p2), p3(p3), n1(n1), n2(n2), n3(n3) {
        material = m;
    }
    void set_normals(vec3 a, vec3 b, vec3 c) { n1 = a; n2 = b; n3 = c; }
    vec3 getMinCoords() const override;
    vec3 getMaxCoords() const override;
};

class Diamond : public Fragment {  // main.cpp:425-479: a quad as two triangles (p1,p2,p4), (p2,p3,p4)
  public:
    std::unique_ptr<Triangle> t1, t2;
    Diamond(vec3 p1, vec3 p2, vec3 p3, vec3 p4);
    Diamond(vec3 p1, vec3 p2, vec3 p3, vec3 p4, vec3 n1, vec3 n2, vec3 n3, vec3 n4);
    Diamond(vec3 p1, vec3 p2, vec3 p3, vec3 p4, vec3 n1, vec3 n2, vec3 n3, vec3 n4, vec2 uv1, vec2 uv2, vec2 uv3, vec2 uv4);
    void setMaterial(Material m) override { t1->setMaterial(m); t2->setMaterial(m); }
    vec3 getMinCoords() const override;
    vec3 getMaxCoords() const override;
};

class Mesh : public Object {  // main.cpp:481-548
  public:
    std::vector<Fragment *> fragments;
    bool owns_fragments = false;  // set by getMeshFromOBJ: the mesh deletes its fragments
    explicit Mesh(std::vector<Fragment *> fragments) : fragments(std::move(fragments)) {}
    ~Mesh() override;
    void setMaterial(Material m) override { for (auto f : fragments) f->setMaterial(m); }
    vec3 getMinCoords() const override;
    vec3 getMaxCoords() const override;
    // every triangle of the mesh in fragment order (a quad contributes two)
    void triangles(std::vector<const Triangle *> &out, std::vector<int> &frag, std::vector<int> &sub) const;
};

class Box : public Object {  // main.cpp:561-619: a one-sided 12-triangle gate in front of `obj`
  public:
    vec3 v1, v2;
    Object *obj;
    Box(vec3 v1, vec3 v2, Object *obj) : v1(v1), v2(v2), obj(obj) {}
    vec3 getMinCoords() const override;
    vec3 getMaxCoords() const override;
};

class Light {  // main.cpp:553-559
  public:
    vec3 position, color;
    explicit Light(vec3 position) : position(position), color(1.0f) {}
    Light(vec3 position, vec3 color) : position(position), color(color) {}
};

// getMeshFromOBJ (main.cpp:887-913).  compat = true reproduces the reference parser exactly,
// including that vt/vn lines are also appended to the vertex list; compat = false keeps the three
// lists apart.  A missing file gives a mesh without faces, as in the reference.
Mesh *getMeshFromOBJ(const std::string &filename, bool compat = true);

// ---- scene --------------------------------------------------------------------------------------
// The reference keeps `objects`, `lights` and `ambient_light` as globals (main.cpp:621-623).
struct FlatScene;  // owns the arrays an rt_scene_desc points to

class Scene {
  public:
    std::vector<Object *> objects;
    std::vector<Light *> lights;
    vec3 ambient_light = vec3(0.001f);
    // Lowers the object graph to the plain arrays of the C ABI.  Returns false (and a reason) for
    // constructs the GPU path does not take: a Box around something that is not a Mesh, or a bare
    // Triangle / Diamond in the object list.
    bool flatten(FlatScene &out, std::string &err) const;
};

struct FlatScene {
    std::vector<rt_object> objects;
    std::vector<rt_sphere> spheres;
    std::vector<rt_plane> planes;
    std::vector<rt_cone> cones;
    std::vector<rt_mesh> meshes;
    std::vector<float> tri_pos, tri_nrm, tri_uv;
    std::vector<int32_t> tri_material, tri_frag, tri_sub;
    std::vector<rt_material> materials;
    std::vector<rt_texture> textures;
    std::vector<const Texture *> texture_keys;
    std::vector<float> light_pos, light_color;
    rt_scene_desc desc{};
    void finish();  // points desc at the vectors
};

// ---- image sink (Image.h:22-68, Image.cpp) -------------------------------------------------------
class Image {
    int width, height;
    std::vector<int> data;

  public:
    Image(int width, int height) : width(width), height(height), data(3 * (size_t)width * height, 0) {}
    void setPixel(int x, int y, int r, int g, int b);
    void setPixel(int x, int y, float r, float g, float b);  // int = (float)(255 * c), truncation
    void setPixel(int x, int y, vec3 color) { setPixel(x, y, color.x, color.y, color.z); }
    void setPixels8(const uint8_t *rgb8);   // a whole frame as the renderer delivers it
    void writeImage(const char *path) const;  // ASCII P3, byte-for-byte the reference's layout
    void writeImageBinary(const char *path) const;  // P6, same pixels
    bool writeImagePNG(const char *path) const;     // 8-bit RGB PNG, same pixels
    int w() const { return width; }
    int h() const { return height; }
    const std::vector<int> &pixels() const { return data; }
};

// ---- renderer: RAII wrapper around the C ABI -----------------------------------------------------
class Renderer {
    rt_ctx *ctx = nullptr;

  public:
    explicit Renderer(int device = 0);
    ~Renderer();
    Renderer(const Renderer &) = delete;
    Renderer &operator=(const Renderer &) = delete;
    rt_ctx *handle() { return ctx; }
    void upload(const Scene &scene);  // flatten + rt_scene_upload + rt_build_accel; throws std::runtime_error
    void update(const Scene &scene);  // same geometry, new transformations / materials / lights: rt_scene_update
    void render(const rt_render_params &p, float *rgb_linear, uint8_t *rgb8);
    void trace(const std::vector<Ray> &rays, int max_depth, std::vector<Hit> *hits, std::vector<vec3> *radiance,
               const Scene &scene);
    rt_counters counters();
};

// Scene presets (see presets.cpp): "analytic", "teapot", "teapot_head", "shells", "textured".
// Objects, textures and decoded images are owned by the returned holder.
struct PresetHolder {
    Scene scene;
    std::vector<std::unique_ptr<Object>> owned_objects;
    std::vector<std::unique_ptr<Light>> owned_lights;
    std::vector<std::unique_ptr<Texture>> owned_textures;
    std::vector<std::unique_ptr<PNG_Image_t>> owned_images;
};
bool build_preset(const std::string &name, const std::string &asset_dir, PresetHolder &out, std::string &err);

}  // namespace rta
