// presets.cpp -- scene presets built through the object API.
//
// The reference hard-codes ONE scene in sceneDefinition() (main.cpp:918-1177); most of its objects
// are constructed but left out of the object list by commented-out push_back lines.  The presets
// below are fixed selections of those objects:
//   shells       exactly the scene main.cpp builds: two shell meshes behind Box gates, the
//                refractive textured water plane, the textured sand plane, one light
//   analytic     the spheres (main.cpp:1061-1072) and cones (1157-1171), the commented planes
//                (1150-1155, the back wall untextured) plus a floor at y = -3, all four lights
//   teapot       teapot.obj behind its Box gate (969-977) with its flat normals negated through
//                Triangle::set_normals -- as loaded they face away from every ray and the mesh is
//                invisible -- five walls, two lights
//   teapot_head  the same with the normals as loaded
//   textured     the three image-textured spheres (1119-1135), the refractive sphere, a mirror
//                variant of the blue sphere (935), textured floor and back wall, two lights
// Values are those of main.cpp; anything this build chose itself is marked "chosen here".
#include <cmath>

#include "scene.hpp"

namespace rta {
namespace {

struct Builder {
    PresetHolder &h;
    std::string assets;
    explicit Builder(PresetHolder &holder, std::string dir) : h(holder), assets(std::move(dir)) {}

    template <class T> T *own(T *o) { h.owned_objects.emplace_back(o); return o; }
    void add(Object *o) { h.scene.objects.push_back(o); }
    void light(vec3 p, float c) { h.owned_lights.emplace_back(new Light(p, vec3(c))); h.scene.lights.push_back(h.owned_lights.back().get()); }
    void plane(vec3 p, vec3 n, const Material &m) { add(own(new Plane(p, n, m))); }

    bool image_texture(const char *const names[5], Texture *&out, std::string &err) {
        PNG_Image_t *img[5];
        for (int i = 0; i < 5; i++) {
            std::string path = assets + "/textures/png/" + names[i] + ".png";
            auto im = std::make_unique<PNG_Image_t>();
            std::string e;
            if (!decode_png_rgba8(path, *im, e)) { err = path + ": " + e; return false; }
            img[i] = im.get();
            h.owned_images.push_back(std::move(im));
        }
        h.owned_textures.emplace_back(new ImageTexture(*img[0], *img[1], *img[2], *img[3], *img[4]));
        out = h.owned_textures.back().get();
        return true;
    }
    // a mesh from an OBJ file behind the Box gate the reference puts in front of it
    Mesh *boxed_mesh(const std::string &file, const Material &m, const mat4 &xf, bool flip) {
        Mesh *mesh = getMeshFromOBJ(assets + "/" + file);
        h.owned_objects.emplace_back(mesh);
        mesh->setMaterial(m);
        if (flip)
            for (auto f : mesh->fragments)
                if (auto t = dynamic_cast<Triangle *>(f)) t->set_normals(-t->n1, -t->n2, -t->n3);
        mesh->setTransformation(xf);
        Box *box = own(new Box(mesh->getMinCoords(), mesh->getMaxCoords(), mesh));
        box->setTransformation(xf);
        add(box);
        return mesh;
    }
};

Material mk(vec3 ambient, vec3 diffuse, float specular, float shininess) {
    Material m;
    m.ambient = ambient; m.diffuse = diffuse; m.specular = vec3(specular); m.shininess = shininess;
    return m;
}

const char *const kSand[5] = {"Sand_005_baseColor", "Sand_005_height", "Sand_005_normal", "Sand_005_ambientOcclusion", "Sand_005_roughness"};
const char *const kWater[5] = {"Water_002_COLOR", "Water_002_DISP", "Water_002_NORM", "Water_002_OCC", "Water_002_ROUGH"};
const char *const kStone[5] = {"Stylized_Stone_Floor_005_basecolor", "Stylized_Stone_Floor_005_height", "Stylized_Stone_Floor_005_normal",
                               "Stylized_Stone_Floor_005_ambientOcclusion", "Stylized_Stone_Floor_005_roughness"};
const char *const kWaffle[5] = {"Waffle_001_basecolor", "Waffle_001_height", "Waffle_001_normal", "Waffle_001_ambientOcclusion", "Waffle_001_roughness"};

}  // namespace

bool build_preset(const std::string &name, const std::string &asset_dir, PresetHolder &out, std::string &err) {
    Builder b(out, asset_dir);
    // main.cpp:920-941, 1057-1059, 1138-1144
    Material green_diffuse = mk(vec3(0.03f, 0.1f, 0.03f), vec3(0.3f, 1.0f, 0.3f), 0, 0);
    Material red_specular = mk(vec3(0.01f, 0.02f, 0.02f), vec3(1.0f, 0.2f, 0.2f), 0.5f, 10.0f);
    Material blue_specular = mk(vec3(0.02f, 0.02f, 0.1f), vec3(0.2f, 0.2f, 1.0f), 0.6f, 100.0f);
    Material highly_specular_yellow = mk(vec3(0.1f, 0.1f, 0.03f), vec3(.6f, .6f, 0.1f), .6f, 100);
    Material refractive;
    refractive.reflectiveness = 0.1f; refractive.refractiveness = 2.0f;
    Material red_diffuse = mk(vec3(0.09f, 0.06f, 0.06f), vec3(0.9f, 0.6f, 0.6f), 0, 0);
    Material blue_diffuse = mk(vec3(0.06f, 0.06f, 0.09f), vec3(0.6f, 0.6f, 0.9f), 0, 0);
    Material plain;

    auto walls = [&](const Material &back, const Material &floor) {
        b.plane(vec3(0, -3, 0), vec3(0, 1, 0), floor);  // chosen here: the reference has no floor
        b.plane(vec3(0, 1, 30), vec3(0.0, 0.0, -1.0), back);
        b.plane(vec3(-15, 1, 0), vec3(1.0, 0.0, 0.0), red_diffuse);
        b.plane(vec3(15, 1, 0), vec3(-1.0, 0.0, 0.0), blue_diffuse);
        b.plane(vec3(0, 27, 0), vec3(0.0, -1, 0), plain);
    };

    if (name == "analytic") {
        Sphere *blue = b.own(new Sphere(blue_specular));
        blue->setTransformation(translate(vec3(1, -2, 8)));
        b.add(blue);
        Sphere *red = b.own(new Sphere(red_specular));
        red->setTransformation(scale(translate(vec3(-1, -2.5, 6)), vec3(.5)));
        b.add(red);
        Sphere *glass = b.own(new Sphere(refractive));
        glass->setTransformation(translate(vec3(-3, -1, 8)) * scale(vec3(2)));
        b.add(glass);
        Cone *cone1 = b.own(new Cone(green_diffuse));
        cone1->setTransformation(scale(translate(vec3(6, -3, 7)) * rotate((float)std::atan(3.0), vec3(0, 0, 1)), vec3(1, 3, 1)));
        b.add(cone1);
        Cone *cone2 = b.own(new Cone(highly_specular_yellow));
        cone2->setTransformation(scale(translate(vec3(-5, 9, 14)) * rotate(3.1415f, vec3(0, 0, 1)), vec3(3, 12, 3)));
        b.add(cone2);
        walls(green_diffuse, plain);
        b.plane(vec3(0, 1, -0.01), vec3(0.0, 0.0, 1.0), green_diffuse);
        b.light(vec3(10, 16, -7), .5f);
        b.light(vec3(2, 1, -15), .5f);
        b.light(vec3(6, 1, 17), 0.3f);
        b.light(vec3(2, 7, 1), 0.2f);
        return true;
    }
    if (name == "teapot" || name == "teapot_head") {
        b.boxed_mesh("teapot.obj", red_specular, translate(vec3(0, 1, 10)), name == "teapot");
        walls(green_diffuse, plain);
        b.light(vec3(10, 16, -7), .5f);
        b.light(vec3(2, 7, 1), 0.2f);  // chosen here: second light from main.cpp:1176
        return true;
    }
    if (name == "shells") {
        Material shell_textured = mk(vec3(0.01f, 0.02f, 0.02f), vec3(0.8f, 0.2f, 0.2f), 0.5f, 10.0f);
        Material shell3_textured = mk(vec3(0.01f, 0.02f, 0.02f), vec3(148, 106, 154) / 255.f, 0.5f, 10.0f);
        float angle = (float)std::atan(-3.0);
        b.boxed_mesh("seashell_obj.obj", shell_textured, translate(vec3(-10, 5, 24.7)) * rotate(angle, vec3(1, 0, 0)) * scale(vec3(0.15)), false);
        b.boxed_mesh("shell.obj", shell3_textured, translate(vec3(9, -8, 16.7)) * rotate(angle, vec3(1, .5, 0)) * scale(vec3(1)), false);
        Texture *sand = nullptr, *water = nullptr;
        if (!b.image_texture(kSand, sand, err) || !b.image_texture(kWater, water, err)) return false;
        Material water_textured;
        water_textured.specular = vec3(.6f); water_textured.refractiveness = .825f; water_textured.refractive_ratio = .2f;
        water_textured.texture = water;
        Material sand_textured;
        sand_textured.specular = vec3(.2f); sand_textured.shininess = 100; sand_textured.texture = sand;
        b.plane(vec3(0, 0, 10), normalize(vec3(0.0, 1, -2)), water_textured);
        b.plane(vec3(0, 0, 25), normalize(vec3(0, 1, -3)), sand_textured);
        b.light(vec3(10, 16, -7), .5f);
        return true;
    }
    if (name == "textured") {
        Texture *sand = nullptr, *water = nullptr, *stone = nullptr, *waffle = nullptr;
        if (!b.image_texture(kSand, sand, err) || !b.image_texture(kWater, water, err) || !b.image_texture(kStone, stone, err) ||
            !b.image_texture(kWaffle, waffle, err))
            return false;
        Material stone_textured; stone_textured.specular = vec3(.6f); stone_textured.shininess = 100; stone_textured.texture = stone;
        Material waffle_textured; waffle_textured.specular = vec3(.6f); waffle_textured.shininess = 100; waffle_textured.texture = waffle;
        Material water_textured; water_textured.specular = vec3(.6f); water_textured.refractiveness = .825f;
        water_textured.refractive_ratio = .2f; water_textured.texture = water;
        Material sand_textured; sand_textured.specular = vec3(.2f); sand_textured.shininess = 100; sand_textured.texture = sand;
        Material mirror_blue = blue_specular; mirror_blue.reflectiveness = .9f;
        mat4 rotation = rotate(.7f, vec3(1, 0, 0)), scaling = scale(vec3(1.5));
        Sphere *water_sphere = b.own(new Sphere(water_textured));
        water_sphere->setTransformation(translate(vec3(0, 1.5, 10)) * rotation * scaling);
        b.add(water_sphere);
        Sphere *waffle_sphere = b.own(new Sphere(waffle_textured));
        waffle_sphere->setTransformation(translate(vec3(-5, 3.5, 12)) * rotation * scaling);  // position chosen here
        b.add(waffle_sphere);
        Sphere *stone_sphere = b.own(new Sphere(stone_textured));
        stone_sphere->setTransformation(translate(vec3(4, 2, 10)) * rotate(.2f, vec3(0, 1, 0)) * scale(vec3(3)));
        b.add(stone_sphere);
        Sphere *glass = b.own(new Sphere(refractive));
        glass->setTransformation(translate(vec3(-3, -1, 8)) * scale(vec3(2)));
        b.add(glass);
        Sphere *mirror = b.own(new Sphere(mirror_blue));
        mirror->setTransformation(translate(vec3(1, -2, 8)));
        b.add(mirror);
        walls(stone_textured, sand_textured);
        b.light(vec3(10, 16, -7), .5f);
        b.light(vec3(2, 7, 1), 0.2f);
        return true;
    }
    err = "unknown preset '" + name + "'";
    return false;
}

}  // namespace rta
