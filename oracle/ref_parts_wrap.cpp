// TEST INFRASTRUCTURE ONLY.  C entry points around the two reference translation units that
// compile from their own sources (Textures.cpp, Image.cpp + the vendored GLM).  The reference
// sources are compiled where they lie; nothing of them is copied here.  Used to validate the
// oracle's texture samplers and its quantisation / PPM writer against the real code.
#include "Textures.h"
#include "Image.h"
#include <cstdint>
#include <cstring>

extern "C" {

// which: 0 texture(), 1 normal(), 2 occlusion(), 3 roughness() of an ImageTexture whose five maps
// all alias one RGBA8 image; 4 CheckerBoardTexture::texture, 5 RainbowTexture::texture
void ref_sample(int which, unsigned w, unsigned h, const uint8_t *rgba, int n, const float *uv, float *out3) {
    PNG_Image_t img;
    img.width = w;
    img.height = h;
    if (rgba) img.data.assign(rgba, rgba + (size_t)w * h * 4);
    ImageTexture it(img, img, img, img, img);
    struct Ck : CheckerBoardTexture { glm::vec3 call(glm::vec2 uv) { return ((Texture *)this)->texture(uv); } } ck;
    struct Rb : RainbowTexture { glm::vec3 call(glm::vec2 uv) { return ((Texture *)this)->texture(uv); } } rb;
    for (int i = 0; i < n; i++) {
        glm::vec2 q(uv[2 * i], uv[2 * i + 1]);
        glm::vec3 r;
        switch (which) {
        case 0: r = it.texture(q); break;
        case 1: r = it.normal(q); break;
        case 2: r = it.occlusion(q); break;
        case 3: r = glm::vec3(it.roughness(q)); break;
        case 4: r = ck.call(q); break;
        default: r = rb.call(q); break;
        }
        out3[3 * i] = r.x; out3[3 * i + 1] = r.y; out3[3 * i + 2] = r.z;
    }
}

// Image::setPixel(x, y, vec3) for every pixel, then Image::writeImage(path)
void ref_write_image(const char *path, int w, int h, const float *rgb01) {
    Image im(w, h);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const float *p = rgb01 + 3 * ((size_t)y * w + x);
            im.setPixel(x, y, glm::vec3(p[0], p[1], p[2]));
        }
    im.writeImage(path);
}
}
