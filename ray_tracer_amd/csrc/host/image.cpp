// image.cpp -- the image sink (Image.cpp:4-39): an int RGB buffer, truncating float -> int
// conversion, and the ASCII P3 writer with the reference's exact byte layout.
#include <cstdio>
#include <fstream>

#include "scene.hpp"

namespace rta {

void Image::setPixel(int x, int y, int r, int g, int b) {
    size_t p = 3 * ((size_t)y * width + x);
    data[p] = r; data[p + 1] = g; data[p + 2] = b;
}
// Image.cpp:30-39: data = (float)(255 * c), i.e. truncation toward zero
void Image::setPixel(int x, int y, float r, float g, float b) {
    size_t p = 3 * ((size_t)y * width + x);
    data[p] = (int)(float)(255 * r); data[p + 1] = (int)(float)(255 * g); data[p + 2] = (int)(float)(255 * b);
}
void Image::setPixels8(const uint8_t *rgb8) {
    for (size_t i = 0; i < data.size(); i++) data[i] = rgb8[i];
}
// Image.cpp:8-23: "P3\nW H\n255\n", then "r g b  " per pixel and a newline per row
void Image::writeImage(const char *path) const {
    std::FILE *f = std::fopen(path, "wb");
    if (!f) return;
    std::fprintf(f, "P3\n%d %d\n255\n", width, height);
    std::string row;
    for (int h = 0; h < height; h++) {
        row.clear();
        for (int w = 0; w < width; w++) {
            const int *p = &data[3 * ((size_t)h * width + w)];
            row += std::to_string(p[0]); row += ' ';
            row += std::to_string(p[1]); row += ' ';
            row += std::to_string(p[2]); row += "  ";
        }
        row += '\n';
        std::fwrite(row.data(), 1, row.size(), f);
    }
    std::fclose(f);
}
void Image::writeImageBinary(const char *path) const {
    std::FILE *f = std::fopen(path, "wb");
    if (!f) return;
    std::fprintf(f, "P6\n%d %d\n255\n", width, height);
    std::vector<unsigned char> row(3 * (size_t)width);
    for (int h = 0; h < height; h++) {
        for (size_t i = 0; i < row.size(); i++) {
            int v = data[3 * (size_t)h * width + i];
            row[i] = (unsigned char)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
        std::fwrite(row.data(), 1, row.size(), f);
    }
    std::fclose(f);
}

}  // namespace rta
