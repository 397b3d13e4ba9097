// image.cpp -- the image sink (Image.cpp:4-39): an int RGB buffer, truncating float -> int
// conversion, and the ASCII P3 writer with the reference's exact byte layout.
#include <zlib.h>

#include <cstdio>
#include <cstring>
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

// 8-bit RGB PNG (colour type 2, no interlace, filter 0 on every row), deflated with zlib: a 4K frame is
// a few MB instead of the ~100 MB of the ASCII P3 file.  Same pixels as writeImage.
bool Image::writeImagePNG(const char *path) const {
    std::vector<unsigned char> raw((size_t)height * (1 + 3 * (size_t)width));
    for (int h = 0; h < height; h++) {
        unsigned char *row = &raw[(size_t)h * (1 + 3 * (size_t)width)];
        row[0] = 0;
        for (size_t i = 0; i < 3 * (size_t)width; i++) {
            int v = data[3 * (size_t)h * width + i];
            row[1 + i] = (unsigned char)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
    }
    uLongf zlen = compressBound((uLong)raw.size());
    std::vector<unsigned char> z(zlen);
    if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 6) != Z_OK) return false;
    std::FILE *f = std::fopen(path, "wb");
    if (!f) return false;
    auto be32 = [](unsigned char *p, uint32_t v) { p[0] = v >> 24; p[1] = v >> 16; p[2] = v >> 8; p[3] = v; };
    auto chunk = [&](const char *type, const unsigned char *body, uint32_t n) {
        unsigned char head[8];
        be32(head, n);
        std::memcpy(head + 4, type, 4);
        std::fwrite(head, 1, 8, f);
        if (n) std::fwrite(body, 1, n, f);
        uLong c = crc32(0L, head + 4, 4);
        if (n) c = crc32(c, body, n);
        unsigned char tail[4];
        be32(tail, (uint32_t)c);
        std::fwrite(tail, 1, 4, f);
    };
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    std::fwrite(sig, 1, 8, f);
    unsigned char ihdr[13];
    be32(ihdr, (uint32_t)width); be32(ihdr + 4, (uint32_t)height);
    ihdr[8] = 8; ihdr[9] = 2; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;
    chunk("IHDR", ihdr, 13);
    chunk("IDAT", z.data(), (uint32_t)zlen);
    chunk("IEND", nullptr, 0);
    bool ok = std::ferror(f) == 0;
    std::fclose(f);
    return ok;
}

}  // namespace rta
