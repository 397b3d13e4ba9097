// png_decode.cpp -- PNG -> RGBA8 (what the reference obtains from lodepng::decode in decodeOneStep,
// main.cpp:764-794; lodepng itself is an absent submodule of the reference).  Handles every
// colour type (grey, RGB, palette, grey+alpha, RGBA), bit depths 1-16, tRNS transparency and Adam7
// interlacing.  16-bit samples keep their high byte.  zlib does the inflate.
#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>

#include "scene.hpp"

namespace rta {
namespace {

uint32_t be32(const unsigned char *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

int paeth(int a, int b, int c) {
    int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// undo the per-scanline filters of one (sub)image in place; returns false on a bad filter type
bool unfilter(unsigned char *data, size_t rows, size_t row_bytes, size_t bpp) {
    std::vector<unsigned char> zero(row_bytes, 0);
    const unsigned char *prev = zero.data();
    for (size_t y = 0; y < rows; y++) {
        unsigned char *line = data + y * (row_bytes + 1);
        unsigned char ft = line[0];
        unsigned char *cur = line + 1;
        for (size_t x = 0; x < row_bytes; x++) {
            int a = x >= bpp ? cur[x - bpp] : 0, b = prev[x], c = x >= bpp ? prev[x - bpp] : 0;
            int add;
            switch (ft) {
            case 0: add = 0; break;
            case 1: add = a; break;
            case 2: add = b; break;
            case 3: add = (a + b) >> 1; break;
            case 4: add = paeth(a, b, c); break;
            default: return false;
            }
            cur[x] = (unsigned char)(cur[x] + add);
        }
        prev = cur;
    }
    return true;
}

struct Header {
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = 0, interlace = 0;
    int channels() const { return ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : 4; }
    size_t bits_per_pixel() const { return (size_t)channels() * depth; }
};

// sample s (0-based, `depth` bits each, MSB first) of a scanline
unsigned sample(const unsigned char *row, size_t s, int depth) {
    if (depth == 8) return row[s];
    if (depth == 16) return ((unsigned)row[2 * s] << 8) | row[2 * s + 1];
    size_t bit = s * depth;
    unsigned byte = row[bit >> 3];
    int shift = 8 - depth - (int)(bit & 7);
    return (byte >> shift) & ((1u << depth) - 1);
}

struct Trns {
    bool present = false;
    unsigned grey = 0, r = 0, g = 0, b = 0;
    std::vector<unsigned char> alpha;  // palette
};

// converts pixel x of a scanline to RGBA8
void to_rgba(const Header &hd, const unsigned char *row, size_t x, const std::vector<unsigned char> &plte, const Trns &tr,
             unsigned char *out) {
    int d = hd.depth;
    auto to8 = [&](unsigned v) -> unsigned char {
        if (d == 8) return (unsigned char)v;
        if (d == 16) return (unsigned char)(v >> 8);
        return (unsigned char)((v * 255u) / ((1u << d) - 1));
    };
    switch (hd.ctype) {
    case 0: {
        unsigned v = sample(row, x, d);
        out[0] = out[1] = out[2] = to8(v);
        out[3] = (tr.present && v == tr.grey) ? 0 : 255;
        break;
    }
    case 2: {
        unsigned r = sample(row, 3 * x, d), g = sample(row, 3 * x + 1, d), b = sample(row, 3 * x + 2, d);
        out[0] = to8(r); out[1] = to8(g); out[2] = to8(b);
        out[3] = (tr.present && r == tr.r && g == tr.g && b == tr.b) ? 0 : 255;
        break;
    }
    case 3: {
        unsigned i = sample(row, x, d);
        if (3 * (size_t)i + 2 < plte.size()) { out[0] = plte[3 * i]; out[1] = plte[3 * i + 1]; out[2] = plte[3 * i + 2]; }
        else { out[0] = out[1] = out[2] = 0; }
        out[3] = i < tr.alpha.size() ? tr.alpha[i] : 255;
        break;
    }
    case 4: {
        unsigned v = sample(row, 2 * x, d), a = sample(row, 2 * x + 1, d);
        out[0] = out[1] = out[2] = to8(v);
        out[3] = to8(a);
        break;
    }
    default: {
        out[0] = to8(sample(row, 4 * x, d)); out[1] = to8(sample(row, 4 * x + 1, d));
        out[2] = to8(sample(row, 4 * x + 2, d)); out[3] = to8(sample(row, 4 * x + 3, d));
    }
    }
}

}  // namespace

bool decode_png_rgba8(const std::string &path, PNG_Image_t &out, std::string &err) {
    std::ifstream f(path, std::ios::binary);
    if (!f) { err = "cannot open file"; return false; }
    std::vector<unsigned char> buf((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    static const unsigned char sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    if (buf.size() < 8 + 25 || std::memcmp(buf.data(), sig, 8) != 0) { err = "not a PNG file"; return false; }
    Header hd;
    std::vector<unsigned char> idat, plte;
    Trns tr;
    bool have_ihdr = false;
    size_t pos = 8;
    while (pos + 12 <= buf.size()) {
        uint32_t len = be32(&buf[pos]);
        const unsigned char *type = &buf[pos + 4], *data = &buf[pos + 8];
        if (pos + 12 + (size_t)len > buf.size()) { err = "truncated chunk"; return false; }
        if (!std::memcmp(type, "IHDR", 4)) {
            if (len < 13) { err = "bad IHDR"; return false; }
            hd.w = be32(data); hd.h = be32(data + 4); hd.depth = data[8]; hd.ctype = data[9]; hd.interlace = data[12];
            have_ihdr = true;
        } else if (!std::memcmp(type, "PLTE", 4)) {
            plte.assign(data, data + len);
        } else if (!std::memcmp(type, "tRNS", 4)) {
            tr.present = true;
            if (hd.ctype == 3) tr.alpha.assign(data, data + len);
            else if (hd.ctype == 0 && len >= 2) tr.grey = ((unsigned)data[0] << 8) | data[1];
            else if (hd.ctype == 2 && len >= 6) { tr.r = ((unsigned)data[0] << 8) | data[1]; tr.g = ((unsigned)data[2] << 8) | data[3]; tr.b = ((unsigned)data[4] << 8) | data[5]; }
        } else if (!std::memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), data, data + len);
        } else if (!std::memcmp(type, "IEND", 4)) {
            break;
        }
        pos += 12 + (size_t)len;
    }
    if (!have_ihdr || hd.w == 0 || hd.h == 0 || hd.w > 65536 || hd.h > 65536) { err = "missing or unreasonable IHDR"; return false; }
    bool depth_ok = (hd.ctype == 0 && (hd.depth == 1 || hd.depth == 2 || hd.depth == 4 || hd.depth == 8 || hd.depth == 16)) ||
                    (hd.ctype == 3 && (hd.depth == 1 || hd.depth == 2 || hd.depth == 4 || hd.depth == 8)) ||
                    ((hd.ctype == 2 || hd.ctype == 4 || hd.ctype == 6) && (hd.depth == 8 || hd.depth == 16));
    if (!depth_ok || hd.interlace > 1) { err = "unsupported colour type / bit depth / interlace method"; return false; }

    // pass geometry: one pass for a plain image, seven for Adam7
    static const int px0[7] = {0, 4, 0, 2, 0, 1, 0}, py0[7] = {0, 0, 4, 0, 2, 0, 1};
    static const int pdx[7] = {8, 8, 4, 4, 2, 2, 1}, pdy[7] = {8, 8, 8, 4, 4, 2, 2};
    int npass = hd.interlace ? 7 : 1;
    size_t bpp_bits = hd.bits_per_pixel(), bpp = (bpp_bits + 7) / 8;
    size_t raw_size = 0;
    size_t pw[7], ph[7];
    for (int p = 0; p < npass; p++) {
        pw[p] = hd.interlace ? (hd.w + pdx[p] - 1 - px0[p]) / pdx[p] : hd.w;
        ph[p] = hd.interlace ? (hd.h + pdy[p] - 1 - py0[p]) / pdy[p] : hd.h;
        if (pw[p] && ph[p]) raw_size += ph[p] * (1 + (pw[p] * bpp_bits + 7) / 8);
    }
    std::vector<unsigned char> raw(raw_size);
    uLongf got = (uLongf)raw_size;
    int zr = uncompress(raw.data(), &got, idat.data(), (uLong)idat.size());
    if (zr != Z_OK || got != raw_size) { err = "inflate failed or size mismatch"; return false; }

    out.width = hd.w; out.height = hd.h;
    out.data.assign((size_t)hd.w * hd.h * 4, 0);
    size_t off = 0;
    for (int p = 0; p < npass; p++) {
        if (!pw[p] || !ph[p]) continue;
        size_t row_bytes = (pw[p] * bpp_bits + 7) / 8;
        if (!unfilter(&raw[off], ph[p], row_bytes, bpp)) { err = "bad filter type"; return false; }
        for (size_t y = 0; y < ph[p]; y++) {
            const unsigned char *row = &raw[off + y * (row_bytes + 1) + 1];
            for (size_t x = 0; x < pw[p]; x++) {
                size_t ox = hd.interlace ? px0[p] + x * pdx[p] : x, oy = hd.interlace ? py0[p] + y * pdy[p] : y;
                to_rgba(hd, row, x, plte, tr, &out.data[(oy * hd.w + ox) * 4]);
            }
        }
        off += ph[p] * (row_bytes + 1);
    }
    return true;
}

// decodeOneStep, main.cpp:764-794
PNG_Image_t *decodeOneStep(const char *filename) {
    PNG_Image_t *image = new PNG_Image_t();
    std::string err;
    if (!decode_png_rgba8(filename, *image, err)) {
        std::cout << "decoder error: " << err << "(file: '" << filename << "')" << std::endl;
        delete image;
        return nullptr;
    }
    return image;
}

}  // namespace rta
