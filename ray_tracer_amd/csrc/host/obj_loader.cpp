// obj_loader.cpp -- Wavefront OBJ subset reader (getMeshFromOBJ, main.cpp:796-913).
//
// Supported, as in the reference: `v x y z`, `vt u v`, `vn x y z`, triangles `f a b c` (optionally
// a/b/c, only the first index is used) and quads `f a/b/c ×4`.
//
// compat mode reproduces observable behaviour of the reference parser that scenes depend on:
//   * every line whose first character is 'v' -- so vt and vn lines too -- is appended to the
//     position list (main.cpp:899-906), which shifts face indices in files that interleave blocks;
//   * a vt line appended that way has no third number; the reference reads an uninitialised float
//     there, this reader uses 0;
//   * a triangle face takes its normals from the normal list at the POSITION indices
//     (main.cpp:844-847).
// Non-compat mode keeps separate lists and honours v/vt/vn indices for triangles as well.
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "scene.hpp"

namespace rta {
namespace {

bool is_ws(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n' || c == '\v' || c == '\f'; }

// tokens as `std::istream >> std::string` splits them
std::vector<std::string> tokens(const std::string &line) {
    std::vector<std::string> out;
    size_t i = 0;
    while (i < line.size()) {
        while (i < line.size() && is_ws(line[i])) i++;
        size_t j = i;
        while (j < line.size() && !is_ws(line[j])) j++;
        if (j > i) out.push_back(line.substr(i, j - i));
        i = j;
    }
    return out;
}
float number(const std::vector<std::string> &t, size_t i) { return i < t.size() ? std::strtof(t[i].c_str(), nullptr) : 0.0f; }
// k-th '/'-separated field as atoi would read it (0 when absent)
int field(const std::string &tok, int k) {
    size_t pos = 0;
    for (int i = 0; i < k; i++) {
        pos = tok.find('/', pos);
        if (pos == std::string::npos) return 0;
        pos++;
    }
    return std::atoi(tok.c_str() + pos);
}
template <class T>
const T &at(const std::vector<T> &v, int i, bool &ok) {
    static const T zero{};
    if (i < 0 || i >= (int)v.size()) { ok = false; return zero; }
    return v[i];
}

}  // namespace

Mesh *getMeshFromOBJ(const std::string &filename, bool compat) {
    std::vector<vec3> vertices, normals;
    std::vector<vec2> vts;
    std::vector<Fragment *> faces;
    std::ifstream file(filename, std::ios::binary);
    std::string line;
    while (std::getline(file, line)) {  // drops '\n', keeps a '\r'
        if (line.empty() || line[0] == '#') continue;
        bool is_vt = line.compare(0, 2, "vt") == 0, is_vn = line.compare(0, 2, "vn") == 0;
        if (line[0] == 'v' && (compat || (!is_vt && !is_vn))) {
            auto t = tokens(line);
            vertices.push_back(vec3(number(t, 1), number(t, 2), number(t, 3)));
        }
        if (is_vt) {
            auto t = tokens(line);
            vts.push_back(vec2(number(t, 1), number(t, 2)));
        }
        if (is_vn) {
            auto t = tokens(line);
            normals.push_back(normalize(vec3(number(t, 1), number(t, 2), number(t, 3))));  // get_normals, main.cpp:817
        }
        if (line[0] != 'f') continue;
        auto t = tokens(line);
        bool ok = true;
        int spaces = 0;
        for (char c : line) spaces += (c == ' ');
        Fragment *frag = nullptr;
        if (spaces == 3 && t.size() >= 4) {  // main.cpp:835-847
            int i1 = field(t[1], 0) - 1, i2 = field(t[2], 0) - 1, i3 = field(t[3], 0) - 1;
            vec3 a = at(vertices, i1, ok), b = at(vertices, i2, ok), c = at(vertices, i3, ok);
            if (!normals.empty()) {
                int n1 = i1, n2 = i2, n3 = i3;
                if (!compat && field(t[1], 2) > 0) { n1 = field(t[1], 2) - 1; n2 = field(t[2], 2) - 1; n3 = field(t[3], 2) - 1; }
                vec3 na = at(normals, n1, ok), nb = at(normals, n2, ok), nc = at(normals, n3, ok);
                if (ok) frag = new Triangle(a, b, c, na, nb, nc);
            } else if (ok) {
                frag = new Triangle(a, b, c);
            }
        } else if (t.size() >= 5) {  // main.cpp:848-884
            int v[4], vt[4], n[4];
            for (int k = 0; k < 4; k++) { v[k] = field(t[k + 1], 0) - 1; vt[k] = field(t[k + 1], 1) - 1; n[k] = field(t[k + 1], 2) - 1; }
            vec3 p[4];
            for (int k = 0; k < 4; k++) p[k] = at(vertices, v[k], ok);
            if (!normals.empty()) {
                vec3 nn[4]; vec2 uv[4];
                for (int k = 0; k < 4; k++) { nn[k] = at(normals, n[k], ok); uv[k] = at(vts, vt[k], ok); }
                if (ok) frag = new Diamond(p[0], p[1], p[2], p[3], nn[0], nn[1], nn[2], nn[3], uv[0], uv[1], uv[2], uv[3]);
            } else if (ok) {
                frag = new Diamond(p[0], p[1], p[2], p[3]);
            }
        }
        // a face with an out-of-range index is undefined behaviour in the reference; it is skipped here
        if (frag) faces.push_back(frag);
    }
    Mesh *m = new Mesh(faces);
    m->owns_fragments = true;
    return m;
}

}  // namespace rta
