// glmini.hpp -- the handful of GLM 0.9.9.7 operations the reference's scene construction uses
// (vec3, mat4, translate / rotate / scale, inverse, determinant, transpose), written out with the
// same operation order as the vendored GLM so that matrices built here round exactly like the
// reference's.  Citations are into the reference checkout's glm/ directory.
#pragma once

#include <cmath>

namespace rta {

struct vec2 {
    float x = 0, y = 0;
    vec2() = default;
    vec2(float x_, float y_) : x(x_), y(y_) {}
};

struct vec3 {
    float x = 0, y = 0, z = 0;
    vec3() = default;
    explicit vec3(float s) : x(s), y(s), z(s) {}
    vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    float &operator[](int i) { return (&x)[i]; }
    const float &operator[](int i) const { return (&x)[i]; }
};
inline vec3 operator+(vec3 a, vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline vec3 operator-(vec3 a, vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline vec3 operator-(vec3 a) { return {-a.x, -a.y, -a.z}; }
inline vec3 operator*(vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline vec3 operator*(float s, vec3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline vec3 operator/(vec3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
// detail/func_geometric.inl:48-55
inline float dot(vec3 a, vec3 b) { float tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z; return tx + ty + tz; }
// detail/func_geometric.inl:70-78
inline vec3 cross(vec3 x, vec3 y) { return {x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y}; }
// detail/func_geometric.inl:84-89 with inversesqrt = 1 / sqrt
inline vec3 normalize(vec3 v) { return v * (1.0f / std::sqrt(dot(v, v))); }

struct vec4 {
    float x = 0, y = 0, z = 0, w = 0;
    vec4() = default;
    vec4(float x_, float y_, float z_, float w_) : x(x_), y(y_), z(z_), w(w_) {}
    vec4(vec3 v, float w_) : x(v.x), y(v.y), z(v.z), w(w_) {}
    float &operator[](int i) { return (&x)[i]; }
    const float &operator[](int i) const { return (&x)[i]; }
};
inline vec4 operator+(vec4 a, vec4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
inline vec4 operator-(vec4 a, vec4 b) { return {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
inline vec4 operator*(vec4 a, vec4 b) { return {a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }
inline vec4 operator*(vec4 a, float s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }

struct mat4 {  // column-major: c[j] is column j
    vec4 c[4];
    mat4() : mat4(1.0f) {}
    explicit mat4(float d) { c[0] = {d, 0, 0, 0}; c[1] = {0, d, 0, 0}; c[2] = {0, 0, d, 0}; c[3] = {0, 0, 0, d}; }
    vec4 &operator[](int j) { return c[j]; }
    const vec4 &operator[](int j) const { return c[j]; }
};

// detail/type_mat4x4.inl:561-572
inline vec4 operator*(const mat4 &m, vec4 v) {
    vec4 add0 = m[0] * v.x + m[1] * v.y;
    vec4 add1 = m[2] * v.z + m[3] * v.w;
    return add0 + add1;
}
// detail/type_mat4x4.inl:630-648
inline mat4 operator*(const mat4 &a, const mat4 &b) {
    mat4 r;
    for (int j = 0; j < 4; j++) r[j] = a[0] * b[j].x + a[1] * b[j].y + a[2] * b[j].z + a[3] * b[j].w;
    return r;
}
// ext/matrix_transform.inl:10-15 via gtx/transform.inl:6-9
inline mat4 translate(const mat4 &m, vec3 v) {
    mat4 r = m;
    r[3] = m[0] * v.x + m[1] * v.y + m[2] * v.z + m[3];
    return r;
}
inline mat4 translate(vec3 v) { return translate(mat4(1.0f), v); }
// ext/matrix_transform.inl:18-46 via gtx/transform.inl:12-15
inline mat4 rotate(const mat4 &m, float angle, vec3 v) {
    float c = std::cos(angle), s = std::sin(angle);
    vec3 axis = normalize(v);
    vec3 temp = (1.0f - c) * axis;
    float r00 = c + temp.x * axis.x, r01 = temp.x * axis.y + s * axis.z, r02 = temp.x * axis.z - s * axis.y;
    float r10 = temp.y * axis.x - s * axis.z, r11 = c + temp.y * axis.y, r12 = temp.y * axis.z + s * axis.x;
    float r20 = temp.z * axis.x + s * axis.y, r21 = temp.z * axis.y - s * axis.x, r22 = c + temp.z * axis.z;
    mat4 r;
    r[0] = m[0] * r00 + m[1] * r01 + m[2] * r02;
    r[1] = m[0] * r10 + m[1] * r11 + m[2] * r12;
    r[2] = m[0] * r20 + m[1] * r21 + m[2] * r22;
    r[3] = m[3];
    return r;
}
inline mat4 rotate(float angle, vec3 v) { return rotate(mat4(1.0f), angle, v); }
// ext/matrix_transform.inl:78-86 via gtx/transform.inl:18-21
inline mat4 scale(const mat4 &m, vec3 v) {
    mat4 r;
    r[0] = m[0] * v.x; r[1] = m[1] * v.y; r[2] = m[2] * v.z; r[3] = m[3];
    return r;
}
inline mat4 scale(vec3 v) { return scale(mat4(1.0f), v); }

inline mat4 transpose(const mat4 &m) {
    mat4 r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) r[i][j] = m[j][i];
    return r;
}
// detail/func_matrix.inl:223-244
inline float determinant(const mat4 &m) {
    float s00 = m[2][2] * m[3][3] - m[3][2] * m[2][3];
    float s01 = m[2][1] * m[3][3] - m[3][1] * m[2][3];
    float s02 = m[2][1] * m[3][2] - m[3][1] * m[2][2];
    float s03 = m[2][0] * m[3][3] - m[3][0] * m[2][3];
    float s04 = m[2][0] * m[3][2] - m[3][0] * m[2][2];
    float s05 = m[2][0] * m[3][1] - m[3][0] * m[2][1];
    float c0 = +(m[1][1] * s00 - m[1][2] * s01 + m[1][3] * s02);
    float c1 = -(m[1][0] * s00 - m[1][2] * s03 + m[1][3] * s04);
    float c2 = +(m[1][0] * s01 - m[1][1] * s03 + m[1][3] * s05);
    float c3 = -(m[1][0] * s02 - m[1][1] * s04 + m[1][2] * s05);
    return m[0][0] * c0 + m[0][1] * c1 + m[0][2] * c2 + m[0][3] * c3;
}
// detail/func_matrix.inl:294-352
inline mat4 inverse(const mat4 &m) {
    float c00 = m[2][2] * m[3][3] - m[3][2] * m[2][3], c02 = m[1][2] * m[3][3] - m[3][2] * m[1][3];
    float c03 = m[1][2] * m[2][3] - m[2][2] * m[1][3], c04 = m[2][1] * m[3][3] - m[3][1] * m[2][3];
    float c06 = m[1][1] * m[3][3] - m[3][1] * m[1][3], c07 = m[1][1] * m[2][3] - m[2][1] * m[1][3];
    float c08 = m[2][1] * m[3][2] - m[3][1] * m[2][2], c10 = m[1][1] * m[3][2] - m[3][1] * m[1][2];
    float c11 = m[1][1] * m[2][2] - m[2][1] * m[1][2], c12 = m[2][0] * m[3][3] - m[3][0] * m[2][3];
    float c14 = m[1][0] * m[3][3] - m[3][0] * m[1][3], c15 = m[1][0] * m[2][3] - m[2][0] * m[1][3];
    float c16 = m[2][0] * m[3][2] - m[3][0] * m[2][2], c18 = m[1][0] * m[3][2] - m[3][0] * m[1][2];
    float c19 = m[1][0] * m[2][2] - m[2][0] * m[1][2], c20 = m[2][0] * m[3][1] - m[3][0] * m[2][1];
    float c22 = m[1][0] * m[3][1] - m[3][0] * m[1][1], c23 = m[1][0] * m[2][1] - m[2][0] * m[1][1];
    vec4 f0(c00, c00, c02, c03), f1(c04, c04, c06, c07), f2(c08, c08, c10, c11);
    vec4 f3(c12, c12, c14, c15), f4(c16, c16, c18, c19), f5(c20, c20, c22, c23);
    vec4 v0(m[1][0], m[0][0], m[0][0], m[0][0]), v1(m[1][1], m[0][1], m[0][1], m[0][1]);
    vec4 v2(m[1][2], m[0][2], m[0][2], m[0][2]), v3(m[1][3], m[0][3], m[0][3], m[0][3]);
    vec4 i0 = v1 * f0 - v2 * f1 + v3 * f2;
    vec4 i1 = v0 * f0 - v2 * f3 + v3 * f4;
    vec4 i2 = v0 * f1 - v1 * f3 + v3 * f5;
    vec4 i3 = v0 * f2 - v1 * f4 + v2 * f5;
    vec4 sa(+1, -1, +1, -1), sb(-1, +1, -1, +1);
    mat4 inv;
    inv[0] = i0 * sa; inv[1] = i1 * sb; inv[2] = i2 * sa; inv[3] = i3 * sb;
    vec4 row0(inv[0][0], inv[1][0], inv[2][0], inv[3][0]);
    vec4 d0 = m[0] * row0;
    float d1 = (d0.x + d0.y) + (d0.z + d0.w);
    float ood = 1.0f / d1;
    for (int j = 0; j < 4; j++) inv[j] = inv[j] * ood;
    return inv;
}

}  // namespace rta
