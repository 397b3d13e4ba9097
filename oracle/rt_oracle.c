/*
 * rt_oracle.c -- TEST INFRASTRUCTURE ONLY (see rt_oracle.h).
 *
 * CPU restatement, in plain C, of the reference renderer's hot path.  Every
 * function cites the reference lines it follows (paths relative to the
 * reference checkout: main.cpp, Textures.cpp, Image.cpp, glm/...).  The
 * arithmetic keeps the reference's evaluation order and its float/double
 * promotions so that results can be compared bit-for-bit with values recorded
 * from the reference; compile with -ffp-contract=off and without fast-math.
 *
 * Parity pinning: see the header of rt_oracle.h.
 */
#define _GNU_SOURCE
#include "rt_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_EPS 0.0001f /* main.cpp:23 */
#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------ vectors */

typedef struct { float x, y; } v2;
typedef struct { float x, y, z; } v3;
typedef struct { float x, y, z, w; } v4;
typedef struct { v4 c[4]; } m4; /* column-major, as glm::mat4 */

static inline v3 V3(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v4 V4(float x, float y, float z, float w) { v4 r = {x, y, z, w}; return r; }
static inline v3 add3(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub3(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 mul3(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 scl3(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline v3 div3(v3 a, float s) { return V3(a.x / s, a.y / s, a.z / s); }
static inline v3 neg3(v3 a) { return V3(-a.x, -a.y, -a.z); }
static inline v4 add4(v4 a, v4 b) { return V4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
static inline v4 sub4(v4 a, v4 b) { return V4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
static inline v4 mul4(v4 a, v4 b) { return V4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
static inline v4 scl4(v4 a, float s) { return V4(a.x * s, a.y * s, a.z * s, a.w * s); }

/* glm/detail/func_geometric.inl:48-55: tmp = a*b; tmp.x + tmp.y + tmp.z */
static inline float dot3(v3 a, v3 b) {
    float tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z;
    return tx + ty + tz;
}
static inline float dot2(v2 a, v2 b) {
    float tx = a.x * b.x, ty = a.y * b.y;
    return tx + ty;
}
/* func_geometric.inl:70-78 */
static inline v3 cross3(v3 x, v3 y) {
    return V3(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
/* func_geometric.inl:10-13 */
static inline float len3(v3 v) { return sqrtf(dot3(v, v)); }
static inline float len2(v2 v) { return sqrtf(dot2(v, v)); }
/* func_geometric.inl:84-89 with inversesqrt = 1/sqrt (func_exponential.inl:48) */
static inline v3 norm3(v3 v) { return scl3(v, 1.0f / sqrtf(dot3(v, v))); }
/* func_geometric.inl:19-22: length(p1 - p0) */
static inline float dist3(v3 p0, v3 p1) { return len3(sub3(p1, p0)); }
/* func_geometric.inl:106-109: I - N * dot(N, I) * 2 */
static inline v3 reflect3(v3 I, v3 N) { return sub3(I, scl3(scl3(N, dot3(N, I)), 2.0f)); }

/* glm/detail/type_mat4x4.inl:561-572 */
static inline v4 m4_mul_v4(const m4 *m, v4 v) {
    v4 mul0 = scl4(m->c[0], v.x);
    v4 mul1 = scl4(m->c[1], v.y);
    v4 add0 = add4(mul0, mul1);
    v4 mul2 = scl4(m->c[2], v.z);
    v4 mul3 = scl4(m->c[3], v.w);
    v4 add1 = add4(mul2, mul3);
    return add4(add0, add1);
}
static inline v3 xform_point(const m4 *m, v3 p) {
    v4 r = m4_mul_v4(m, V4(p.x, p.y, p.z, 1.0f));
    return V3(r.x, r.y, r.z);
}
static inline v3 xform_dir(const m4 *m, v3 p) {
    v4 r = m4_mul_v4(m, V4(p.x, p.y, p.z, 0.0f));
    return V3(r.x, r.y, r.z);
}
/* type_mat4x4.inl:630-648 */
static m4 m4_mul(const m4 *a, const m4 *b) {
    m4 r;
    for (int j = 0; j < 4; j++) {
        v4 bj = b->c[j];
        r.c[j] = add4(add4(add4(scl4(a->c[0], bj.x), scl4(a->c[1], bj.y)), scl4(a->c[2], bj.z)),
                      scl4(a->c[3], bj.w));
    }
    return r;
}
static m4 m4_identity(void) {
    m4 r;
    r.c[0] = V4(1, 0, 0, 0);
    r.c[1] = V4(0, 1, 0, 0);
    r.c[2] = V4(0, 0, 1, 0);
    r.c[3] = V4(0, 0, 0, 1);
    return r;
}
#define E(m, col, row) (((const float *)&(m)->c[col])[row])
/* glm/detail/func_matrix.inl:223-244 */
static float m4_det(const m4 *m) {
    float s00 = E(m, 2, 2) * E(m, 3, 3) - E(m, 3, 2) * E(m, 2, 3);
    float s01 = E(m, 2, 1) * E(m, 3, 3) - E(m, 3, 1) * E(m, 2, 3);
    float s02 = E(m, 2, 1) * E(m, 3, 2) - E(m, 3, 1) * E(m, 2, 2);
    float s03 = E(m, 2, 0) * E(m, 3, 3) - E(m, 3, 0) * E(m, 2, 3);
    float s04 = E(m, 2, 0) * E(m, 3, 2) - E(m, 3, 0) * E(m, 2, 2);
    float s05 = E(m, 2, 0) * E(m, 3, 1) - E(m, 3, 0) * E(m, 2, 1);
    float c0 = +(E(m, 1, 1) * s00 - E(m, 1, 2) * s01 + E(m, 1, 3) * s02);
    float c1 = -(E(m, 1, 0) * s00 - E(m, 1, 2) * s03 + E(m, 1, 3) * s04);
    float c2 = +(E(m, 1, 0) * s01 - E(m, 1, 1) * s03 + E(m, 1, 3) * s05);
    float c3 = -(E(m, 1, 0) * s02 - E(m, 1, 1) * s04 + E(m, 1, 2) * s05);
    return E(m, 0, 0) * c0 + E(m, 0, 1) * c1 + E(m, 0, 2) * c2 + E(m, 0, 3) * c3;
}
/* func_matrix.inl:294-352 */
static m4 m4_inverse(const m4 *m) {
    float c00 = E(m, 2, 2) * E(m, 3, 3) - E(m, 3, 2) * E(m, 2, 3);
    float c02 = E(m, 1, 2) * E(m, 3, 3) - E(m, 3, 2) * E(m, 1, 3);
    float c03 = E(m, 1, 2) * E(m, 2, 3) - E(m, 2, 2) * E(m, 1, 3);
    float c04 = E(m, 2, 1) * E(m, 3, 3) - E(m, 3, 1) * E(m, 2, 3);
    float c06 = E(m, 1, 1) * E(m, 3, 3) - E(m, 3, 1) * E(m, 1, 3);
    float c07 = E(m, 1, 1) * E(m, 2, 3) - E(m, 2, 1) * E(m, 1, 3);
    float c08 = E(m, 2, 1) * E(m, 3, 2) - E(m, 3, 1) * E(m, 2, 2);
    float c10 = E(m, 1, 1) * E(m, 3, 2) - E(m, 3, 1) * E(m, 1, 2);
    float c11 = E(m, 1, 1) * E(m, 2, 2) - E(m, 2, 1) * E(m, 1, 2);
    float c12 = E(m, 2, 0) * E(m, 3, 3) - E(m, 3, 0) * E(m, 2, 3);
    float c14 = E(m, 1, 0) * E(m, 3, 3) - E(m, 3, 0) * E(m, 1, 3);
    float c15 = E(m, 1, 0) * E(m, 2, 3) - E(m, 2, 0) * E(m, 1, 3);
    float c16 = E(m, 2, 0) * E(m, 3, 2) - E(m, 3, 0) * E(m, 2, 2);
    float c18 = E(m, 1, 0) * E(m, 3, 2) - E(m, 3, 0) * E(m, 1, 2);
    float c19 = E(m, 1, 0) * E(m, 2, 2) - E(m, 2, 0) * E(m, 1, 2);
    float c20 = E(m, 2, 0) * E(m, 3, 1) - E(m, 3, 0) * E(m, 2, 1);
    float c22 = E(m, 1, 0) * E(m, 3, 1) - E(m, 3, 0) * E(m, 1, 1);
    float c23 = E(m, 1, 0) * E(m, 2, 1) - E(m, 2, 0) * E(m, 1, 1);

    v4 f0 = V4(c00, c00, c02, c03), f1 = V4(c04, c04, c06, c07), f2 = V4(c08, c08, c10, c11);
    v4 f3 = V4(c12, c12, c14, c15), f4 = V4(c16, c16, c18, c19), f5 = V4(c20, c20, c22, c23);
    v4 a0 = V4(E(m, 1, 0), E(m, 0, 0), E(m, 0, 0), E(m, 0, 0));
    v4 a1 = V4(E(m, 1, 1), E(m, 0, 1), E(m, 0, 1), E(m, 0, 1));
    v4 a2 = V4(E(m, 1, 2), E(m, 0, 2), E(m, 0, 2), E(m, 0, 2));
    v4 a3 = V4(E(m, 1, 3), E(m, 0, 3), E(m, 0, 3), E(m, 0, 3));

    v4 i0 = add4(sub4(mul4(a1, f0), mul4(a2, f1)), mul4(a3, f2));
    v4 i1 = add4(sub4(mul4(a0, f0), mul4(a2, f3)), mul4(a3, f4));
    v4 i2 = add4(sub4(mul4(a0, f1), mul4(a1, f3)), mul4(a3, f5));
    v4 i3 = add4(sub4(mul4(a0, f2), mul4(a1, f4)), mul4(a2, f5));
    v4 sa = V4(+1, -1, +1, -1), sb = V4(-1, +1, -1, +1);
    m4 inv;
    inv.c[0] = mul4(i0, sa);
    inv.c[1] = mul4(i1, sb);
    inv.c[2] = mul4(i2, sa);
    inv.c[3] = mul4(i3, sb);
    v4 row0 = V4(inv.c[0].x, inv.c[1].x, inv.c[2].x, inv.c[3].x);
    v4 d0 = mul4(m->c[0], row0);
    float d1 = (d0.x + d0.y) + (d0.z + d0.w);
    float ood = 1.0f / d1;
    for (int j = 0; j < 4; j++) inv.c[j] = scl4(inv.c[j], ood);
    return inv;
}
static m4 m4_transpose(const m4 *m) {
    m4 r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) ((float *)&r.c[i])[j] = E(m, j, i);
    return r;
}
/* glm/ext/matrix_transform.inl:10-15 */
static m4 m4_translate(const m4 *m, v3 v) {
    m4 r = *m;
    r.c[3] = add4(add4(add4(scl4(m->c[0], v.x), scl4(m->c[1], v.y)), scl4(m->c[2], v.z)), m->c[3]);
    return r;
}
/* matrix_transform.inl:18-46 */
static m4 m4_rotate(const m4 *m, float angle, v3 v) {
    float c = cosf(angle), s = sinf(angle);
    v3 axis = norm3(v);
    v3 temp = scl3(axis, 1.0f - c); /* (T(1) - c) * axis */
    float r00 = c + temp.x * axis.x;
    float r01 = temp.x * axis.y + s * axis.z;
    float r02 = temp.x * axis.z - s * axis.y;
    float r10 = temp.y * axis.x - s * axis.z;
    float r11 = c + temp.y * axis.y;
    float r12 = temp.y * axis.z + s * axis.x;
    float r20 = temp.z * axis.x + s * axis.y;
    float r21 = temp.z * axis.y - s * axis.x;
    float r22 = c + temp.z * axis.z;
    m4 r;
    r.c[0] = add4(add4(scl4(m->c[0], r00), scl4(m->c[1], r01)), scl4(m->c[2], r02));
    r.c[1] = add4(add4(scl4(m->c[0], r10), scl4(m->c[1], r11)), scl4(m->c[2], r12));
    r.c[2] = add4(add4(scl4(m->c[0], r20), scl4(m->c[1], r21)), scl4(m->c[2], r22));
    r.c[3] = m->c[3];
    return r;
}
/* matrix_transform.inl:78-86 */
static m4 m4_scale(const m4 *m, v3 v) {
    m4 r;
    r.c[0] = scl4(m->c[0], v.x);
    r.c[1] = scl4(m->c[1], v.y);
    r.c[2] = scl4(m->c[2], v.z);
    r.c[3] = m->c[3];
    return r;
}

void orc_mat_identity(float *m) { m4 r = m4_identity(); memcpy(m, &r, sizeof r); }
void orc_mat_translate(const float *m, const float *v, float *out) {
    m4 a; memcpy(&a, m, sizeof a);
    m4 r = m4_translate(&a, V3(v[0], v[1], v[2]));
    memcpy(out, &r, sizeof r);
}
void orc_mat_rotate(const float *m, float angle, const float *axis, float *out) {
    m4 a; memcpy(&a, m, sizeof a);
    m4 r = m4_rotate(&a, angle, V3(axis[0], axis[1], axis[2]));
    memcpy(out, &r, sizeof r);
}
void orc_mat_scale(const float *m, const float *v, float *out) {
    m4 a; memcpy(&a, m, sizeof a);
    m4 r = m4_scale(&a, V3(v[0], v[1], v[2]));
    memcpy(out, &r, sizeof r);
}
void orc_mat_mul(const float *a, const float *b, float *out) {
    m4 x, y; memcpy(&x, a, sizeof x); memcpy(&y, b, sizeof y);
    m4 r = m4_mul(&x, &y);
    memcpy(out, &r, sizeof r);
}

/* ------------------------------------------------------------------ scene */

enum { K_SPHERE, K_PLANE, K_CONE, K_MESH, K_BOX };
enum { T_CHECKER, T_RAINBOW, T_IMAGE };

typedef struct {
    int kind;
    unsigned w, h;
    uint8_t *maps[5]; /* base, height, normal, occlusion, roughness (RGBA8) */
} texture_t;

typedef struct {
    v3 p1, p2, p3;
    v3 n1, n2, n3;
    v2 uv1, uv2, uv3;
    int frag, sub;
    orc_material mat;
} tri_t;

typedef struct {
    int ntri;
    tri_t *tri;
    int nfrag;
} mesh_t;

typedef struct object_s {
    int kind;
    m4 M, Minv, N; /* Object::setTransformation, main.cpp:85-93 */
    orc_material mat;
    /* plane (also the cone's cap) */
    v3 point, normal, b1, b2;
    /* mesh */
    mesh_t mesh;
    /* box */
    v3 v1, v2;
    tri_t box_tri[12];
    struct object_s *child;
} object_t;

struct orc_scene {
    int nobj, cap_obj;
    object_t **obj; /* the reference's `objects` list */
    int nlight, cap_light;
    v3 *light_pos, *light_col;
    v3 ambient; /* main.cpp:622 */
    int ntex, cap_tex;
    texture_t *tex;
};

typedef struct {
    int hit;
    v3 normal, point;
    float distance;
    const object_t *object;
    const tri_t *tri;   /* triangle hit inside a mesh */
    int sub;            /* cone cap */
    const orc_material *mat;
    v2 uv;
    int from_outside;
} hit_t;

void orc_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
int orc_get_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void orc_material_default(orc_material *m) {
    memset(m, 0, sizeof *m);
    m->diffuse[0] = m->diffuse[1] = m->diffuse[2] = 1.0f;
    m->texture = -1;
}

orc_scene *orc_scene_new(void) {
    orc_scene *s = (orc_scene *)calloc(1, sizeof *s);
    s->ambient = V3(0.001f, 0.001f, 0.001f);
    return s;
}

static void free_object(object_t *o) {
    if (!o) return;
    if (o->kind == K_BOX) free_object(o->child);
    free(o->mesh.tri);
    free(o);
}

void orc_scene_free(orc_scene *s) {
    if (!s) return;
    for (int i = 0; i < s->nobj; i++) free_object(s->obj[i]);
    free(s->obj);
    free(s->light_pos);
    free(s->light_col);
    for (int i = 0; i < s->ntex; i++)
        for (int k = 0; k < 5; k++) free(s->tex[i].maps[k]);
    free(s->tex);
    free(s);
}

static int push_object(orc_scene *s, object_t *o) {
    if (s->nobj == s->cap_obj) {
        s->cap_obj = s->cap_obj ? 2 * s->cap_obj : 16;
        s->obj = (object_t **)realloc(s->obj, sizeof(object_t *) * s->cap_obj);
    }
    s->obj[s->nobj] = o;
    return s->nobj++;
}

static int push_texture(orc_scene *s, texture_t t) {
    if (s->ntex == s->cap_tex) {
        s->cap_tex = s->cap_tex ? 2 * s->cap_tex : 8;
        s->tex = (texture_t *)realloc(s->tex, sizeof(texture_t) * s->cap_tex);
    }
    s->tex[s->ntex] = t;
    return s->ntex++;
}

int orc_add_checker_texture(orc_scene *s) { texture_t t; memset(&t, 0, sizeof t); t.kind = T_CHECKER; return push_texture(s, t); }
int orc_add_rainbow_texture(orc_scene *s) { texture_t t; memset(&t, 0, sizeof t); t.kind = T_RAINBOW; return push_texture(s, t); }

int orc_add_image_texture(orc_scene *s, unsigned w, unsigned h, const uint8_t *base,
                          const uint8_t *height, const uint8_t *normal, const uint8_t *occlusion,
                          const uint8_t *roughness) {
    const uint8_t *src[5] = {base, height, normal, occlusion, roughness};
    texture_t t;
    memset(&t, 0, sizeof t);
    t.kind = T_IMAGE;
    t.w = w;
    t.h = h;
    size_t n = (size_t)w * h * 4;
    for (int k = 0; k < 5; k++) {
        if (!src[k]) continue;
        t.maps[k] = (uint8_t *)malloc(n);
        memcpy(t.maps[k], src[k], n);
    }
    return push_texture(s, t);
}

/* Object::setTransformation, main.cpp:85-93 */
static void set_transformation(object_t *o, const m4 *M) {
    o->M = *M;
    o->Minv = m4_inverse(M);
    if (fabsf(m4_det(M)) == 1.0f)
        o->N = *M;
    else
        o->N = m4_transpose(&o->Minv);
}

static object_t *new_object(int kind, const orc_material *mat) {
    object_t *o = (object_t *)calloc(1, sizeof *o);
    o->kind = kind;
    if (mat) o->mat = *mat; else orc_material_default(&o->mat);
    m4 I = m4_identity();
    o->M = o->Minv = o->N = I;
    return o;
}

/* Plane::computeB1 / computeBaseVectors, main.cpp:198-210 */
static v3 compute_b1(v3 n) {
    v3 a = cross3(n, V3(1, 0, 0));
    v3 b = cross3(n, V3(0, 1, 0));
    v3 max_ab = dot3(a, a) > dot3(b, b) ? a : b;
    v3 c = cross3(n, V3(0, 0, 1));
    return norm3(dot3(max_ab, max_ab) < dot3(c, c) ? c : max_ab);
}
static void plane_init(object_t *o, v3 point, v3 normal) {
    o->point = point;
    o->normal = normal;
    o->b1 = scl3(compute_b1(normal), .05f);
    o->b2 = scl3(cross3(normal, o->b1), .1f);
}

int orc_add_sphere(orc_scene *s, const float *M, const orc_material *mat) {
    object_t *o = new_object(K_SPHERE, mat);
    m4 m; memcpy(&m, M, sizeof m);
    set_transformation(o, &m);
    return push_object(s, o);
}
int orc_add_plane(orc_scene *s, const float *p, const float *n, const orc_material *mat) {
    object_t *o = new_object(K_PLANE, mat);
    plane_init(o, V3(p[0], p[1], p[2]), V3(n[0], n[1], n[2]));
    return push_object(s, o);
}
int orc_add_cone(orc_scene *s, const float *M, const orc_material *mat) {
    object_t *o = new_object(K_CONE, mat);
    plane_init(o, V3(0, 1, 0), V3(0, 1, 0)); /* Cone::base, main.cpp:264 */
    m4 m; memcpy(&m, M, sizeof m);
    set_transformation(o, &m);
    return push_object(s, o);
}
int orc_add_light(orc_scene *s, const float *pos, const float *color) {
    if (s->nlight == s->cap_light) {
        s->cap_light = s->cap_light ? 2 * s->cap_light : 4;
        s->light_pos = (v3 *)realloc(s->light_pos, sizeof(v3) * s->cap_light);
        s->light_col = (v3 *)realloc(s->light_col, sizeof(v3) * s->cap_light);
    }
    s->light_pos[s->nlight] = V3(pos[0], pos[1], pos[2]);
    s->light_col[s->nlight] = color ? V3(color[0], color[1], color[2]) : V3(1, 1, 1);
    return s->nlight++;
}
void orc_set_ambient(orc_scene *s, const float *rgb) { s->ambient = V3(rgb[0], rgb[1], rgb[2]); }
int orc_num_objects(const orc_scene *s) { return s->nobj; }
int orc_num_lights(const orc_scene *s) { return s->nlight; }

/* Triangle 3-point constructor, main.cpp:338-340 */
static tri_t tri_flat(v3 p1, v3 p2, v3 p3) {
    tri_t t;
    memset(&t, 0, sizeof t);
    t.p1 = p1; t.p2 = p2; t.p3 = p3;
    t.n1 = t.n2 = t.n3 = norm3(cross3(sub3(p3, p1), sub3(p2, p1)));
    /* uv1..3 are never initialised by the reference for this constructor; zero is used here and
     * makes Triangle::intersect take its else-branch for uv (main.cpp:406-412). */
    orc_material_default(&t.mat);
    return t;
}

/* Box::Box, main.cpp:568-595 */
static void box_init(object_t *o, v3 v1, v3 v2, object_t *child) {
    o->v1 = v1; o->v2 = v2; o->child = child;
    v3 A = v1, B = V3(v2.x, v1.y, v1.z), C = V3(v2.x, v2.y, v1.z), D = V3(v1.x, v2.y, v1.z);
    v3 G = v2, Ee = V3(v1.x, v1.y, v2.z), F = V3(v2.x, v1.y, v2.z), H = V3(v1.x, v2.y, v2.z);
    v3 q[12][3] = {{B, A, D}, {C, B, D}, {F, B, C}, {G, F, C}, {Ee, F, G}, {H, Ee, G},
                   {A, Ee, H}, {D, A, H}, {C, D, H}, {G, C, H}, {F, Ee, A}, {B, F, A}};
    for (int i = 0; i < 12; i++) o->box_tri[i] = tri_flat(q[i][0], q[i][1], q[i][2]);
}

/* Triangle::getMin/MaxCoords + Diamond + Mesh, main.cpp:353-364, 446-461, 495-523 */
static void mesh_bounds(const mesh_t *m, v3 *bmin, v3 *bmax) {
    v3 lo = V3(INFINITY, INFINITY, INFINITY), hi = V3(-INFINITY, -INFINITY, -INFINITY);
    for (int i = 0; i < m->ntri; i++) {
        const tri_t *t = &m->tri[i];
        lo.x = fminf(lo.x, fminf(t->p1.x, fminf(t->p2.x, t->p3.x)));
        lo.y = fminf(lo.y, fminf(t->p1.y, fminf(t->p2.y, t->p3.y)));
        lo.z = fminf(lo.z, fminf(t->p1.z, fminf(t->p2.z, t->p3.z)));
        hi.x = fmaxf(hi.x, fmaxf(t->p1.x, fmaxf(t->p2.x, t->p3.x)));
        hi.y = fmaxf(hi.y, fmaxf(t->p1.y, fmaxf(t->p2.y, t->p3.y)));
        hi.z = fmaxf(hi.z, fmaxf(t->p1.z, fmaxf(t->p2.z, t->p3.z)));
    }
    *bmin = lo; *bmax = hi;
}

static int finish_mesh(orc_scene *s, object_t *mesh, const float *M, int boxed) {
    m4 m; memcpy(&m, M, sizeof m);
    set_transformation(mesh, &m);
    if (!boxed) return push_object(s, mesh);
    object_t *box = new_object(K_BOX, NULL);
    v3 lo, hi;
    mesh_bounds(&mesh->mesh, &lo, &hi);
    box_init(box, lo, hi, mesh);
    set_transformation(box, &m);
    return push_object(s, box);
}

int orc_add_mesh_tris(orc_scene *s, int ntri, const float *pos, const float *nrm,
                      const orc_material *mats, const float *M, const orc_material *mat, int boxed) {
    object_t *o = new_object(K_MESH, mat);
    o->mesh.ntri = ntri;
    o->mesh.nfrag = ntri;
    o->mesh.tri = (tri_t *)calloc((size_t)ntri, sizeof(tri_t));
    for (int i = 0; i < ntri; i++) {
        tri_t *t = &o->mesh.tri[i];
        const float *p = pos + 9 * i, *n = nrm + 9 * i;
        t->p1 = V3(p[0], p[1], p[2]); t->p2 = V3(p[3], p[4], p[5]); t->p3 = V3(p[6], p[7], p[8]);
        t->n1 = V3(n[0], n[1], n[2]); t->n2 = V3(n[3], n[4], n[5]); t->n3 = V3(n[6], n[7], n[8]);
        t->frag = i; t->sub = 0;
        t->mat = mats ? mats[i] : (mat ? *mat : o->mat);
    }
    return finish_mesh(s, o, M, boxed);
}

/* A mesh from explicit fragments: triangle i has corners pos[i], normals nrm[i], texture coordinates
 * uv[i] (NULL: zeros) and belongs to fragment fragsub[2i] as its fragsub[2i+1]-th triangle; two
 * consecutive triangles of one fragment are a Diamond (main.cpp:425-479).  This is how the tests hand
 * the oracle the faces that ANOTHER parser produced (the renderer's non-polluting OBJ mode). */
int orc_add_mesh_frags(orc_scene *s, int ntri, const float *pos, const float *nrm, const float *uv,
                       const int *fragsub, const float *M, const orc_material *mat, int boxed) {
    object_t *o = new_object(K_MESH, mat);
    o->mesh.ntri = ntri;
    o->mesh.nfrag = 0;
    o->mesh.tri = (tri_t *)calloc((size_t)(ntri > 0 ? ntri : 1), sizeof(tri_t));
    for (int i = 0; i < ntri; i++) {
        tri_t *t = &o->mesh.tri[i];
        const float *p = pos + 9 * i, *n = nrm + 9 * i;
        t->p1 = V3(p[0], p[1], p[2]); t->p2 = V3(p[3], p[4], p[5]); t->p3 = V3(p[6], p[7], p[8]);
        t->n1 = V3(n[0], n[1], n[2]); t->n2 = V3(n[3], n[4], n[5]); t->n3 = V3(n[6], n[7], n[8]);
        if (uv) { t->uv1.x = uv[6 * i]; t->uv1.y = uv[6 * i + 1]; t->uv2.x = uv[6 * i + 2]; t->uv2.y = uv[6 * i + 3];
                  t->uv3.x = uv[6 * i + 4]; t->uv3.y = uv[6 * i + 5]; }
        t->frag = fragsub[2 * i]; t->sub = fragsub[2 * i + 1];
        if (t->frag + 1 > o->mesh.nfrag) o->mesh.nfrag = t->frag + 1;
        t->mat = mat ? *mat : o->mat;
    }
    return finish_mesh(s, o, M, boxed);
}

/* ------------------------------------------------------------------ OBJ loader */

typedef struct { v3 *a; int n, cap; } v3list;
typedef struct { v2 *a; int n, cap; } v2list;
static void push3(v3list *l, v3 v) {
    if (l->n == l->cap) { l->cap = l->cap ? 2 * l->cap : 1024; l->a = (v3 *)realloc(l->a, sizeof(v3) * l->cap); }
    l->a[l->n++] = v;
}
static void push2(v2list *l, v2 v) {
    if (l->n == l->cap) { l->cap = l->cap ? 2 * l->cap : 1024; l->a = (v2 *)realloc(l->a, sizeof(v2) * l->cap); }
    l->a[l->n++] = v;
}
static int is_ws(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n' || c == '\v' || c == '\f'; }
/* next whitespace-delimited token as `istream >> std::string` would deliver it */
static const char *next_token(const char *p, char *buf, size_t cap) {
    while (*p && is_ws(*p)) p++;
    size_t n = 0;
    while (*p && !is_ws(*p)) { if (n + 1 < cap) buf[n++] = *p; p++; }
    buf[n] = 0;
    return p;
}
/* `istream >> float`: on failure (nothing left on the line) the variable keeps its old value */
static const char *next_float(const char *p, float *out) {
    while (*p && is_ws(*p)) p++;
    if (!*p) return p;
    char *end;
    float v = strtof(p, &end);
    if (end == p) { *out = 0.0f; return p; }
    *out = v;
    return end;
}
/* get_vertex, main.cpp:796-805.  A missing third component is an uninitialised read in the
 * reference (observed value ~4.6e-41, the high half of a stack address); 0 is used here. */
static v3 parse_vertex(const char *line) {
    char tok[64];
    float p1 = 0, p2 = 0, p3 = 0;
    const char *p = next_token(line, tok, sizeof tok);
    p = next_float(p, &p1);
    p = next_float(p, &p2);
    p = next_float(p, &p3);
    return V3(p1, p2, p3);
}
static v2 parse_vt(const char *line) { /* get_vt, main.cpp:807-815 */
    v3 v = parse_vertex(line);
    v2 r = {v.x, v.y};
    return r;
}
static int count_char(const char *s, char c) { int n = 0; for (; *s; s++) n += (*s == c); return n; }
/* atoi of the k-th '/'-separated field */
static int slash_field(const char *tok, int k) {
    const char *p = tok;
    for (int i = 0; i < k; i++) { p = strchr(p, '/'); if (!p) return 0; p++; }
    return atoi(p);
}

/* getMeshFromOBJ + get_face, main.cpp:830-913.  Every line starting with 'v' -- including vt and
 * vn lines -- is appended to the vertex list (main.cpp:899-906); faces index that list. */
static int load_obj(const char *path, mesh_t *out, const orc_material *mat) {
    memset(out, 0, sizeof *out);
    FILE *f = fopen(path, "rb");
    if (!f) return 0; /* std::ifstream on a missing file: zero faces */
    v3list vertices = {0}, normals = {0};
    v2list vts = {0};
    int cap = 0;
    char *line = NULL;
    size_t lcap = 0;
    ssize_t len;
    while ((len = getline(&line, &lcap, f)) >= 0) {
        if (len > 0 && line[len - 1] == '\n') line[--len] = 0; /* std::getline drops '\n', keeps '\r' */
        if (line[0] == '#') continue;
        if (line[0] == 'v') push3(&vertices, parse_vertex(line));
        if (line[0] == 'v' && line[1] == 't') push2(&vts, parse_vt(line));
        if (line[0] == 'v' && line[1] == 'n') push3(&normals, norm3(parse_vertex(line)));
        if (line[0] == 'f') {
            if (out->ntri + 2 > cap) { cap = cap ? 2 * cap : 4096; out->tri = (tri_t *)realloc(out->tri, sizeof(tri_t) * cap); }
            char tok[5][128];
            const char *p = line;
            for (int k = 0; k < 5; k++) p = next_token(p, tok[k], sizeof tok[k]);
            int frag = out->nfrag++;
            if (count_char(line, ' ') == 3) {
                int i1 = slash_field(tok[1], 0), i2 = slash_field(tok[2], 0), i3 = slash_field(tok[3], 0);
                tri_t t;
                if (normals.n > 0) {
                    memset(&t, 0, sizeof t);
                    t.p1 = vertices.a[i1 - 1]; t.p2 = vertices.a[i2 - 1]; t.p3 = vertices.a[i3 - 1];
                    t.n1 = normals.a[i1 - 1]; t.n2 = normals.a[i2 - 1]; t.n3 = normals.a[i3 - 1];
                } else {
                    t = tri_flat(vertices.a[i1 - 1], vertices.a[i2 - 1], vertices.a[i3 - 1]);
                }
                t.frag = frag; t.sub = 0; t.mat = *mat;
                out->tri[out->ntri++] = t;
            } else {
                int v[4], vt[4], n[4];
                for (int k = 0; k < 4; k++) {
                    v[k] = slash_field(tok[k + 1], 0) - 1;
                    vt[k] = slash_field(tok[k + 1], 1) - 1;
                    n[k] = slash_field(tok[k + 1], 2) - 1;
                }
                /* Diamond, main.cpp:430-444: t1 = (p1,p2,p4), t2 = (p2,p3,p4) */
                int idx[2][3] = {{0, 1, 3}, {1, 2, 3}};
                for (int q = 0; q < 2; q++) {
                    tri_t t;
                    int a = idx[q][0], b = idx[q][1], c = idx[q][2];
                    if (normals.n > 0) {
                        memset(&t, 0, sizeof t);
                        t.p1 = vertices.a[v[a]]; t.p2 = vertices.a[v[b]]; t.p3 = vertices.a[v[c]];
                        t.n1 = normals.a[n[a]]; t.n2 = normals.a[n[b]]; t.n3 = normals.a[n[c]];
                        t.uv1 = vts.a[vt[a]]; t.uv2 = vts.a[vt[b]]; t.uv3 = vts.a[vt[c]];
                    } else {
                        t = tri_flat(vertices.a[v[a]], vertices.a[v[b]], vertices.a[v[c]]);
                    }
                    t.frag = frag; t.sub = q; t.mat = *mat;
                    out->tri[out->ntri++] = t;
                }
            }
        }
    }
    free(line);
    fclose(f);
    free(vertices.a); free(normals.a); free(vts.a);
    return 1;
}

int orc_add_mesh_obj(orc_scene *s, const char *path, const float *M, const orc_material *mat,
                     int boxed, int flip_normals) {
    object_t *o = new_object(K_MESH, mat);
    load_obj(path, &o->mesh, &o->mat);
    if (flip_normals) /* Triangle::set_normals(-n1,-n2,-n3), main.cpp:366-370 */
        for (int i = 0; i < o->mesh.ntri; i++) {
            tri_t *t = &o->mesh.tri[i];
            t->n1 = neg3(t->n1); t->n2 = neg3(t->n2); t->n3 = neg3(t->n3);
        }
    return finish_mesh(s, o, M, boxed);
}

static const object_t *mesh_of(const orc_scene *s, int obj) {
    if (obj < 0 || obj >= s->nobj) return NULL;
    const object_t *o = s->obj[obj];
    if (o->kind == K_BOX) o = o->child;
    return o->kind == K_MESH ? o : NULL;
}
int orc_mesh_info(const orc_scene *s, int obj, float *bmin, float *bmax) {
    const object_t *o = mesh_of(s, obj);
    if (!o) return 0;
    v3 lo, hi;
    mesh_bounds(&o->mesh, &lo, &hi);
    if (bmin) { bmin[0] = lo.x; bmin[1] = lo.y; bmin[2] = lo.z; }
    if (bmax) { bmax[0] = hi.x; bmax[1] = hi.y; bmax[2] = hi.z; }
    return o->mesh.ntri;
}
int orc_mesh_copy(const orc_scene *s, int obj, float *pos, float *nrm, float *uv, int *fragsub) {
    const object_t *o = mesh_of(s, obj);
    if (!o) return 0;
    for (int i = 0; i < o->mesh.ntri; i++) {
        const tri_t *t = &o->mesh.tri[i];
        if (pos) memcpy(pos + 9 * i, &t->p1, 9 * sizeof(float));
        if (nrm) memcpy(nrm + 9 * i, &t->n1, 9 * sizeof(float));
        if (uv) memcpy(uv + 6 * i, &t->uv1, 6 * sizeof(float));
        if (fragsub) { fragsub[2 * i] = t->frag; fragsub[2 * i + 1] = t->sub; }
    }
    return o->mesh.ntri;
}
void orc_plane_basis(const orc_scene *s, int obj, float *b1, float *b2) {
    const object_t *o = s->obj[obj];
    b1[0] = o->b1.x; b1[1] = o->b1.y; b1[2] = o->b1.z;
    b2[0] = o->b2.x; b2[1] = o->b2.y; b2[2] = o->b2.z;
}

/* ------------------------------------------------------------------ textures */

/* ImageTexture::getRGBAat, Textures.cpp:27-35.  `int % unsigned` converts the int to unsigned
 * first, so negative coordinates wrap modulo 2^32 and then modulo the image size. */
static v4 get_rgba_at(const texture_t *t, int map, v2 uv) {
    const uint8_t *data = t->maps[map];
    /* TEXTURE_REPETITION * uv.x * width: int*float -> float, float*unsigned -> float */
    float fx = 2 * uv.x * (float)t->w;
    float fy = 2 * uv.y * (float)t->h;
    int xi = (int)roundf(fx), yi = (int)roundf(fy);
    int x = (int)((unsigned)xi % t->w);
    int y = (int)((unsigned)yi % t->h);
    /* (y * image.width + x) * 4 is evaluated in unsigned arithmetic and stored to int */
    int index = (int)(((unsigned)y * t->w + (unsigned)x) * 4u);
    return V4((float)data[index + 0] / 255.f, (float)data[index + 1] / 255.f,
              (float)data[index + 2] / 255.f, (float)data[index + 3] / 255.f);
}
/* Texture::texture(uv): Textures.cpp:9-24, 41-43 */
static v3 tex_color(const texture_t *t, v2 uv) {
    switch (t->kind) {
    case T_CHECKER: {
        int v = (int)(floorf(16 * uv.x) + floorf(16 * uv.y)) % 2;
        return V3((float)v, (float)v, (float)v);
    }
    case T_RAINBOW:
        switch ((int)floorf(18 * (uv.x + uv.y)) % 3) {
        case 0: return V3(1, 0, 0);
        case 1: return V3(0, 1, 0);
        case 2: return V3(0, 0, 1);
        default: return V3(0, 0, 0);
        }
    default: {
        v4 c = get_rgba_at(t, 0, uv);
        return V3(powf(c.x, 2.2f), powf(c.y, 2.2f), powf(c.z, 2.2f));
    }
    }
}
static v3 tex_normal(const texture_t *t, v2 uv) { v4 c = get_rgba_at(t, 2, uv); return V3(c.x, c.y, c.z); }
static v3 tex_occlusion(const texture_t *t, v2 uv) { v4 c = get_rgba_at(t, 3, uv); return V3(c.x, c.y, c.z); }
static float tex_roughness(const texture_t *t, v2 uv) { return get_rgba_at(t, 4, uv).x; }

static const texture_t *image_texture_of(const orc_scene *s, const orc_material *m) {
    if (m->texture < 0) return NULL;
    const texture_t *t = &s->tex[m->texture];
    return t->kind == T_IMAGE ? t : NULL;
}

void orc_sample_texture(const orc_scene *s, int tex, int which, int n, const float *uv, float *out3) {
    const texture_t *t = &s->tex[tex];
    for (int i = 0; i < n; i++) {
        v2 q = {uv[2 * i], uv[2 * i + 1]};
        v3 r;
        if (which == 0) r = tex_color(t, q);
        else if (which == 1) r = tex_normal(t, q);
        else if (which == 2) r = tex_occlusion(t, q);
        else { float f = tex_roughness(t, q); r = V3(f, f, f); }
        out3[3 * i] = r.x; out3[3 * i + 1] = r.y; out3[3 * i + 2] = r.z;
    }
}

/* ------------------------------------------------------------------ intersections */

static hit_t miss(void) {
    hit_t h;
    memset(&h, 0, sizeof h);
    h.from_outside = 1; /* the only member with a default, main.cpp:52 */
    return h;
}

/* mat3(c0,c1,c2) * v, glm/detail/type_mat3x3.inl:468-474 */
static inline v3 m3_mul(v3 c0, v3 c1, v3 c2, v3 v) {
    return V3(c0.x * v.x + c1.x * v.y + c2.x * v.z, c0.y * v.x + c1.y * v.y + c2.y * v.z,
              c0.z * v.x + c1.z * v.y + c2.z * v.z);
}

/* Plane::intersect, main.cpp:222-257 */
static hit_t plane_intersect(const orc_scene *s, const object_t *obj, v3 point, v3 normal, v3 b1,
                             v3 b2, const orc_material *mat, v3 ro, v3 rd) {
    hit_t h = miss();
    float DdotN = dot3(rd, normal);
    if (DdotN < 0) {
        float PdotN = dot3(sub3(point, ro), normal);
        float t = PdotN / DdotN;
        if (t > 0) {
            h.hit = 1;
            h.normal = normal;
            h.distance = t;
            h.object = obj;
            h.mat = mat;
            h.point = add3(scl3(rd, t), ro);
            h.uv.x = dot3(b1, h.point);
            h.uv.y = dot3(b2, h.point);
            const texture_t *it = s ? image_texture_of(s, mat) : NULL;
            if (it) {
                v3 tn = tex_normal(it, h.uv);
                tn = sub3(scl3(tn, 2.f), V3(1, 1, 1));
                h.normal = m3_mul(b1, b2, normal, tn);
            }
            h.normal = norm3(h.normal);
        }
    }
    return h;
}

/* Sphere::intersect, main.cpp:122-189.  pow(x, 2) on floats is double arithmetic. */
static hit_t sphere_intersect(const orc_scene *s, const object_t *obj, v3 ro, v3 rd) {
    hit_t h = miss();
    v3 o = xform_point(&obj->Minv, ro);
    v3 d = xform_dir(&obj->Minv, rd);
    float a = (float)((double)d.x * d.x + (double)d.y * d.y + (double)d.z * d.z);
    float b = 2 * (d.x * o.x + d.y * o.y + d.z * o.z);
    float c = (float)((double)o.x * o.x + (double)o.y * o.y + (double)o.z * o.z - 1);
    float delta = (float)((double)b * b - (double)(4 * a * c));
    if (delta < 0) return h;
    float t1 = (-b + sqrtf(delta)) / (2 * a);
    float t2 = (-b - sqrtf(delta)) / (2 * a);
    if (t1 < 0 && t2 < 0) return h;
    float ta = t1 < 0 ? INFINITY : t1, tb = t2 < 0 ? INFINITY : t2;
    float t = tb < ta ? tb : ta; /* std::min(ta, tb) */
    v3 i = add3(o, scl3(d, t));
    h.point = i;
    h.normal = i;
    h.uv.x = (float)(((double)atan2f(i.z, i.x) + M_PI) / M_PI);
    h.uv.y = (float)(((double)asinf(i.y) + M_PI / 2) / M_PI);
    int from_inside = fabsf(o.x) < 1 && fabsf(o.y) < 1 && fabsf(o.z) < 1;
    h.from_outside = !from_inside;
    h.point = xform_point(&obj->M, h.point);
    const texture_t *it = image_texture_of(s, &obj->mat);
    if (it) {
        v3 tangent = V3(sinf(h.uv.x), 0, cosf(h.uv.x));
        v3 bitangent = cross3(h.normal, tangent);
        v3 tn = tex_normal(it, h.uv);
        tn = sub3(scl3(tn, 2.f), V3(1, 1, 1));
        tn = m3_mul(norm3(tangent), norm3(bitangent), norm3(h.normal), tn);
        h.normal = xform_dir(&obj->N, tn);
    } else {
        h.normal = xform_dir(&obj->N, h.normal);
    }
    h.normal = norm3(h.normal);
    h.distance = dist3(ro, h.point);
    h.object = obj;
    h.mat = &obj->mat;
    h.hit = 1;
    return h;
}

/* Cone::intersect, main.cpp:269-325 */
static hit_t cone_intersect(const orc_scene *s, const object_t *obj, v3 ro, v3 rd) {
    (void)s;
    hit_t h = miss();
    v3 d = xform_dir(&obj->Minv, rd);
    v3 o = xform_point(&obj->Minv, ro);
    float a = (float)((double)d.x * d.x - (double)d.y * d.y + (double)d.z * d.z);
    float b = 2 * (d.x * o.x - d.y * o.y + d.z * o.z);
    float c = (float)((double)o.x * o.x - (double)o.y * o.y + (double)o.z * o.z);
    float delta = (float)((double)b * b - (double)(4 * a * c));
    if (delta < 0) return h;
    float t1 = (-b + sqrtf(delta)) / (2 * a);
    float t2 = (-b - sqrtf(delta)) / (2 * a);
    if (t1 < 0 && t2 < 0) return h;
    v3 i1 = add3(o, scl3(d, t1)), i2 = add3(o, scl3(d, t2));
    if ((i1.y < 0 && i2.y < 0) || (i1.y > 1 && i2.y > 1)) return h;
    float ta = t1 < 0 ? INFINITY : t1, tb = t2 < 0 ? INFINITY : t2;
    float t = tb < ta ? tb : ta;
    v3 i = add3(o, scl3(d, t));
    if (i.y > 1) {
        /* cap: the inner Plane is hit with the LOCAL ray; the image-texture branch of the plane
         * is reachable in principle because the cap shares the cone's material */
        h = plane_intersect(s, obj, obj->point, obj->normal, obj->b1, obj->b2, &obj->mat, o, d);
        h.sub = 1;
    } else {
        h.hit = 1;
        h.object = obj;
        h.mat = &obj->mat;
        h.point = i;
        float y = (float)(((double)i.x * i.x + (double)i.z * i.z) / i.y + i.y);
        h.normal = sub3(i, V3(0, y, 0));
        h.uv.x = atan2f(i.x, i.z); /* glm::atan(y, x) */
        h.uv.y = i.y;
    }
    h.point = xform_point(&obj->M, h.point);
    h.normal = xform_dir(&obj->N, h.normal);
    h.normal = norm3(h.normal);
    h.distance = dist3(ro, h.point);
    return h;
}

/* glm::sign on a float: (0 < x) - (x < 0); NaN gives 0 (glm/detail/func_common.inl:144-149) */
static inline float sign_f(float x) { return (float)(0.0f < x) - (float)(x < 0.0f); }

/* Triangle::intersect, main.cpp:372-417 */
static hit_t triangle_intersect(const tri_t *tr, v3 ro, v3 rd) {
    hit_t h = miss();
    h.distance = INFINITY;
    v3 sn = cross3(sub3(tr->p2, tr->p1), sub3(tr->p3, tr->p1));
    /* temporary Plane(p1, sn).intersect(ray): main.cpp:226-237 */
    float DdotN = dot3(rd, sn);
    if (!(DdotN < 0)) return h;
    float PdotN = dot3(sub3(tr->p1, ro), sn);
    float t = PdotN / DdotN;
    if (!(t > 0)) return h;
    v3 p = add3(scl3(rd, t), ro);
    v3 sn1 = cross3(sub3(tr->p2, p), sub3(tr->p3, p));
    v3 sn2 = cross3(sub3(tr->p3, p), sub3(tr->p1, p));
    v3 sn3 = cross3(sub3(tr->p1, p), sub3(tr->p2, p));
    float s1 = sign_f(dot3(sn1, sn)), s2 = sign_f(dot3(sn2, sn)), s3 = sign_f(dot3(sn3, sn));
    if (s1 < 0 || s2 < 0 || s3 < 0) return h;
    h.hit = 1;
    h.point = p;
    /* interpolate(): length(sn1)*n1 + length(sn2)*n2 + length(sn3)*n3, main.cpp:419-422 */
    float l1 = len3(sn1), l2 = len3(sn2), l3 = len3(sn3);
    h.normal = norm3(add3(add3(scl3(tr->n1, l1), scl3(tr->n2, l2)), scl3(tr->n3, l3)));
    h.distance = t;
    h.tri = tr;
    h.mat = &tr->mat;
    float W = len3(sn);
    if (len2(tr->uv1) > 0) {
        float w1 = len3(div3(sn1, W)), w2 = len3(div3(sn2, W)), w3 = len3(div3(sn3, W));
        v3 u1 = V3(tr->uv1.x, tr->uv1.y, 0), u2 = V3(tr->uv2.x, tr->uv2.y, 0), u3 = V3(tr->uv3.x, tr->uv3.y, 0);
        v3 r = add3(add3(scl3(u1, w1), scl3(u2, w2)), scl3(u3, w3));
        h.uv.x = r.x; h.uv.y = r.y;
    } else {
        h.uv.x = l1 / W; h.uv.y = l2 / W;
    }
    h.from_outside = dot3(h.normal, rd) < 0;
    return h;
}

/* Mesh::intersect incl. Diamond::intersect, main.cpp:468-478, 525-547 */
static hit_t mesh_intersect(const object_t *obj, v3 ro, v3 rd) {
    hit_t h = miss();
    h.distance = INFINITY;
    v3 d = xform_dir(&obj->Minv, rd);
    v3 o = xform_point(&obj->Minv, ro);
    const mesh_t *m = &obj->mesh;
    for (int i = 0; i < m->ntri;) {
        hit_t fh;
        if (i + 1 < m->ntri && m->tri[i + 1].frag == m->tri[i].frag) {
            hit_t h1 = triangle_intersect(&m->tri[i], o, d);
            hit_t h2 = triangle_intersect(&m->tri[i + 1], o, d);
            if (!(h1.hit || h2.hit)) fh = miss();
            else fh = (h1.distance < h2.distance) ? h1 : h2;
            i += 2;
        } else {
            fh = triangle_intersect(&m->tri[i], o, d);
            i += 1;
        }
        if (fh.hit && fh.distance < h.distance) h = fh;
    }
    /* applied to hits and misses alike in the reference; only hits are observable */
    h.point = xform_point(&obj->M, h.point);
    h.normal = xform_dir(&obj->N, h.normal);
    h.normal = norm3(h.normal);
    h.distance = dist3(ro, h.point);
    return h;
}

static hit_t object_intersect(const orc_scene *s, const object_t *obj, v3 ro, v3 rd);

/* Box::intersect, main.cpp:599-618 */
static hit_t box_intersect(const orc_scene *s, const object_t *obj, v3 ro, v3 rd) {
    v3 d = xform_dir(&obj->Minv, rd);
    v3 o = xform_point(&obj->Minv, ro);
    int any = 0;
    for (int i = 0; i < 12; i++) {
        hit_t th = triangle_intersect(&obj->box_tri[i], o, d);
        if (th.hit) any = 1;
    }
    if (!any) { hit_t h = miss(); h.distance = INFINITY; return h; }
    return object_intersect(s, obj->child, ro, rd);
}

static hit_t object_intersect(const orc_scene *s, const object_t *obj, v3 ro, v3 rd) {
    switch (obj->kind) {
    case K_SPHERE: return sphere_intersect(s, obj, ro, rd);
    case K_PLANE: return plane_intersect(s, obj, obj->point, obj->normal, obj->b1, obj->b2, &obj->mat, ro, rd);
    case K_CONE: return cone_intersect(s, obj, ro, rd);
    case K_MESH: return mesh_intersect(obj, ro, rd);
    default: return box_intersect(s, obj, ro, rd);
    }
}

/* find_closest_hit, main.cpp:697-709 */
static hit_t find_closest_hit(const orc_scene *s, v3 ro, v3 rd, int *obj_index) {
    hit_t best = miss();
    best.distance = INFINITY;
    int bi = -1;
    for (int k = 0; k < s->nobj; k++) {
        hit_t h = object_intersect(s, s->obj[k], ro, rd);
        if (h.hit && h.from_outside && h.distance > 0.01f && h.distance < best.distance) {
            best = h;
            bi = k;
        }
    }
    if (obj_index) *obj_index = bi;
    return best;
}

static void export_hit(const hit_t *h, int obj_index, orc_hit *out) {
    memset(out, 0, sizeof *out);
    out->hit = h->hit;
    out->object = h->hit ? obj_index : -1;
    out->frag = (h->hit && h->tri) ? h->tri->frag : 0;
    out->sub = (h->hit && h->tri) ? h->tri->sub : (h->hit ? h->sub : 0);
    out->from_outside = h->from_outside;
    out->distance = h->distance;
    if (h->hit) {
        out->normal[0] = h->normal.x; out->normal[1] = h->normal.y; out->normal[2] = h->normal.z;
        out->point[0] = h->point.x; out->point[1] = h->point.y; out->point[2] = h->point.z;
        out->uv[0] = h->uv.x; out->uv[1] = h->uv.y;
    }
}

void orc_intersect_object(const orc_scene *s, int obj, int n, const float *o, const float *d, orc_hit *out) {
    for (int i = 0; i < n; i++) {
        hit_t h = object_intersect(s, s->obj[obj], V3(o[3 * i], o[3 * i + 1], o[3 * i + 2]),
                                   V3(d[3 * i], d[3 * i + 1], d[3 * i + 2]));
        export_hit(&h, obj, &out[i]);
    }
}

void orc_find_closest_hit(const orc_scene *s, int n, const float *o, const float *d, orc_hit *out) {
#pragma omp parallel for schedule(dynamic, 64)
    for (int i = 0; i < n; i++) {
        int bi;
        hit_t h = find_closest_hit(s, V3(o[3 * i], o[3 * i + 1], o[3 * i + 2]),
                                   V3(d[3 * i], d[3 * i + 1], d[3 * i + 2]), &bi);
        export_hit(&h, bi, &out[i]);
    }
}

/* ------------------------------------------------------------------ shading */

/* refract, main.cpp:627-633 (not glm::refract; no total-internal-reflection branch) */
static v3 refract_dir(v3 i, v3 n, float index) {
    v3 a = scl3(n, dot3(i, n));
    v3 b = sub3(i, a);
    float beta = 1 / index;
    float alpha = (float)sqrt(1 + (1 - (double)beta * beta) * (double)(dot3(b, b) / dot3(a, a)));
    return add3(scl3(a, alpha), scl3(b, beta));
}

void orc_refract(const float *i, const float *n, float index, float *out) {
    v3 r = refract_dir(V3(i[0], i[1], i[2]), V3(n[0], n[1], n[2]), index);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

static inline float clamp01(float x) { /* glm::clamp = min(max(x, 0), 1) */
    float m = (x < 0.0f) ? 0.0f : x;
    return (1.0f < m) ? 1.0f : m;
}

/* PhongModel, main.cpp:642-695 */
static v3 phong(const orc_scene *s, v3 point, v3 normal, v2 uv, v3 view, const orc_material *m,
                uint64_t *rays) {
    v3 color = mul3(s->ambient, V3(m->ambient[0], m->ambient[1], m->ambient[2]));
    const texture_t *tex = m->texture >= 0 ? &s->tex[m->texture] : NULL;
    const texture_t *img = image_texture_of(s, m);
    for (int l = 0; l < s->nlight; l++) {
        v3 lp = s->light_pos[l];
        v3 L = norm3(sub3(lp, point));
        (*rays)++;
        hit_t sh = find_closest_hit(s, add3(point, scl3(L, ORC_EPS)), L, NULL);
        if (sh.distance < dist3(point, lp)) continue;
        v3 R = reflect3(neg3(L), normal);
        float NdotL = clamp01(dot3(normal, L));
        float VdotR = clamp01(dot3(view, R));
        v3 dc = V3(m->diffuse[0], m->diffuse[1], m->diffuse[2]);
        if (tex) dc = tex_color(tex, uv);
        v3 diffuse = mul3(dc, V3(NdotL, NdotL, NdotL));
        float shininess = m->shininess;
        if (img) {
            diffuse = scl3(diffuse, 5.f);
            shininess = (float)(.5 / pow((double)tex_roughness(img, uv), 4.0) - .5);
        }
        float sp = powf(VdotR, shininess);
        v3 specular = mul3(V3(m->specular[0], m->specular[1], m->specular[2]), V3(sp, sp, sp));
        float r = dist3(point, lp);
        r = (r < 0.1f) ? 0.1f : r; /* std::max(r, 0.1f) */
        v3 term = mul3(s->light_col[l], add3(diffuse, specular));
        term = scl3(term, 1.3f);
        term = div3(term, r);
        term = div3(term, r);
        color = add3(color, term);
    }
    if (img) color = mul3(color, tex_occlusion(img, uv));
    else color = scl3(color, 4.0f);
    color = scl3(color, 2.0f);
    return color;
}

/* trace_ray, main.cpp:716-761.  depth counts from 0 for primary rays; rays deeper than max_depth
 * are not traced (build-defined cap; max_depth < 0 = unbounded as the reference). */
static v3 trace_ray(const orc_scene *s, v3 ro, v3 rd, int depth, int max_depth, uint64_t *rays) {
    v3 color = V3(0, 0, 0);
    if (max_depth >= 0 && depth > max_depth) return color;
    (*rays)++;
    hit_t h = find_closest_hit(s, ro, rd, NULL);
    if (!h.hit) return color;
    const orc_material *m = h.mat;
    color = phong(s, h.point, h.normal, h.uv, norm3(neg3(rd)), m, rays);
    v3 view = neg3(rd);
    v3 normal = h.normal, point = h.point;
    if (m->refractiveness > 0) {
        float index = h.from_outside ? m->refractiveness : 1 / m->refractiveness;
        v3 rdir = refract_dir(neg3(view), normal, index);
        v3 rc = trace_ray(s, add3(point, scl3(rdir, ORC_EPS)), rdir, depth + 1, max_depth, rays);
        color = add3(scl3(color, m->refractive_ratio), scl3(rc, 1.f - m->refractive_ratio));
    }
    if (m->reflectiveness > 0) {
        v3 rc = V3(0, 0, 0);
        v3 rdir = reflect3(neg3(view), normal);
        if (h.from_outside)
            rc = trace_ray(s, add3(point, scl3(rdir, ORC_EPS)), rdir, depth + 1, max_depth, rays);
        color = add3(scl3(color, 1 - m->reflectiveness), scl3(rc, m->reflectiveness));
    }
    return color;
}

void orc_trace_rays(const orc_scene *s, int n, const float *o, const float *d, int max_depth,
                    float *radiance, uint64_t *rays_out) {
#pragma omp parallel for schedule(dynamic, 16)
    for (int i = 0; i < n; i++) {
        uint64_t rays = 0;
        v3 c = trace_ray(s, V3(o[3 * i], o[3 * i + 1], o[3 * i + 2]),
                         V3(d[3 * i], d[3 * i + 1], d[3 * i + 2]), 0, max_depth, &rays);
        radiance[3 * i] = c.x; radiance[3 * i + 1] = c.y; radiance[3 * i + 2] = c.z;
        if (rays_out) rays_out[i] = rays;
    }
}

/* ------------------------------------------------------------------ camera / image */

/* toneMapping, main.cpp:1184-1189 */
static v3 tone_map(v3 c) {
    float gamma = (float)(1.0 / 1.8);
    return V3(clamp01(18.0f * powf(c.x, gamma)), clamp01(18.0f * powf(c.y, gamma)),
              clamp01(18.0f * powf(c.z, gamma)));
}
/* Image::setPixel(vec3), Image.cpp:35-39: int = (float)(255 * c), truncation */
static inline uint8_t quantise(float c) { return (uint8_t)(int)(float)(255 * c); }

void orc_tonemap_quantise(int n, const float *rgb, uint8_t *rgb8) {
    for (int i = 0; i < n; i++) {
        v3 t = tone_map(V3(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]));
        rgb8[3 * i] = quantise(t.x); rgb8[3 * i + 1] = quantise(t.y); rgb8[3 * i + 2] = quantise(t.z);
    }
}

typedef struct { float s, X, Y; } camera_t;
/* main.cpp:1231-1233 */
static camera_t make_camera(int w, int h, float fov) {
    camera_t c;
    c.s = (float)(2 * tan(0.5 * (double)fov / 180 * M_PI) / w);
    c.X = -c.s * w / 2;
    c.Y = c.s * h / 2;
    return c;
}
static int taps_of(int mode, int m) { return mode == ORC_MODE_REF_1SPP ? 1 : mode == ORC_MODE_REF_AA9 ? 9 : m * m; }
/* primary ray direction of tap k of pixel (i, j); main.cpp:1241-1266.  GRID mode is defined by
 * this build: corner as in the AA path, taps at ((a+.5)/m*s, -(b+.5)/m*s), a = k%m, b = k/m. */
static v3 primary_dir(const camera_t *c, int mode, int m, int i, int j, int k, float *weight) {
    float s = c->s;
    if (mode == ORC_MODE_REF_1SPP) {
        float dx = c->X + i * s + s / 2;
        float dy = c->Y - j * s + s / 2;
        *weight = 1.0f;
        return V3(dx, dy, 1);
    }
    float dx = c->X + i * s, dy = c->Y - j * s;
    if (mode == ORC_MODE_REF_AA9) {
        static const float wts[9] = {1, 2, 1, 2, 4, 2, 1, 2, 1};
        float tx = dx + (k % 3) * s / 2;
        float ty = dy - ((int)(k / 3)) * s / 2;
        *weight = wts[k];
        return norm3(V3(tx, ty, 1));
    }
    int a = k % m, b = k / m;
    float tx = dx + ((float)a + 0.5f) / (float)m * s;
    float ty = dy - ((float)b + 0.5f) / (float)m * s;
    *weight = 1.0f;
    return norm3(V3(tx, ty, 1));
}

int orc_camera_rays(int w, int h, float fov, int mode, int grid_m, int x0, int y0, int x1, int y1,
                    float *o, float *d, float *weights) {
    camera_t cam = make_camera(w, h, fov);
    int taps = taps_of(mode, grid_m);
    size_t r = 0;
    for (int j = y0; j < y1; j++)
        for (int i = x0; i < x1; i++)
            for (int k = 0; k < taps; k++, r++) {
                float wt;
                v3 dir = primary_dir(&cam, mode, grid_m, i, j, k, &wt);
                o[3 * r] = o[3 * r + 1] = o[3 * r + 2] = 0;
                d[3 * r] = dir.x; d[3 * r + 1] = dir.y; d[3 * r + 2] = dir.z;
                if (weights && j == y0 && i == x0) weights[k] = wt;
            }
    return taps;
}

uint64_t orc_render(const orc_scene *s, int w, int h, float fov, int mode, int grid_m, int max_depth,
                    int x0, int y0, int x1, int y1, float *rgb, uint8_t *rgb8, int nthreads) {
    camera_t cam = make_camera(w, h, fov);
    int taps = taps_of(mode, grid_m);
    uint64_t total = 0;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : total)
    for (int i = x0; i < x1; i++) { /* the reference walks columns in the outer loop */
        for (int j = y0; j < y1; j++) {
            uint64_t rays = 0;
            v3 color = V3(0, 0, 0);
            if (mode == ORC_MODE_REF_1SPP) {
                float wt;
                v3 dir = primary_dir(&cam, mode, grid_m, i, j, 0, &wt);
                color = trace_ray(s, V3(0, 0, 0), dir, 0, max_depth, &rays);
            } else {
                for (int k = 0; k < taps; k++) {
                    float wt;
                    v3 dir = primary_dir(&cam, mode, grid_m, i, j, k, &wt);
                    v3 c = trace_ray(s, V3(0, 0, 0), dir, 0, max_depth, &rays);
                    color = add3(color, scl3(c, wt));
                }
                color = div3(color, mode == ORC_MODE_REF_AA9 ? 16.f : (float)taps);
            }
            size_t p = 3 * ((size_t)j * w + i);
            if (rgb) { rgb[p] = color.x; rgb[p + 1] = color.y; rgb[p + 2] = color.z; }
            if (rgb8) {
                v3 t = tone_map(color);
                rgb8[p] = quantise(t.x); rgb8[p + 1] = quantise(t.y); rgb8[p + 2] = quantise(t.z);
            }
            total += rays;
        }
    }
    return total;
}

/* Image::writeImage, Image.cpp:8-23 */
int orc_write_ppm(const char *path, int w, int h, const uint8_t *rgb8) {
    FILE *f = fopen(path, "wb");
    if (!f) return -1;
    fprintf(f, "P3\n%d %d\n255\n", w, h);
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            const uint8_t *p = rgb8 + 3 * ((size_t)y * w + x);
            fprintf(f, "%d %d %d  ", p[0], p[1], p[2]);
        }
        fputc('\n', f);
    }
    fclose(f);
    return 0;
}

/* ------------------------------------------------------------------ presets */

static orc_material mk(float ar, float ag, float ab, float dr, float dg, float db, float sp, float sh) {
    orc_material m;
    orc_material_default(&m);
    m.ambient[0] = ar; m.ambient[1] = ag; m.ambient[2] = ab;
    m.diffuse[0] = dr; m.diffuse[1] = dg; m.diffuse[2] = db;
    m.specular[0] = m.specular[1] = m.specular[2] = sp;
    m.shininess = sh;
    return m;
}
static int add_texset(orc_scene *s, const orc_texset *t) {
    return orc_add_image_texture(s, t->w, t->h, t->maps[0], t->maps[1], t->maps[2], t->maps[3], t->maps[4]);
}
static void add_plane3(orc_scene *s, float px, float py, float pz, v3 n, const orc_material *m) {
    float p[3] = {px, py, pz}, nn[3] = {n.x, n.y, n.z};
    orc_add_plane(s, p, nn, m);
}
static void add_light3(orc_scene *s, float x, float y, float z, float c) {
    float p[3] = {x, y, z}, col[3] = {c, c, c};
    orc_add_light(s, p, col);
}

int orc_build_preset(orc_scene *s, const char *name, const char *asset_dir, const orc_texset *tex) {
    /* materials, main.cpp:920-941, 1057-1059, 1138-1144 */
    orc_material green_diffuse = mk(0.03f, 0.1f, 0.03f, 0.3f, 1.0f, 0.3f, 0, 0);
    orc_material red_specular = mk(0.01f, 0.02f, 0.02f, 1.0f, 0.2f, 0.2f, 0.5f, 10.0f);
    orc_material blue_specular = mk(0.02f, 0.02f, 0.1f, 0.2f, 0.2f, 1.0f, 0.6f, 100.0f);
    orc_material yellow = mk(0.1f, 0.1f, 0.03f, .6f, .6f, 0.1f, .6f, 100);
    orc_material refractive; orc_material_default(&refractive);
    refractive.reflectiveness = 0.1f; refractive.refractiveness = 2.0f;
    orc_material red_diffuse = mk(0.09f, 0.06f, 0.06f, 0.9f, 0.6f, 0.6f, 0, 0);
    orc_material blue_diffuse = mk(0.06f, 0.06f, 0.09f, 0.6f, 0.6f, 0.9f, 0, 0);
    orc_material plain; orc_material_default(&plain);
    m4 I = m4_identity();
    char path[1024];

    if (!strcmp(name, "analytic")) {
        /* spheres main.cpp:1061-1072, cones 1157-1171, planes 1150-1155 (+ a floor), lights 1173-1176 */
        m4 t = m4_translate(&I, V3(1, -2, 8));
        orc_add_sphere(s, (float *)&t, &blue_specular);
        t = m4_translate(&I, V3(-1, -2.5, 6)); t = m4_scale(&t, V3(.5f, .5f, .5f));
        orc_add_sphere(s, (float *)&t, &red_specular);
        m4 a = m4_translate(&I, V3(-3, -1, 8)), b = m4_scale(&I, V3(2, 2, 2));
        t = m4_mul(&a, &b);
        orc_add_sphere(s, (float *)&t, &refractive);
        a = m4_translate(&I, V3(6, -3, 7)); b = m4_rotate(&I, (float)atan(3.0), V3(0, 0, 1));
        t = m4_mul(&a, &b); t = m4_scale(&t, V3(1, 3, 1));
        orc_add_cone(s, (float *)&t, &green_diffuse);
        a = m4_translate(&I, V3(-5, 9, 14)); b = m4_rotate(&I, 3.1415f, V3(0, 0, 1));
        t = m4_mul(&a, &b); t = m4_scale(&t, V3(3, 12, 3));
        orc_add_cone(s, (float *)&t, &yellow);
        add_plane3(s, 0, -3, 0, V3(0, 1, 0), &plain);
        add_plane3(s, 0, 1, 30, V3(0, 0, -1), &green_diffuse);
        add_plane3(s, -15, 1, 0, V3(1, 0, 0), &red_diffuse);
        add_plane3(s, 15, 1, 0, V3(-1, 0, 0), &blue_diffuse);
        add_plane3(s, 0, 27, 0, V3(0, -1, 0), &plain);
        add_plane3(s, 0, 1, -0.01f, V3(0, 0, 1), &green_diffuse);
        add_light3(s, 10, 16, -7, .5f);
        add_light3(s, 2, 1, -15, .5f);
        add_light3(s, 6, 1, 17, 0.3f);
        add_light3(s, 2, 7, 1, 0.2f);
        return 0;
    }
    if (!strcmp(name, "teapot") || !strcmp(name, "teapot_head")) {
        /* teapot main.cpp:969-977; flat normals flipped so the mesh is visible ("teapot"), or kept
         * as the reference builds them ("teapot_head": invisible and shadowless) */
        m4 t = m4_translate(&I, V3(0, 1, 10));
        snprintf(path, sizeof path, "%s/teapot.obj", asset_dir);
        orc_add_mesh_obj(s, path, (float *)&t, &red_specular, 1, !strcmp(name, "teapot"));
        add_plane3(s, 0, -3, 0, V3(0, 1, 0), &plain);
        add_plane3(s, 0, 1, 30, V3(0, 0, -1), &green_diffuse);
        add_plane3(s, -15, 1, 0, V3(1, 0, 0), &red_diffuse);
        add_plane3(s, 15, 1, 0, V3(-1, 0, 0), &blue_diffuse);
        add_plane3(s, 0, 27, 0, V3(0, -1, 0), &plain);
        add_light3(s, 10, 16, -7, .5f);
        add_light3(s, 2, 7, 1, 0.2f);
        return 0;
    }
    if (!strcmp(name, "shells")) {
        /* the scene main.cpp builds at HEAD: main.cpp:985-1031, 1052-1054, 1082-1118, 1145-1148, 1173 */
        if (!tex) return -2;
        orc_material shell_textured = mk(0.01f, 0.02f, 0.02f, 0.8f, 0.2f, 0.2f, 0.5f, 10.0f);
        orc_material shell3 = mk(0.01f, 0.02f, 0.02f, 148 / 255.f, 106 / 255.f, 154 / 255.f, 0.5f, 10.0f);
        float angle = (float)atan(-3);
        m4 a = m4_translate(&I, V3(-10, 5, 24.7)), b = m4_rotate(&I, angle, V3(1, 0, 0)), c = m4_scale(&I, V3(0.15, 0.15, 0.15));
        m4 ab = m4_mul(&a, &b), t = m4_mul(&ab, &c);
        snprintf(path, sizeof path, "%s/seashell_obj.obj", asset_dir);
        orc_add_mesh_obj(s, path, (float *)&t, &shell_textured, 1, 0);
        a = m4_translate(&I, V3(9, -8, 16.7)); b = m4_rotate(&I, angle, V3(1, .5, 0)); c = m4_scale(&I, V3(1, 1, 1));
        ab = m4_mul(&a, &b); t = m4_mul(&ab, &c);
        snprintf(path, sizeof path, "%s/shell.obj", asset_dir);
        orc_add_mesh_obj(s, path, (float *)&t, &shell3, 1, 0);
        int sand = add_texset(s, &tex[0]), water = add_texset(s, &tex[1]);
        orc_material water_textured; orc_material_default(&water_textured);
        water_textured.specular[0] = water_textured.specular[1] = water_textured.specular[2] = .6f;
        water_textured.refractiveness = .825f; water_textured.refractive_ratio = .2f;
        water_textured.texture = water;
        orc_material sand_textured; orc_material_default(&sand_textured);
        sand_textured.specular[0] = sand_textured.specular[1] = sand_textured.specular[2] = .2f;
        sand_textured.shininess = 100; sand_textured.texture = sand;
        add_plane3(s, 0, 0, 10, norm3(V3(0.0, 1, -2)), &water_textured);
        add_plane3(s, 0, 0, 25, norm3(V3(0, 1, -3)), &sand_textured);
        add_light3(s, 10, 16, -7, .5f);
        return 0;
    }
    if (!strcmp(name, "textured")) {
        /* textured spheres main.cpp:1075-1135 with reflective/refractive companions */
        if (!tex) return -2;
        int sand = add_texset(s, &tex[0]), water = add_texset(s, &tex[1]);
        int stone = add_texset(s, &tex[2]), waffle = add_texset(s, &tex[3]);
        orc_material stone_textured = mk(0, 0, 0, 1, 1, 1, .6f, 100); stone_textured.texture = stone;
        orc_material waffle_textured = mk(0, 0, 0, 1, 1, 1, .6f, 100); waffle_textured.texture = waffle;
        orc_material water_textured = mk(0, 0, 0, 1, 1, 1, .6f, 0);
        water_textured.refractiveness = .825f; water_textured.refractive_ratio = .2f; water_textured.texture = water;
        orc_material sand_textured = mk(0, 0, 0, 1, 1, 1, .2f, 100); sand_textured.texture = sand;
        orc_material mirror_blue = blue_specular; mirror_blue.reflectiveness = .9f; /* main.cpp:935 */
        m4 tr = m4_translate(&I, V3(0, 1.5, 10)), ro = m4_rotate(&I, .7f, V3(1, 0, 0)), sc = m4_scale(&I, V3(1.5, 1.5, 1.5));
        m4 trro = m4_mul(&tr, &ro), t = m4_mul(&trro, &sc);
        orc_add_sphere(s, (float *)&t, &water_textured);
        tr = m4_translate(&I, V3(-5, 3.5, 12)); trro = m4_mul(&tr, &ro); t = m4_mul(&trro, &sc);
        orc_add_sphere(s, (float *)&t, &waffle_textured);
        tr = m4_translate(&I, V3(4, 2, 10)); ro = m4_rotate(&I, .2f, V3(0, 1, 0)); sc = m4_scale(&I, V3(3, 3, 3));
        trro = m4_mul(&tr, &ro); t = m4_mul(&trro, &sc);
        orc_add_sphere(s, (float *)&t, &stone_textured);
        m4 a = m4_translate(&I, V3(-3, -1, 8)), b = m4_scale(&I, V3(2, 2, 2));
        t = m4_mul(&a, &b);
        orc_add_sphere(s, (float *)&t, &refractive);
        t = m4_translate(&I, V3(1, -2, 8));
        orc_add_sphere(s, (float *)&t, &mirror_blue);
        add_plane3(s, 0, -3, 0, V3(0, 1, 0), &sand_textured);
        add_plane3(s, 0, 1, 30, V3(0, 0, -1), &stone_textured);
        add_plane3(s, -15, 1, 0, V3(1, 0, 0), &red_diffuse);
        add_plane3(s, 15, 1, 0, V3(-1, 0, 0), &blue_diffuse);
        add_plane3(s, 0, 27, 0, V3(0, -1, 0), &plain);
        add_light3(s, 10, 16, -7, .5f);
        add_light3(s, 2, 7, 1, 0.2f);
        return 0;
    }
    return -1;
}
