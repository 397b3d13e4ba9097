/*
 * rt_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C, CPU, single-precision restatement of the per-pixel ray-tracing
 * loop of karma-riuk/ray_tracer (main.cpp pixel loop -> trace_ray ->
 * find_closest_hit -> Object::intersect / PhongModel, plus the texture
 * samplers, the OBJ loader with its observable quirks, tone mapping and the
 * 8-bit quantisation rule).  It exists to CHECK the HIP renderer and to be
 * timed as the CPU baseline; nothing in the shipped library links or calls it.
 *
 * Pinning status: the reference ships no tests or golden images, and its
 * main.cpp cannot be compiled in this image (it includes an un-vendored
 * lodepng header).  The restatement is pinned against (a) values recorded
 * from a run of the unmodified reference (see tests/golden/reference_anchors.json)
 * and (b) Textures.cpp / Image.cpp of the reference compiled as they are
 * (oracle/_ref).  Everything those do not cover is "parity unpinned".
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_scene orc_scene;

/* Material.h:17-26 */
typedef struct {
    float ambient[3];
    float diffuse[3];
    float specular[3];
    float shininess;
    float reflectiveness;
    float refractiveness;
    float refractive_ratio;
    int   texture; /* -1 = none, otherwise a handle from orc_add_*_texture */
} orc_material;

/* Hit (main.cpp:45-53) with the Object* replaced by indices. */
typedef struct {
    int   hit;
    int   object;       /* index into the scene's object list, -1 on a miss   */
    int   frag;         /* mesh fragment index (f-line order), else 0         */
    int   sub;          /* 0/1: triangle inside a quad; 1: cone cap plane     */
    int   from_outside;
    float distance;
    float normal[3];
    float point[3];
    float uv[2];
} orc_hit;

enum { ORC_MODE_REF_1SPP = 0, ORC_MODE_REF_AA9 = 1, ORC_MODE_GRID = 2 };

orc_scene *orc_scene_new(void);
void orc_scene_free(orc_scene *s);
void orc_material_default(orc_material *m);
/* OpenMP threads used by the batch entry points (orc_trace_rays, orc_find_closest_hit, orc_render) */
void orc_set_threads(int n);
int orc_get_threads(void);

/* glm restatements: column-major float[16] matrices. */
void orc_mat_identity(float *m);
void orc_mat_translate(const float *m, const float *v, float *out);
void orc_mat_rotate(const float *m, float angle, const float *axis, float *out);
void orc_mat_scale(const float *m, const float *v, float *out);
void orc_mat_mul(const float *a, const float *b, float *out);

/* Five RGBA8 maps of w*h texels each (copied). height may be NULL (never sampled). */
int orc_add_image_texture(orc_scene *s, unsigned w, unsigned h, const uint8_t *base,
                          const uint8_t *height, const uint8_t *normal,
                          const uint8_t *occlusion, const uint8_t *roughness);
int orc_add_checker_texture(orc_scene *s);
int orc_add_rainbow_texture(orc_scene *s);

int orc_add_sphere(orc_scene *s, const float *M, const orc_material *mat);
int orc_add_plane(orc_scene *s, const float *point, const float *normal, const orc_material *mat);
int orc_add_cone(orc_scene *s, const float *M, const orc_material *mat);
/* Mesh from an OBJ file (reference-compatible parser).  boxed: wrap in the 12-triangle Box gate
 * built from the mesh's local bounds, sharing M.  flip_normals: negate every per-vertex normal
 * after loading (Triangle::set_normals).  Returns object index or <0. */
int orc_add_mesh_obj(orc_scene *s, const char *path, const float *M, const orc_material *mat,
                     int boxed, int flip_normals);
/* Mesh from explicit triangles: 9 floats positions + 9 floats normals per triangle, optional
 * per-triangle material (NULL: all use mat). */
int orc_add_mesh_tris(orc_scene *s, int ntri, const float *pos, const float *nrm,
                      const orc_material *mats, const float *M, const orc_material *mat, int boxed);
/* explicit fragments: per triangle corners, normals, uv (or NULL) and (fragment, sub) ids */
int orc_add_mesh_frags(orc_scene *s, int ntri, const float *pos, const float *nrm, const float *uv,
                       const int *fragsub, const float *M, const orc_material *mat, int boxed);
int orc_add_light(orc_scene *s, const float *pos, const float *color);
void orc_set_ambient(orc_scene *s, const float *rgb);

int orc_num_objects(const orc_scene *s);
int orc_num_lights(const orc_scene *s);
/* number of triangles of mesh object obj (0 for others) and its local bounds */
int orc_mesh_info(const orc_scene *s, int obj, float *bmin, float *bmax);
/* copy triangle data of mesh obj: pos[ntri*9], nrm[ntri*9], uv[ntri*6], fragsub[ntri*2] */
int orc_mesh_copy(const orc_scene *s, int obj, float *pos, float *nrm, float *uv, int *fragsub);
void orc_plane_basis(const orc_scene *s, int obj, float *b1, float *b2);

/* Object::intersect on one object (no closest-hit filters). */
void orc_intersect_object(const orc_scene *s, int obj, int n, const float *o, const float *d,
                          orc_hit *out);
/* find_closest_hit (main.cpp:697-709) */
void orc_find_closest_hit(const orc_scene *s, int n, const float *o, const float *d, orc_hit *out);
/* trace_ray (main.cpp:716-761); max_depth<0: unbounded (as the reference).  rays_out (optional)
 * receives the number of find_closest_hit evaluations per input ray. */
void orc_trace_rays(const orc_scene *s, int n, const float *o, const float *d, int max_depth,
                    float *radiance, uint64_t *rays_out);
/* refract (main.cpp:627-633) on one incident direction */
void orc_refract(const float *i, const float *n, float index, float *out);
/* texture samplers (Textures.cpp:9-55): which = 0 texture(), 1 normal(), 2 occlusion(), 3 roughness() */
void orc_sample_texture(const orc_scene *s, int tex, int which, int n, const float *uv, float *out3);

/* Camera + sampling + tone map + quantisation over the sub-rectangle [x0,x1) x [y0,y1) of a w*h
 * frame (main.cpp:1216-1271).  rgb (optional): w*h*3 linear radiance; rgb8 (optional): w*h*3
 * quantised (Image.cpp:35-39).  Pixels outside the rectangle are left untouched.
 * Returns the number of rays (find_closest_hit evaluations). */
uint64_t orc_render(const orc_scene *s, int w, int h, float fov, int mode, int grid_m,
                    int max_depth, int x0, int y0, int x1, int y1, float *rgb, uint8_t *rgb8,
                    int nthreads);
/* generate the primary rays of a frame without tracing them (same arithmetic as orc_render):
 * o,d: [h*w*taps*3] in (y, x, tap) order; weights: [taps] */
int orc_camera_rays(int w, int h, float fov, int mode, int grid_m, int x0, int y0, int x1, int y1,
                    float *o, float *d, float *weights);
void orc_tonemap_quantise(int n, const float *rgb, uint8_t *rgb8);
/* ASCII P3 writer with the byte layout of Image.cpp:8-23 */
int orc_write_ppm(const char *path, int w, int h, const uint8_t *rgb8);

/* Scene presets.  tex: 4 texture sets in the order sand, water, stone, waffle; each set is
 * 5 RGBA8 maps (base, height, normal, occlusion, roughness) of tex_w*tex_h texels, or NULL when
 * the preset uses no image textures. */
typedef struct {
    unsigned w, h;
    const uint8_t *maps[5];
} orc_texset;
int orc_build_preset(orc_scene *s, const char *name, const char *asset_dir, const orc_texset *tex);

#ifdef __cplusplus
}
#endif
#endif
