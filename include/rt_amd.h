/*
 * rt_amd.h -- C ABI of the MI355X renderer (librt_amd.so).
 *
 * The reference (karma-riuk/ray_tracer) has no FFI layer: its seam is C++-internal.  main()'s pixel
 * loop (main.cpp:1237-1273) calls `glm::vec3 trace_ray(Ray)` (main.cpp:716), which reads the globals
 * `objects`, `lights`, `ambient_light` (main.cpp:621-623) that sceneDefinition() fills through the
 * object constructors.  This header is the replacement for that seam: the host keeps the object API
 * (see ray_tracer_amd/csrc/host/scene.hpp), lowers it to the plain arrays below, and every ray is
 * traced by HIP kernels.  All entry points return 0 on success or a negative rt_status; nothing
 * throws across the boundary.  A context is bound to one GPU and must be used from one host thread
 * at a time.
 */
#ifndef RT_AMD_H
#define RT_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    RT_OK = 0,
    RT_ERR_INVALID = -1,   /* bad argument / inconsistent scene description   */
    RT_ERR_NO_DEVICE = -2, /* no usable GPU                                    */
    RT_ERR_HIP = -3,       /* a HIP runtime call failed (see rt_last_error)    */
    RT_ERR_STATE = -4,     /* call order violated (e.g. render before upload)  */
    RT_ERR_LIMIT = -5      /* a fixed limit was exceeded (BVH depth, sizes)    */
} rt_status;

typedef struct rt_ctx rt_ctx;

/* ---- scene description: plain arrays, copied during rt_scene_upload ------------------------- */

/* Material.h:17-26.  texture: index into rt_scene_desc.textures or -1. */
typedef struct {
    float ambient[3];
    float diffuse[3];
    float specular[3];
    float shininess;
    float reflectiveness;
    float refractiveness;
    float refractive_ratio;
    int32_t texture;
} rt_material;

/* Textures.h:23-50.  Image maps are RGBA8, width*height texels, row-major (what
 * lodepng::decode hands to decodeOneStep, main.cpp:764-794).  The height map of the
 * reference's ImageTexture is never sampled and is not part of the description. */
enum { RT_TEX_CHECKER = 0, RT_TEX_RAINBOW = 1, RT_TEX_IMAGE = 2 };
typedef struct {
    int32_t kind;
    uint32_t width, height;
    const uint8_t *base_color;
    const uint8_t *normal;
    const uint8_t *occlusion;
    const uint8_t *roughness;
} rt_texture;

/* Affine maps are stored as the four columns of the reference's glm::mat4 with the last row
 * dropped: column j occupies [3*j, 3*j+2].  to_local = inverseTransformationMatrix, to_world =
 * transformationMatrix, normal_mat = normalMatrix of Object::setTransformation (main.cpp:85-93). */
typedef struct {
    float to_local[12];
    float to_world[12];
    float normal_mat[12];
    int32_t material;
} rt_sphere; /* Sphere, main.cpp:103-190 */

typedef struct {
    float point[3];
    float normal[3];
    float b1[3]; /* Plane::computeBaseVectors, main.cpp:207-210 */
    float b2[3];
    int32_t material;
} rt_plane; /* Plane, main.cpp:192-258 */

typedef struct {
    float to_local[12];
    float to_world[12];
    float normal_mat[12];
    rt_plane cap; /* Cone::base = Plane((0,1,0),(0,1,0)), main.cpp:261-266; intersected in local space */
    int32_t material;
} rt_cone; /* Cone, main.cpp:260-326 */

/* Mesh (main.cpp:481-548), optionally behind a Box gate (main.cpp:561-619).  Triangles
 * [tri_offset, tri_offset+tri_count) of the scene's triangle arrays, in fragment order. */
typedef struct {
    float to_local[12];
    float to_world[12];
    float normal_mat[12];
    int32_t tri_offset;
    int32_t tri_count;
    int32_t boxed;
    float box_to_local[12]; /* the Box's own inverse transformation */
    float box_min[3];       /* Box::v1 */
    float box_max[3];       /* Box::v2 */
} rt_mesh;

enum { RT_OBJ_SPHERE = 0, RT_OBJ_PLANE = 1, RT_OBJ_CONE = 2, RT_OBJ_MESH = 3 };
typedef struct {
    int32_t kind;
    int32_t index; /* into the array of that kind */
} rt_object;

typedef struct {
    int32_t n_objects;
    const rt_object *objects; /* the order of the reference's `objects` vector; hit IDs index it */
    int32_t n_spheres;
    const rt_sphere *spheres;
    int32_t n_planes;
    const rt_plane *planes;
    int32_t n_cones;
    const rt_cone *cones;
    int32_t n_meshes;
    const rt_mesh *meshes;
    int32_t n_triangles;
    const float *tri_pos;        /* [n][9]  p1 p2 p3 (Triangle, main.cpp:330-351)            */
    const float *tri_nrm;        /* [n][9]  n1 n2 n3                                         */
    const float *tri_uv;         /* [n][6]  uv1 uv2 uv3 (zeros when the face has none)        */
    const int32_t *tri_material; /* [n]                                                      */
    const int32_t *tri_frag;     /* [n]     fragment (f-line) index inside its mesh           */
    const int32_t *tri_sub;      /* [n]     0 / 1: first / second triangle of a quad (Diamond) */
    int32_t n_materials;
    const rt_material *materials;
    int32_t n_textures;
    const rt_texture *textures;
    int32_t n_lights;
    const float *light_pos;   /* [n][3] Light, main.cpp:553-559 */
    const float *light_color; /* [n][3] */
    float ambient[3];         /* ambient_light, main.cpp:622 */
} rt_scene_desc;

/* ---- rendering -------------------------------------------------------------------------------- */

enum {
    RT_MODE_REF_1SPP = 0, /* main.cpp:1262-1267: one un-normalised ray through (i+.5, j-.5)   */
    RT_MODE_REF_AA9 = 1,  /* main.cpp:1243-1260: nine taps on the pixel's edges, tent weights  */
    RT_MODE_GRID = 2      /* grid_m x grid_m equal-weight taps (no counterpart in the reference) */
};
enum { RT_ACCEL_LBVH = 0, RT_ACCEL_BRUTE = 1 };

typedef struct {
    int32_t width, height;
    float fov;         /* degrees; the reference uses 90 (main.cpp:1218) */
    int32_t mode;      /* RT_MODE_*                                                   */
    int32_t grid_m;    /* RT_MODE_GRID only                                           */
    int32_t max_depth; /* rays deeper than this are not traced; the reference is unbounded */
    int32_t accel;     /* RT_ACCEL_*                                                  */
    int32_t rank;      /* this context renders local tile q = tile q * world_size + (rank + q) % world_size, q = 0, 1, ... */
    int32_t world_size;
    int32_t count_work; /* non-zero: also count intersection tests (slower kernel variant) */
    int32_t kernel;     /* RT_KERNEL_*: which render kernel; they all write the same pixels */
} rt_render_params;

/* RT_KERNEL_AUTO picks by a fixed rule (scenes with meshes and 4 or more taps per pixel: QUEUE, and PIXEL for
 * the first frame of a view on one GPU, which records the tiles' times; else PIXEL).
 * PIXEL: one lane per pixel, a block per tile.  SPLIT / SPLIT2: a pixel's taps spread over the waves of a
 * block (a quadrant / half a quadrant per block).  PER_TILE: PIXEL or SPLIT per tile, by the tile's time in
 * the previous frame.  QUEUE: persistent blocks taking tiles and quadrants from a work list, longest first. */
enum { RT_KERNEL_AUTO = 0, RT_KERNEL_PIXEL = 1, RT_KERNEL_SPLIT = 2, RT_KERNEL_SPLIT2 = 3, RT_KERNEL_PER_TILE = 4, RT_KERNEL_QUEUE = 5 };

#define RT_TILE_W 16
#define RT_TILE_H 16
#define RT_MAX_DEPTH 16

/* Hit (main.cpp:45-53) with the Object* replaced by indices. */
typedef struct {
    int32_t hit;
    int32_t object; /* index into rt_scene_desc.objects, -1 on a miss                     */
    int32_t frag;   /* fragment index inside the mesh (0 for other objects)               */
    int32_t sub;    /* 0/1: triangle of a quad; 1: the cap plane of a cone                */
    int32_t from_outside;
    float distance;
    float normal[3];
    float point[3];
    float uv[2];
} rt_hit_record;

typedef struct {
    uint64_t rays_primary;   /* one ray = one find_closest_hit evaluation (main.cpp:653, 718) */
    uint64_t rays_shadow;
    uint64_t rays_secondary; /* reflection + refraction */
    /* Shadow rays whose closest-hit query was run.  The others (rays_shadow - shadow_queries) were
     * answered without one: nothing in the scene can lie between their origin and the light. */
    uint64_t shadow_queries;
    /* the following are filled only when count_work was set */
    uint64_t tri_tests;   /* ray-triangle tests entered (mesh triangles)          */
    uint64_t box_tests;   /* ray-AABB slab tests (two per BVH node visited)        */
    uint64_t gate_tests;  /* Box gates evaluated with the exact triangle test      */
    uint64_t gate_tri_tests; /* triangle tests entered while evaluating those gates */
    uint64_t quadric_tests; /* sphere + cone tests */
    uint64_t plane_tests;
    uint64_t shade_evals; /* (hit, light) Phong evaluations                        */
    uint64_t tex_fetches;
    uint64_t node_steps;       /* tree nodes visited by the packet walk, counted once per wave of 64 rays      */
    uint64_t node_lane_visits; /* rays that took part in those visits: / (64 * node_steps) = how full the packets are */
    float last_trace_ms;  /* device time of the last trace kernel (HIP events)     */
    float last_build_ms;  /* device time of the last rt_build_accel                */
    int32_t bvh_max_depth;
    int32_t render_variant; /* kernel of the last render launch, as RT_KERNEL_* - 1 */
} rt_counters;

/* Create / destroy a context on GPU `device` (a HIP device ordinal). */
int rt_context_create(int device, rt_ctx **out);
int rt_context_destroy(rt_ctx *ctx);
const char *rt_last_error(const rt_ctx *ctx);

/* Replaces filling the globals objects / lights / ambient_light (main.cpp:621-623, 918-1177):
 * copies the description to the GPU.  The caller keeps ownership of every pointer. */
int rt_scene_upload(rt_ctx *ctx, const rt_scene_desc *scene);

/* A new frame of the same geometry: `scene` must have the triangle positions / normals / uvs, mesh
 * triangle ranges and object / light / texture counts of the uploaded scene; transformations (of spheres, cones,
 * meshes and their boxes), planes, materials, lights and the ambient term may differ.  Only those small
 * tables are copied; triangles, textures and the LBVHs (built in each mesh's own space) are kept, so a
 * moving object costs no rebuild.  The reference renders one still image; its .gitignore:3-4 (frames/,
 * *.mp4) shows the frame loop this serves, which with the reference means re-running main() per frame
 * (main.cpp:1212-1288). */
int rt_scene_update(rt_ctx *ctx, const rt_scene_desc *scene);

/* Builds one bounding-volume tree per mesh on the GPU, in the mesh's own space (replaces the linear
 * fragment scan of Mesh::intersect, main.cpp:534-539, with identical closest-hit results): leaves in
 * Morton order, topology by binned surface-area splits (environment RT_BUILD=radix: the radix tree
 * over the Morton codes instead).  Idempotent. */
int rt_build_accel(rt_ctx *ctx);

/* Replaces main()'s pixel loop (main.cpp:1237-1273) including toneMapping (main.cpp:1184-1189) and
 * Image::setPixel quantisation (Image.cpp:35-39).  Host buffers, row-major 3*(y*w+x); either may be
 * NULL.  Blocking.  With world_size > 1 only this rank's tiles are written: the rest of each buffer
 * comes back as the caller passed it, so the ranks of a frame can be rendered into one buffer in turn. */
int rt_render(rt_ctx *ctx, const rt_render_params *p, float *rgb_linear, uint8_t *rgb8);

/* Device-buffer variant for multi-GPU use: renders this rank's tiles into tile-major device
 * buffers ([local_tile][RT_TILE_H][RT_TILE_W][3]); buffers must hold rt_shard_tiles() tiles.
 * Asynchronous on the context's stream; rt_sync waits. */
int rt_shard_tiles(const rt_render_params *p, int32_t *tiles_per_rank, int32_t *tiles_total);
int rt_render_shard_device(rt_ctx *ctx, const rt_render_params *p, float *d_rgb_tiles,
                           uint8_t *d_rgb8_tiles);
/* The same for callers that cannot allocate device memory (tight serving loops, harnesses that forbid
 * hipMalloc): no counterpart in the reference.  It always renders with the RT_KERNEL_PER_TILE kernel and
 * keeps that kernel's per-tile times in a static device array of the library (520 KiB per GPU) instead of
 * memory allocated on demand, and calls no allocating routine.  Limit: at most 65536 tiles (16x16 pixels
 * each) per launch, e.g. 4096 x 4096 pixels on one GPU; RT_ERR_LIMIT beyond that.  count_work is not
 * supported.  All contexts of a GPU share the array, so launches through this entry point run one after
 * the other on that GPU, whatever streams their contexts use (each waits for an event the one before
 * recorded); contexts that alternate frames of different sizes or scenes find no usable tile times and
 * render every tile the slower, split way, with the same pixels.  Not meant for HIP graph capture: the
 * captured launches would keep replaying the tile times of the frame they were captured with. */
int rt_render_shard_device_noalloc(rt_ctx *ctx, const rt_render_params *p, float *d_rgb_tiles,
                                   uint8_t *d_rgb8_tiles);
/* De-interleaves gathered shards ([rank][local_tile][16][16][3], rank-major as all_gather delivers
 * them) into row-major frames on the device.  Either pair may be NULL. */
int rt_untile_device(rt_ctx *ctx, const rt_render_params *p, const float *d_rgb_shards,
                     float *d_rgb_frame, const uint8_t *d_rgb8_shards, uint8_t *d_rgb8_frame);
/* The same on a stream the caller names (hipStream_t; NULL = the default stream) instead of the
 * context's: lets the exchange + untile of one frame run beside the rendering of the next.  The
 * caller orders it after the render / gather that produced the shards (events).  No counterpart in
 * the reference. */
int rt_untile_device_on(rt_ctx *ctx, void *hip_stream, const rt_render_params *p,
                        const float *d_rgb_shards, float *d_rgb_frame,
                        const uint8_t *d_rgb8_shards, uint8_t *d_rgb8_frame);
int rt_sync(rt_ctx *ctx);
/* The HIP stream (hipStream_t) the context launches on, for event timing by the caller. */
void *rt_stream(rt_ctx *ctx);
/* Makes the context launch on a caller-owned stream (e.g. the one a collective library uses), or
 * back on its own when stream is NULL.  Waits for work already queued on the previous stream. */
int rt_set_stream(rt_ctx *ctx, void *hip_stream);

/* trace_ray (main.cpp:716-761) and find_closest_hit (main.cpp:697-709) for arbitrary rays: host
 * arrays o[n][3], d[n][3]; hits and/or radiance[n][3] may be NULL.  The parity entry point. */
int rt_trace_rays(rt_ctx *ctx, int64_t n, const float *o, const float *d, int32_t max_depth,
                  int32_t accel, rt_hit_record *hits, float *radiance);

/* Object::intersect (main.cpp:72) of one object for a batch of rays, without the filters of
 * find_closest_hit.  For a boxed mesh this is Box::intersect (main.cpp:599-618). */
int rt_intersect_object(rt_ctx *ctx, int32_t object, int64_t n, const float *o, const float *d,
                        int32_t accel, rt_hit_record *hits);

/* Counters accumulated since the last rt_reset_counters. */
int rt_get_counters(rt_ctx *ctx, rt_counters *out);
int rt_reset_counters(rt_ctx *ctx);

/* Copies the LBVH of mesh `mesh` back to the host for inspection (tests): nodes as 16 floats each
 * (lo0[3] hi0[3] lo1[3] hi1[3] child0 child1 pad pad, children as int32 bit patterns) and the
 * triangle order (sorted position -> index into the scene's triangle arrays). */
int rt_debug_copy_bvh(rt_ctx *ctx, int32_t mesh, float *nodes, int32_t *n_nodes, int32_t *order,
                      int32_t *n_tris);

#ifdef __cplusplus
}
#endif
#endif
