// rt_core.h -- per-ray arithmetic of the renderer: intersection tests, LBVH traversal, Phong
// shading and the iterative form of the reference's recursive trace_ray.
//
// Everything here is a pure inline function over plain device arrays, written once and used by
// the HIP kernels in rt_device.hip.  The same header compiles as host C++ so that the logic can be
// unit-tested on a machine without a GPU (tests/cpu_twin); that build is test scaffolding and is
// never linked into the shipped library.
//
// Arithmetic follows the reference's evaluation order and float/double promotions (citations at
// each function; paths are relative to the reference checkout).  Build with -ffp-contract=off:
// comparisons must keep IEEE NaN semantics (a NaN refraction direction has to miss everything,
// main.cpp:627-633) and hit/no-hit decisions should round like the reference's.
#pragma once

#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include "rt_amd.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RT_HD __host__ __device__ __forceinline__
#else
#define RT_HD inline
#endif

#if defined(__HIP_DEVICE_COMPILE__)
#define RT_AS4 __attribute__((address_space(4)))
#else
#define RT_AS4
#endif

namespace rtc {

constexpr float kEps = 0.0001f;     // EPS, main.cpp:23
constexpr float kMinDist = 0.01f;   // find_closest_hit threshold, main.cpp:705
constexpr int kStackDepth = 64;     // LBVH traversal stack entries per wave: a tree over unique 64-bit keys is never higher
constexpr double kPi = 3.14159265358979323846;

// Scene features a kernel instantiation supports.  A scene is rendered by the leanest
// instantiation that covers what it contains (fewer live registers, smaller code).
enum : int {
    F_MESH = 1,       // triangle meshes
    F_QUADRIC = 2,    // spheres and cones
    F_TEXTURE = 4,    // any textured material
    F_SECONDARY = 8,  // any reflective or refractive material
    F_ALL = 15
};

struct v2 { float x, y; };
struct v3 { float x, y, z; };
struct alignas(16) f4 { float x, y, z, w; };
struct alignas(16) u4 { uint32_t x, y, z, w; };

RT_HD v3 V3(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
RT_HD v3 ld3(const float *p) { return V3(p[0], p[1], p[2]); }
RT_HD v3 operator+(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_HD v3 operator-(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_HD v3 operator*(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
RT_HD v3 operator*(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
RT_HD v3 operator/(v3 a, float s) { return V3(a.x / s, a.y / s, a.z / s); }
RT_HD v3 operator-(v3 a) { return V3(-a.x, -a.y, -a.z); }
// glm/detail/func_geometric.inl:48-55: (x + y) + z
RT_HD float dot(v3 a, v3 b) { float tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z; return tx + ty + tz; }
// func_geometric.inl:70-78
RT_HD v3 cross(v3 x, v3 y) { return V3(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y); }
RT_HD float length(v3 v) { return sqrtf(dot(v, v)); }
// func_geometric.inl:84-89: v * (1 / sqrt(dot(v, v)))
RT_HD v3 normalize(v3 v) { return v * (1.0f / sqrtf(dot(v, v))); }
// func_geometric.inl:19-22
RT_HD float distance(v3 p0, v3 p1) { return length(p1 - p0); }
// func_geometric.inl:106-109
RT_HD v3 reflect(v3 I, v3 N) { return I - (N * dot(N, I)) * 2.0f; }
RT_HD int32_t f2i(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __float_as_int(f);
#else
    int32_t i; __builtin_memcpy(&i, &f, 4); return i;
#endif
}
// ---- wave-level primitives (a "wave" is one lane in the host build)
RT_HD uint64_t wave_ballot(bool p) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ballot_w64(p);
#else
    return p ? 1ull : 0ull;
#endif
}
// the lane's bit of a wave-uniform mask, as a per-lane condition (no vector instruction on the device)
RT_HD bool wave_mask_bit(uint64_t mask) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_inverse_ballot_w64(mask);
#else
    return (mask & 1ull) != 0;
#endif
}
RT_HD int32_t wave_uniform(int32_t v) {  // the value of the first active lane, as a scalar
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_readfirstlane(v);
#else
    return v;
#endif
}
// A copy of a wave-uniform value that the optimiser cannot trace back to it.  A lane that finds `mine == u` for a
// uniform u = wave_uniform(mine) must go on using u (a scalar register: addresses formed from it give scalar loads);
// an optimiser that knows the two are equal there is free to substitute the per-lane value.  Comparing with
// uniform_alias(u) instead keeps u itself out of the equality.
RT_HD uint64_t uniform_alias64(uint64_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm("" : "+s"(u));
#endif
    return u;
}
RT_HD int32_t uniform_alias(int32_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm("" : "+s"(u));
#endif
    return u;
}
// the value a lane of `mask` holds (the lowest one), as a scalar; mask != 0
RT_HD int32_t wave_value_at(int32_t v, uint64_t mask) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_readlane(v, __builtin_ctzll(mask));
#else
    (void)mask;
    return v;
#endif
}
RT_HD int wave_lane() {
#if defined(__HIP_DEVICE_COMPILE__)
    return (int)__lane_id();
#else
    return 0;
#endif
}
RT_HD int popc64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __popcll(x);
#else
    return __builtin_popcountll(x);
#endif
}
// A value fetched through a wave-uniform address: keeps it in scalar registers and makes the
// compiler complete the fetch HERE, so that several fetches issued just before share one wait
// instead of each being delayed to (and waited for at) its first use.
#if defined(__HIP_DEVICE_COMPILE__)
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
#else
struct u32x16 {
    uint32_t e[16];
    uint32_t &operator[](int k) { return e[k]; }
    const uint32_t &operator[](int k) const { return e[k]; }
};
#endif
RT_HD void pin_scalar(u32x16 &v) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm("" : "+s"(v));
#else
    (void)v;
#endif
}
// 64 bytes of a scene table through a wave-uniform address.  The tables are never written while a
// trace kernel runs, so the read goes through the constant address space: always one scalar load.
RT_HD u32x16 load_u32x16(const void *p) {  // p: 64-byte aligned on the device
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const u32x16 __attribute__((address_space(4))) *const_ptr;
    return *(const_ptr)(uintptr_t)p;
#else
    u32x16 v; memcpy(&v, p, sizeof v); return v;
#endif
}
// the same 64 bytes through an address of the lane's own (vector loads)
RT_HD u32x16 load_u32x16_lane(const void *p) {
#if defined(__HIP_DEVICE_COMPILE__)
    return *reinterpret_cast<const u32x16 *>(p);
#else
    u32x16 v; memcpy(&v, p, sizeof v); return v;
#endif
}
RT_HD float u2f(uint32_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(u);
#else
    float f; memcpy(&f, &u, 4); return f;
#endif
}
#if defined(__HIP_DEVICE_COMPILE__)
#define RT_UNREACHABLE() __builtin_unreachable()
#else
#define RT_UNREACHABLE() ((void)0)
#endif
RT_HD bool has_nan(v3 a) { return !(a.x == a.x) || !(a.y == a.y) || !(a.z == a.z); }

// mat4 * vec4 of glm/detail/type_mat4x4.inl:561-572 on a matrix stored as four 3-row columns:
// (c0*x + c1*y) + (c2*z + c3*w)
RT_HD v3 xform(const float *m, v3 p, float w) {
    v3 add0 = ld3(m) * p.x + ld3(m + 3) * p.y;
    v3 add1 = ld3(m + 6) * p.z + ld3(m + 9) * w;
    return add0 + add1;
}
RT_HD v3 xform_point(const float *m, v3 p) { return xform(m, p, 1.0f); }
// w = 0: the translation column contributes c3 * 0 = +-0, which can only change the sign of a zero
RT_HD v3 xform_dir(const float *m, v3 p) { return (ld3(m) * p.x + ld3(m + 3) * p.y) + ld3(m + 6) * p.z; }
// mat3(c0,c1,c2) * v, glm/detail/type_mat3x3.inl:468-474
RT_HD v3 m3_mul(v3 c0, v3 c1, v3 c2, v3 v) {
    return V3(c0.x * v.x + c1.x * v.y + c2.x * v.z, c0.y * v.x + c1.y * v.y + c2.y * v.z,
              c0.z * v.x + c1.z * v.y + c2.z * v.z);
}
// c ? a : b by components (the conditional operator on the structs themselves selects between their ADDRESSES, and
// the compiler then keeps both in private memory and loads through the selected address)
RT_HD v3 select3(bool c, v3 a, v3 b) { return V3(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z); }
// v as a value the optimiser must treat as computed HERE from `with`: what is derived from it is derived again, not
// kept in registers from an earlier, identical derivation (used where recomputing a few values after a traversal is
// cheaper than holding them through it)
RT_HD v3 again(v3 v, float with) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm("" : "+v"(v.x), "+v"(v.y), "+v"(v.z) : "v"(with));
#else
    (void)with;
#endif
    return v;
}
RT_HD float clamp01(float x) { float m = (x < 0.0f) ? 0.0f : x; return (1.0f < m) ? 1.0f : m; }

// ---------------------------------------------------------------- device-side scene

// LBVH node, 64 bytes: the boxes of both children, then the children.  In the packet traversal the
// node index is wave-uniform, so a node is fetched with ONE scalar load into SGPRs and its twelve
// bounds feed the per-lane slab tests as scalar operands.
// Every mesh stores its nodes eight times, once per ray-direction octant o (bit a set: direction
// component a negative): variant o holds, in the lo* / hi* slots, the NEAR / FAR face of each box for
// that octant (variant 0 is the plain lo / hi node).  A wave whose rays share an octant reads the
// matching variant and needs no per-lane min / max to order the slab distances.  Variant o of node i
// of a mesh with n nodes is nodes[node_offset + o * n + i].
constexpr int kOctants = 8;
// What can shadow light l, summarised for a whole wave of shadow rays.  A shadow ray ends at the
// light, so seen from the light it lies on a line through the light; it can only meet a mesh if that
// line enters the cone from the light around the mesh's bounding sphere.  cone[k] = {axis (negated:
// shadow rays run TOWARDS the light), cos of the half angle} for mesh k; cos = 2 marks an unused
// slot.  Light l is `cullable` when nothing but at most kCullMeshes meshes can shadow it (every plane
// has the light on its front side, there are no quadrics): then a wave whose rays are all outside
// all cones needs no shadow query at all.
constexpr int kCullMeshes = 4;
struct alignas(64) LightCull { f4 cone[kCullMeshes]; };
// Everything the light loop reads of one light, 128 bytes: two scalar loads at a known place (Light, main.cpp:553-559,
// plus the cones of its LightCull).
struct alignas(64) LightRec {
    float pos[3];
    uint32_t flags;  // kLightCullable
    float color[3], pad1;
    uint32_t pad2[8];
    LightCull cull;
};
static_assert(sizeof(LightRec) == 128, "LightRec layout");
constexpr uint32_t kLightCullable = 1u;  // LightRec::flags: the cones decide alone whether a shadow query towards this light is needed
constexpr uint32_t kUnusedCone = 0x40000000u;  // cos = 2.0f: the slot of a mesh that is not there
struct BvhNode {
    float lo0[3], hi0[3];
    float lo1[3], hi1[3];
    int32_t c0, c1;  // >= 0: internal node index (mesh-relative); < 0: ~c = sorted triangle position
    int32_t pad0, pad1;
};

// One mesh.  The first 64 bytes (MeshHead) are all that a ray needs to reject the mesh and to set up
// the traversal; they are fetched as one scalar load.  The second block (to_local and
// the gate boxes) is read only by waves with a ray inside the world box.
struct MeshHead {
    float world_lo[3], world_hi[3];  // padded world-space bounds of the (gate box of the) mesh: first, cheapest rejection
    int32_t object;                  // index in the object list, -1: not in the list
    int32_t tri_count, boxed;
    int32_t same_xform;              // the Box and the mesh share one transformation (always so in the reference's scenes)
    int32_t tri_offset, node_offset;
    int32_t gate_tri_offset;         // first of this mesh's 12 gate triangles in DevScene::gate_tri
    int32_t root;                    // the tree's root node (mesh-relative, like the child links)
    int32_t pad[2];
};
struct alignas(64) DevMesh : MeshHead {
    float to_local[12];
    float gate_lo[3], gate_hi[3];        // padded Box bounds for the conservative pre-test
    float gate_in_lo[3], gate_in_hi[3];  // shrunken Box bounds: an origin inside sees only back faces
    float box_to_local[12], to_world[12], normal_mat[12];
    int32_t pad2[4];
};
static_assert(sizeof(MeshHead) == 64 && sizeof(DevMesh) == 320, "DevMesh layout");

// What the plane pass of find_closest reads.  Planes whose normal is exactly +-e_a (the walls of a room) are kept as
// pairs: the plane with normal +e_a and the one with normal -e_a of one axis a, 32 bytes, either of them possibly
// absent.  Pairs 0, 1, 2 are the first such planes of the axes x, y, z (tested with the ray component known at compile
// time); further planes of an axis follow as overflow pairs.  The array is padded to whole 64-byte loads.  All other
// planes take 32 bytes each and arrive two per load.
// cannot_shadow: a shadow ray runs from a surface point to light l and only asks whether something
// lies closer than the light.  A plane is hit on its front face only, i.e. by a ray that passes from
// its front side to its back side; when the light is on the front side (by a margin far above
// rounding) the crossing lies beyond the light, so the plane cannot change the answer and the test is
// skipped for that light's shadow rays.  In a room lit from inside this removes every wall from
// every shadow ray.
struct AxisPair {
    float coord_pos;     // point[a] of the plane with normal +e_a
    int32_t object_pos;  // its index in the object list, -1: no such plane
    uint32_t cs_pos;     // bit l: the plane can never lie between a surface point and light l (l < 32)
    float coord_neg;     // the same for the plane with normal -e_a
    int32_t object_neg;
    uint32_t cs_neg;
    int32_t axis;        // a
    uint32_t pad;
};
struct PlaneRec {
    float n[3], p[3];
    int32_t object;  // index in the object list, -1: padding
    uint32_t cannot_shadow;
};
constexpr int kAxisPairGroup = 2, kPlaneGroup = 2, kFixedAxisPairs = 4;  // (pairs 0-2 and an empty one: two loads; overflow pairs after them)

// What shading needs of a material, 64 bytes: one scalar load when all lanes shade the same material.
struct MatRec {
    float ambient[3];  // ambient_light * material.ambient: the colour PhongModel starts from (main.cpp:646)
    float diffuse[3], specular[3];
    float shininess, reflectiveness, refractiveness, refractive_ratio;
    uint32_t tex_wh;          // image texture: width | height << 16
    uint32_t tex_lo, tex_hi;  // image texture: the address of its texel records (16-byte aligned); the low 4 bits of
                              // tex_lo say what the texture is: 0 none, 1 + RT_TEX_* otherwise
};
// What turning a closest hit into a shaded surface needs of an object, 128 bytes (two scalar loads when all lanes
// resolve the same object): the geometry that is not in the tables of the object's kind, and the material.
struct alignas(64) ObjRec {
    int32_t kind, index;  // rt_object
    int32_t material;     // index of `mat` in the material table; -1: a mesh whose triangles have materials of their own
    uint32_t pad0;
    float n[3];           // plane: the normal as given (main.cpp:212-218)
    float nn[3];          // plane: normalize(n), what Plane::intersect reports without a normal map (main.cpp:250)
    float b1[3], b2[3];   // plane: texture base vectors (main.cpp:207-210)
    MatRec mat;
};
static_assert(sizeof(MatRec) == 64 && sizeof(ObjRec) == 128 && sizeof(AxisPair) == 32, "record layout");

struct DevTexture {
    int32_t kind;
    uint32_t w, h;
    uint32_t pad;
    // image textures: one 16-byte record per texel holding the RGBA8 texels of all four maps
    // (x base colour, y normal, z occlusion, w roughness), so that a shaded hit costs one fetch
    const u4 *texels;
    const void *pad2[3];
};

// Per sphere: its index in `objects` (-1: not in the object list) and a ball in world space that contains
// it with room to spare (sphere_culls() in rt_prepare.h says how much): rays that pass the ball, leave it
// behind or cannot reach it within the best distance so far skip the sphere's exact test.
struct SphereCull {
    float c[3];
    float r2;        // squared radius
    float r;
    float keep;      // factors <= 1 on the squared distance of the ray's origin from the centre: the room the
    float keep_far;  // rounding of the exact test needs, which grows with that distance (0: never skip)
    int32_t object;
};

// The scene record.  It lives in device memory and the kernels read it through scalar loads, so the small tables
// that every sample needs are part of the record itself (one load at a known offset, no pointer to fetch first):
// the axis-plane pairs, and the first few general planes, mesh heads, object records and lights.  Entries beyond
// those capacities are read through the pointers next to them.  Fields are grouped by the phase that reads them.
constexpr int kEmbGenPlanes = 4, kEmbMeshes = 4, kEmbObjects = 8, kEmbLights = 8;
struct alignas(64) DevScene {
    // ---- closest-hit query
    uint32_t planes_skip_lights;  // bit l: no plane can shadow light l, its shadow rays skip the plane pass (l < 32)
    int32_t n_axis_overflow, n_gen_planes;  // overflow pairs / general planes, both padded to whole loads
    int32_t n_meshes, n_spheres, n_cones, n_planes, n_objects;
    const AxisPair *axis_overflow;  // pairs beyond the fixed ones
    const PlaneRec *gen_planes;     // all general planes (the first kEmbGenPlanes also in gen_first)
    const DevMesh *meshes;          // all meshes (the heads of the first kEmbMeshes also in mesh_heads)
    const SphereCull *sphere_cull;
    AxisPair axis_pairs[kFixedAxisPairs];
    PlaneRec gen_first[kEmbGenPlanes];
    MeshHead mesh_heads[kEmbMeshes];
    const rt_sphere *spheres;
    const rt_cone *cones;
    const int32_t *plane_obj, *cone_obj;  // index of each plane / cone in `objects` (meshes: MeshHead::object)
    const rt_plane *planes;
    // ---- tree walk and Box gate
    const BvhNode *nodes;
    const f4 *tri_test;        // 3 x f4 per sorted triangle: p1 p2 p3 sn
    const uint32_t *tri_key;   // per sorted triangle: 2*frag + (1 - sub), the tie-break order
    const f4 *gate_tri;        // 3 x f4 per Box triangle, 12 per mesh
    int32_t unused_gate_order;  // (keeps the layout)
    int32_t pad0;
    // ---- from a closest hit to a shading context
    const ObjRec *obj_recs;       // per object (the first kEmbObjects also in obj_first)
    const MatRec *mat_recs;       // per material
    const int32_t *tri_orig;   // sorted position -> index into the arrays below
    const float *tri_nrm;      // [n][9]
    const float *tri_uv;       // [n][6]
    const int32_t *tri_material, *tri_frag, *tri_sub;
    const float *srgb_lut;     // [256] pow(b / 255, 2.2): the base-colour decoding of Textures.cpp:41-43 per byte value
    const rt_object *objects;      // (the entry points that report hit records)
    const rt_material *materials;
    const DevTexture *textures;
    ObjRec obj_first[kEmbObjects];
    // ---- lights
    int32_t n_lights;
    uint32_t cullable_lights;       // bit l: the cones of light l decide alone whether a shadow query is needed (l < 32)
    const void *unused_lights;      // (keeps the layout)
    uint32_t pad1[12];
    // The records of ALL lights: the scene record is allocated with room for max(n_lights, kEmbLights) of them, so light
    // l is at a fixed offset from the record's own address (no table pointer to load, no case to tell apart).
    LightRec light_first[kEmbLights];
};
// bytes of a scene record for a scene of n_lights lights
RT_HD size_t scene_record_bytes(int32_t n_lights) { return sizeof(DevScene) + (size_t)(n_lights > kEmbLights ? n_lights - kEmbLights : 0) * sizeof(LightRec); }
// How the per-ray code sees the scene: a reference to the record in device memory, read through the constant
// address space (always scalar loads).  The kernels receive its address, two scalar registers, instead of the
// record itself: sixty-odd kernel-argument registers that stayed live through every loop pushed as many other
// scalar values into vector-register lanes.  fresh() hands out the same reference in a form the optimiser cannot
// connect with earlier uses: what a phase (a closest-hit query, the resolution of a hit, one light) reads of the
// record is fetched when the phase starts and occupies registers only while the phase runs.  `with`: a value of
// the phase's own (its ray, its hit): the reference counts as computed from it, so it cannot be formed before the
// phase.  (The statement emits nothing and touches no memory: loads that follow it stay scalar loads.)
#if defined(__HIP_DEVICE_COMPILE__)
typedef const DevScene RT_AS4 &SceneRef;
RT_HD SceneRef fresh(SceneRef S, float with) {
    const DevScene RT_AS4 *p = &S;
    asm("" : "+s"(p) : "v"(with));
    return *p;
}
RT_HD SceneRef scene_ref(const DevScene *p) { return *(const DevScene RT_AS4 *)(uintptr_t)p; }
#else
typedef const DevScene &SceneRef;
RT_HD SceneRef fresh(SceneRef S, float) { return S; }
RT_HD SceneRef scene_ref(const DevScene *p) { return *p; }
#endif

static_assert(offsetof(DevScene, n_axis_overflow) == 4 && offsetof(DevScene, n_gen_planes) == 8 && offsetof(DevScene, n_meshes) == 12 &&
                  offsetof(DevScene, n_spheres) == 16 && offsetof(DevScene, n_cones) == 20 && offsetof(DevScene, axis_overflow) == 32 &&
                  offsetof(DevScene, axis_pairs) == 64 && offsetof(DevScene, mesh_heads) % 64 == 0 && offsetof(DevScene, obj_first) % 64 == 0 &&
                  offsetof(DevScene, light_first) % 64 == 0,
              "find_closest reads the first 64 bytes of the scene record as one load");
// Where record i of a table lives: in the scene record itself (an address known without a load), or behind the
// table's pointer.  The pointer is read in the branch that needs it, through a reference of its own (fresh), so that
// the common case never waits for it.
RT_HD const char *gen_plane_at(SceneRef S, int32_t i, float with) {
    if (i < kEmbGenPlanes) return (const char *)(uintptr_t)&S.gen_first[i];
    return (const char *)(fresh(S, with).gen_planes + i);
}
RT_HD const char *mesh_head_at(SceneRef S, int32_t i, float with) {
    if (i < kEmbMeshes) return (const char *)(uintptr_t)&S.mesh_heads[i];
    return (const char *)(fresh(S, with).meshes + i);
}
RT_HD const char *obj_rec_at(SceneRef S, int32_t i, float with) {
    // (the embedded address first, from an unsigned 32-bit offset, and the table's in its place for the objects beyond:
    // as two returns the compiler builds the case from a flag and two branches, eleven scalar instructions a sample)
    const uint32_t off = (uint32_t)i * (uint32_t)sizeof(ObjRec);
    const char *p = (const char *)(uintptr_t)&S.obj_first[0] + off;
    if (i >= kEmbObjects) p = (const char *)fresh(S, with).obj_recs + off;
    return p;
}
RT_HD const char *light_rec_at(SceneRef S, int32_t i) {
    return (const char *)(uintptr_t)&S.light_first[0] + (uint32_t)i * (uint32_t)sizeof(LightRec);  // (an unsigned 32-bit offset: no 64-bit shift)
}

// per-thread work counters; the counting variant is a separate template instantiation
struct Work {
    uint32_t rays_primary, rays_shadow, rays_secondary, shadow_q;
    uint32_t tri, box, gate, gate_tri, quadric, plane, shade, tex;
    uint32_t node_steps, node_lanes;  // packet walk: node visits counted once per wave, and the lanes that took part in them
};
RT_HD void work_zero(Work &w) {
    w.rays_primary = w.rays_shadow = w.rays_secondary = w.shadow_q = 0;
    w.tri = w.box = w.gate = w.gate_tri = w.quadric = w.plane = w.shade = w.tex = 0;
    w.node_steps = w.node_lanes = 0;
}

// ---------------------------------------------------------------- textures

// An image texture as the samplers need it: size and texel records (wave-uniform values when they come from a
// MatRec that was fetched for the whole wave, per-lane values otherwise).
struct TexRef { uint32_t w, h; const u4 *texels; };
// ImageTexture::getRGBAat (Textures.cpp:26-39) for all four maps at once: the texel record at uv.
// `int % unsigned` is unsigned arithmetic.
template <bool COUNT>
RT_HD u4 tex_texel(TexRef t, v2 uv, Work &wk) {
    if (COUNT) wk.tex++;
    float fx = 2 * uv.x * (float)t.w;
    float fy = 2 * uv.y * (float)t.h;
    int xi = (int)roundf(fx), yi = (int)roundf(fy);
    uint32_t x = (uint32_t)xi % t.w;
    uint32_t y = (uint32_t)yi % t.h;
    return t.texels[y * t.w + x];
}
RT_HD f4 unpack_rgba8(uint32_t px) {
    f4 r;
    r.x = (float)(px & 255u) / 255.f; r.y = (float)((px >> 8) & 255u) / 255.f;
    r.z = (float)((px >> 16) & 255u) / 255.f; r.w = (float)(px >> 24) / 255.f;
    return r;
}
RT_HD v3 unpack_rgb8(uint32_t px) { f4 c = unpack_rgba8(px); return V3(c.x, c.y, c.z); }
// Texture::texture(uv) of the procedural textures: Textures.cpp:9-24
RT_HD v3 tex_procedural(int kind, v2 uv) {
    if (kind == RT_TEX_CHECKER) {
        int v = (int)(floorf(16 * uv.x) + floorf(16 * uv.y)) % 2;
        return V3((float)v, (float)v, (float)v);
    }
    int v = (int)floorf(18 * (uv.x + uv.y)) % 3;
    return V3(v == 0 ? 1.0f : 0.0f, v == 1 ? 1.0f : 0.0f, v == 2 ? 1.0f : 0.0f);
}
// ImageTexture::texture (Textures.cpp:41-43): pow(channel, 2.2) of the base colour, from the table
RT_HD v3 tex_base_color(SceneRef S, uint32_t px) {
    return V3(S.srgb_lut[px & 255u], S.srgb_lut[(px >> 8) & 255u], S.srgb_lut[(px >> 16) & 255u]);
}
RT_HD const DevTexture *image_texture_of(SceneRef S, const rt_material &m) {
    if (m.texture < 0) return nullptr;
    const DevTexture *t = &S.textures[m.texture];
    return t->kind == RT_TEX_IMAGE ? t : nullptr;
}
RT_HD TexRef tex_ref(const DevTexture &t) { TexRef r; r.w = t.w; r.h = t.h; r.texels = t.texels; return r; }
// what a MatRec says about its texture
RT_HD int mat_tex_kind(uint32_t tex_lo) { return (int)(tex_lo & 15u) - 1; }  // -1: none, else RT_TEX_*
RT_HD TexRef mat_tex_ref(uint32_t tex_wh, uint32_t tex_lo, uint32_t tex_hi) {
    TexRef r;
    r.w = tex_wh & 0xffffu; r.h = tex_wh >> 16;
    r.texels = reinterpret_cast<const u4 *>((uintptr_t)(((uint64_t)tex_hi << 32) | (uint64_t)(tex_lo & ~15u)));
    return r;
}

// ---------------------------------------------------------------- analytic primitives

// A resolved hit without its material (Hit's point, normal, uv and from_outside, main.cpp:45-53) plus the
// texture record the surface function read on the way (normal-mapped surfaces), so that shading need not read
// it again.  Plain scalars only: the struct lives in registers.
struct SurfGeo {
    v3 point, normal;
    v2 uv;
    uint32_t tex_color, tex_normal, tex_occlusion, tex_roughness;  // the four RGBA8 texels at uv, when has_texel
    bool has_texel, from_outside;
};
RT_HD void geo_set_texel(SurfGeo &g, u4 t) { g.tex_color = t.x; g.tex_normal = t.y; g.tex_occlusion = t.z; g.tex_roughness = t.w; g.has_texel = true; }
struct Surf : SurfGeo {  // a resolved hit as the hit records report it
    int32_t material;
    int32_t frag, sub;
};

// Plane::intersect, main.cpp:222-257.  Returns the ray parameter (the plane's `distance`).
RT_HD bool plane_hit(const rt_plane &pl, v3 ro, v3 rd, float &t) {
    v3 n = ld3(pl.normal);
    float DdotN = dot(rd, n);
    float PdotN = dot(ld3(pl.point) - ro, n);
    t = PdotN / DdotN;  // evaluated unconditionally (no divergent exit); only used when DdotN < 0
    return (DdotN < 0) & (t > 0);
}
// Plane::intersect, main.cpp:231-250: point, uv and normal of a hit at ray parameter t.  n: the plane's normal as
// given, nn: normalize(n) (what the plane reports unless its material has an image texture: then the normal map's
// texel is carried into the basis (b1, b2, n), main.cpp:241-249).
template <bool COUNT>
RT_HD void plane_geo(v3 n, v3 nn, v3 b1, v3 b2, bool image, TexRef tex, v3 ro, v3 rd, float t, SurfGeo &s, Work &wk) {
    s.point = rd * t + ro;
    s.uv.x = dot(b1, s.point);
    s.uv.y = dot(b2, s.point);
    s.has_texel = false;
    s.normal = nn;
    if (image) {
        geo_set_texel(s, tex_texel<COUNT>(tex, s.uv, wk));
        v3 tn = unpack_rgb8(s.tex_normal);
        tn = tn * 2.f - V3(1, 1, 1);
        s.normal = normalize(m3_mul(b1, b2, n, tn));
    }
    s.from_outside = true;
}

struct QuadricHit {
    float t;  // local ray parameter of the chosen root
    bool cap; // cone: the root lies above y = 1 and the cap plane is used
    bool from_outside;
    float distance;
    v3 o, d, i; // local ray and local hit point
};

// True when the exact test below cannot produce a hit that find_closest would keep: the ray's line passes
// the ball, or the ball lies behind the origin, or its nearest point is farther than the best distance so
// far (a sphere reports the Euclidean distance of its hit point, main.cpp:183).  The exact test is float
// arithmetic on local coordinates: its discriminant is uncertain by 6e-7 |o|^2 (o: the local origin) in
// units of the squared local distance between line and centre, so far-away rays "hit" a shell around the
// sphere that grows with |o|.  `keep` and `keep_far` shrink |oc|^2 by that shell (with margins, see
// sphere_culls() in rt_prepare.h), so that every ray the exact test could accept passes this test.
// NaN operands make every comparison false.
RT_HD bool misses_ball(const SphereCull &k, v3 ro, v3 rd, float best_distance) {
    v3 oc = V3(k.c[0], k.c[1], k.c[2]) - ro;
    float occ = dot(oc, oc);
    float b = dot(oc, rd), dd = dot(rd, rd);
    float out = occ * k.keep - k.r2;  // > 0: the origin is outside the ball and its shell
    float reach = best_distance + k.r;
    return (dd * out > b * b) | ((out > 0) & (b < 0)) | (occ * k.keep_far > reach * reach);
}

// Sphere::intersect, main.cpp:122-160 + 183 (geometry, acceptance inputs).  pow(x, 2) on floats is
// double arithmetic in the reference.
RT_HD bool sphere_geom(const rt_sphere &sp, v3 ro, v3 rd, QuadricHit &q) {
    v3 o = xform_point(sp.to_local, ro);
    v3 d = xform_dir(sp.to_local, rd);
    float a = (float)((double)d.x * d.x + (double)d.y * d.y + (double)d.z * d.z);
    float b = 2 * (d.x * o.x + d.y * o.y + d.z * o.z);
    float c = (float)((double)o.x * o.x + (double)o.y * o.y + (double)o.z * o.z - 1);
    float delta = (float)((double)b * b - (double)(4 * a * c));
    if (delta < 0) return false;
    float t1 = (-b + sqrtf(delta)) / (2 * a);
    float t2 = (-b - sqrtf(delta)) / (2 * a);
    if (t1 < 0 && t2 < 0) return false;
    float ta = t1 < 0 ? INFINITY : t1, tb = t2 < 0 ? INFINITY : t2;
    float t = tb < ta ? tb : ta;
    q.t = t;
    q.o = o; q.d = d;
    q.i = o + d * t;
    q.cap = false;
    q.from_outside = !(fabsf(o.x) < 1 && fabsf(o.y) < 1 && fabsf(o.z) < 1);
    q.distance = distance(ro, xform_point(sp.to_world, q.i));
    return true;
}
// Sphere::intersect, main.cpp:153-181 (uv, normal mapping, world transform)
template <bool COUNT>
RT_HD void sphere_geo(const rt_sphere &sp, const QuadricHit &q, bool image, TexRef tex, SurfGeo &s, Work &wk) {
    v3 i = q.i;
    s.uv.x = (float)(((double)atan2f(i.z, i.x) + kPi) / kPi);
    s.uv.y = (float)(((double)asinf(i.y) + kPi / 2) / kPi);
    s.point = xform_point(sp.to_world, i);
    v3 n;
    s.has_texel = false;
    if (image) {
        v3 tangent = V3(sinf(s.uv.x), 0, cosf(s.uv.x));
        v3 bitangent = cross(i, tangent);
        geo_set_texel(s, tex_texel<COUNT>(tex, s.uv, wk));
        v3 tn = unpack_rgb8(s.tex_normal);
        tn = tn * 2.f - V3(1, 1, 1);
        tn = m3_mul(normalize(tangent), normalize(bitangent), normalize(i), tn);
        n = xform_dir(sp.normal_mat, tn);
    } else {
        n = xform_dir(sp.normal_mat, i);
    }
    s.normal = normalize(n);
    s.from_outside = q.from_outside;
}

// Cone::intersect, main.cpp:269-325
RT_HD bool cone_geom(const rt_cone &cn, v3 ro, v3 rd, QuadricHit &q) {
    v3 d = xform_dir(cn.to_local, rd);
    v3 o = xform_point(cn.to_local, ro);
    float a = (float)((double)d.x * d.x - (double)d.y * d.y + (double)d.z * d.z);
    float b = 2 * (d.x * o.x - d.y * o.y + d.z * o.z);
    float c = (float)((double)o.x * o.x - (double)o.y * o.y + (double)o.z * o.z);
    float delta = (float)((double)b * b - (double)(4 * a * c));
    if (delta < 0) return false;
    float t1 = (-b + sqrtf(delta)) / (2 * a);
    float t2 = (-b - sqrtf(delta)) / (2 * a);
    if (t1 < 0 && t2 < 0) return false;
    v3 i1 = o + d * t1, i2 = o + d * t2;
    if ((i1.y < 0 && i2.y < 0) || (i1.y > 1 && i2.y > 1)) return false;
    float ta = t1 < 0 ? INFINITY : t1, tb = t2 < 0 ? INFINITY : t2;
    float t = tb < ta ? tb : ta;
    v3 i = o + d * t;
    q.o = o; q.d = d;
    q.cap = i.y > 1;
    q.from_outside = true;
    if (q.cap) {
        float tp;
        if (!plane_hit(cn.cap, o, d, tp)) return false;  // the cap is tested with the local ray
        q.t = tp;
        q.i = d * tp + o;
    } else {
        q.t = t;
        q.i = i;
    }
    q.distance = distance(ro, xform_point(cn.to_world, q.i));
    return true;
}
// cap_image / cap_tex: the image texture of the cap plane's material (the cap is a Plane of its own, main.cpp:261-266)
template <bool COUNT>
RT_HD void cone_geo(const rt_cone &cn, const QuadricHit &q, bool cap_image, TexRef cap_tex, SurfGeo &s, Work &wk) {
    v3 n;
    if (q.cap) {
        SurfGeo ps;
        v3 cap_n = ld3(cn.cap.normal);
        plane_geo<COUNT>(cap_n, normalize(cap_n), ld3(cn.cap.b1), ld3(cn.cap.b2), cap_image, cap_tex, q.o, q.d, q.t, ps, wk);
        s.uv = ps.uv;
        n = ps.normal;
        s.point = ps.point;
    } else {
        v3 i = q.i;
        float y = (float)(((double)i.x * i.x + (double)i.z * i.z) / i.y + i.y);
        n = i - V3(0, y, 0);
        s.uv.x = atan2f(i.x, i.z);
        s.uv.y = i.y;
        s.point = i;
    }
    s.has_texel = false;  // (the cap's texel, if any, was read at the cap's own uv for the cap's own material: shading reads its own)
    s.point = xform_point(cn.to_world, s.point);
    s.normal = normalize(xform_dir(cn.normal_mat, n));
    s.from_outside = true;
}

// ---------------------------------------------------------------- triangles and meshes

struct TriBest {
    float t;
    int32_t k;     // sorted triangle position, -1: none
    uint32_t key;  // tie-break order of k
};

// Triangle::intersect, main.cpp:372-395 (hit / no-hit and the ray parameter).  The candidate
// replaces the current best when its t is smaller, or equal with a smaller key: Mesh::intersect
// keeps the first fragment among equal t (main.cpp:536) and Diamond::intersect the second
// triangle of a quad (main.cpp:476); keys encode exactly that order, so the result does not
// depend on the order in which triangles are visited.
template <bool COUNT>
RT_HD void tri_test(const f4 *tt, int32_t k, uint32_t key, v3 o, v3 d, TriBest &best, Work &wk) {
    if (COUNT) wk.tri++;
    f4 A = tt[3 * k], B = tt[3 * k + 1], C = tt[3 * k + 2];
    v3 p1 = V3(A.x, A.y, A.z), p2 = V3(A.w, B.x, B.y), p3 = V3(B.z, B.w, C.x), sn = V3(C.y, C.z, C.w);
    float DdotN = dot(d, sn);
    if (!(DdotN < 0)) return;
    float PdotN = dot(p1 - o, sn);
    float t = PdotN / DdotN;
    if (!(t > 0)) return;
    if (t > best.t || (t == best.t && key > best.key)) return;
    v3 p = d * t + o;
    v3 sn1 = cross(p2 - p, p3 - p);
    v3 sn2 = cross(p3 - p, p1 - p);
    v3 sn3 = cross(p1 - p, p2 - p);
    // glm::sign(x) < 0 is x < 0; zero and NaN pass (main.cpp:390-395)
    if (dot(sn1, sn) < 0 || dot(sn2, sn) < 0 || dot(sn3, sn) < 0) return;
    best.t = t;
    best.k = k;
    best.key = key;
}

// hit / no-hit only (the Box gate, main.cpp:609-615)
RT_HD bool tri_hits(const f4 *tt, int32_t k, v3 o, v3 d) {
    TriBest b;
    b.t = INFINITY; b.k = -1; b.key = 0xffffffffu;
    Work wk;
    tri_test<false>(tt, k, 0u, o, d, b, wk);
    return b.k >= 0;
}

// Triangle::intersect, main.cpp:397-413: interpolated normal, uv, from_outside (local space)
RT_HD void tri_surface(SceneRef S, int32_t k, float t, v3 o, v3 d, v3 &p, v3 &n, v2 &uv,
                       bool &from_outside, int32_t &orig) {
    const f4 *tt = S.tri_test;
    f4 A = tt[3 * k], B = tt[3 * k + 1], C = tt[3 * k + 2];
    v3 p1 = V3(A.x, A.y, A.z), p2 = V3(A.w, B.x, B.y), p3 = V3(B.z, B.w, C.x), sn = V3(C.y, C.z, C.w);
    p = d * t + o;
    v3 sn1 = cross(p2 - p, p3 - p);
    v3 sn2 = cross(p3 - p, p1 - p);
    v3 sn3 = cross(p1 - p, p2 - p);
    orig = S.tri_orig[k];
    const float *nr = S.tri_nrm + 9 * (size_t)orig;
    float l1 = length(sn1), l2 = length(sn2), l3 = length(sn3);
    n = normalize((ld3(nr) * l1 + ld3(nr + 3) * l2) + ld3(nr + 6) * l3);
    float W = length(sn);
    const float *tu = S.tri_uv + 6 * (size_t)orig;
    v3 u1 = V3(tu[0], tu[1], 0), u2 = V3(tu[2], tu[3], 0), u3 = V3(tu[4], tu[5], 0);
    if (sqrtf(u1.x * u1.x + u1.y * u1.y) > 0) {
        float w1 = length(sn1 / W), w2 = length(sn2 / W), w3 = length(sn3 / W);
        v3 r = (u1 * w1 + u2 * w2) + u3 * w3;
        uv.x = r.x; uv.y = r.y;
    } else {
        uv.x = l1 / W; uv.y = l2 / W;
    }
    from_outside = dot(n, d) < 0;
}

// Ray/box slab test against bounds that were padded at build time.  NaN products (0 * inf) are
// dropped by fminf/fmaxf, which makes the test conservative for rays lying in a bounding plane.
RT_HD bool slab(const float *lo, const float *hi, v3 o, v3 inv, float tmax, float &tnear) {
    float ax = (lo[0] - o.x) * inv.x, bx = (hi[0] - o.x) * inv.x;
    float ay = (lo[1] - o.y) * inv.y, by = (hi[1] - o.y) * inv.y;
    float az = (lo[2] - o.z) * inv.z, bz = (hi[2] - o.z) * inv.z;
    float tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.0f));
    float tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), tmax));
    tnear = tn;
    return tn <= tf * 1.0000004f;
}

RT_HD float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
// fminf(a, b) for a b that is known not to be a signalling NaN but not known so to the compiler, which would first pass
// it through an instruction that quiets it
RT_HD float fmin_plain(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    return fminf(a, b);
#endif
}
// Reciprocal for the conservative box tests only (never for a hit decision): 1-ulp hardware rcp on
// the GPU; a zero component becomes a huge finite value so that the fused slab test cannot produce
// inf - inf.
RT_HD float safe_rcp(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return fabsf(x) >= 1e-30f ? __builtin_amdgcn_rcpf(x) : copysignf(1e30f, x);
#else
    return fabsf(x) >= 1e-30f ? 1.0f / x : copysignf(1e30f, x);
#endif
}

RT_HD float safe_rcp_of(float x, float rcp_x) { return fabsf(x) >= 1e-30f ? rcp_x : copysignf(1e30f, x); }  // safe_rcp(x), given the reciprocal

// The same test in fused form for the traversal loop: t = bound * inv - o * inv.  The boxes of the
// tree are padded by ~100 ulps of the mesh's largest coordinate, far more than the rounding
// difference between the fused and the unfused form, so the test stays conservative.
RT_HD bool slab_fused(float lox, float loy, float loz, float hix, float hiy, float hiz, v3 inv, v3 oi, float tmax, float &tnear) {
    float ax = fma_(lox, inv.x, -oi.x), bx = fma_(hix, inv.x, -oi.x);
    float ay = fma_(loy, inv.y, -oi.y), by = fma_(hiy, inv.y, -oi.y);
    float az = fma_(loz, inv.z, -oi.z), bz = fma_(hiz, inv.z, -oi.z);
    float tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.0f));
    float tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), tmax));
    tnear = tn;
    return tn <= tf;
}

// Branch-free form of tri_test for the traversal loop: every lane that holds a leaf runs the whole
// test and the outcome is folded into one predicate (identical decisions, no divergent exits).
template <bool COUNT>
RT_HD void tri_test_flat(const f4 *tt, int32_t k, uint32_t key, v3 o, v3 d, bool in, TriBest &best, Work &wk) {
    // in: the lane's ray reached this leaf.  The other lanes of the wave run the same instructions and keep what they had:
    // a branch around the test would be the one divergent branch of the walk's loop, and with it the compiler rebuilds
    // the whole loop's control flow from flag registers (some ten scalar instructions more per node).
    if (COUNT) wk.tri += in ? 1u : 0u;
    f4 A = tt[3 * k], B = tt[3 * k + 1], C = tt[3 * k + 2];
    v3 p1 = V3(A.x, A.y, A.z), p2 = V3(A.w, B.x, B.y), p3 = V3(B.z, B.w, C.x), sn = V3(C.y, C.z, C.w);
    float DdotN = dot(d, sn);
    float PdotN = dot(p1 - o, sn);
    float t = PdotN / DdotN;
    bool ok = in & (DdotN < 0) & (t > 0) & !((t > best.t) | ((t == best.t) & (key > best.key)));
    v3 p = d * t + o;
    v3 sn1 = cross(p2 - p, p3 - p);
    v3 sn2 = cross(p3 - p, p1 - p);
    v3 sn3 = cross(p1 - p, p2 - p);
    float e1 = dot(sn1, sn), e2 = dot(sn2, sn), e3 = dot(sn3, sn);
    ok = ok & !((e1 < 0) | (e2 < 0) | (e3 < 0));
    best.t = ok ? t : best.t;
    best.k = ok ? k : best.k;
    best.key = ok ? key : best.key;
}

// The walk itself.  ONE_OCTANT: all rays point into one octant and `nodes` is that octant's copy of the tree, whose
// boxes hold near / far faces and whose children are stored front to back; otherwise the plain copy, per-lane
// min / max in the slab test and the stored order.  The control decisions are kept to a few scalar instructions per
// step: the scalar unit serves all four SIMDs of a compute unit at one instruction per cycle, and the walk's
// steps were bound by it (about 48 scalar against 32 vector instructions per node with a voted visiting order).
template <bool COUNT, bool ONE_OCTANT, class Stack>
RT_HD void bvh_walk(const BvhNode *nodes, const f4 *tt, const uint32_t *keys, int32_t start, v3 o, v3 d, v3 inv, v3 oi, uint64_t inm,
                    TriBest &best, Stack &stk, Work &wk) {
    int sp = 0;  // wave-uniform stack height
    int32_t cur = start;
    for (;;) {
        if (cur >= 0) {
            // the whole node in scalar registers first: bounds 0-11 (lo0 hi0 lo1 hi1), children 12-13
            u32x16 nd = load_u32x16(reinterpret_cast<const char *>(nodes) + ((uint32_t)cur << 6));
            pin_scalar(nd);
            const bool in = wave_mask_bit(inm);
            if (COUNT && in) { wk.box += 2; wk.node_lanes++; }
            if (COUNT && wave_uniform(wave_lane()) == wave_lane()) wk.node_steps++;  // (one lane of the wave counts the visit)
            // a lane that is not in the visit gets an empty interval: its tests below fail by themselves
            float tm = in ? best.t : -1.0f;
            uint64_t b0, b1;
            if (ONE_OCTANT) {
                float n0 = fmaxf(fmaxf(fma_(u2f(nd[0]), inv.x, -oi.x), fma_(u2f(nd[1]), inv.y, -oi.y)), fmaxf(fma_(u2f(nd[2]), inv.z, -oi.z), 0.0f));
                float f0 = fminf(fminf(fma_(u2f(nd[3]), inv.x, -oi.x), fma_(u2f(nd[4]), inv.y, -oi.y)), fmin_plain(fma_(u2f(nd[5]), inv.z, -oi.z), tm));
                float n1 = fmaxf(fmaxf(fma_(u2f(nd[6]), inv.x, -oi.x), fma_(u2f(nd[7]), inv.y, -oi.y)), fmaxf(fma_(u2f(nd[8]), inv.z, -oi.z), 0.0f));
                float f1 = fminf(fminf(fma_(u2f(nd[9]), inv.x, -oi.x), fma_(u2f(nd[10]), inv.y, -oi.y)), fmin_plain(fma_(u2f(nd[11]), inv.z, -oi.z), tm));
                b0 = wave_ballot(n0 <= f0); b1 = wave_ballot(n1 <= f1);
            } else {
                float n0, n1;
                bool h0 = slab_fused(u2f(nd[0]), u2f(nd[1]), u2f(nd[2]), u2f(nd[3]), u2f(nd[4]), u2f(nd[5]), inv, oi, tm, n0);
                bool h1 = slab_fused(u2f(nd[6]), u2f(nd[7]), u2f(nd[8]), u2f(nd[9]), u2f(nd[10]), u2f(nd[11]), inv, oi, tm, n1);
                b0 = wave_ballot(h0); b1 = wave_ballot(h1);
            }
            // (A wave-uniform condition that is looked at in two blocks is kept as a lane mask and inverted by two vector
            // instructions; a mask that is tested twice therefore goes through uniform_alias64 the second time.)
            if (b0 != 0) {
                if (b1 != 0) stk.put(sp++, (int32_t)nd[13], b1);
                cur = (int32_t)nd[12]; inm = b0;
                continue;
            }
            const uint64_t b1_alone = uniform_alias64(b1);
            if (b1_alone != 0) {
                cur = (int32_t)nd[13]; inm = b1_alone;
                continue;
            }
        } else {
            int32_t k = ~cur;
            tri_test_flat<COUNT>(tt, k, keys[k], o, d, wave_mask_bit(inm), best, wk);
        }
        if (sp == 0) break;
        stk.get(--sp, cur, inm);
    }
}

// Closest triangle of a mesh for the local-space rays of a whole wave: PACKET traversal of the
// mesh's LBVH.  The wave walks the tree once, with one wave-uniform node index and one stack; every
// lane tests the node's two child boxes against its own ray, and a child is visited when ANY lane
// reaches it (the lanes that do are recorded as a 64-bit mask with the stack entry).  Node and
// triangle records are therefore fetched by scalar loads and there is no per-lane stack and no
// divergent control flow.  A lane only tests triangles of leaves whose box its own ray entered
// within its current best t, exactly the triangles a private traversal would test, so the result
// is identical to the reference's linear scan.  Rays of one wave come from one 8x8 pixel block (or
// its shadow / bounce rays) and share most of their path through the tree.
// `wants`: the lanes whose rays take part (a wave-uniform mask).  Stack: put(slot, node, mask) / get(slot, node, mask).
template <bool COUNT, class Stack>
RT_HD void bvh_closest_packet(SceneRef S, const MeshHead &m, v3 o, v3 d, uint64_t wants, TriBest &best, Stack &stk, Work &wk) {
    if (wants == 0) return;
    const BvhNode *nodes = S.nodes + m.node_offset;
    const f4 *tt = S.tri_test + 3 * (size_t)m.tri_offset;
    const uint32_t *keys = S.tri_key + m.tri_offset;
    // A zero direction component would give inf - inf = NaN in the fused slab test; a huge finite
    // reciprocal keeps the sign logic (the ray then counts as parallel to that slab).
    v3 inv = V3(safe_rcp(d.x), safe_rcp(d.y), safe_rcp(d.z));
    v3 oi = o * inv;
    // When all participating rays point into the same octant, which face of a box is the near one
    // is a wave-uniform fact: the wave reads the node variant stored for that octant and the per-lane
    // min / max pairs of the slab test disappear (same values, same decisions: t(lo) <= t(hi) for a
    // positive reciprocal, the reverse for a negative one).
    const uint64_t negx = wave_ballot(inv.x < 0) & wants, negy = wave_ballot(inv.y < 0) & wants, negz = wave_ballot(inv.z < 0) & wants;
    // per axis: 0 all positive, 1 all negative, 2 mixed (plain integers: conditions kept as such become lane masks)
    const int32_t sx = negx == 0 ? 0 : negx == wants ? 1 : 2, sy = negy == 0 ? 0 : negy == wants ? 1 : 2, sz = negz == 0 ? 0 : negz == wants ? 1 : 2;
    const int32_t start = m.tri_count == 1 ? ~0 : m.root;
    if ((sx | sy | sz) < 2) {
        nodes += (size_t)(sx | (sy << 1) | (sz << 2)) * (size_t)(m.tri_count - 1);
        bvh_walk<COUNT, true>(nodes, tt, keys, start, o, d, inv, oi, wants, best, stk, wk);
    } else {
        bvh_walk<COUNT, false>(nodes, tt, keys, start, o, d, inv, oi, wants, best, stk, wk);
    }
}

// Mesh::intersect (main.cpp:525-547) behind the optional Box gate (main.cpp:599-618): geometry and
// the acceptance inputs of find_closest_hit.
struct MeshHit {
    float t;        // local ray parameter (Triangle's `distance`)
    int32_t k;      // sorted triangle position (scene-wide)
    float distance; // Euclidean, world space (main.cpp:545)
    bool from_outside;
};

// clip: hits with a local ray parameter beyond it cannot matter to the caller (shadow rays: the
// light's distance); INFINITY for radiance rays.  The closest triangle within the clip range is
// the mesh's closest triangle whenever that one lies within the range, so the result the caller
// uses is unchanged.
//
// Box gate = "some one-sided Box face is hit".  Evaluated as: (1) a conservative slab test on the
// padded box rejects rays that pass it by; (2) a ray whose origin lies strictly inside the box
// only sees back faces, so the gate is closed; (3) for the remaining rays the tree is searched
// first and the 12 exact triangle tests run only for rays that found a triangle (the gate is a
// conjunct of the result, so the order does not matter).
template <bool COUNT, class Stack>
RT_HD bool mesh_geom(SceneRef S, const MeshHead &h, const DevMesh &m, v3 ro, v3 rd, v3 winv, v3 woi, uint64_t want, int accel, float clip, MeshHit &mh, Stack &stk, Work &wk) {
    // want: the lanes whose rays take part, a wave-uniform mask (every test below is one comparison whose lane mask is
    // combined with it by scalar instructions)
    // phase 1, per lane: cheap rejections
    float tn;
    if (COUNT && wave_mask_bit(want)) wk.box++;
    if (h.tri_count <= 0) return false;
    want &= wave_ballot(slab_fused(h.world_lo[0], h.world_lo[1], h.world_lo[2], h.world_hi[0], h.world_hi[1], h.world_hi[2], winv, woi, INFINITY, tn));
    if (want == 0) return false;  // the whole wave misses the mesh's world box
    v3 d = xform_dir(m.to_local, rd);
    v3 o = xform_point(m.to_local, ro);
    {   // evaluated for every mesh (no branch, so that to_local and the boxes arrive in one batch of loads);
        // the outcome only counts for boxed meshes
        v3 bo = o, bd = d;  // the ray in the Box's own space
        if (!h.same_xform) {
            bd = xform_dir(m.box_to_local, rd);
            bo = xform_point(m.box_to_local, ro);
        }
        v3 binv = V3(safe_rcp(bd.x), safe_rcp(bd.y), safe_rcp(bd.z));
        if (COUNT && wave_mask_bit(want) && h.boxed != 0) wk.box++;
        const uint64_t g = wave_ballot(slab_fused(m.gate_lo[0], m.gate_lo[1], m.gate_lo[2], m.gate_hi[0], m.gate_hi[1], m.gate_hi[2], binv, bo * binv, INFINITY, tn));
        const uint64_t inside = wave_ballot(bo.x > m.gate_in_lo[0]) & wave_ballot(bo.x < m.gate_in_hi[0]) & wave_ballot(bo.y > m.gate_in_lo[1]) &
                                wave_ballot(bo.y < m.gate_in_hi[1]) & wave_ballot(bo.z > m.gate_in_lo[2]) & wave_ballot(bo.z < m.gate_in_hi[2]);
        if (h.boxed != 0) want &= g & ~inside;
    }
    // phase 2, whole wave: the tree
    TriBest best;
    best.t = clip; best.k = -1; best.key = 0xffffffffu;
    if (accel == RT_ACCEL_BRUTE) {
        if (wave_mask_bit(want)) {
            const f4 *tt = S.tri_test + 3 * (size_t)h.tri_offset;
            const uint32_t *keys = S.tri_key + h.tri_offset;
            for (int32_t k = 0; k < h.tri_count; k++) tri_test<COUNT>(tt, k, keys[k], o, d, best, wk);
        }
    } else {
        bvh_closest_packet<COUNT>(S, h, o, d, want, best, stk, wk);
    }
    // phase 3, per lane: the exact Box gate and the acceptance inputs, for lanes that found a triangle
    if (best.k < 0) return false;
    if (h.boxed) {
        if (COUNT) wk.gate++;
        // the ray in the Box's space once more (the same arithmetic as above: formed again here rather than held in six
        // registers through the traversal)
        const float *to_box = h.same_xform ? m.to_local : m.box_to_local;
        const v3 rd2 = again(rd, best.t), ro2 = again(ro, best.t);
        const v3 bd = xform_dir(to_box, rd2), bo = xform_point(to_box, ro2);
        const f4 *gt = S.gate_tri + 3 * (size_t)h.gate_tri_offset;
        bool any = false;
        // The gate is the OR of the 12 one-sided triangle tests, so their order is free: the two triangles of the face the
        // ray enters through (largest entry parameter of the three slabs) come first and nearly always answer; the others
        // run only for lanes they left undecided (teapot frame: 24 M instead of 55 M executed gate triangle tests, shells
        // 340 M instead of 1595 M).  Faces in make_gate_tris' order: z lo, x hi, z hi, x lo, y hi, y lo.
        int32_t face;
        {
            float ex = ((bd.x > 0 ? m.gate_lo[0] : m.gate_hi[0]) - bo.x) * safe_rcp(bd.x);
            float ey = ((bd.y > 0 ? m.gate_lo[1] : m.gate_hi[1]) - bo.y) * safe_rcp(bd.y);
            float ez = ((bd.z > 0 ? m.gate_lo[2] : m.gate_hi[2]) - bo.z) * safe_rcp(bd.z);
            face = (ex >= ey && ex >= ez) ? (bd.x > 0 ? 3 : 1) : (ey >= ez) ? (bd.y > 0 ? 5 : 4) : (bd.z > 0 ? 0 : 2);
            if (COUNT) wk.gate_tri += 2;
            any = tri_hits(gt, 2 * face, bo, bd) | tri_hits(gt, 2 * face + 1, bo, bd);
        }
        for (int i = 0; i < 12; i++) {
            if (any) break;
            if ((i >> 1) == face) continue;
            if (COUNT) wk.gate_tri++;
            any = tri_hits(gt, i, bo, bd);
        }
        if (!any) return false;
    }
    mh.t = best.t;
    mh.k = best.k + h.tri_offset;
    v3 p, n; v2 uv; int32_t orig;
    tri_surface(S, mh.k, mh.t, o, d, p, n, uv, mh.from_outside, orig);
    mh.distance = distance(ro, xform_point(m.to_world, p));
    return true;
}
RT_HD void mesh_geo(SceneRef S, const DevMesh &m, v3 ro, v3 rd, float t, int32_t k, SurfGeo &s, int32_t &orig) {
    s.has_texel = false;
    v3 d = xform_dir(m.to_local, rd);
    v3 o = xform_point(m.to_local, ro);
    v3 p, n;
    tri_surface(S, k, t, o, d, p, n, s.uv, s.from_outside, orig);
    s.point = xform_point(m.to_world, p);
    s.normal = normalize(xform_dir(m.normal_mat, n));
}

// ---------------------------------------------------------------- closest hit

// Shading-only arithmetic.  Decisions (which object is hit, shadowed or not) and every value that
// feeds a later ray use the reference's exact operations above; the colour arithmetic below may
// use the hardware's 1-ulp reciprocal and exp2/log2 on the GPU, which stays orders of magnitude
// inside the 1e-4 relative radiance bar.
RT_HD float shade_rcp(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(x);
#else
    return 1.0f / x;
#endif
}
RT_HD float shade_rsq(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rsqf(x);
#else
    return 1.0f / sqrtf(x);
#endif
}
// pow(x, y) for x in [0, 1], y >= 0 (Phong lobe, main.cpp:678)
// fast: no lane of the wave has an exponent above 256 (shade_pow_fast).  One decision for the wave: the library's powf is
// some 150 instructions, and exponents above 256 do not occur in the reference's materials.
RT_HD uint32_t shade_pow_fast(float y) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ballot_w64(!(y <= 256.0f)) == 0 ? 1u : 0u;
#else
    (void)y;
    return 0u;
#endif
}
RT_HD float shade_pow(float x, float y, uint32_t fast) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (fast != 0) {
        const float p = __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x));
        return y == 0.0f ? 1.0f : p;
    }
#else
    (void)fast;
#endif
    return powf(x, y);
}


// The winner of find_closest_hit, reduced to what is needed to rebuild the full hit record:
// t is the ray parameter in the object's own space; k is the sorted triangle position (mesh) or
// 1 for a cone's cap plane.
struct Closest {
    float distance;
    int32_t object;  // -1: miss
    float t;
    int32_t k;
};

// find_closest_hit, main.cpp:697-709.  The reference walks the object list in order and keeps the
// first of equal distances; here objects are visited kind by kind (planes, quadrics, meshes) and a
// candidate replaces the best when its distance is smaller, or equal with a smaller object index,
// which selects the same object.
// occl: a shadow ray only needs to know whether something accepted lies closer than the light;
// once the best distance is below occl the (expensive) meshes are skipped.  -INFINITY disables it.
RT_HD bool closer(float dist, int32_t idx, const Closest &best) {
    return (dist > kMinDist) & ((dist < best.distance) | ((dist == best.distance) & (idx < best.object)));
}
// ---- planes (Plane::intersect, main.cpp:222-237)
//
// The reference divides PdotN by DdotN for every plane the ray faces and keeps the smallest quotient above 0.01
// (ties: the plane that comes first in the object list).  Here one pass over the plane records forms, per lane and
// without a branch, a 1-ulp ESTIMATE of each quotient (PdotN times the hardware reciprocal of DdotN) and keeps the
// smallest one with its PdotN, DdotN and object, plus the second smallest estimate.  The exact quotient is then
// formed once, for the lane's winner.  An estimate is within 2e-7 relative of its quotient, so the winner is the
// reference's whenever the runner-up is more than 1e-5 relative behind and the 0.01 cut is not within 1e-5 relative
// of the winner; any lane for which that cannot be said sends the whole wave through planes_exact, the reference's
// own sequence of divisions (a ray within 1e-5 relative of an edge where two planes meet: practically never).
// +-0 for a vector of finite components, NaN otherwise
RT_HD float zero_or_nan(v3 v) { return __builtin_fmaf(v.x, 0.0f, __builtin_fmaf(v.y, 0.0f, v.z * 0.0f)); }
struct PlaneCand {
    float est;       // smallest estimate so far (INFINITY: none)
    float second;    // second smallest
    float P, D;      // PdotN and DdotN of the plane with the smallest estimate
    int32_t object;
};
constexpr float kEstMin = kMinDist * (1.0f - 1e-5f);
RT_HD uint32_t umin32(uint32_t a, uint32_t b) { return a < b ? a : b; }
RT_HD float med3(float a, float b, float c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_fmed3f(a, b, c);
#else
    return fmaxf(fminf(a, b), fminf(fmaxf(a, b), c));
#endif
}
// est: NaN (or anything not above kEstMin) unless the lane's ray faces the plane and the plane lies ahead
RT_HD void plane_candidate(PlaneCand &c, float est, float P, float D, int32_t object) {
    float e = est > kEstMin ? est : INFINITY;
    bool lt = e < c.est;
    c.second = med3(c.second, e, c.est);  // (second >= est always: the median of the three is the new second smallest)
    c.est = lt ? e : c.est; c.P = lt ? P : c.P; c.D = lt ? D : c.D; c.object = lt ? object : c.object;
}
// The two planes of one axis a, normals +e_a and -e_a.  A ray faces at most one of them: +e_a when its direction
// component d_a is negative, -e_a when it is positive (with d_a == 0 neither: the estimate is then infinite or NaN),
// so one estimate serves the pair.  For a normal +-e_a the reference's dot products reduce exactly to the a-th
// components (x * 0 = +-0 and v + +-0 = v, finite components: the caller handles the others), and for -e_a the
// quotient (-q) / (-d_a) equals q / d_a bit for bit: the candidate carries q and d_a themselves.
// coord_pos / coord_neg: NaN when there is no such plane (or none that can shadow this wave's light); r_a: 1-ulp 1 / d_a.
template <bool COUNT>
RT_HD void plane_axis_pair(float coord_pos, int32_t object_pos, float coord_neg, int32_t object_neg, float o_a, float d_a, float r_a,
                           PlaneCand &c, Work &wk) {
    const bool down = d_a < 0;
    const float coord = down ? coord_pos : coord_neg;
    const int32_t object = down ? object_pos : object_neg;
    if (COUNT) wk.plane += (coord_pos == coord_pos ? 1u : 0u) + (coord_neg == coord_neg ? 1u : 0u);  // one estimate decides both tests
    const float q = coord - o_a;
    plane_candidate(c, q * r_a, q, d_a, object);
}
// words `at` .. `at + 7` of g: an AxisPair; a plane that cannot shadow the wave's light counts as absent
RT_HD float pair_coord(uint32_t coord, uint32_t cs, uint32_t light_bit) { return u2f((cs & light_bit) != 0 ? 0x7fc00000u : coord); }  // (a select of scalars)
template <bool COUNT>
RT_HD void plane_general(const u32x16 &g, int at, uint32_t light_bit, v3 ro, v3 rd, PlaneCand &c, Work &wk) {
    int32_t idx = (int32_t)g[at + 6];
    // padding, or cannot shadow this wave's light (wave-uniform; formed as one integer: a condition built from two stays
    // a pair of lane masks)
    if ((((uint32_t)idx >> 31) | (g[at + 7] & light_bit)) != 0) return;
    if (COUNT) wk.plane++;
    v3 n = V3(u2f(g[at]), u2f(g[at + 1]), u2f(g[at + 2]));
    float DdotN = dot(rd, n);
    float PdotN = dot(V3(u2f(g[at + 3]), u2f(g[at + 4]), u2f(g[at + 5])) - ro, n);
    float est = PdotN * shade_rcp(DdotN);
    plane_candidate(c, DdotN < 0 ? est : NAN, PdotN, DdotN, idx);
}
// The reference's own sequence for every listed plane (the rare waves with a lane too close to call).
template <bool COUNT>
RT_HD void planes_exact(SceneRef S, v3 ro, v3 rd, Closest &best, Work &wk) {
    best.distance = INFINITY; best.object = -1; best.t = 0; best.k = 0;
    for (int32_t i = 0; i < S.n_planes; i++) {
        int32_t idx = S.plane_obj[i];
        if (idx < 0) continue;
        if (COUNT) wk.plane++;
        float t;
        bool ok = plane_hit(S.planes[i], ro, rd, t) && closer(t, idx, best);
        best.distance = ok ? t : best.distance; best.object = ok ? idx : best.object; best.t = ok ? t : best.t;
    }
}
// rinv: 1-ulp reciprocals of the direction's components; g0, g1: the fixed axis pairs (S.axis_pairs as two loads);
// n_axis_overflow, n_gen_planes, axis_overflow: the scene record's fields of these names
template <bool COUNT>
RT_HD void planes_pass(SceneRef S, const u32x16 &g0, const u32x16 &g1, int32_t n_axis_overflow, int32_t n_gen_planes, const AxisPair *axis_overflow,
                       v3 ro, v3 rd, v3 rinv, uint32_t light_bit, Closest &best, Work &wk) {
    PlaneCand c;
    c.est = INFINITY; c.second = INFINITY; c.P = 0; c.D = 1; c.object = -1;
    {
        // (a shadow ray's wave skips a pair neither of whose planes can shadow its light: wave-uniform; for other rays
        // light_bit is a constant 0 and the checks fold away)
        // (a pair is dead when neither of its planes is there and can shadow the wave's light: one integer per plane, 0 = alive)
        // (for rays that are no shadow rays light_bit is the constant 0 once find_closest is inlined, and every pair is evaluated
        // without a look: an empty pair holds NaN coordinates and yields no candidate)
#define RT_PAIR_LIVE(g, at) (__builtin_constant_p(light_bit) || umin32((g[at + 1] >> 31) | (g[at + 2] & light_bit), (g[at + 4] >> 31) | (g[at + 5] & light_bit)) == 0)
        if (RT_PAIR_LIVE(g0, 0))
            plane_axis_pair<COUNT>(pair_coord(g0[0], g0[2], light_bit), (int32_t)g0[1], pair_coord(g0[3], g0[5], light_bit), (int32_t)g0[4], ro.x, rd.x, rinv.x, c, wk);
        if (RT_PAIR_LIVE(g0, 8))
            plane_axis_pair<COUNT>(pair_coord(g0[8], g0[10], light_bit), (int32_t)g0[9], pair_coord(g0[11], g0[13], light_bit), (int32_t)g0[12], ro.y, rd.y, rinv.y, c, wk);
        if (RT_PAIR_LIVE(g1, 0))
            plane_axis_pair<COUNT>(pair_coord(g1[0], g1[2], light_bit), (int32_t)g1[1], pair_coord(g1[3], g1[5], light_bit), (int32_t)g1[4], ro.z, rd.z, rinv.z, c, wk);
#undef RT_PAIR_LIVE
    }
    // overflow pairs (an axis with several planes of one orientation): the axis is a wave-uniform value
    for (int32_t i = 0; i < n_axis_overflow; i += 2) {
        u32x16 g = load_u32x16(axis_overflow + i);
        pin_scalar(g);
        for (int at = 0; at < 16; at += 8) {
            const int32_t a = (int32_t)g[at + 6];
            float o_a = a == 0 ? ro.x : a == 1 ? ro.y : ro.z, d_a = a == 0 ? rd.x : a == 1 ? rd.y : rd.z;
            float r_a = a == 0 ? rinv.x : a == 1 ? rinv.y : rinv.z;
            plane_axis_pair<COUNT>(pair_coord(g[at], g[at + 2], light_bit), (int32_t)g[at + 1], pair_coord(g[at + 3], g[at + 5], light_bit),
                                   (int32_t)g[at + 4], o_a, d_a, r_a, c, wk);
        }
    }
    for (int32_t i = 0; i < n_gen_planes; i += kPlaneGroup) {
        u32x16 g = load_u32x16(gen_plane_at(S, i, rd.y));
        pin_scalar(g);
        plane_general<COUNT>(g, 0, light_bit, ro, rd, c, wk);
        plane_general<COUNT>(g, 8, light_bit, ro, rd, c, wk);
    }
    if (light_bit != 0 && wave_ballot(c.est < INFINITY) == 0) {  // (shadow rays mostly find no plane ahead at all)
        best.distance = INFINITY; best.object = -1; best.t = 0;
        return;
    }
    const float t = c.P / c.D;  // the reference's quotient (main.cpp:233) for the lane's winner
    // With a non-finite component the reference's dot products with +-e_a are NaN and the plane is missed (the general
    // planes' own dot products say so themselves): nf is +-0 for a finite ray and NaN otherwise, and a ray with a
    // non-finite component misses every plane.
    // (not a value of the reference's arithmetic, only +-0 or NaN: fused, three instructions a vector)
    const float nf = zero_or_nan(rd) + zero_or_nan(ro);
    const float e = c.est + nf;
    const bool has = e < INFINITY;
    // (an estimate so large that e * 1.00001f is infinite counts as unsure too: the exact quotient may not be finite)
    const uint64_t unsure = wave_ballot(e < INFINITY) & (wave_ballot(!(c.second > e * 1.00001f)) | wave_ballot(!(t > kMinDist)));
    if (unsure != 0) {
        planes_exact<COUNT>(S, ro, rd, best, wk);
        return;
    }
    best.distance = has ? t : INFINITY; best.object = has ? c.object : -1; best.t = has ? t : 0.0f;
}

template <int F, bool COUNT, class Stack>
RT_HD void find_closest(SceneRef scene, v3 ro, v3 rd, int accel, int light, float clip, float occl, Closest &best, Stack &stk, Work &wk) {
    SceneRef S = fresh(scene, rd.x);
    // light: >= 0 when the wave's rays are shadow rays towards that light (wave-uniform: the light loop keeps the lanes
    // of a wave in step); planes that cannot shadow it are skipped.
    const uint32_t light_bit = (uint32_t)(1ull << umin32((uint32_t)light, 63u));  // (bit `light` of 32, 0 for any other value: a minimum and a shift)
    best.distance = INFINITY;
    best.object = -1;
    best.t = 0;
    best.k = 0;
    const v3 rinv = V3(shade_rcp(rd.x), shade_rcp(rd.y), shade_rcp(rd.z));  // estimates and box tests only, never a hit decision
    // everything the query reads of the scene record in one batch of loads: its first 64 bytes (counts and table
    // pointers), the fixed axis pairs and the first mesh's head
    u32x16 hdr = load_u32x16((const void *)(uintptr_t)&S.planes_skip_lights);
    u32x16 g0 = load_u32x16((const void *)(uintptr_t)&S.axis_pairs[0]), g1 = load_u32x16((const void *)(uintptr_t)&S.axis_pairs[2]);
    u32x16 head0 = (F & F_MESH) ? load_u32x16((const void *)(uintptr_t)&S.mesh_heads[0]) : g0;
    pin_scalar(g0); pin_scalar(g1);  // (not the head: pinned as a whole it is copied register by register to where its fields are used)
    const int32_t n_axis_overflow = (int32_t)hdr[1], n_gen_planes = (int32_t)hdr[2], n_meshes = (int32_t)hdr[3];
    const int32_t n_spheres = (int32_t)hdr[4], n_cones = (int32_t)hdr[5];
    if ((hdr[0] & light_bit) == 0)
        planes_pass<COUNT>(S, g0, g1, n_axis_overflow, n_gen_planes, reinterpret_cast<const AxisPair *>((uintptr_t)(((uint64_t)hdr[9] << 32) | hdr[8])),
                           ro, rd, rinv, light_bit, best, wk);
    const bool meshes = (F & F_MESH) && n_meshes > 0;
    if (F & F_QUADRIC)
    for (int32_t i = 0; i < n_spheres; i++) {
        QuadricHit q;
        const SphereCull &k = S.sphere_cull[i];
        int32_t idx = k.object;
        if (idx < 0) continue;
        if (misses_ball(k, ro, rd, best.distance)) continue;
        if (COUNT) wk.quadric++;
        if (sphere_geom(S.spheres[i], ro, rd, q) && q.from_outside && closer(q.distance, idx, best)) {
            best.distance = q.distance; best.object = idx; best.t = q.t; best.k = 0;
        }
    }
    if (F & F_QUADRIC)
    for (int32_t i = 0; i < n_cones; i++) {
        QuadricHit q;
        if (COUNT) wk.quadric++;
        int32_t idx = S.cone_obj[i];
        if (idx < 0) continue;
        if (cone_geom(S.cones[i], ro, rd, q) && closer(q.distance, idx, best)) {
            best.distance = q.distance; best.object = idx; best.t = q.t; best.k = q.cap ? 1 : 0;
        }
    }
    if (meshes) {
    // every triangle test fails on NaN inputs (an unordered comparison finds a NaN in either operand)
    const uint64_t valid = ~(wave_ballot(__builtin_isunordered(ro.x, ro.y)) | wave_ballot(__builtin_isunordered(ro.z, rd.x)) |
                             wave_ballot(__builtin_isunordered(rd.y, rd.z)));
    u32x16 next_head = head0;  // the first mesh's, from the batch of loads above; a later one is fetched when the mesh before it is done
    // reciprocals for the world-box tests: the first mesh takes the plane pass's, a later mesh forms them again (they are
    // not held through the traversal of the mesh before)
    v3 winv = V3(safe_rcp_of(rd.x, rinv.x), safe_rcp_of(rd.y, rinv.y), safe_rcp_of(rd.z, rinv.z));
    for (int32_t i = 0; i < n_meshes; i++) {
        const v3 woi = ro * winv;
        union { u32x16 raw; MeshHead h; } head;
        pin_scalar(next_head);
        head.raw = next_head;
        const int32_t idx = head.h.object;
        if (idx >= 0) {  // (wave-uniform) in the object list
            MeshHit mh;
            const uint64_t want = valid & ~wave_ballot(best.distance < occl);
            if (mesh_geom<COUNT>(S, head.h, S.meshes[i], ro, rd, winv, woi, want, accel, clip, mh, stk, wk) && mh.from_outside && closer(mh.distance, idx, best)) {
                best.distance = mh.distance; best.object = idx; best.t = mh.t; best.k = mh.k;
            }
        }
        if (i + 1 < n_meshes) {
            next_head = load_u32x16(mesh_head_at(S, i + 1, best.distance));
            const v3 rdi = again(rd, best.distance);
            winv = V3(safe_rcp(rdi.x), safe_rcp(rdi.y), safe_rcp(rdi.z));
        }
    }
    }
}

// The full hit record of a closest hit, lane by lane (the entry points that report records; the render kernels
// shade through surface_shade below and never build one).
template <int F, bool COUNT>
RT_HD void resolve(SceneRef S, const Closest &c, v3 ro, v3 rd, Surf &s, Work &wk) {
    rt_object ob = S.objects[c.object];
    s.frag = 0;
    s.sub = 0;
    TexRef none;
    none.w = none.h = 1; none.texels = nullptr;
    if (ob.kind == RT_OBJ_PLANE) {
        const rt_plane &pl = S.planes[ob.index];
        const DevTexture *it = (F & F_TEXTURE) ? image_texture_of(S, S.materials[pl.material]) : nullptr;
        v3 n = ld3(pl.normal);
        plane_geo<COUNT>(n, normalize(n), ld3(pl.b1), ld3(pl.b2), it != nullptr, it ? tex_ref(*it) : none, ro, rd, c.t, s, wk);
        s.material = pl.material;
    } else if ((F & F_MESH) && ob.kind == RT_OBJ_MESH) {
        int32_t orig;
        mesh_geo(S, S.meshes[ob.index], ro, rd, c.t, c.k, s, orig);
        s.material = S.tri_material[orig];
        s.frag = S.tri_frag[orig];
        s.sub = S.tri_sub[orig];
    } else if (!(F & F_QUADRIC)) {
        // unreachable: the scene has no object of another kind
    } else if (ob.kind == RT_OBJ_SPHERE) {
        const rt_sphere &sp = S.spheres[ob.index];
        QuadricHit q;
        q.o = xform_point(sp.to_local, ro); q.d = xform_dir(sp.to_local, rd);
        q.t = c.t; q.cap = false; q.i = q.o + q.d * c.t;
        q.from_outside = !(fabsf(q.o.x) < 1 && fabsf(q.o.y) < 1 && fabsf(q.o.z) < 1);
        const DevTexture *it = (F & F_TEXTURE) ? image_texture_of(S, S.materials[sp.material]) : nullptr;
        sphere_geo<COUNT>(sp, q, it != nullptr, it ? tex_ref(*it) : none, s, wk);
        s.material = sp.material;
    } else {
        const rt_cone &cn = S.cones[ob.index];
        QuadricHit q;
        q.d = xform_dir(cn.to_local, rd); q.o = xform_point(cn.to_local, ro);
        q.t = c.t; q.cap = c.k != 0; q.from_outside = true;
        q.i = select3(q.cap, q.d * c.t + q.o, q.o + q.d * c.t);
        const DevTexture *it = (F & F_TEXTURE) ? image_texture_of(S, S.materials[cn.cap.material]) : nullptr;
        cone_geo<COUNT>(cn, q, it != nullptr, it ? tex_ref(*it) : none, s, wk);
        s.material = cn.material;
        s.sub = q.cap ? 1 : 0;
    }
}

RT_HD void fill_hit(rt_hit_record &h, const Closest &c, const Surf &s) {
    h.hit = c.object >= 0; h.object = c.object; h.from_outside = 1; h.distance = c.distance;
    h.frag = 0; h.sub = 0;
    for (int i = 0; i < 3; i++) { h.normal[i] = 0; h.point[i] = 0; }
    h.uv[0] = h.uv[1] = 0;
    if (c.object >= 0) {
        h.frag = s.frag; h.sub = s.sub; h.from_outside = s.from_outside ? 1 : 0;
        h.normal[0] = s.normal.x; h.normal[1] = s.normal.y; h.normal[2] = s.normal.z;
        h.point[0] = s.point.x; h.point[1] = s.point.y; h.point[2] = s.point.z;
        h.uv[0] = s.uv.x; h.uv[1] = s.uv.y;
    }
}
// the record of a closest-hit query (a miss included)
template <int F, bool COUNT>
RT_HD void hit_record(SceneRef S, const Closest &c, v3 ro, v3 rd, rt_hit_record &h, Work &wk) {
    Surf s;
    s.frag = 0; s.sub = 0; s.from_outside = true; s.has_texel = false; s.material = 0;
    s.point = V3(0, 0, 0); s.normal = V3(0, 0, 0); s.uv.x = s.uv.y = 0;
    if (c.object >= 0) resolve<F, COUNT>(S, c, ro, rd, s, wk);
    fill_hit(h, c, s);
}

// Object::intersect for one object without the closest-hit filters (the per-object parity entry)
template <bool COUNT, class Stack>
RT_HD void intersect_object(SceneRef S, int32_t k, v3 ro, v3 rd, int accel, rt_hit_record &out, Stack &stk, Work &wk) {
    Closest c;
    c.object = -1; c.distance = INFINITY; c.t = 0; c.k = 0;
    bool from_outside = true;
    rt_object ob = S.objects[k];
    bool hit = false;
    if (ob.kind == RT_OBJ_PLANE) {
        float t;
        hit = plane_hit(S.planes[ob.index], ro, rd, t);
        c.t = t; c.distance = t;
    } else if (ob.kind == RT_OBJ_MESH) {
        MeshHit mh;
        v3 winv = V3(safe_rcp(rd.x), safe_rcp(rd.y), safe_rcp(rd.z));
        hit = mesh_geom<COUNT>(S, S.meshes[ob.index], S.meshes[ob.index], ro, rd, winv, ro * winv, ~0ull, accel, INFINITY, mh, stk, wk);
        if (hit) { c.t = mh.t; c.k = mh.k; c.distance = mh.distance; from_outside = mh.from_outside; }
    } else {
        QuadricHit q;
        hit = ob.kind == RT_OBJ_SPHERE ? sphere_geom(S.spheres[ob.index], ro, rd, q) : cone_geom(S.cones[ob.index], ro, rd, q);
        if (hit) { c.t = q.t; c.k = q.cap ? 1 : 0; c.distance = q.distance; from_outside = q.from_outside; }
    }
    Surf s;
    s.frag = 0; s.sub = 0; s.from_outside = from_outside; s.has_texel = false; s.material = 0;
    s.point = V3(0, 0, 0); s.normal = V3(0, 0, 0); s.uv.x = s.uv.y = 0;
    if (hit) {
        c.object = k;
        resolve<F_ALL, COUNT>(S, c, ro, rd, s, wk);
    }
    fill_hit(out, c, s);
    out.from_outside = from_outside ? 1 : 0;
    if (!hit) out.distance = INFINITY;
}

// ---------------------------------------------------------------- shading

// refract, main.cpp:627-633 (not glm::refract; no total-internal-reflection branch)
RT_HD v3 refract_dir(v3 i, v3 n, float index) {
    v3 a = n * dot(i, n);
    v3 b = i - a;
    float beta = 1 / index;
    float alpha = (float)sqrt(1 + (1 - (double)beta * beta) * (double)(dot(b, b) / dot(a, a)));
    return a * alpha + b * beta;
}

// What PhongModel (main.cpp:642-695) and the recursion of trace_ray (main.cpp:728-757) need of a hit: the surface
// and the material's terms, per lane, in registers.
struct ShadeCtx {
    v3 point, normal, d_in;  // d_in: direction of the ray that produced the hit (the view vector is normalize(-d_in))
    v3 color;                // running sum, starts at ambient_light * material.ambient
    v3 dc;                   // diffuse colour (texture(uv) or material.diffuse), x5 for image textures
    v3 spec;                 // material.specular
    float shininess;
    uint32_t occlusion_px;   // RGBA8 texel of the occlusion map (image-textured surfaces)
    uint32_t flags;          // kShadeImage, kShadeOutside
    float refl, refr, ratio; // material.reflectiveness / refractiveness / refractive_ratio (kernels with secondary rays)
};
enum : uint32_t { kShadeImage = 1u, kShadeOutside = 2u };

// The material of a hit in the form shading uses it.  Fetched for a whole wave at once the fields are scalar
// registers; the same code serves values fetched per lane.
struct Mat {
    v3 ambient, diffuse, specular;
    float shininess, refl, refr, ratio;
    uint32_t tex_wh, tex_lo, tex_hi;
};
RT_HD Mat mat_of(const u32x16 &m) {
    Mat r;
    r.ambient = V3(u2f(m[0]), u2f(m[1]), u2f(m[2])); r.diffuse = V3(u2f(m[3]), u2f(m[4]), u2f(m[5]));
    r.specular = V3(u2f(m[6]), u2f(m[7]), u2f(m[8]));
    r.shininess = u2f(m[9]); r.refl = u2f(m[10]); r.refr = u2f(m[11]); r.ratio = u2f(m[12]);
    r.tex_wh = m[13]; r.tex_lo = m[14]; r.tex_hi = m[15];
    return r;
}

// The start of PhongModel for one hit (main.cpp:646, 664-676): ambient term, diffuse colour, shininess.
template <int F, bool COUNT>
RT_HD void shade_begin(SceneRef S, const Mat &m, const SurfGeo &g, v3 d_in, ShadeCtx &sc, Work &wk) {
    sc.point = g.point; sc.normal = g.normal; sc.d_in = d_in; sc.occlusion_px = 0;
    sc.flags = g.from_outside ? kShadeOutside : 0u;
    sc.color = m.ambient;
    sc.dc = m.diffuse;
    sc.spec = m.specular;
    sc.shininess = m.shininess;
    sc.refl = m.refl; sc.refr = m.refr; sc.ratio = m.ratio;
    const int kind = (F & F_TEXTURE) ? mat_tex_kind(m.tex_lo) : -1;
    if (kind == RT_TEX_IMAGE) {
        uint32_t color = g.tex_color, occlusion = g.tex_occlusion, roughness = g.tex_roughness;
        if (!g.has_texel) {  // surfaces without a normal-mapped branch (cones, meshes)
            u4 t = tex_texel<COUNT>(mat_tex_ref(m.tex_wh, m.tex_lo, m.tex_hi), g.uv, wk);
            color = t.x; occlusion = t.z; roughness = t.w;
        }
        sc.occlusion_px = occlusion;
        sc.flags |= kShadeImage;
        sc.dc = tex_base_color(S, color) * 5.f;  // diffuse *= 5 (main.cpp:673) commutes with the NdotL factor
        // .5 / pow(roughness, 4) - .5 in double (main.cpp:674): r*r is exact in double, its square rounds once
        double r2 = (double)unpack_rgba8(roughness).x;
        r2 = r2 * r2;
        sc.shininess = (float)(.5 / (r2 * r2) - .5);
    } else if (kind >= 0) {
        sc.dc = tex_procedural(kind, g.uv);
    }
}

// While a shadow query of a hit is traced its ShadeCtx is parked in per-lane storage supplied by the Stack
// policy (LDS columns on the GPU: cold_put(slot, float) / cold_get(slot)), so that it holds no registers during
// the traversal.  Only a query that is actually made parks: most shadow rays of a lit room are answered without
// one (LightCull) and their hits never leave the registers.
RT_HD float i2f(int32_t i) { return u2f((uint32_t)i); }
template <class Stack>
RT_HD void cold_put3(Stack &stk, int slot, v3 v) { stk.cold_put(slot, v.x); stk.cold_put(slot + 1, v.y); stk.cold_put(slot + 2, v.z); }
template <class Stack>
RT_HD v3 cold_get3(Stack &stk, int slot) { return V3(stk.cold_get(slot), stk.cold_get(slot + 1), stk.cold_get(slot + 2)); }
enum : int { kColdPoint = 0, kColdNormal = 3, kColdDirIn = 6, kColdColor = 9, kColdDc = 12, kColdSpec = 15, kColdShininess = 18,
             kColdOcclusion = 19, kColdFlags = 20, kColdRefl = 21, kColdRefr = 22, kColdRatio = 23, kColdSlots = 24 };
// scenes without textures and secondary rays park neither texel nor flags nor the blend factors
constexpr int cold_slots(int features) { return (features & F_SECONDARY) ? kColdSlots : (features & F_TEXTURE) ? kColdRefl : kColdOcclusion; }
template <int F, class Stack>
RT_HD void park_shade(Stack &stk, const ShadeCtx &sc) {
    cold_put3(stk, kColdPoint, sc.point); cold_put3(stk, kColdNormal, sc.normal);
    cold_put3(stk, kColdDirIn, sc.d_in); cold_put3(stk, kColdColor, sc.color); cold_put3(stk, kColdDc, sc.dc);
    cold_put3(stk, kColdSpec, sc.spec); stk.cold_put(kColdShininess, sc.shininess);
    if (F & (F_TEXTURE | F_SECONDARY)) { stk.cold_put(kColdOcclusion, u2f(sc.occlusion_px)); stk.cold_put(kColdFlags, u2f(sc.flags)); }
    if (F & F_SECONDARY) { stk.cold_put(kColdRefl, sc.refl); stk.cold_put(kColdRefr, sc.refr); stk.cold_put(kColdRatio, sc.ratio); }
}
template <int F, class Stack>
RT_HD void unpark_shade(Stack &stk, ShadeCtx &sc) {
    sc.point = cold_get3(stk, kColdPoint); sc.normal = cold_get3(stk, kColdNormal);
    sc.d_in = cold_get3(stk, kColdDirIn); sc.color = cold_get3(stk, kColdColor); sc.dc = cold_get3(stk, kColdDc);
    sc.spec = cold_get3(stk, kColdSpec); sc.shininess = stk.cold_get(kColdShininess);
    if (F & (F_TEXTURE | F_SECONDARY)) { sc.occlusion_px = (uint32_t)f2i(stk.cold_get(kColdOcclusion)); sc.flags = (uint32_t)f2i(stk.cold_get(kColdFlags)); }
    else { sc.occlusion_px = 0; sc.flags = kShadeOutside; }  // (no image textures, and nobody asks which side was hit)
    if (F & F_SECONDARY) { sc.refl = stk.cold_get(kColdRefl); sc.refr = stk.cold_get(kColdRefr); sc.ratio = stk.cold_get(kColdRatio); }
    else { sc.refl = sc.refr = sc.ratio = 0; }
}

// One object's part of turning closest hits into shading contexts (Hit's fields, main.cpp:45-53, then the start of
// PhongModel), for the lanes that hit the object whose record (g: first half of its ObjRec, mr: its material) was
// fetched for the whole wave.
template <int F, bool COUNT>
RT_HD void shade_object(SceneRef S, const u32x16 &g, const u32x16 &mr, const Closest &c, v3 ro, v3 rd, ShadeCtx &sc, Work &wk) {
    const int32_t kind = (int32_t)g[0], index = (int32_t)g[1];
    Mat m = mat_of(mr);
    const bool image = (F & F_TEXTURE) && mat_tex_kind(m.tex_lo) == RT_TEX_IMAGE;
    const TexRef tex = mat_tex_ref(m.tex_wh, m.tex_lo, m.tex_hi);
    // every kind leaves its geometry in s (and a mesh with materials per triangle its lane's material in m); the shading
    // context is then formed at one place, not once per kind with the results merged afterwards
    SurfGeo s;
    if (kind == RT_OBJ_PLANE) {
        plane_geo<COUNT>(V3(u2f(g[4]), u2f(g[5]), u2f(g[6])), V3(u2f(g[7]), u2f(g[8]), u2f(g[9])), V3(u2f(g[10]), u2f(g[11]), u2f(g[12])),
                         V3(u2f(g[13]), u2f(g[14]), u2f(g[15])), image, tex, ro, rd, c.t, s, wk);
    } else if ((F & F_MESH) && kind == RT_OBJ_MESH) {
        int32_t orig;
        mesh_geo(S, S.meshes[index], ro, rd, c.t, c.k, s, orig);
        if ((int32_t)g[2] < 0)  // triangles with materials of their own: every lane reads its own
            m = mat_of(load_u32x16_lane(S.mat_recs + S.tri_material[orig]));
    } else if (!(F & F_QUADRIC)) {
        // The scene has no object of another kind (its feature set says so, and the host checked every object's kind).
        // Telling the compiler keeps it from carrying values "as they were" through a path nobody takes.
        RT_UNREACHABLE();
    } else if (kind == RT_OBJ_SPHERE) {
        const rt_sphere &sp = S.spheres[index];
        QuadricHit q;
        q.o = xform_point(sp.to_local, ro); q.d = xform_dir(sp.to_local, rd);
        q.t = c.t; q.cap = false; q.i = q.o + q.d * c.t;
        q.from_outside = !(fabsf(q.o.x) < 1 && fabsf(q.o.y) < 1 && fabsf(q.o.z) < 1);
        sphere_geo<COUNT>(sp, q, image, tex, s, wk);
    } else {
        const rt_cone &cn = S.cones[index];
        QuadricHit q;
        q.d = xform_dir(cn.to_local, rd); q.o = xform_point(cn.to_local, ro);
        q.t = c.t; q.cap = c.k != 0; q.from_outside = true;
        q.i = select3(q.cap, q.d * c.t + q.o, q.o + q.d * c.t);
        cone_geo<COUNT>(cn, q, image, tex, s, wk);
    }
    shade_begin<F, COUNT>(S, m, s, rd, sc, wk);
}

// From the closest hits of a wave to shading contexts.  The lanes of a wave mostly hit the same object (an 8x8
// pixel block on one wall): the wave fetches that object's record and material ONCE through scalar loads and every
// lane resolves its hit with them -- no per-lane chain of dependent loads (object -> plane -> material ->
// texture).  In a wave whose lanes hit several objects every lane reads the record of its own object (vector loads)
// and the lanes go through the object kinds side by side.  A lane without a hit gets a NaN point (its shadow rays miss
// everything); its other fields are not meaningful and nobody uses them.
template <int F, bool COUNT, class Stack>
RT_HD void surface_shade(SceneRef scene, const Closest &c, uint64_t hits, v3 ro, v3 rd, ShadeCtx &sc, Stack &stk, Work &wk) {
    // hits: the lanes with a hit, as a wave-uniform mask.  (Predicates travel as masks formed by one comparison each and
    // are combined by scalar instructions: a ballot of a combined condition costs two more vector instructions.)
    SceneRef S = fresh(scene, c.t);
    const bool hit = wave_mask_bit(hits);
    if (hits != 0) {
        const int32_t obj = wave_value_at(c.object, hits);  // (of the first lane with a hit)
        if ((wave_ballot(c.object != uniform_alias(obj)) & hits) == 0) {
            // one object: straight code for all lanes (a lane without a hit has c.t = 0, c.k = 0 and computes a value nobody uses)
            const char *rec = obj_rec_at(S, obj, c.distance);
            u32x16 g = load_u32x16(rec), mr = load_u32x16(rec + 64);
            pin_scalar(g); pin_scalar(mr);
            shade_object<F, COUNT>(S, g, mr, c, ro, rd, sc, wk);
        } else if (hit) {
            // several objects: every lane reads the record of its own (vector loads) and the lanes go through the
            // object kinds side by side.  (Taking the objects in turn with scalar records repeats a kind's arithmetic
            // for every object of that kind, and a wave on a silhouette has three or four.)
            const char *rec = reinterpret_cast<const char *>(S.obj_recs + c.object);
            u32x16 g = load_u32x16_lane(rec), mr = load_u32x16_lane(rec + 64);
            shade_object<F, COUNT>(S, g, mr, c, ro, rd, sc, wk);
        }
    }
    sc.point = select3(hit, sc.point, V3(NAN, NAN, NAN));  // (three selects: as a conditional assignment it was two branches with their mask bookkeeping)
}

// one light's contribution (main.cpp:658-685); L is the normalised direction to the light, r its distance
template <bool COUNT>
RT_HD void shade_light(ShadeCtx &sc, v3 light_color, v3 L, float r, uint32_t pow_fast, Work &wk) {
    if (COUNT) wk.shade++;
    float NdotL = clamp01(dot(sc.normal, L));
    v3 term = sc.dc * NdotL;
    // the view vector, the mirrored light direction and the lobe only where a material has a specular term, decided for
    // the wave (its lanes mostly share a material; a zero coefficient multiplies a finite lobe value: the sum is the same
    // with and without it)
    if (wave_ballot((fabsf(sc.spec.x) + fabsf(sc.spec.y)) + fabsf(sc.spec.z) != 0.0f) != 0) {
        v3 view = -sc.d_in * shade_rsq(dot(sc.d_in, sc.d_in));  // normalize(-d_in), colour only
        v3 R = reflect(-L, sc.normal);
        float VdotR = clamp01(dot(view, R));
        term = term + sc.spec * shade_pow(VdotR, sc.shininess, pow_fast);
    }
    r = (r < 0.1f) ? 0.1f : r;
    float inv_r = shade_rcp(r);
    term = light_color * term;
    sc.color = sc.color + term * (1.3f * inv_r * inv_r);
}
// the end of PhongModel (main.cpp:688-692)
template <int F = F_ALL>
RT_HD v3 shade_end(const ShadeCtx &sc) {
    v3 color = sc.color;
    if ((F & F_TEXTURE) && (sc.flags & kShadeImage)) color = color * unpack_rgb8(sc.occlusion_px);  // (no image textures without F_TEXTURE)
    else color = color * 4.0f;
    return color * 2.0f;
}

// One shadow query per light for the hits of a wave (main.cpp:650-685), all lanes in step: the light index is
// wave-uniform.  A lane without a hit (sc.point NaN) sends NaN rays, which miss everything at no cost.
//
// Direction and distance to the light: L = normalize(light - point) and |light - point| exactly as the reference
// forms them for every lane whose shadow ray is traced -- they are that ray.  A lane whose ray lies outside every
// cone of a cullable light (LightCull) traces nothing; there the two values only enter the colour arithmetic and
// come from the hardware's 1-ulp reciprocal square root.  Which form a lane gets depends on its own ray alone, so a
// pixel's value does not depend on which other rays share its wave.
template <int F, bool COUNT, class Stack>
RT_HD void shade_lights(SceneRef scene, uint64_t hits, int accel, ShadeCtx &sc, Stack &stk, Work &wk) {
    const bool hit = wave_mask_bit(hits);
    SceneRef S0 = fresh(scene, sc.point.x);
    const int32_t n_lights = S0.n_lights;
    const uint32_t pow_fast = shade_pow_fast(sc.shininess);  // (the exponent does not change in the loop: as an integer, not a lane mask)
    for (int32_t l = 0; l < n_lights; l++) {
        const char *rec = light_rec_at(fresh(scene, sc.point.y), l);
        u32x16 lr = load_u32x16(rec), lc = load_u32x16(rec + 64);  // (both halves in one batch: the second is only looked at for a cullable light)
        pin_scalar(lr);
        v3 light_color = V3(u2f(lr[4]), u2f(lr[5]), u2f(lr[6]));
        v3 Lv = V3(u2f(lr[0]), u2f(lr[1]), u2f(lr[2])) - sc.point;
        float d2 = dot(Lv, Lv);
        const float ri_approx = shade_rsq(d2);
        const v3 La = Lv * ri_approx;  // the 1-ulp direction: for the cone tests, and for the shading of lanes whose ray is not traced
        uint64_t traced_mask;  // the lanes whose shadow ray has to be traced
        if ((lr[3] & kLightCullable) != 0) {  // (wave-uniform)
            pin_scalar(lc);
            // slots are filled in order and an unused one holds cos = 2 (never inside): stop at the first
            uint64_t in = wave_ballot(!(dot(La, V3(u2f(lc[0]), u2f(lc[1]), u2f(lc[2]))) < u2f(lc[3])));
            if (lc[7] != kUnusedCone) {
                in |= wave_ballot(!(dot(La, V3(u2f(lc[4]), u2f(lc[5]), u2f(lc[6]))) < u2f(lc[7])));
                if (lc[11] != kUnusedCone) {
                    in |= wave_ballot(!(dot(La, V3(u2f(lc[8]), u2f(lc[9]), u2f(lc[10]))) < u2f(lc[11])));
                    if (lc[15] != kUnusedCone) in |= wave_ballot(!(dot(La, V3(u2f(lc[12]), u2f(lc[13]), u2f(lc[14]))) < u2f(lc[15])));
                }
            }
            traced_mask = hits & in;
        } else {
            traced_mask = hits;
        }
        wk.rays_shadow += hit ? 1u : 0u;
        wk.shadow_q += wave_mask_bit(traced_mask) ? 1u : 0u;
        uint64_t lit = hits;  // the lanes the light reaches
        v3 L;
        float r;
        // (wave-uniform; marked as the rarer case: the register copies at the join of the two cases go where the compiler
        // thinks less time is spent, and without the mark that was, after an unrelated change, the common case: 22 per light)
        if (__builtin_expect(traced_mask != 0, 0)) {
            {
                const float len = sqrtf(d2), inv_len = 1.0f / len;
                v3 Le = Lv * inv_len;  // glm::normalize(light - point)
                if (!wave_mask_bit(traced_mask)) Le = V3(NAN, NAN, NAN);
                park_shade<F>(stk, sc);
                Closest cs;
                // |L| = 1, so the local ray parameter equals the world distance up to rounding: clip there
                find_closest<F, COUNT>(scene, sc.point + Le * kEps, Le, accel, l, len * 1.001f + 1e-3f, len, cs, stk, wk);
                unpark_shade<F>(stk, sc);
                // (everything that follows is formed again from the parked point and the light's record: nothing but
                // the parked values lives across the query)
                u32x16 lr2 = load_u32x16(light_rec_at(fresh(scene, sc.point.y), l));
                pin_scalar(lr2);
                light_color = V3(u2f(lr2[4]), u2f(lr2[5]), u2f(lr2[6]));
                v3 Lv2 = V3(u2f(lr2[0]), u2f(lr2[1]), u2f(lr2[2])) - sc.point;
                const float len2 = len;  // (the length and its reciprocal are the two values held through the query)
                const bool mine = wave_mask_bit(traced_mask);
                lit = hits & ~(traced_mask & wave_ballot(cs.distance < len2));
                L = Lv2 * inv_len;
                r = len2;
                if ((lr2[3] & kLightCullable) != 0) {  // the lanes whose ray was not traced: the 1-ulp forms, as in a wave that traces nothing
                    float d2b = dot(Lv2, Lv2);
                    float ri = shade_rsq(d2b);
                    L = select3(mine, L, Lv2 * ri);
                    r = mine ? r : d2b * ri;
                }
            }
        } else {
            L = La;
            r = d2 * ri_approx;
        }
        if (wave_mask_bit(lit)) shade_light<COUNT>(sc, light_color, L, r, pow_fast, wk);
    }
}

// ---------------------------------------------------------------- trace loop

// trace_ray (main.cpp:716-761) for a stream of primary rays, without recursion: an outer loop takes
// one radiance ray (primary, reflection or refraction) per lane and iteration; after its closest-hit
// query all lanes run PhongModel's light loop in step (shade_lights: one shadow query per light, the
// light index wave-uniform); then the lane continues with a child ray, a pending ray or its next
// primary ray.  The reference's result is linear in the radiance of the child rays, so each pending
// ray carries the product of the blend factors on its path:
//   local shading      weight * (1 - refl) * (ratio if refractive else 1)
//   refraction child   weight * (1 - refl) * (1 - ratio)
//   reflection child   weight * refl
// Rays deeper than max_depth are dropped (the reference recurses without bound).
//
// Source supplies primary rays and receives their radiance:
//   bool next(v3 &o, v3 &d);  void done(v3 radiance);  rt_hit_record *first_hit();
struct Pending { v3 o, d; float w; int32_t depth; };

// The same for scenes without reflective or refractive materials, where every primary ray ends in
// direct lighting: primary query, then one shadow query per light, as plain nested loops.  The lanes
// of a wave stay in step (they have the same number of taps and lights), so the light index is a
// wave-uniform value and neither a ray stage nor a pending stack has to be carried.
template <int F, bool COUNT, class Stack, class Source>
RT_HD void trace_loop_direct(SceneRef S, int accel, Stack &stk, Work &wk, Source &src) {
    for (;;) {
        v3 o, d;
        if (!src.more()) break;            // (wave-uniform: the lanes of a wave have the same number of rays)
        const bool live = src.next(o, d);  // (all lanes call next() alike, in step; a source may hand a lane no ray while others get one)
        Closest c;
        find_closest<F, COUNT>(S, o, d, accel, -1, INFINITY, -INFINITY, c, stk, wk);
        wk.rays_primary += live ? 1u : 0u;
        rt_hit_record *fh = src.first_hit();
        if (fh && live) hit_record<F, COUNT>(S, c, o, d, *fh, wk);
        c.object = live ? c.object : -1;   // (a lane without a ray has no hit: from here on "hit" is one comparison)
        const uint64_t hits = wave_ballot(c.object >= 0);
        ShadeCtx sc;
        surface_shade<F, COUNT>(S, c, hits, o, d, sc, stk, wk);
        shade_lights<F, COUNT>(S, hits, accel, sc, stk, wk);
        v3 radiance = V3(0, 0, 0);
        if (wave_mask_bit(hits)) radiance = shade_end<F>(sc);
        if (live) src.done(radiance);
    }
}

template <int F, bool COUNT, class Stack, class Source>
RT_HD void trace_loop(SceneRef S, int max_depth, int accel, Stack &stk, Work &wk, Source &src) {
    if (!(F & F_SECONDARY)) {
        trace_loop_direct<F, COUNT>(S, accel, stk, wk, src);
        return;
    }
    Pending pend[RT_MAX_DEPTH + 1];
    int np = 0;
    v3 o, d;
    if (!src.next(o, d)) return;
    v3 radiance = V3(0, 0, 0);
    float weight = 1.0f;
    int depth = 0;
    bool primary = true;
    for (;;) {
        // one radiance ray (primary, reflection or refraction) of every lane
        Closest c;
        find_closest<F, COUNT>(S, o, d, accel, -1, INFINITY, -INFINITY, c, stk, wk);
        if (primary) wk.rays_primary++; else wk.rays_secondary++;
        bool hit = c.object >= 0;
        if (primary) {
            rt_hit_record *fh = src.first_hit();
            if (fh) hit_record<F, COUNT>(S, c, o, d, *fh, wk);
        }
        primary = false;
        ShadeCtx sc;
        const uint64_t hits = wave_ballot(c.object >= 0);
        surface_shade<F, COUNT>(S, c, hits, o, d, sc, stk, wk);
        shade_lights<F, COUNT>(S, hits, accel, sc, stk, wk);
        if (hit) {
            v3 local = shade_end<F>(sc);
            float refl = sc.refl, refr = sc.refr;
            float ratio = sc.ratio;
            const bool from_outside = (sc.flags & kShadeOutside) != 0;
            float w_after_refl = refl > 0 ? weight * (1 - refl) : weight;
            float w_local = refr > 0 ? w_after_refl * ratio : w_after_refl;
            radiance = radiance + local * w_local;
            int cd = depth + 1;
            bool deeper = cd <= max_depth, have_next = false;
            if (refr > 0 && deeper) {
                float index = from_outside ? refr : 1 / refr;
                v3 rdir = refract_dir(sc.d_in, sc.normal, index);
                o = sc.point + rdir * kEps; d = rdir;
                have_next = true;
            }
            if (refl > 0 && from_outside && deeper) {
                v3 rdir = reflect(sc.d_in, sc.normal);
                v3 ro2 = sc.point + rdir * kEps;
                float rw = weight * refl;
                if (have_next) {
                    if (np <= RT_MAX_DEPTH) { pend[np].o = ro2; pend[np].d = rdir; pend[np].w = rw; pend[np].depth = cd; np++; }
                } else {
                    o = ro2; d = rdir; weight = rw; depth = cd;
                    continue;
                }
            }
            if (have_next) { weight = w_after_refl * (1.f - ratio); depth = cd; continue; }
        }
        // this ray's subtree is finished: resume a pending ray, else hand the radiance over
        if (np > 0) {
            np--;
            o = pend[np].o; d = pend[np].d; weight = pend[np].w; depth = pend[np].depth;
            continue;
        }
        src.done(radiance);
        if (!src.next(o, d)) break;
        radiance = V3(0, 0, 0); weight = 1.0f; depth = 0; primary = true;
    }
}

// a single ray (the arbitrary-ray entry points)
struct OneRay {
    v3 o, d, radiance;
    rt_hit_record *hit;
    bool taken;
    RT_HD bool more() const { return !taken; }
    RT_HD bool next(v3 &ro, v3 &rd) { if (taken) return false; taken = true; ro = o; rd = d; return true; }
    RT_HD void done(v3 r) { radiance = r; }
    RT_HD rt_hit_record *first_hit() { return hit; }
};
template <int F, bool COUNT, class Stack>
RT_HD v3 trace(SceneRef S, v3 ro, v3 rd, int max_depth, int accel, Stack &stk, Work &wk, rt_hit_record *first_hit) {
    OneRay src;
    src.o = ro; src.d = rd; src.radiance = V3(0, 0, 0); src.hit = first_hit; src.taken = false;
    trace_loop<F, COUNT>(S, max_depth, accel, stk, wk, src);
    return src.radiance;
}

// ---------------------------------------------------------------- camera, tone mapping

struct Camera { float s, X, Y; };
// main.cpp:1231-1233
RT_HD Camera make_camera(int w, int h, float fov) {
    Camera c;
    c.s = (float)(2 * tan(0.5 * (double)fov / 180 * kPi) / w);
    c.X = -c.s * w / 2;
    c.Y = c.s * h / 2;
    return c;
}
RT_HD int taps_of(int mode, int m) { return mode == RT_MODE_REF_1SPP ? 1 : mode == RT_MODE_REF_AA9 ? 9 : m * m; }
// main.cpp:1241-1266; GRID: corner as in the AA path, taps at ((a+.5)/m*s, -(b+.5)/m*s)
// weight of tap k in the pixel's sum (main.cpp:1249-1262: 1 2 1 / 2 4 2 / 1 2 1 for the 9-tap mode)
RT_HD float tap_weight(int mode, int k) {
    if (mode != RT_MODE_REF_AA9) return 1.0f;
    int kx = k % 3, ky = k / 3;
    return (float)((kx == 1 ? 2 : 1) * (ky == 1 ? 2 : 1));
}
// offset of column / row a of the m x m GRID taps from the pixel's corner
RT_HD float grid_offset(int a, int m, float s) { return ((float)a + 0.5f) / (float)m * s; }
// dx, dy: the pixel's corner X + i*s, Y - j*s (main.cpp:1244-1245; the one-tap mode adds its half pixel to the same sums, main.cpp:1263-1264)
RT_HD float pixel_dx(const Camera &c, int i) { return c.X + i * c.s; }
RT_HD float pixel_dy(const Camera &c, int j) { return c.Y - j * c.s; }
// The direction of tap k (main.cpp:1241-1266); GRID: tap (col, row) of the m x m taps, its offsets from a table
// (grid_off[a] = grid_offset(a, m, s): no integer and no float division per tap), col and row advanced here.  One chain of wave-uniform cases that leave the unnormalised direction, then one normalize for the modes that
// have one: as nested calls with a normalize each, every tap went through a dozen scalar instructions of dispatch.
RT_HD v3 tap_dir(const Camera &c, int mode, int m, float dx, float dy, int k, int &col, int &row, const float *grid_off, float &weight) {
    float tx, ty;
    weight = 1.0f;
    if (mode == RT_MODE_GRID) {
        tx = dx + grid_off[col];
        ty = dy - grid_off[row];
        if (++col == m) { col = 0; row++; }
    } else if (mode == RT_MODE_REF_AA9) {
        int kx = k % 3, ky = k / 3;
        tx = dx + kx * c.s / 2;
        ty = dy - ky * c.s / 2;
        weight = tap_weight(mode, k);
    } else {
        return V3(dx + c.s / 2, dy + c.s / 2, 1);
    }
    return normalize(V3(tx, ty, 1));
}
constexpr int kMaxGrid = 64;  // largest m of the GRID mode
RT_HD void fill_grid_offsets(float *grid_off, int m, float s) {
    for (int a = 0; a < kMaxGrid; a++) grid_off[a] = a < m ? grid_offset(a, m, s) : 0.0f;
}
// toneMapping, main.cpp:1184-1189.  The result only feeds the 8-bit quantisation: on the GPU pow is exp2(g * log2(x))
// with the hardware's 1-ulp functions (a few 1e-7 relative, i.e. 1 LSB off on about one pixel in ten thousand,
// against some 250 instructions per call of the library's powf); 0 -> 0, negative and NaN -> NaN, inf -> inf as powf.
RT_HD float tone_pow(float x, float g) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_exp2f(g * __builtin_amdgcn_logf(x));
#else
    return powf(x, g);
#endif
}
RT_HD v3 tone_map(v3 c) {
    float gamma = (float)(1.0 / 1.8);
    return V3(clamp01(18.0f * tone_pow(c.x, gamma)), clamp01(18.0f * tone_pow(c.y, gamma)), clamp01(18.0f * tone_pow(c.z, gamma)));
}
// Image::setPixel(vec3), Image.cpp:35-39: int = (float)(255 * c), truncation
RT_HD uint8_t quantise(float c) { return (uint8_t)(int)(float)(255 * c); }

// One pixel: all taps in order, weighted sum, normalisation (main.cpp:1243-1270).  The lanes of a wave walk their
// taps in step: every lane calls next() alike, so the tap counters are wave-uniform (scalar registers), and a lane
// whose pixel lies outside the frame (`inside` false) simply gets no ray.
// PARKED: the pixel's corner and its running sum live in the Stack policy's per-lane storage (LDS columns on the
// GPU) instead of registers: five values that are touched once per tap and would otherwise be held, or spilled to
// scratch memory, through every traversal.
constexpr int kTapSlots = 5;  // dx, dy, colour: after the cold_slots() columns of the kernel's feature set
template <class Stack, bool PARKED, int BASE>
struct PixelTaps {
    enum : int { kColdTapDx = BASE, kColdTapDy = BASE + 1, kColdTapColor = BASE + 2 };
    Camera cam;
    int mode, grid_m, taps, k;
    int col, row;           // GRID: k = row * grid_m + col
    const float *grid_off;  // GRID: grid_offset(a, grid_m, cam.s) for a < grid_m
    Stack *stk;
    float dx, dy;           // the pixel's corner (not PARKED)
    bool inside;
    float wt;
    v3 color;               // (not PARKED)
    RT_HD void start(float dx0, float dy0) {
        if (PARKED) { stk->cold_put(kColdTapDx, dx0); stk->cold_put(kColdTapDy, dy0); cold_put3(*stk, kColdTapColor, V3(0, 0, 0)); }
        else { dx = dx0; dy = dy0; color = V3(0, 0, 0); }
    }
    RT_HD bool more() const { return k < taps; }
    RT_HD bool next(v3 &ro, v3 &rd) {
        if (k >= taps) return false;
        ro = V3(0, 0, 0);
        const float x = PARKED ? stk->cold_get(kColdTapDx) : dx, y = PARKED ? stk->cold_get(kColdTapDy) : dy;
        rd = tap_dir(cam, mode, grid_m, x, y, k, col, row, grid_off, wt);
        k++;
        return inside;
    }
    RT_HD void done(v3 r) {
        if (PARKED) {
            v3 c = cold_get3(*stk, kColdTapColor);
            cold_put3(*stk, kColdTapColor, mode == RT_MODE_REF_1SPP ? r : c + r * wt);
        } else {
            color = mode == RT_MODE_REF_1SPP ? r : color + r * wt;
        }
    }
    RT_HD rt_hit_record *first_hit() { return nullptr; }
    RT_HD v3 result() {
        v3 c = PARKED ? cold_get3(*stk, kColdTapColor) : color;
        return mode == RT_MODE_REF_1SPP ? c : c / (mode == RT_MODE_REF_AA9 ? 16.f : (float)taps);
    }
};
// inside: false for a lane without a pixel (it takes part in the wave's loop and traces nothing)
template <int F, bool COUNT, bool PARKED, class Stack>
RT_HD v3 render_pixel(SceneRef S, const Camera &cam, int mode, int grid_m, int max_depth, int accel,
                      int i, int j, bool inside, const float *grid_off, Stack &stk, Work &wk) {
    PixelTaps<Stack, PARKED, cold_slots(F)> src;
    src.cam = cam; src.mode = mode; src.grid_m = grid_m; src.taps = taps_of(mode, grid_m);
    src.stk = &stk; src.inside = inside;
    src.k = 0; src.col = 0; src.row = 0; src.grid_off = grid_off; src.wt = 1.0f;
    src.start(pixel_dx(cam, i), pixel_dy(cam, j));
    trace_loop<F, COUNT>(S, max_depth, accel, stk, wk, src);
    return src.result();
}

}  // namespace rtc
