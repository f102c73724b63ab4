// Flattened scene: the POD arrays the kernels (and the CPU checker oracle/o2)
// walk.  Produced on the host by host/flatten.cpp from the reference-shaped
// object graph, uploaded once to HBM, read-only during a render.
//
// Layout rules: every struct is a multiple of 8 bytes, little-endian, no
// pointers -- cross references are 32-bit indices into sibling arrays, so the
// same bytes are valid in host memory, HBM and LDS.
#pragma once
#include "rt_config.hpp"
#include "vec3.hpp"

namespace rt {

// ---- primitives -----------------------------------------------------------
enum PrimType : uint32_t {
  PRIM_SPHERE = 0,         // hit.rs:180-245
  PRIM_MOVING_SPHERE = 1,  // hit.rs:247-328
  PRIM_RECT = 2,           // hit.rs:446-639 (axis selects Xy/Xz/Yz)
  PRIM_TRIANGLE = 3,       // hit.rs:87-178
  PRIM_GRAVITY_SPHERE = 4, // hit.rs:330-444 (the bouncing ball of the video scene)
};

// A primitive reference: type in the top 3 bits, index into that type's array below.
typedef uint32_t PrimRef;
RT_HD PrimRef make_primref(uint32_t type, uint32_t index) { return (type << 29) | index; }
RT_HD uint32_t primref_type(PrimRef r) { return r >> 29; }
RT_HD uint32_t primref_index(PrimRef r) { return r & 0x1fffffffu; }

struct FlatSphere {  // 40 B
  real cx, cy, cz, radius;
  int32_t mat;
  int32_t pad;
};
struct FlatMovingSphere {  // 80 B
  real c0[3], c1[3];
  real time0, time1, radius;
  int32_t mat;
  int32_t pad;
};
// hit.rs:330-336.  The ball's height lives in a table of its own (GravitySphere::new simulates 100 time units in steps of
// 0.001 and stores every height: ~100 002 doubles per ball); times past the table fall back to the reference's second,
// slightly different simulation loop (hit.rs:378-389).
struct FlatGravitySphere {  // 64 B
  real sx, sy, sz;   // start
  real time0, radius;
  int32_t mat;
  int32_t pad;
  int64_t table_first; // index into SceneView::gravity_y
  int64_t table_len;
};
enum RectAxis : int32_t { RECT_XY = 0, RECT_XZ = 1, RECT_YZ = 2 };
struct FlatRect {  // 48 B.  (a0,a1,b0,b1,k) are the constructor's (x0,x1,y0,y1,k).
  real a0, a1, b0, b1, k;
  int32_t axis;
  int32_t mat;
};
struct FlatTriangle {  // 104 B
  real v0[3], v1[3], v2[3], normal[3];  // normal = unit((v1-v0) x (v2-v0)), hit.rs:96-107
  int32_t mat;
  int32_t pad;
};

// ---- BVH --------------------------------------------------------------------
// Binary BVH; a node stores the boxes of its two children.  child[c] >= 0 is an
// internal node index; child[c] < 0 encodes a leaf: bits 0..2 = count-1,
// bits 3..30 = first slot in the BVH's primitive-reference list (relative to
// FlatBvh.first_ref).  Leaves hold 1..8 primitives.
struct FlatNode {  // 112 B
  real bmin[2][3];
  real bmax[2][3];
  int32_t child[2];
  int32_t pad[2];
};
// The same tree with child boxes rounded OUTWARD to f32 (64 B per node): the culling structure of
// the fast kernels.  Node i of `nodes32` mirrors node i of `nodes`.  See core/cull32.hpp for why
// culling in f32 cannot change a result.
struct FlatNode32 {  // 64 B
  float lo[2][3];
  float hi[2][3];
  int32_t child[2];
  int32_t axis;
  int32_t pad;
};
// Time-aware culling boxes, node i mirroring node i of `nodes` (present when a BVH of the scene holds MovingSpheres, whose
// centre is LINEAR in time, hit.rs:275-278).  With s = (ray.time - time0) / (time1 - time0) of the BVH's own interval
// (BvhNode::from_list(list, time0, time1), bvh.rs:85-93; FlatEntry::f[0], f[1]), child c's box at that instant is contained in
//     [lo0 + s dlo, hi0 + s dhi]    for every s in [0, 1]
// evaluated in f32 with one fma per plane: the end boxes are the unions of the subtree's primitive boxes AT time0 and AT time1
// (the lerp of two unions contains the union of the lerps: min of linear functions is concave), pushed outward by 2^-21 of the
// axis' largest coordinate, which covers the rounding of s, of the differences and of the fma.  The reference's own boxes are
// the unions OVER the interval (hit.rs:317-327): a sphere that rises 5 units during the shutter (Book-1 at HEAD) has a box 27
// times its size there, and every ray tests it whatever its own time.  A BVH is a culling structure: the image cannot change.
struct FlatMotion32 {  // 96 B
  float lo0[2][3], hi0[2][3];
  float dlo[2][3], dhi[2][3];
};
RT_HD bool node_child_is_leaf(int32_t c) { return c < 0; }
RT_HD uint32_t leaf_first(int32_t c) { return ((uint32_t)c & 0x7fffffffu) >> 3; }
RT_HD uint32_t leaf_count(int32_t c) { return ((uint32_t)c & 7u) + 1u; }
RT_HD int32_t make_leaf(uint32_t first, uint32_t count) {
  return (int32_t)(0x80000000u | (first << 3) | (count - 1u));
}

// ---- entries: what a HittableList slot can hold ----------------------------------
enum EntryKind : int32_t {
  ENTRY_PRIM = 0,    // a = PrimRef
  ENTRY_GROUP = 1,   // ordered list semantics (HittableList / RectPrism): a = first ref, b = count
  ENTRY_BVH = 2,     // a = root node, b = first ref of its prim-ref list, c = ref count; f[0], f[1] = time0, time1 of from_list
  ENTRY_XFORM = 3,   // a = child entry (PRIM/GROUP/BVH), b = number of ops (outermost first, <= RT_MAX_XFORM_OPS)
  ENTRY_MEDIUM = 4,  // a = boundary entry (PRIM/GROUP/BVH/XFORM), b = phase material; f[0] = -1/density
};
enum XformOp : int32_t { XFORM_TRANSLATE = 0, XFORM_ROTATE_Y = 1 };
#define RT_MAX_XFORM_OPS 4  // Translate / RotateY wrappers around one object (hit.rs:787-936); deeper chains: RTX_EUNSUPPORTED
struct FlatXformOp {
  int32_t op;
  int32_t pad;
  real v[3];  // translate: offset; rotate_y: v[0] = sin_theta, v[1] = cos_theta
};
struct FlatEntry {
  int32_t kind;
  int32_t a, b, c;
  real f[2];
  FlatXformOp ops[RT_MAX_XFORM_OPS];
};

// ---- shading --------------------------------------------------------------------
enum MaterialKind : int32_t {
  MAT_LAMBERTIAN = 0,     // hit.rs:1020-1052
  MAT_METAL = 1,          // hit.rs:1054-1084
  MAT_DIELECTRIC = 2,     // hit.rs:1086-1127
  MAT_DIFFUSE_LIGHT = 3,  // hit.rs:1129-1152
  MAT_ISOTROPIC = 4,      // hit.rs:992-1011
};
struct FlatMaterial {  // 48 B
  int32_t kind;
  int32_t tex;       // albedo / emit texture (Lambertian, DiffuseLight, Isotropic)
  real albedo[3];  // Metal; for the texture-carrying kinds the colour of `tex` when that is a SolidColor (resolved by the flattener); Dielectric: 1 / ir and the two r0^2 of reflectance (shading.hpp: dielectric_constants)
  real param;      // Metal: fuzz (already clamped to <= 1); Dielectric: ir
  int32_t needs_uv;  // 1 if the texture tree below `tex` contains an Image texture
  int32_t pad;
};
enum TextureKind : int32_t {
  TEX_SOLID = 0,    // texture.rs:11-31
  TEX_CHECKER = 1,  // texture.rs:33-64   a = even, b = odd
  TEX_NOISE = 2,    // texture.rs:66-88   a = perlin table, scale
  TEX_IMAGE = 3,    // texture.rs:90-122  a = image
};
struct FlatTexture {  // 48 B
  int32_t kind;
  int32_t a, b;
  int32_t pad;
  real color[3];
  real scale;
};
struct FlatPerlin {  // 9216 B, perlin.rs:6-11
  real ranvec[256][3];
  int32_t perm_x[256], perm_y[256], perm_z[256];
};
struct FlatImage {  // texels are f64 triples exactly as Screen::from_ppm_p3 parses them
  int32_t width, height;
  int64_t first_texel;  // index (in texels) into the texel array
};

// camera.rs:6-17
struct FlatCamera {  // 192 B
  Vec3 origin, lower_left_corner, horizontal, vertical, u, v, w;
  real lens_radius, time1, time2;
};

// ---- a view over one flattened scene (pointers into host memory, HBM or LDS) ---
struct SceneView {
  const FlatSphere* spheres;
  const FlatMovingSphere* moving_spheres;
  const FlatRect* rects;
  const FlatTriangle* triangles;
  const FlatNode* nodes;
  const FlatNode32* nodes32;
  const FlatMotion32* motion32;  // null unless a BVH holds MovingSpheres
  const PrimRef* refs;
  const FlatEntry* entries;
  const int32_t* top_level;  // entry indices, in HittableList order
  const FlatMaterial* materials;
  const FlatTexture* textures;
  const FlatPerlin* perlins;
  const FlatImage* images;
  const real* texels;  // 3 doubles per texel
  // f32 bounding box (lo xyz, hi xyz; rounded outward) of every top-level slot that is a plain static primitive,
  // (-inf, +inf) for every other slot: lets the list scan skip a primitive test no lane of the wave can pass
  // (core/cull32.hpp: conservative, cannot change a result).  May be null.
  const float* top_box32;
  const FlatGravitySphere* gravity_spheres;
  const real* gravity_y;  // the height tables of all gravity spheres, one after another
  int32_t n_top_level;
  int32_t max_stack;  // deepest traversal stack any BVH of this scene needs
  uint32_t features;  // Feature bits the scene can reach
  uint32_t pad;
};

// Per-render constants.
struct RenderParams {
  FlatCamera cam;
  Color background;
  int32_t image_width, image_height;
  int32_t samples_per_pixel, max_depth;
  uint64_t seed;
};

// Feature bits: which code paths a scene can reach.  The flattener computes the mask of a
// scene; the kernels are compiled for a few preset masks so that, e.g., a sphere-only scene
// does not carry triangle / medium / Perlin code (and registers).  A path is only ever
// compiled OUT when the scene cannot reach it, so results do not depend on the preset.
enum Feature : uint32_t {
  F_SPHERE = 1u << 0, F_MOVING_SPHERE = 1u << 1, F_RECT = 1u << 2, F_TRIANGLE = 1u << 3,
  F_PRIM_ENTRY = 1u << 4, F_GROUP = 1u << 5, F_BVH = 1u << 6, F_XFORM = 1u << 7, F_MEDIUM = 1u << 8,
  F_LAMBERTIAN = 1u << 9, F_METAL = 1u << 10, F_DIELECTRIC = 1u << 11, F_LIGHT = 1u << 12,
  F_ISOTROPIC = 1u << 13,
  F_CHECKER = 1u << 14, F_NOISE = 1u << 15, F_IMAGE = 1u << 16,
  F_GRAVITY_SPHERE = 1u << 17,
  F_MEDIUM_GENERAL = 1u << 18,  // a medium whose boundary is anything but one plain static sphere (a box, a BVH, a moved object)
  F_MEDIUM_SPHERE = 1u << 19,   // a medium whose boundary is one plain static sphere (Book-2's smoke ball and fog)
  F_ALL = (1u << 20) - 1u,
};

// Work counters for the algorithmic-bytes model.
struct TraceCounters {
  unsigned long long box_tests;      // reference-equivalent "node visits" (one Aabb::hit each)
  unsigned long long sphere_tests, moving_sphere_tests, rect_tests, triangle_tests;  // gravity spheres count as moving spheres
  unsigned long long scatters;       // material evaluations
  unsigned long long texels;         // image texel fetches
  unsigned long long perlin_calls;   // Perlin::noise calls (8 gradient fetches each)
  unsigned long long rays;           // world.hit calls (bounces)
  unsigned long long samples;
};

}  // namespace rt
