// Ray/primitive intersection, AABB slab test, BVH traversal and the ordered
// top-level object table -- the L1 half of the hot path.
//
// Reference: /root/reference/src/hit.rs (Hittable impls), aabb.rs:23-61,
// bvh.rs:97-112.  Every primitive test performs the reference's arithmetic in
// the reference's order; what is re-designed is the control structure around it:
//   * intersection is split into "find t" (candidate distance) and "finalize"
//     (hit point, normal, face, uv), so a traversal only carries (t, primitive)
//     and builds ONE HitRecord per ray instead of one per accepted candidate;
//   * the recursive, pointer-chasing BvhNode::hit becomes an iterative walk over
//     a flat node array with an explicit index stack (LDS on the device);
//   * HittableList / RectPrism become ordered groups, Translate/RotateY become
//     op lists on an entry, ConstantMedium becomes an entry kind.
// Closest-hit results are independent of the traversal order; exact ties inside
// one BVH are resolved as "higher slot in the BVH's flattened primitive list
// wins", an order-independent stand-in for the reference's "right child wins"
// (bvh.rs:105), whose outcome depends on its random split axes.  Ordered lists
// keep the reference's "later object wins" exactly (hit.rs:676-680).
#pragma once
#include "flat_types.hpp"
#include "cull32.hpp"
#include "rng.hpp"
#include "rt_math.hpp"

namespace rt {

// hit.rs:10-18
struct HitRecord {
  Point3 p;
  Vec3 normal;
  real t, u, v;
  bool front_face;
  int32_t mat;
};

// hit.rs:69-79
RT_HD void create_normal_face(const Ray& r, Vec3 outward_normal, Vec3* normal, bool* front_face) {
  bool ff = dot(r.direction, outward_normal) < real(0.0);
  *normal = ff ? outward_normal : -outward_normal;
  *front_face = ff;
}

// hit.rs:195-200.  Out of line: only Image-textured spheres use uv, and acos/atan2 are long.
RT_HD void get_sphere_uv(Point3 p, real* u, real* v) {
  real theta = rt_acos(-p.y);
  real phi = rt_atan2(-p.z, p.x) + RT_PI;
  *u = phi / (real(2.0) * RT_PI);
  *v = theta / RT_PI;
}

RT_HD Vec3 load_v3(const real* p) { return v3(p[0], p[1], p[2]); }

// hit.rs:275-278
RT_HD Point3 moving_sphere_center(const FlatMovingSphere& s, real time) {
  Point3 c0 = load_v3(s.c0), c1 = load_v3(s.c1);
  return c0 + ((time - s.time0) / (s.time1 - s.time0)) * (c1 - c0);
}

// GravitySphere::get_center (hit.rs:369-391), literally: the table when `(time / incr) as usize + 1 <= stored.len()`
// (a saturating cast: negative or NaN times index entry 0), otherwise the brute-force loop with its own constants
// (2 x radius, restitution 0.8 -- the reference's "radius x2 bug" comment).
RT_HD Point3 gravity_sphere_center(const FlatGravitySphere& s, const real* table, real time) {
  const real incr = real(0.001);
  const real q = time / incr;
  uint64_t idx;  // Rust `as usize`: saturating, NaN -> 0
  if (!(q == q) || q <= real(0.0)) idx = 0;
  else if (q >= real(18446744073709551615.0)) idx = ~0ull;
  else idx = (uint64_t)q;
  if (idx < ~0ull && idx + 1 <= (uint64_t)s.table_len) return v3(s.sx, table[s.table_first + (int64_t)idx], s.sz);
  // The clock of the loop is kept in double in BOTH compilations: `t += 0.001f` stops advancing at t >= 32768 (the increment is
  // below half an ulp there) and the loop would never end; in the f64 build this is the reference's arithmetic unchanged.
  // Renders reject shutter times more than 10 s past the stored table (render.hip: gravity_time_limit), which bounds the trip count.
  double t = (double)s.time0;
  real y = s.sy, vel = real(0.0);
  while (t < (double)time) {
    t += 0.001;
    vel -= real(0.000001);
    if (y - real(2.0) * s.radius <= real(0.0)) vel *= -real(0.8);
    y = rt_fmax(real(2.0) * s.radius, y + vel);
  }
  return v3(s.sx, y, s.sz);
}

// The self-intersection guard of ray_color: world.hit(r, 0.001, infinity) (world.rs:77).  In double precision this is the
// reference's constant.  A single-precision hit point is only known to about |p| 2^-23 per coordinate, so a ray that leaves
// a surface at the shallow angle theta finds the same surface again at t ~ |p| 2^-23 / cos(theta), and once that exceeds
// 0.001 the path dies inside the object it just left (measured on Book-2, coordinates of several hundred: 1 % of the
// light lost per sphere bounce).  The fast mode widens the guard with the magnitude of the ray's origin: 512 ulps (objects
// under a rotation and a translation lose a few more bits on the way in and out), which leaves only rays shallower than
// about 1 : 100 exposed.
RT_HD real ray_t_min(const Ray& r) {
#if defined(RT_F32)
  return real(0.001) + real(0x1.0p-14) * rt_fmax(rt_fabs(r.origin.x), rt_fmax(rt_fabs(r.origin.y), rt_fabs(r.origin.z)));
#else
  (void)r;
  return real(0.001);
#endif
}

// Shared by Sphere::hit (hit.rs:204-222) and MovingSphere::hit (hit.rs:282-300).
RT_HD bool sphere_root(Point3 center, real radius, const Ray& r, real t_min, real t_max,
                       real* t_out) {
  Vec3 oc = r.origin - center;
  real a = length_squared(r.direction);
  real half_b = dot(oc, r.direction);
  real c = length_squared(oc) - radius * radius;
  real discriminant = half_b * half_b - a * c;
  if (discriminant < real(0.0)) return false;
#if defined(RT_F32)
  // In single precision c carries an absolute error of a few ulps of |oc|^2 + r^2, so a root is only known to about
  // that error / |half_b|: a ray leaving the surface of a big sphere at a shallow angle (the ground of Book-1: r = 1000)
  // would find the sphere again at t ~ 1e-3 -- beyond the 0.001 guard -- and die inside it (measured: every sphere scene
  // 0.2 - 1.2 % darker).  The fast mode raises the guard by that uncertainty.
  t_min += real(0x1.0p-22) * (length_squared(oc) + radius * radius) / rt_fabs(half_b);
#endif
  real sqrtd = rt_sqrt(discriminant);
  real root = (-half_b - sqrtd) / a;
  if (root < t_min || t_max < root) {
    root = (-half_b + sqrtd) / a;
    if (root < t_min || t_max < root) return false;
  }
  *t_out = root;
  return true;
}

// sphere_root in two halves, for callers that ask the same sphere about the same ray more than once (a ConstantMedium asks
// its boundary twice, hit.rs:961-967, and Book-2 lists each boundary sphere a third time as a glass ball of its own): the
// quadratic is solved once and every query picks from the two roots exactly as sphere_root would -- same expressions on
// the same operands, so the same bits.
// *guard: what sphere_root adds to a query's t_min (0 in double precision; the fast mode's root uncertainty).
RT_HD bool sphere_roots(Point3 center, real radius, const Ray& r, real* root1, real* root2, real* guard) {
  Vec3 oc = r.origin - center;
  real a = length_squared(r.direction);
  real half_b = dot(oc, r.direction);
  real c = length_squared(oc) - radius * radius;
  real discriminant = half_b * half_b - a * c;
  if (discriminant < real(0.0)) return false;
#if defined(RT_F32)
  *guard = real(0x1.0p-22) * (length_squared(oc) + radius * radius) / rt_fabs(half_b);
#else
  *guard = real(0.0);
#endif
  real sqrtd = rt_sqrt(discriminant);
  *root1 = (-half_b - sqrtd) / a;
  *root2 = (-half_b + sqrtd) / a;
  return true;
}
RT_HD bool sphere_pick(real root1, real root2, real t_min, real t_max, real* t_out) {
  real root = root1;
  if (root < t_min || t_max < root) {
    root = root2;
    if (root < t_min || t_max < root) return false;
  }
  *t_out = root;
  return true;
}

// hit.rs:111-149: plane hit, range test, three inside-edge tests.
RT_HD bool triangle_t(const FlatTriangle& tr, const Ray& r, real t_min, real t_max,
                      real* t_out) {
  Vec3 n = load_v3(tr.normal), v0 = load_v3(tr.v0), v1 = load_v3(tr.v1), v2 = load_v3(tr.v2);
  real n_dot_d = dot(n, r.direction);
  if (rt_fabs(n_dot_d) < real(0.0001)) return false;
  real d = -dot(n, v0);
  real t = -(dot(n, r.origin) + d) / n_dot_d;
  if (t < t_min || t > t_max) return false;
  Point3 p = ray_at(r, t);
  Vec3 c = cross(v1 - v0, p - v0);
  if (dot(n, c) < real(0.0)) return false;
  c = cross(v2 - v1, p - v1);
  if (dot(n, c) < real(0.0)) return false;
  c = cross(v0 - v2, p - v2);
  if (dot(n, c) < real(0.0)) return false;
  *t_out = t;
  return true;
}

// hit.rs:476-485 (Xy), 541-550 (Xz), 606-615 (Yz).  One body per axis, each naming the ray
// components it reads (a runtime-selected member turns into an indexed stack load on the GPU).
RT_HD bool rect_core(const FlatRect& q, real ok, real dk, real oa, real da, real ob, real db,
                     real t_min, real t_max, real* t_out) {
  real t = (q.k - ok) / dk;
  if (t < t_min || t > t_max) return false;
  real x = oa + t * da;
  real y = ob + t * db;
  if (x < q.a0 || x > q.a1 || y < q.b0 || y > q.b1) return false;
  *t_out = t;
  return true;
}
RT_HD bool rect_t(const FlatRect& q, const Ray& r, real t_min, real t_max, real* t_out) {
  if (q.axis == RECT_XY)
    return rect_core(q, r.origin.z, r.direction.z, r.origin.x, r.direction.x, r.origin.y, r.direction.y, t_min, t_max, t_out);
  if (q.axis == RECT_XZ)
    return rect_core(q, r.origin.y, r.direction.y, r.origin.x, r.direction.x, r.origin.z, r.direction.z, t_min, t_max, t_out);
  return rect_core(q, r.origin.x, r.direction.x, r.origin.y, r.direction.y, r.origin.z, r.direction.z, t_min, t_max, t_out);
}

template <uint32_t F, bool COUNT>
RT_HD bool prim_t(const SceneView& sv, PrimRef ref, const Ray& r, real t_min, real t_max,
                  real* t_out, TraceCounters* cnt) {
  uint32_t idx = primref_index(ref);
  uint32_t type = primref_type(ref);
  if ((F & F_SPHERE) && (type == PRIM_SPHERE || !(F & (F_MOVING_SPHERE | F_RECT | F_TRIANGLE | F_GRAVITY_SPHERE)))) {
    if (COUNT) cnt->sphere_tests++;
    const FlatSphere& s = sv.spheres[idx];
    return sphere_root(v3(s.cx, s.cy, s.cz), s.radius, r, t_min, t_max, t_out);
  }
  if ((F & F_MOVING_SPHERE) && (type == PRIM_MOVING_SPHERE || !(F & (F_RECT | F_TRIANGLE | F_GRAVITY_SPHERE)))) {
    if (COUNT) cnt->moving_sphere_tests++;
    const FlatMovingSphere& s = sv.moving_spheres[idx];
    return sphere_root(moving_sphere_center(s, r.time), s.radius, r, t_min, t_max, t_out);
  }
  if ((F & F_RECT) && (type == PRIM_RECT || !(F & (F_TRIANGLE | F_GRAVITY_SPHERE)))) {
    if (COUNT) cnt->rect_tests++;
    return rect_t(sv.rects[idx], r, t_min, t_max, t_out);
  }
  if ((F & F_TRIANGLE) && (type == PRIM_TRIANGLE || !(F & F_GRAVITY_SPHERE))) {
    if (COUNT) cnt->triangle_tests++;
    return triangle_t(sv.triangles[idx], r, t_min, t_max, t_out);
  }
  if (F & F_GRAVITY_SPHERE) {  // hit.rs:394-412
    if (COUNT) cnt->moving_sphere_tests++;
    const FlatGravitySphere& s = sv.gravity_spheres[idx];
    return sphere_root(gravity_sphere_center(s, sv.gravity_y, r.time), s.radius, r, t_min, t_max, t_out);
  }
  return false;
}

// Build the HitRecord of the winning primitive (the tail of each Hittable::hit).
template <uint32_t F>
RT_HD void prim_finalize(const SceneView& sv, PrimRef ref, const Ray& r, real t, HitRecord* rec) {
  uint32_t idx = primref_index(ref);
  uint32_t type = primref_type(ref);
  rec->t = t;
  rec->p = ray_at(r, t);
  if ((F & F_SPHERE) && (type == PRIM_SPHERE || !(F & (F_MOVING_SPHERE | F_RECT | F_TRIANGLE | F_GRAVITY_SPHERE)))) {
    // hit.rs:222-236
    const FlatSphere& s = sv.spheres[idx];
    Vec3 outward = (rec->p - v3(s.cx, s.cy, s.cz)) / s.radius;
    create_normal_face(r, outward, &rec->normal, &rec->front_face);
    rec->mat = s.mat;
    rec->u = real(0.0); rec->v = real(0.0);  // only read by Image textures
    if (F & F_IMAGE) {
      if (sv.materials[s.mat].needs_uv) get_sphere_uv(outward, &rec->u, &rec->v);
    }
    return;
  }
  if ((F & F_MOVING_SPHERE) && (type == PRIM_MOVING_SPHERE || !(F & (F_RECT | F_TRIANGLE | F_GRAVITY_SPHERE)))) {
    // hit.rs:301-314 (u = v = 0)
    const FlatMovingSphere& s = sv.moving_spheres[idx];
    Vec3 outward = (rec->p - moving_sphere_center(s, r.time)) / s.radius;
    create_normal_face(r, outward, &rec->normal, &rec->front_face);
    rec->mat = s.mat;
    rec->u = real(0.0); rec->v = real(0.0);
    return;
  }
  if ((F & F_RECT) && (type == PRIM_RECT || !(F & (F_TRIANGLE | F_GRAVITY_SPHERE)))) {
    // hit.rs:486-500, 551-565, 616-630
    const FlatRect& q = sv.rects[idx];
    rec->u = real(0.0); rec->v = real(0.0);
    if (q.axis == RECT_XY) {
      if ((F & F_IMAGE) && sv.materials[q.mat].needs_uv) {
        rec->u = ((r.origin.x + t * r.direction.x) - q.a0) / (q.a1 - q.a0);
        rec->v = ((r.origin.y + t * r.direction.y) - q.b0) / (q.b1 - q.b0);
      }
      create_normal_face(r, v3(0, 0, 1), &rec->normal, &rec->front_face);
    } else if (q.axis == RECT_XZ) {
      if ((F & F_IMAGE) && sv.materials[q.mat].needs_uv) {
        rec->u = ((r.origin.x + t * r.direction.x) - q.a0) / (q.a1 - q.a0);
        rec->v = ((r.origin.z + t * r.direction.z) - q.b0) / (q.b1 - q.b0);
      }
      create_normal_face(r, v3(0, 1, 0), &rec->normal, &rec->front_face);
    } else {
      if ((F & F_IMAGE) && sv.materials[q.mat].needs_uv) {
        rec->u = ((r.origin.y + t * r.direction.y) - q.a0) / (q.a1 - q.a0);
        rec->v = ((r.origin.z + t * r.direction.z) - q.b0) / (q.b1 - q.b0);
      }
      create_normal_face(r, v3(1, 0, 0), &rec->normal, &rec->front_face);
    }
    rec->mat = q.mat;
    return;
  }
  if ((F & F_TRIANGLE) && (type == PRIM_TRIANGLE || !(F & F_GRAVITY_SPHERE))) {
    // hit.rs:151-161 (u = v = 1)
    const FlatTriangle& tr = sv.triangles[idx];
    create_normal_face(r, load_v3(tr.normal), &rec->normal, &rec->front_face);
    rec->mat = tr.mat;
    rec->u = real(1.0); rec->v = real(1.0);
    return;
  }
  if (F & F_GRAVITY_SPHERE) {
    // hit.rs:413-428 (u = v = 0)
    const FlatGravitySphere& s = sv.gravity_spheres[idx];
    Vec3 outward = (rec->p - gravity_sphere_center(s, sv.gravity_y, r.time)) / s.radius;
    create_normal_face(r, outward, &rec->normal, &rec->front_face);
    rec->mat = s.mat;
    rec->u = real(0.0); rec->v = real(0.0);
  }
}

// aabb.rs:23-61.  inv_d is the reference's per-axis 1.0/direction (hoisted out of
// the node loop: same value every time).  The reference returns false at the
// first axis where t_max <= t_min; since t_min only grows and t_max only shrinks
// that is equivalent to testing once after the third axis.
//
// STRICT = true is Aabb::hit literally (an empty OR single-point interval misses).  The walkers use
// STRICT = false ("may hit": only a provably empty interval misses), the same rule the f32 culling
// test applies (core/cull32.hpp).  Why the walkers do not cull on a single-point interval: a box test is
// an optimisation here, every hit is decided by the primitive tests, and this build tests the box of
// every LEAF (a node stores its children's boxes) whereas BvhNode::hit (bvh.rs:97-112) only ever tests
// the union box of two children.  A zero-thickness leaf box -- an axis-aligned triangle, hit.rs:164-177
// pads nothing -- would be culled by the literal rule for EVERY ray, although the reference hits such a
// triangle whenever its sibling is not coplanar with it (which of its random trees make it invisible is
// topology luck, Q7).  With "may hit" the closest hit is the closest primitive hit, independent of tree
// shape and of which walker (f64 or f32 culling) runs.  tests/test_oracle_pairs.py::test_axis_aligned_triangles.
template <bool STRICT>
RT_HD bool aabb_interval(const real* bmin, const real* bmax, Point3 o, Vec3 inv_d, real t_min,
                         real t_max) {
  {
    real t0 = (bmin[0] - o.x) * inv_d.x, t1 = (bmax[0] - o.x) * inv_d.x;
    if (inv_d.x < real(0.0)) { real s = t0; t0 = t1; t1 = s; }
    t_min = t0 > t_min ? t0 : t_min;
    t_max = t1 < t_max ? t1 : t_max;
  }
  {
    real t0 = (bmin[1] - o.y) * inv_d.y, t1 = (bmax[1] - o.y) * inv_d.y;
    if (inv_d.y < real(0.0)) { real s = t0; t0 = t1; t1 = s; }
    t_min = t0 > t_min ? t0 : t_min;
    t_max = t1 < t_max ? t1 : t_max;
  }
  {
    real t0 = (bmin[2] - o.z) * inv_d.z, t1 = (bmax[2] - o.z) * inv_d.z;
    if (inv_d.z < real(0.0)) { real s = t0; t0 = t1; t1 = s; }
    t_min = t0 > t_min ? t0 : t_min;
    t_max = t1 < t_max ? t1 : t_max;
  }
  return STRICT ? !(t_max <= t_min) : !(t_max < t_min);
}
RT_HD bool aabb_hit(const real* bmin, const real* bmax, Point3 o, Vec3 inv_d, real t_min, real t_max) {
  return aabb_interval<true>(bmin, bmax, o, inv_d, t_min, t_max);
}
RT_HD bool aabb_may_hit(const real* bmin, const real* bmax, Point3 o, Vec3 inv_d, real t_min, real t_max) {
  return aabb_interval<false>(bmin, bmax, o, inv_d, t_min, t_max);
}

// A traversal's running answer.
struct Closest {
  real t;
  PrimRef ref;
  uint32_t order;  // slot in the owning list (tie break)
  bool hit;
};

// Offer one primitive to the running closest hit.  Range rejection is the
// reference's (strict on both ends, so t == closest is accepted); `order`
// makes the tie outcome independent of the visiting order.
template <uint32_t F, bool COUNT>
RT_HD void offer_prim(const SceneView& sv, PrimRef ref, uint32_t order, const Ray& r, real t_min,
                      Closest* best, TraceCounters* cnt) {
  real t;
  if (!prim_t<F, COUNT>(sv, ref, r, t_min, best->t, &t, cnt)) return;
  if (best->hit && t == best->t && order < best->order) return;
  best->t = t;
  best->ref = ref;
  best->order = order;
  best->hit = true;
}

// One step of the iterative closest-hit walk of a flattened BVH: visit `*node` (test its two
// child boxes near-first along the split axis, intersect leaf children, choose where to go next).
// Returns false when the walk is complete.  STACK provides reset() / push(int32_t) / pop() /
// empty(); the device instantiates it over LDS.  The visiting order is pure scheduling: the
// closest hit does not depend on it.  Exposed as a step so the device can interleave the walks
// of many rays per lane (hip/render.hip, k_trace_stream); bvh_closest below simply loops it.
template <uint32_t F, bool COUNT, class STACK>
RT_HD bool bvh_step(const SceneView& sv, uint32_t first_ref, const Ray& r, Vec3 inv_d, uint32_t dir_neg,
                    real t_min, int32_t* node, Closest* best, STACK& stack, TraceCounters* cnt) {
  const FlatNode& n = sv.nodes[*node];
  if (COUNT) cnt->box_tests += 2;
  int first = (int)((dir_neg >> (uint32_t)n.pad[0]) & 1u);  // near child along the split axis
  bool hf = aabb_may_hit(n.bmin[first], n.bmax[first], r.origin, inv_d, t_min, best->t);
  int32_t cf = n.child[first];
  int32_t next = -1;
  bool have_next = false;
  if (hf) {
    if (node_child_is_leaf(cf)) {
      uint32_t f = leaf_first(cf), k = leaf_count(cf);
      for (uint32_t i = 0; i < k; ++i)
        offer_prim<F, COUNT>(sv, sv.refs[first_ref + f + i], f + i, r, t_min, best, cnt);
    } else {
      next = cf; have_next = true;
    }
  }
  // the far box is tested after the near leaf may have shrunk best->t
  bool hs = aabb_may_hit(n.bmin[1 - first], n.bmax[1 - first], r.origin, inv_d, t_min, best->t);
  int32_t cs = n.child[1 - first];
  if (hs) {
    if (node_child_is_leaf(cs)) {
      uint32_t f = leaf_first(cs), k = leaf_count(cs);
      for (uint32_t i = 0; i < k; ++i)
        offer_prim<F, COUNT>(sv, sv.refs[first_ref + f + i], f + i, r, t_min, best, cnt);
    } else if (have_next) {
      stack.push(cs);
    } else {
      next = cs; have_next = true;
    }
  }
  if (have_next) { *node = next; return true; }
  if (stack.empty()) return false;
  *node = stack.pop();
  return true;
}

RT_HD Vec3 ray_inv_dir(const Ray& r) {  // aabb.rs:48: inv_d = 1.0 / direction, per axis
  return v3(real(1.0) / r.direction.x, real(1.0) / r.direction.y, real(1.0) / r.direction.z);
}
RT_HD uint32_t ray_dir_neg(const Ray& r) {  // bit a set: the ray travels towards -a
  return (r.direction.x < real(0.0) ? 1u : 0u) | (r.direction.y < real(0.0) ? 2u : 0u) | (r.direction.z < real(0.0) ? 4u : 0u);
}

template <uint32_t F, bool COUNT, class STACK>
RT_HD void bvh_closest(const SceneView& sv, int32_t root, uint32_t first_ref, const Ray& r,
                       real t_min, Closest* best, STACK& stack, TraceCounters* cnt) {
  Vec3 inv_d = ray_inv_dir(r);
  uint32_t dir_neg = ray_dir_neg(r);
  stack.reset();
  int32_t node = root;
  while (bvh_step<F, COUNT>(sv, first_ref, r, inv_d, dir_neg, t_min, &node, best, stack, cnt)) {
  }
}

// How a BVH entry is walked.  SerialWalk is the portable one (CPU checker, simple kernels); the
// device supplies a wave-cooperative walker (hip/render.hip VoteWalk).  Any walker must return the
// closest hit under offer_prim's rule, so the choice is invisible in results.
struct SerialWalk {
  template <uint32_t F, bool COUNT, class STACK>
  RT_HD static void run(const SceneView& sv, int32_t root, uint32_t first_ref, const Ray& r, real t_min,
                        Closest* best, STACK& stack, TraceCounters* cnt) {
    bvh_closest<F, COUNT>(sv, root, first_ref, r, t_min, best, stack, cnt);
  }
};

// PRIM / GROUP / BVH entries: find the closest candidate in [t_min, t_max].
template <uint32_t F, bool COUNT, class STACK, class WALK = SerialWalk>
RT_HD void geom_closest(const SceneView& sv, const FlatEntry& e, const Ray& r, real t_min,
                        real t_max, Closest* best, STACK& stack, TraceCounters* cnt) {
  best->t = t_max;
  best->hit = false;
  best->ref = 0;
  best->order = 0;
  if ((F & F_BVH) && (e.kind == ENTRY_BVH || !(F & (F_PRIM_ENTRY | F_GROUP)))) {
    WALK::template run<F, COUNT>(sv, e.a, (uint32_t)e.b, r, t_min, best, stack, cnt);
    return;
  }
  if ((F & F_PRIM_ENTRY) && (e.kind == ENTRY_PRIM || !(F & F_GROUP))) {
    real t;
    if (prim_t<F, COUNT>(sv, (PrimRef)e.a, r, t_min, t_max, &t, cnt)) {
      best->t = t; best->ref = (PrimRef)e.a; best->hit = true;
    }
    return;
  }
  if (F & F_GROUP) {
    // HittableList::hit (hit.rs:660-690): in order, shrinking closest_so_far, later wins ties.
    for (int32_t i = 0; i < e.b; ++i) {
      PrimRef ref = sv.refs[e.a + i];
      real t;
      if (prim_t<F, COUNT>(sv, ref, r, t_min, best->t, &t, cnt)) {
        best->t = t; best->ref = ref; best->order = (uint32_t)i; best->hit = true;
      }
    }
  }
}

// hit.rs:802-807 / 892-904: the ray seen by the child of one transform op.
RT_HD Ray xform_ray(const FlatXformOp& op, const Ray& r) {
  if (op.op == XFORM_TRANSLATE) {
    return make_ray(r.origin - load_v3(op.v), r.direction, r.time);
  }
  real sin_t = op.v[0], cos_t = op.v[1];
  Vec3 o = v3(cos_t * r.origin.x - sin_t * r.origin.z, r.origin.y,
              sin_t * r.origin.x + cos_t * r.origin.z);
  Vec3 d = v3(cos_t * r.direction.x - sin_t * r.direction.z, r.direction.y,
              sin_t * r.direction.x + cos_t * r.direction.z);
  return make_ray(o, d, r.time);
}

// hit.rs:808-820 / 909-930: map the child's record back out through one op.
// `child_ray` is the ray the op handed to its child; both ops re-face-forward
// the normal against THAT ray (Translate: same direction as the outer ray;
// RotateY: the rotated ray against the un-rotated normal -- reproduced as is).
RT_HD void xform_record(const FlatXformOp& op, const Ray& child_ray, HitRecord* rec) {
  if (op.op == XFORM_TRANSLATE) {
    create_normal_face(child_ray, rec->normal, &rec->normal, &rec->front_face);
    rec->p = rec->p + load_v3(op.v);
    return;
  }
  real sin_t = op.v[0], cos_t = op.v[1];
  Vec3 p = v3(cos_t * rec->p.x + sin_t * rec->p.z, rec->p.y, -sin_t * rec->p.x + cos_t * rec->p.z);
  Vec3 n = v3(cos_t * rec->normal.x + sin_t * rec->normal.z, rec->normal.y,
              -sin_t * rec->normal.x + cos_t * rec->normal.z);
  rec->p = p;
  create_normal_face(child_ray, n, &rec->normal, &rec->front_face);
}

// The way back out of a chain of ops (outermost first): innermost op first, each against the ray IT handed to its child.  That
// ray is recomputed from the outer ray for every op but the innermost (whose child ray is `innermost_ray`, already at hand)
// instead of being kept live across the walk.
RT_HD void xform_record_chain(const FlatXformOp* ops, int nops, const Ray& outer, const Ray& innermost_ray, HitRecord* rec) {
  xform_record(ops[nops - 1], innermost_ray, rec);
  for (int k = nops - 2; k >= 0; --k) {
    Ray c = outer;
    for (int j = 0; j <= k; ++j) c = xform_ray(ops[j], c);
    xform_record(ops[k], c, rec);
  }
}

// The world: HittableList::hit over the ordered top-level table (hit.rs:660-690), with
// Translate / RotateY (hit.rs:802-823, 892-931) and ConstantMedium (hit.rs:955-986) handled
// around ONE geometry query site so the traversal code exists once in the kernel.
//
// A ConstantMedium asks its boundary twice (rec1 over (-inf, inf), rec2 from rec1.t + 0.0001),
// then draws one uniform from the path's stream -- inside the intersection, exactly where
// the reference draws it (hit.rs:969).
template <uint32_t F, bool COUNT, class STACK, class WALK = SerialWalk>
RT_HD bool world_hit(const SceneView& sv, const Ray& r, real t_min, real t_max, HitRecord* rec,
                     Rng& rng, STACK& stack, TraceCounters* cnt) {
  if (COUNT) cnt->rays++;
  bool hit_anything = false;
  real closest_so_far = t_max;
  // plain primitives in the list: a conservative f32 box test per lane, and the f64 primitive test is skipped when
  // no lane of the wave can pass it (most waves never see Book-2's small spheres)
  const bool cull_prims = (F & F_PRIM_ENTRY) && sv.top_box32 != nullptr;
  Ray32 q32 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (cull_prims) q32 = make_ray32(r, t_min);
  for (int32_t i = 0; i < sv.n_top_level; ++i) {
    // the table walk is wave-uniform: entries are fetched once per wave (rt_load_uniform)
    const FlatEntry e_rec = rt_load_uniform(&sv.entries[rt_load_uniform(&sv.top_level[i])]);
    const FlatEntry* e = &e_rec;
    if (cull_prims && e->kind == ENTRY_PRIM) {
      const float* bx = sv.top_box32 + 6 * i;
      const float lo[3] = {rt_load_uniform(bx + 0), rt_load_uniform(bx + 1), rt_load_uniform(bx + 2)};
      const float hi[3] = {rt_load_uniform(bx + 3), rt_load_uniform(bx + 4), rt_load_uniform(bx + 5)};
      const bool may = cull32_may_hit(lo, hi, q32, cull_round_up(closest_so_far));
      if (!RT_WAVE_ANY(may)) continue;
    }
    const bool is_medium = (F & F_MEDIUM) && e->kind == ENTRY_MEDIUM;
    FlatEntry solid_rec = e_rec;
    if (is_medium) solid_rec = rt_load_uniform(&sv.entries[e->a]);
    const FlatEntry* solid = &solid_rec;
    // the ray as the innermost geometry sees it (up to RT_MAX_XFORM_OPS transform ops, outermost first); the
    // intermediate ray is recomputed for the way back instead of being kept live across the walk
    const bool is_xform = (F & F_XFORM) && solid->kind == ENTRY_XFORM;
    Ray rq = r;
    FlatEntry geom_rec = solid_rec;
    int nops = 0;
    if (is_xform) {
      nops = solid->b;
      geom_rec = rt_load_uniform(&sv.entries[solid->a]);
      for (int k = 0; k < nops; ++k) rq = xform_ray(solid->ops[k], rq);
    }

    Closest best;
    real q_min = is_medium ? -RT_INFINITY : t_min;
    real q_max = is_medium ? RT_INFINITY : closest_so_far;
    real rec1_t = real(0.0);
    bool ok = true;
    const int n_query = is_medium ? 2 : 1;
    for (int q = 0; q < n_query; ++q) {
      geom_closest<F, COUNT, STACK, WALK>(sv, geom_rec, rq, q_min, q_max, &best, stack, cnt);
      if (!best.hit) { ok = false; break; }
      if (q == 0) { rec1_t = best.t; q_min = rec1_t + real(0.0001); }
    }
    if (!ok) continue;

    // every accepted hit replaces the running record (range rejection already compared against
    // closest_so_far), so the record is built in place
    if (is_medium) {
      real rec2_t = best.t;
      real t1 = rt_fmax(rec1_t, t_min);
      real t2 = rt_fmin(rec2_t, closest_so_far);
      if (t1 >= t2) continue;
      if (t1 < real(0.0)) t1 = real(0.0);
      real ray_length = length(r.direction);
      real distance_inside_boundary = (t2 - t1) * ray_length;
      real hit_distance = e->f[0] * rt_log(rng_f64(rng));
      if (hit_distance > distance_inside_boundary) continue;
      real t = t1 + hit_distance / ray_length;
      rec->t = t;
      rec->p = ray_at(r, t);
      rec->normal = v3(0, 0, 0);
      rec->front_face = true;
      rec->u = real(0.0); rec->v = real(0.0);
      rec->mat = e->b;
    } else {
      prim_finalize<F>(sv, best.ref, rq, best.t, rec);
      if (is_xform) xform_record_chain(solid->ops, nops, r, rq, rec);
    }
    hit_anything = true;
    closest_so_far = rec->t;
  }
  return hit_anything;
}

// Host-side stack (the device uses an LDS-backed one, see hip/trace_kernels.hip).
template <int N>
struct LocalStack {
  int32_t data[N];
  int n;
  RT_HD void reset() { n = 0; }
  RT_HD void push(int32_t v) { data[n++] = v; }
  RT_HD int32_t pop() { return data[--n]; }
  RT_HD bool empty() const { return n == 0; }
};

}  // namespace rt
