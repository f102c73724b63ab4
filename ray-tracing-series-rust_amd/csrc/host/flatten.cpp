// Scene graph -> flat arrays (see core/flat_types.hpp for the layout).
//
// Shapes accepted (everything the reference's scene catalogue builds,
// /root/reference/src/world.rs:95-874):
//   world      := LIST | any single object
//   LIST item  := primitive | RECT_PRISM | LIST (inlined: list-in-list hits are order
//                 equivalent to the inlined sequence) | BVH | XFORM chain | CONSTANT_MEDIUM
//   BVH item   := primitive | RECT_PRISM (its 6 rects become BVH primitives) | LIST (inlined)
//   XFORM      := up to RT_MAX_XFORM_OPS nested Translate / RotateY over
//                 primitive | RECT_PRISM | LIST of primitives | BVH
//   MEDIUM     := boundary is any of the above except another medium; media only at top level
// Anything else is reported as unsupported rather than approximated.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include "../core/geometry.hpp"
#include "../core/shading.hpp"
#include "flat_scene.hpp"

namespace rtx {

using rt::Vec3;

size_t FlatScene::total_bytes() const {
  return spheres.size() * sizeof(rt::FlatSphere) + moving_spheres.size() * sizeof(rt::FlatMovingSphere) +
         rects.size() * sizeof(rt::FlatRect) + triangles.size() * sizeof(rt::FlatTriangle) +
         gravity_spheres.size() * sizeof(rt::FlatGravitySphere) + gravity_y.size() * sizeof(double) +
         nodes.size() * sizeof(rt::FlatNode) + nodes32.size() * sizeof(rt::FlatNode32) + refs.size() * sizeof(rt::PrimRef) +
         entries.size() * sizeof(rt::FlatEntry) + top_level.size() * sizeof(int32_t) +
         materials.size() * sizeof(rt::FlatMaterial) + textures.size() * sizeof(rt::FlatTexture) +
         perlins.size() * sizeof(rt::FlatPerlin) + images.size() * sizeof(rt::FlatImage) +
         texels.size() * sizeof(double) + motion32.size() * sizeof(rt::FlatMotion32);
}

namespace {

struct Flattener {
  const SceneGraph& g;
  FlatScene& out;
  BuildOptions opt;
  std::string err;
  std::vector<int64_t> prim_of;  // graph hittable -> PrimRef already emitted (-1: not yet)
  std::vector<int32_t> bvh_entry_of;  // graph BVH hittable -> its entry (-1: not yet): a reused handle (the reference holds
                                      // objects as Arc, so one BvhNode may sit under several Translates) is emitted once

  Flattener(const SceneGraph& gg, FlatScene& o, const BuildOptions& op)
      : g(gg), out(o), opt(op), prim_of(gg.hittables.size(), -1), bvh_entry_of(gg.hittables.size(), -1) {}

  bool fail(const std::string& m) { if (err.empty()) err = m; return false; }

  static bool is_prim(int32_t kind) {
    return kind == H_SPHERE || kind == H_MOVING_SPHERE || kind == H_TRIANGLE || kind == H_XY_RECT ||
           kind == H_XZ_RECT || kind == H_YZ_RECT || kind == H_GRAVITY_SPHERE;
  }

  rt::PrimRef emit_rect(int32_t axis, double a0, double a1, double b0, double b1, double k, int32_t mat) {
    rt::FlatRect q;
    q.a0 = a0; q.a1 = a1; q.b0 = b0; q.b1 = b1; q.k = k; q.axis = axis; q.mat = mat;
    out.rects.push_back(q);
    return rt::make_primref(rt::PRIM_RECT, (uint32_t)out.rects.size() - 1);
  }

  rt::PrimRef emit_prim(int32_t h) {
    if (prim_of[h] >= 0) return (rt::PrimRef)prim_of[h];
    const GHittable& o = g.hittables[h];
    rt::PrimRef ref = 0;
    switch (o.kind) {
      case H_SPHERE: {
        rt::FlatSphere s;
        s.cx = o.f[0]; s.cy = o.f[1]; s.cz = o.f[2]; s.radius = o.f[3]; s.mat = o.mat; s.pad = 0;
        out.spheres.push_back(s);
        ref = rt::make_primref(rt::PRIM_SPHERE, (uint32_t)out.spheres.size() - 1);
        break;
      }
      case H_MOVING_SPHERE: {
        rt::FlatMovingSphere s;
        for (int i = 0; i < 3; ++i) { s.c0[i] = o.f[i]; s.c1[i] = o.f[3 + i]; }
        s.time0 = o.f[6]; s.time1 = o.f[7]; s.radius = o.f[8]; s.mat = o.mat; s.pad = 0;
        out.moving_spheres.push_back(s);
        ref = rt::make_primref(rt::PRIM_MOVING_SPHERE, (uint32_t)out.moving_spheres.size() - 1);
        break;
      }
      case H_TRIANGLE: {
        rt::FlatTriangle t;
        for (int i = 0; i < 3; ++i) { t.v0[i] = o.f[i]; t.v1[i] = o.f[3 + i]; t.v2[i] = o.f[6 + i]; t.normal[i] = o.f[9 + i]; }
        t.mat = o.mat; t.pad = 0;
        out.triangles.push_back(t);
        ref = rt::make_primref(rt::PRIM_TRIANGLE, (uint32_t)out.triangles.size() - 1);
        break;
      }
      case H_GRAVITY_SPHERE: {
        rt::FlatGravitySphere s;
        s.sx = o.f[0]; s.sy = o.f[1]; s.sz = o.f[2]; s.time0 = o.f[3]; s.radius = o.f[4]; s.mat = o.mat; s.pad = 0;
        s.table_first = (int64_t)out.gravity_y.size();
        s.table_len = (int64_t)o.table.size();
        out.gravity_y.insert(out.gravity_y.end(), o.table.begin(), o.table.end());
        out.gravity_spheres.push_back(s);
        ref = rt::make_primref(rt::PRIM_GRAVITY_SPHERE, (uint32_t)out.gravity_spheres.size() - 1);
        break;
      }
      case H_XY_RECT: ref = emit_rect(rt::RECT_XY, o.f[0], o.f[1], o.f[2], o.f[3], o.f[4], o.mat); break;
      case H_XZ_RECT: ref = emit_rect(rt::RECT_XZ, o.f[0], o.f[1], o.f[2], o.f[3], o.f[4], o.mat); break;
      default: ref = emit_rect(rt::RECT_YZ, o.f[0], o.f[1], o.f[2], o.f[3], o.f[4], o.mat); break;
    }
    prim_of[h] = (int64_t)ref;
    return ref;
  }

  // RectPrism::new (hit.rs:720-769): six sides in the reference's add() order.
  void emit_prism(const GHittable& o, std::vector<rt::PrimRef>* refs) {
    const double* p0 = &o.f[0];
    const double* p1 = &o.f[3];
    refs->push_back(emit_rect(rt::RECT_XY, p0[0], p1[0], p0[1], p1[1], p1[2], o.mat));
    refs->push_back(emit_rect(rt::RECT_XY, p0[0], p1[0], p0[1], p1[1], p0[2], o.mat));
    refs->push_back(emit_rect(rt::RECT_XZ, p0[0], p1[0], p0[2], p1[2], p1[1], o.mat));
    refs->push_back(emit_rect(rt::RECT_XZ, p0[0], p1[0], p0[2], p1[2], p0[1], o.mat));
    refs->push_back(emit_rect(rt::RECT_YZ, p0[1], p1[1], p0[2], p1[2], p1[0], o.mat));
    refs->push_back(emit_rect(rt::RECT_YZ, p0[1], p1[1], p0[2], p1[2], p0[0], o.mat));
  }

  // Primitives of h in list order; lists and prisms are inlined.  handles (optional) receives, per reference, the
  // graph hittable it came from (-1 for the rectangles of a RectPrism, which have no handle of their own).
  bool collect_prims(int32_t h, std::vector<rt::PrimRef>* refs, int depth = 0, std::vector<int32_t>* handles = nullptr) {
    if (depth > 64) return fail("hittable lists nested deeper than 64");
    const GHittable& o = g.hittables[h];
    if (is_prim(o.kind)) { refs->push_back(emit_prim(h)); if (handles) handles->push_back(h); return true; }
    if (o.kind == H_RECT_PRISM) { emit_prism(o, refs); if (handles) handles->resize(refs->size(), -1); return true; }
    // A list or another BvhNode inside a BVH (or inside a transformed / medium list): BvhNode::from_list takes any Hittable
    // (bvh.rs:85-93).  The closest hit over nested containers of plain primitives is the closest hit over the primitives, so the
    // inner container dissolves into the outer one's primitive set and ONE tree is built over all of them (a BVH is a culling
    // structure).  What the inner container's own order would decide -- an exact tie between two of its primitives -- follows
    // the outer container's rule (later slot wins).
    if (o.kind == H_LIST || o.kind == H_BVH) {
      for (int32_t c : o.children)
        if (!collect_prims(c, refs, depth + 1, handles)) return false;
      return true;
    }
    return fail("unsupported nesting: a Translate / RotateY / ConstantMedium INSIDE a BVH or inside a transformed or medium list "
                "(an instanced sub-tree); primitives, RectPrisms, lists and BVHs of those may sit there");
  }

  // Reference bounding boxes (hit.rs bounding_box impls) of one flattened primitive.
  void prim_box(rt::PrimRef ref, double time0, double time1, double* b) const {
    uint32_t idx = rt::primref_index(ref);
    switch (rt::primref_type(ref)) {
      case rt::PRIM_SPHERE: {  // hit.rs:239-244
        const rt::FlatSphere& s = out.spheres[idx];
        Vec3 c = rt::v3(s.cx, s.cy, s.cz), r = rt::v3(s.radius, s.radius, s.radius);
        Vec3 lo = c - r, hi = c + r;
        b[0] = lo.x; b[1] = lo.y; b[2] = lo.z; b[3] = hi.x; b[4] = hi.y; b[5] = hi.z;
        break;
      }
      case rt::PRIM_MOVING_SPHERE: {  // hit.rs:317-327
        const rt::FlatMovingSphere& s = out.moving_spheres[idx];
        Vec3 r = rt::v3(s.radius, s.radius, s.radius);
        Vec3 ca = rt::moving_sphere_center(s, time0), cb = rt::moving_sphere_center(s, time1);
        Vec3 lo0 = ca - r, hi0 = ca + r, lo1 = cb - r, hi1 = cb + r;
        b[0] = std::fmin(lo0.x, lo1.x); b[1] = std::fmin(lo0.y, lo1.y); b[2] = std::fmin(lo0.z, lo1.z);
        b[3] = std::fmax(hi0.x, hi1.x); b[4] = std::fmax(hi0.y, hi1.y); b[5] = std::fmax(hi0.z, hi1.z);
        break;
      }
      case rt::PRIM_GRAVITY_SPHERE: {  // hit.rs:430-443: union of the boxes at time0 and time1 (NOT of the trajectory between)
        const rt::FlatGravitySphere& s = out.gravity_spheres[idx];
        Vec3 r = rt::v3(s.radius, s.radius, s.radius);
        Vec3 ca = rt::gravity_sphere_center(s, out.gravity_y.data(), time0), cb = rt::gravity_sphere_center(s, out.gravity_y.data(), time1);
        Vec3 lo0 = ca - r, hi0 = ca + r, lo1 = cb - r, hi1 = cb + r;
        b[0] = std::fmin(lo0.x, lo1.x); b[1] = std::fmin(lo0.y, lo1.y); b[2] = std::fmin(lo0.z, lo1.z);
        b[3] = std::fmax(hi0.x, hi1.x); b[4] = std::fmax(hi0.y, hi1.y); b[5] = std::fmax(hi0.z, hi1.z);
        break;
      }
      case rt::PRIM_RECT: {  // hit.rs:503-508, 568-573, 633-638 (+-0.0001 on the thin axis)
        const rt::FlatRect& q = out.rects[idx];
        if (q.axis == rt::RECT_XY) { b[0] = q.a0; b[1] = q.b0; b[2] = q.k - 0.0001; b[3] = q.a1; b[4] = q.b1; b[5] = q.k + 0.0001; }
        else if (q.axis == rt::RECT_XZ) { b[0] = q.a0; b[1] = q.k - 0.0001; b[2] = q.b0; b[3] = q.a1; b[4] = q.k + 0.0001; b[5] = q.b1; }
        else { b[0] = q.k - 0.0001; b[1] = q.a0; b[2] = q.b0; b[3] = q.k + 0.0001; b[4] = q.a1; b[5] = q.b1; }
        break;
      }
      default: {  // hit.rs:164-177
        const rt::FlatTriangle& t = out.triangles[idx];
        for (int a = 0; a < 3; ++a) {
          b[a] = std::fmin(std::fmin(t.v0[a], t.v1[a]), t.v2[a]);
          b[3 + a] = std::fmax(std::fmax(t.v0[a], t.v1[a]), t.v2[a]);
        }
        break;
      }
    }
  }

  int32_t push_entry(const rt::FlatEntry& e) {
    out.entries.push_back(e);
    return (int32_t)out.entries.size() - 1;
  }
  static rt::FlatEntry blank_entry(int32_t kind) {
    rt::FlatEntry e;
    memset(&e, 0, sizeof(e));
    e.kind = kind;
    return e;
  }

  int32_t emit_group(const std::vector<rt::PrimRef>& refs) {
    rt::FlatEntry e = blank_entry(rt::ENTRY_GROUP);
    e.a = (int32_t)out.refs.size();
    e.b = (int32_t)refs.size();
    out.refs.insert(out.refs.end(), refs.begin(), refs.end());
    return push_entry(e);
  }

  // PRIM / GROUP / BVH entry for h; -1 on failure.
  int32_t emit_geom_entry(int32_t h) {
    const GHittable& o = g.hittables[h];
    if (is_prim(o.kind)) {
      rt::FlatEntry e = blank_entry(rt::ENTRY_PRIM);
      e.a = (int32_t)emit_prim(h);
      return push_entry(e);
    }
    if (o.kind == H_RECT_PRISM || o.kind == H_LIST) {
      std::vector<rt::PrimRef> refs;
      if (!collect_prims(h, &refs)) return -1;
      if (refs.empty()) { fail("empty HittableList under a transform/medium"); return -1; }
      return emit_group(refs);
    }
    if (o.kind == H_BVH) {
      if (bvh_entry_of[h] >= 0) return bvh_entry_of[h];
      std::vector<rt::PrimRef> refs;
      std::vector<int32_t> handles;
      const size_t tris_before = out.triangles.size();  // triangles below this index are referenced by something else already
      for (int32_t c : o.children)
        if (!collect_prims(c, &refs, 0, &handles)) return -1;
      if (refs.size() < 2 && !opt.reference_bvh) return bvh_entry_of[h] = emit_group(refs);
      std::vector<double> boxes(6 * refs.size());
      for (size_t i = 0; i < refs.size(); ++i) prim_box(refs[i], o.f[0], o.f[1], &boxes[6 * i]);
      std::vector<uint32_t> order;
      int32_t depth = 0;
      int32_t root;
      if (opt.reference_bvh) {
        std::vector<double> sort_boxes(6 * refs.size());
        for (size_t i = 0; i < refs.size(); ++i) prim_box(refs[i], 0.0, 0.0, &sort_boxes[6 * i]);
        root = build_bvh_reference(boxes, sort_boxes, opt.bvh_seed + (uint64_t)out.n_bvh, &out.nodes, &order, &depth);
      } else {
        // Leaf size: a sphere or rectangle test costs several box tests, so giving every such primitive its own
        // (tight) box pays; triangle meshes are deep and their boxes loose, two per leaf measured best.
        BuildOptions bo = opt;
        if (bo.max_leaf <= 0) {
          bool has_tri = false;
          for (rt::PrimRef r : refs) has_tri |= rt::primref_type(r) == rt::PRIM_TRIANGLE;
          bo.max_leaf = has_tri ? 2 : 1;
        }
        const auto t0 = std::chrono::steady_clock::now();
        if (opt.gpu_builder && refs.size() >= 1024) {
          std::string gerr;
          root = build_bvh_gpu(boxes, bo.max_leaf, &out.nodes, &order, &depth, &out.bvh_device_ms, &gerr);
          if (root < 0) { fail("GPU BVH builder: " + gerr); return -1; }
        } else {
          // A BVH that holds MovingSpheres: partition by where everything is at the MIDDLE of the interval.  The stored boxes (the
          // reference's unions over the interval) of moving and static spheres overlap wherever the movers started; at one instant they
          // are apart, SAH then keeps movers with movers, and the time-aware boxes of such subtrees (FlatMotion32) stay tight at every
          // instant -- the lerp of a mixed subtree's end boxes spans start AND end.  Topology only: nothing a ray can see.
          std::vector<double> mid_boxes;
          bool movers = false;
          for (rt::PrimRef r : refs) movers |= rt::primref_type(r) == rt::PRIM_MOVING_SPHERE;
          if (movers && o.f[0] < o.f[1] && opt.motion_topology) {
            mid_boxes.resize(boxes.size());
            const double tm = 0.5 * (o.f[0] + o.f[1]);
            for (size_t i = 0; i < refs.size(); ++i) prim_box(refs[i], tm, tm, &mid_boxes[6 * i]);
          }
          root = build_bvh(boxes, bo, &out.nodes, &order, &depth, &out.sah_cost, mid_boxes.empty() ? nullptr : &mid_boxes);
        }
        out.bvh_build_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      }
      rt::FlatEntry e = blank_entry(rt::ENTRY_BVH);
      e.a = root;
      e.b = (int32_t)out.refs.size();
      e.c = (int32_t)refs.size();
      e.f[0] = o.f[0]; e.f[1] = o.f[1];  // BvhNode::from_list(list, time0, time1)
      // Triangles of this BVH are stored so that array order is leaf order: a leaf's triangles become neighbours in
      // memory, and for a pure mesh the slot of a reference is its triangle index minus a constant, which lets the
      // mesh kernels skip the reference fetch.  Only triangles nothing else refers to may be MOVED for that:
      //   * every triangle of this BVH was first emitted by this BVH, each once (the catalogue's meshes): they are
      //     permuted within the slots they occupy, and prim_of follows them so later users of a handle see the move;
      //   * otherwise (a triangle shared with an earlier entry or BVH, or listed twice) nothing is moved: the BVH gets
      //     a private leaf-ordered copy of its triangles at the end of the array.
      {
        std::vector<uint32_t> in_leaf_order;
        for (size_t i = 0; i < refs.size(); ++i)
          if (rt::primref_type(refs[order[i]]) == rt::PRIM_TRIANGLE) in_leaf_order.push_back(rt::primref_index(refs[order[i]]));
        std::vector<uint32_t> sorted_idx = in_leaf_order;
        std::sort(sorted_idx.begin(), sorted_idx.end());
        bool exclusive = true;
        for (size_t k = 0; k < sorted_idx.size(); ++k) {
          if (sorted_idx[k] < tris_before) exclusive = false;
          if (k > 0 && sorted_idx[k] == sorted_idx[k - 1]) exclusive = false;
        }
        if (exclusive) {
          std::vector<rt::FlatTriangle> moved(in_leaf_order.size());
          for (size_t k = 0; k < in_leaf_order.size(); ++k) moved[k] = out.triangles[in_leaf_order[k]];
          for (size_t k = 0; k < in_leaf_order.size(); ++k) out.triangles[sorted_idx[k]] = moved[k];
        } else {
          const uint32_t base = (uint32_t)out.triangles.size();
          for (size_t k = 0; k < in_leaf_order.size(); ++k) {
            const rt::FlatTriangle copy = out.triangles[in_leaf_order[k]];
            out.triangles.push_back(copy);
            sorted_idx[k] = base + (uint32_t)k;
          }
        }
        size_t k = 0;
        for (size_t i = 0; i < refs.size(); ++i) {
          rt::PrimRef r = refs[order[i]];
          if (rt::primref_type(r) == rt::PRIM_TRIANGLE) {
            r = rt::make_primref(rt::PRIM_TRIANGLE, sorted_idx[k++]);
            if (exclusive && handles[order[i]] >= 0) prim_of[handles[order[i]]] = (int64_t)r;
          }
          out.refs.push_back(r);
        }
      }
      out.max_stack = std::max(out.max_stack, depth);
      out.n_bvh++;
      return bvh_entry_of[h] = push_entry(e);
    }
    fail("unsupported object where geometry is expected (transform of a transform chain > 2, medium inside a wrapper, ...)");
    return -1;
  }

  // Any non-medium entry, including Translate/RotateY chains.
  int32_t emit_solid_entry(int32_t h) {
    const GHittable& o = g.hittables[h];
    if (o.kind != H_TRANSLATE && o.kind != H_ROTATE_Y) {
      if (o.kind == H_CONSTANT_MEDIUM) { fail("a ConstantMedium may only appear in the top-level list"); return -1; }
      return emit_geom_entry(h);
    }
    rt::FlatEntry e = blank_entry(rt::ENTRY_XFORM);
    int32_t cur = h;
    int nops = 0;
    while (g.hittables[cur].kind == H_TRANSLATE || g.hittables[cur].kind == H_ROTATE_Y) {
      if (nops == RT_MAX_XFORM_OPS) { fail("more than " + std::to_string(RT_MAX_XFORM_OPS) + " nested Translate/RotateY wrappers around one object"); return -1; }
      const GHittable& w = g.hittables[cur];
      rt::FlatXformOp& op = e.ops[nops++];
      if (w.kind == H_TRANSLATE) { op.op = rt::XFORM_TRANSLATE; op.v[0] = w.f[0]; op.v[1] = w.f[1]; op.v[2] = w.f[2]; }
      else { op.op = rt::XFORM_ROTATE_Y; op.v[0] = w.f[0]; op.v[1] = w.f[1]; op.v[2] = 0.0; }
      cur = w.children[0];
    }
    int32_t child = emit_geom_entry(cur);
    if (child < 0) return -1;
    e.a = child;
    e.b = nops;
    return push_entry(e);
  }

  int32_t emit_entry(int32_t h) {
    const GHittable& o = g.hittables[h];
    if (o.kind != H_CONSTANT_MEDIUM) return emit_solid_entry(h);
    int32_t boundary = emit_solid_entry(o.children[0]);
    if (boundary < 0) return -1;
    rt::FlatEntry e = blank_entry(rt::ENTRY_MEDIUM);
    e.a = boundary;
    e.b = o.mat;
    e.f[0] = o.f[0];
    return push_entry(e);
  }

  bool emit_top(int32_t h, int depth = 0) {
    if (depth > 64) return fail("hittable lists nested deeper than 64");
    const GHittable& o = g.hittables[h];
    if (o.kind == H_LIST) {
      for (int32_t c : o.children)
        if (!emit_top(c, depth + 1)) return false;
      return true;
    }
    int32_t e = emit_entry(h);
    if (e < 0) return false;
    out.top_level.push_back(e);
    return true;
  }

  bool texture_has_image(int32_t t, int depth = 0) const {
    if (t < 0 || depth > 16) return false;
    const GTexture& tx = g.textures[t];
    if (tx.kind == rt::TEX_IMAGE) return true;
    if (tx.kind == rt::TEX_CHECKER) return texture_has_image(tx.a, depth + 1) || texture_has_image(tx.b, depth + 1);
    return false;
  }

  bool run(int32_t world) {
    for (const GTexture& t : g.textures) {
      rt::FlatTexture f;
      memset(&f, 0, sizeof(f));
      f.kind = t.kind; f.a = t.a; f.b = t.b;
      f.color[0] = t.color[0]; f.color[1] = t.color[1]; f.color[2] = t.color[2];
      f.scale = t.scale;
      out.textures.push_back(f);
    }
    for (const GMaterial& m : g.materials) {
      rt::FlatMaterial f;
      memset(&f, 0, sizeof(f));
      f.kind = m.kind; f.tex = m.tex;
      f.albedo[0] = m.albedo[0]; f.albedo[1] = m.albedo[1]; f.albedo[2] = m.albedo[2];
      f.param = m.param;
      f.needs_uv = texture_has_image(m.tex) ? 1 : 0;
      // SolidColor::value (texture.rs:27-31) resolved here: kernels compiled for scenes without checker / noise / image
      // textures read the colour from the material record and never fetch the texture record behind it
      if ((m.kind == rt::MAT_LAMBERTIAN || m.kind == rt::MAT_DIFFUSE_LIGHT || m.kind == rt::MAT_ISOTROPIC) && m.tex >= 0 &&
          (size_t)m.tex < g.textures.size() && g.textures[m.tex].kind == rt::TEX_SOLID) {
        const GTexture& t = g.textures[m.tex];
        f.albedo[0] = t.color[0]; f.albedo[1] = t.color[1]; f.albedo[2] = t.color[2];
      }
      if (m.kind == rt::MAT_DIELECTRIC) rt::dielectric_constants(f.param, f.albedo);  // 1 / ir and the two r0^2 (core/shading.hpp)
      out.materials.push_back(f);
    }
    out.perlins = g.perlins;
    for (const GImage& im : g.images) {
      rt::FlatImage f;
      f.width = im.width; f.height = im.height;
      f.first_texel = (int64_t)(out.texels.size() / 3);
      out.texels.insert(out.texels.end(), im.texels.begin(), im.texels.end());
      out.images.push_back(f);
    }
    // An empty world is legal: HittableList::hit returns None and every path sees the background.
    if (!emit_top(world)) return false;
    uint32_t f = 0;
    if (!out.spheres.empty()) f |= rt::F_SPHERE;
    if (!out.moving_spheres.empty()) f |= rt::F_MOVING_SPHERE;
    if (!out.rects.empty()) f |= rt::F_RECT;
    if (!out.triangles.empty()) f |= rt::F_TRIANGLE;
    if (!out.gravity_spheres.empty()) f |= rt::F_GRAVITY_SPHERE;
    for (const rt::FlatEntry& e : out.entries) {
      if (e.kind == rt::ENTRY_PRIM) f |= rt::F_PRIM_ENTRY;
      else if (e.kind == rt::ENTRY_GROUP) f |= rt::F_GROUP;
      else if (e.kind == rt::ENTRY_BVH) f |= rt::F_BVH;
      else if (e.kind == rt::ENTRY_XFORM) f |= rt::F_XFORM;
      else if (e.kind == rt::ENTRY_MEDIUM) {
        f |= rt::F_MEDIUM;
        const rt::FlatEntry& boundary = out.entries[e.a];
        if (boundary.kind == rt::ENTRY_PRIM && rt::primref_type((rt::PrimRef)boundary.a) == rt::PRIM_SPHERE) f |= rt::F_MEDIUM_SPHERE;
        else f |= rt::F_MEDIUM_GENERAL;
      }
    }
    for (const rt::FlatMaterial& m : out.materials) {
      if (m.kind == rt::MAT_LAMBERTIAN) f |= rt::F_LAMBERTIAN;
      else if (m.kind == rt::MAT_METAL) f |= rt::F_METAL;
      else if (m.kind == rt::MAT_DIELECTRIC) f |= rt::F_DIELECTRIC;
      else if (m.kind == rt::MAT_DIFFUSE_LIGHT) f |= rt::F_LIGHT;
      else if (m.kind == rt::MAT_ISOTROPIC) f |= rt::F_ISOTROPIC;
    }
    for (const rt::FlatTexture& t : out.textures) {
      if (t.kind == rt::TEX_CHECKER) f |= rt::F_CHECKER;
      else if (t.kind == rt::TEX_NOISE) f |= rt::F_NOISE;
      else if (t.kind == rt::TEX_IMAGE) f |= rt::F_IMAGE;
    }
    out.features = f;
    // boxes of the plain static primitives in the top-level list (moving spheres are left unbounded: the camera's
    // shutter interval is not known here)
    out.top_box32.assign(6 * out.top_level.size(), 0.0f);
    for (size_t k = 0; k < out.top_level.size(); ++k) {
      float* bx = &out.top_box32[6 * k];
      for (int a = 0; a < 3; ++a) { bx[a] = -INFINITY; bx[3 + a] = INFINITY; }
      const rt::FlatEntry& e = out.entries[out.top_level[k]];
      if (e.kind != rt::ENTRY_PRIM) continue;
      const rt::PrimRef ref = (rt::PrimRef)e.a;
      if (rt::primref_type(ref) == rt::PRIM_MOVING_SPHERE || rt::primref_type(ref) == rt::PRIM_GRAVITY_SPHERE) continue;
      double b[6];
      prim_box(ref, 0.0, 0.0, b);
      for (int a = 0; a < 3; ++a) {
        float lo = (float)b[a];
        if ((double)lo > b[a]) lo = std::nextafterf(lo, -INFINITY);
        float hi = (float)b[3 + a];
        if ((double)hi < b[3 + a]) hi = std::nextafterf(hi, INFINITY);
        if (lo == lo && hi == hi) { bx[a] = lo; bx[3 + a] = hi; }
      }
    }
    // f32 culling copy of every node: lo rounded down, hi rounded up
    out.nodes32.resize(out.nodes.size());
    for (size_t i = 0; i < out.nodes.size(); ++i) {
      const rt::FlatNode& n = out.nodes[i];
      rt::FlatNode32& m = out.nodes32[i];
      for (int c = 0; c < 2; ++c)
        for (int a = 0; a < 3; ++a) {
          float lo = (float)n.bmin[c][a];
          if ((double)lo > n.bmin[c][a]) lo = std::nextafterf(lo, -INFINITY);
          float hi = (float)n.bmax[c][a];
          if ((double)hi < n.bmax[c][a]) hi = std::nextafterf(hi, INFINITY);
          m.lo[c][a] = lo; m.hi[c][a] = hi;
        }
      m.child[0] = n.child[0]; m.child[1] = n.child[1];
      m.axis = n.pad[0]; m.pad = 0;
    }
    build_motion_boxes();
    return true;
  }

  // ---- time-aware culling boxes (FlatMotion32, core/flat_types.hpp)
  struct Box6 { double b[6]; };
  // the box of `child` (a node index or a leaf code) of BVH entry e at the single instant t; fills boxes_at[node][c] on the way
  Box6 motion_box_at(const rt::FlatEntry& e, int32_t child, double t, std::vector<Box6>* at) {
    Box6 r;
    for (int a = 0; a < 3; ++a) { r.b[a] = INFINITY; r.b[3 + a] = -INFINITY; }
    auto grow = [&](const double* b) { for (int a = 0; a < 3; ++a) { r.b[a] = std::fmin(r.b[a], b[a]); r.b[3 + a] = std::fmax(r.b[3 + a], b[3 + a]); } };
    if (rt::node_child_is_leaf(child)) {
      const uint32_t f = rt::leaf_first(child), k = rt::leaf_count(child);
      for (uint32_t i = 0; i < k; ++i) {
        double b[6];
        prim_box(out.refs[(size_t)e.b + f + i], t, t, b);
        grow(b);
      }
      return r;
    }
    for (int c = 0; c < 2; ++c) {
      const Box6 cb = motion_box_at(e, out.nodes[(size_t)child].child[c], t, at);
      (*at)[2 * (size_t)child + c] = cb;
      grow(cb.b);
    }
    return r;
  }
  void build_motion_boxes() {
    out.motion32.clear();
    bool any = false;
    for (const rt::FlatEntry& e : out.entries) {
      if (e.kind != rt::ENTRY_BVH || e.a < 0 || !(e.f[0] < e.f[1])) continue;
      for (int32_t i = 0; i < e.c; ++i) any |= rt::primref_type(out.refs[(size_t)e.b + i]) == rt::PRIM_MOVING_SPHERE;
    }
    if (!any) return;
    // every node starts as its static box (the reference's union over the interval, as in nodes32) with no slope
    out.motion32.resize(out.nodes.size());
    for (size_t i = 0; i < out.nodes.size(); ++i) {
      rt::FlatMotion32& m = out.motion32[i];
      memset(&m, 0, sizeof(m));
      for (int c = 0; c < 2; ++c)
        for (int a = 0; a < 3; ++a) { m.lo0[c][a] = out.nodes32[i].lo[c][a]; m.hi0[c][a] = out.nodes32[i].hi[c][a]; }
    }
    std::vector<Box6> at0(2 * out.nodes.size()), at1(2 * out.nodes.size());
    for (const rt::FlatEntry& e : out.entries) {
      if (e.kind != rt::ENTRY_BVH || e.a < 0 || !(e.f[0] < e.f[1])) continue;
      bool moving = false, linear = true;
      for (int32_t i = 0; i < e.c; ++i) {
        const uint32_t ty = rt::primref_type(out.refs[(size_t)e.b + i]);
        moving |= ty == rt::PRIM_MOVING_SPHERE;
        linear &= ty != rt::PRIM_GRAVITY_SPHERE;  // a bouncing ball is not linear in time: such a BVH keeps its static boxes
      }
      if (!moving || !linear) continue;
      std::vector<int32_t> todo{e.a};
      (void)motion_box_at(e, e.a, e.f[0], &at0);
      (void)motion_box_at(e, e.a, e.f[1], &at1);
      while (!todo.empty()) {
        const int32_t n = todo.back();
        todo.pop_back();
        rt::FlatMotion32& m = out.motion32[(size_t)n];
        for (int c = 0; c < 2; ++c) {
          const Box6 &b0 = at0[2 * (size_t)n + c], &b1 = at1[2 * (size_t)n + c];
          for (int a = 0; a < 3; ++a) {
            const double mag = std::fmax(std::fmax(std::fabs(b0.b[a]), std::fabs(b0.b[3 + a])), std::fmax(std::fabs(b1.b[a]), std::fabs(b1.b[3 + a])));
            const double margin = mag * 0x1.0p-21 + 1e-30;
            auto down = [](double x) { float f = (float)x; return (double)f > x ? std::nextafterf(f, -INFINITY) : f; };
            auto up = [](double x) { float f = (float)x; return (double)f < x ? std::nextafterf(f, INFINITY) : f; };
            const float lo_a = down(b0.b[a] - margin), lo_b = down(b1.b[a] - margin);
            const float hi_a = up(b0.b[3 + a] + margin), hi_b = up(b1.b[3 + a] + margin);
            m.lo0[c][a] = lo_a; m.dlo[c][a] = lo_b - lo_a;
            m.hi0[c][a] = hi_a; m.dhi[c][a] = hi_b - hi_a;
          }
          if (!rt::node_child_is_leaf(out.nodes[(size_t)n].child[c])) todo.push_back(out.nodes[(size_t)n].child[c]);
        }
      }
    }
  }
};

}  // namespace

bool flatten_scene(const SceneGraph& g, int32_t world, const BuildOptions& opt, FlatScene* out,
                   std::string* err) {
  if (!g.valid_hittable(world)) { *err = "flatten: bad world handle"; return false; }
  *out = FlatScene();
  Flattener f(g, *out, opt);
  if (!f.run(world)) { *err = f.err; return false; }
  return true;
}

}  // namespace rtx
