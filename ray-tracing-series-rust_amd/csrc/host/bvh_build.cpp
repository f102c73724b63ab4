// O(n log n) binned-SAH BVH builder (host).
//
// Replaces BvhNode::new (/root/reference/src/bvh.rs:14-83), which clones the whole
// object vector at every node (O(n^2)), picks a random axis in {x,y} and splits at
// the median.  Only the SET of primitives and their boxes matter for results: the
// BVH is a culling structure, closest-hit answers do not depend on its topology.
// Output layout: core/flat_types.hpp FlatNode (a node holds its two children's boxes).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include "../core/rng.hpp"
#include "flat_scene.hpp"

namespace rtx {

namespace {

struct Box {
  double mn[3], mx[3];
  void reset() {
    for (int a = 0; a < 3; ++a) { mn[a] = std::numeric_limits<double>::infinity(); mx[a] = -mn[a]; }
  }
  void grow(const double* b) {
    for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], b[a]); mx[a] = std::max(mx[a], b[3 + a]); }
  }
  void grow(const Box& o) {
    for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], o.mn[a]); mx[a] = std::max(mx[a], o.mx[a]); }
  }
  double half_area() const {
    double dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
    if (!(dx >= 0.0) || !(dy >= 0.0) || !(dz >= 0.0)) return 0.0;
    return dx * dy + dy * dz + dz * dx;
  }
};

struct Builder {
  const std::vector<double>& boxes;   // what the nodes store
  const std::vector<double>& tboxes;  // what the topology is chosen by (the same boxes unless the caller has better ones)
  const BuildOptions& opt;
  std::vector<rt::FlatNode>* nodes;
  std::vector<uint32_t>& order;
  std::vector<double> centroids;  // 3 per primitive
  double sah = 0.0;

  Builder(const std::vector<double>& b, const std::vector<double>& tb, const BuildOptions& o, std::vector<rt::FlatNode>* n,
          std::vector<uint32_t>& ord)
      : boxes(b), tboxes(tb), opt(o), nodes(n), order(ord) {}

  // Returns the child code for [begin, end) and its box; *depth = internal depth below.
  int32_t build(uint32_t begin, uint32_t end, Box* out_box, int32_t* depth) {
    uint32_t n = end - begin;
    Box bb; bb.reset();   // stored box
    Box cb; cb.reset();
    for (uint32_t i = begin; i < end; ++i) {
      bb.grow(&boxes[6 * (size_t)order[i]]);
      const double* c = &centroids[3 * (size_t)order[i]];
      for (int a = 0; a < 3; ++a) { cb.mn[a] = std::min(cb.mn[a], c[a]); cb.mx[a] = std::max(cb.mx[a], c[a]); }
    }
    *out_box = bb;
    if ((int)n <= opt.max_leaf) {
      *depth = 0;
      sah += bb.half_area() * n;
      return rt::make_leaf(begin, n);
    }
    // ---- binned SAH over the three axes
    const int K = std::max(4, std::min(opt.sah_bins, 64));
    double best_cost = std::numeric_limits<double>::infinity();
    int best_axis = -1, best_split = -1;
    for (int a = 0; a < 3; ++a) {
      double lo = cb.mn[a], hi = cb.mx[a];
      if (!(hi > lo)) continue;
      double scale = (double)K / (hi - lo);
      Box bin_box[64];
      uint32_t bin_cnt[64];
      for (int k = 0; k < K; ++k) { bin_box[k].reset(); bin_cnt[k] = 0; }
      for (uint32_t i = begin; i < end; ++i) {
        double c = centroids[3 * (size_t)order[i] + a];
        int k = (int)((c - lo) * scale);
        if (k < 0) k = 0;
        if (k >= K) k = K - 1;
        bin_box[k].grow(&tboxes[6 * (size_t)order[i]]);
        bin_cnt[k]++;
      }
      double right_area[64];
      uint32_t right_cnt[64];
      Box acc; acc.reset();
      uint32_t cnt = 0;
      for (int k = K - 1; k >= 1; --k) {
        acc.grow(bin_box[k]); cnt += bin_cnt[k];
        right_area[k] = acc.half_area(); right_cnt[k] = cnt;
      }
      acc.reset(); cnt = 0;
      for (int k = 0; k < K - 1; ++k) {
        acc.grow(bin_box[k]); cnt += bin_cnt[k];
        if (cnt == 0 || right_cnt[k + 1] == 0) continue;
        double cost = acc.half_area() * cnt + right_area[k + 1] * right_cnt[k + 1];
        if (cost < best_cost) { best_cost = cost; best_axis = a; best_split = k; }
      }
    }
    uint32_t mid;
    int axis_used = 0;
    if (best_axis >= 0) {
      double lo = cb.mn[best_axis], hi = cb.mx[best_axis];
      double scale = (double)K / (hi - lo);
      const int a = best_axis, split = best_split;
      auto first_right = std::partition(order.begin() + begin, order.begin() + end, [&](uint32_t p) {
        int k = (int)((centroids[3 * (size_t)p + a] - lo) * scale);
        if (k < 0) k = 0;
        if (k >= K) k = K - 1;
        return k <= split;
      });
      mid = (uint32_t)(first_right - order.begin());
      axis_used = a;
    } else {
      mid = begin;  // all centroids coincide
    }
    if (mid == begin || mid == end) {
      // Degenerate: median split on the widest box axis keeps the depth logarithmic.
      int a = 0;
      double ext = -1.0;
      for (int k = 0; k < 3; ++k) { double e = bb.mx[k] - bb.mn[k]; if (e > ext) { ext = e; a = k; } }
      mid = begin + n / 2;
      std::nth_element(order.begin() + begin, order.begin() + mid, order.begin() + end,
                       [&](uint32_t p, uint32_t q) {
                         double cp = centroids[3 * (size_t)p + a], cq = centroids[3 * (size_t)q + a];
                         return cp < cq || (cp == cq && p < q);
                       });
      axis_used = a;
    }
    size_t idx = nodes->size();
    nodes->emplace_back();
    Box b0, b1;
    int32_t d0 = 0, d1 = 0;
    int32_t c0 = build(begin, mid, &b0, &d0);
    int32_t c1 = build(mid, end, &b1, &d1);
    // A single primitive whose box is at least as big as everything beside it (Book-1's r = 1000 ground next to 483 small
    // spheres) is asked FIRST by every ray, whatever its direction: one primitive test bounds the ray before the sibling
    // subtree is walked (a downward ray otherwise walks the whole sphere field with an unbounded interval and meets the ground
    // last), and because every ray of a wave takes it at the same step, that test runs with all lanes.  Encoded as split
    // "axis" 3: the walkers pick the near child as (dir_neg >> axis) & 1 and dir_neg has three bits, so child 0 is always first.
    if (opt.leaf_first) {
      const bool l0 = rt::node_child_is_leaf(c0) && rt::leaf_count(c0) == 1, l1 = rt::node_child_is_leaf(c1) && rt::leaf_count(c1) == 1;
      if (l1 && !l0 && b1.half_area() >= b0.half_area()) { std::swap(c0, c1); std::swap(b0, b1); axis_used = 3; }
      else if (l0 && !l1 && b0.half_area() >= b1.half_area()) axis_used = 3;
    }
    rt::FlatNode& nd = (*nodes)[idx];
    for (int a = 0; a < 3; ++a) {
      nd.bmin[0][a] = b0.mn[a]; nd.bmax[0][a] = b0.mx[a];
      nd.bmin[1][a] = b1.mn[a]; nd.bmax[1][a] = b1.mx[a];
    }
    nd.child[0] = c0; nd.child[1] = c1;
    nd.pad[0] = axis_used;  // split axis: child 0 holds the lower centroids along it (3: child 0 is always visited first)
    nd.pad[1] = 0;
    *depth = 1 + std::max(d0, d1);
    sah += bb.half_area();
    return (int32_t)idx;
  }
};

}  // namespace

int32_t build_bvh(const std::vector<double>& boxes, const BuildOptions& opt,
                  std::vector<rt::FlatNode>* nodes, std::vector<uint32_t>* order, int32_t* depth,
                  double* sah_cost, const std::vector<double>* topology_boxes) {
  const std::vector<double>& tboxes = (topology_boxes && topology_boxes->size() == boxes.size()) ? *topology_boxes : boxes;
  size_t n = boxes.size() / 6;
  order->resize(n);
  for (size_t i = 0; i < n; ++i) (*order)[i] = (uint32_t)i;
  *depth = 0;
  if (n < 2) return -1;
  BuildOptions o = opt;
  if (o.max_leaf < 1) o.max_leaf = 1;
  if (o.max_leaf > 8) o.max_leaf = 8;
  Builder b(boxes, tboxes, o, nodes, *order);
  b.centroids.resize(3 * n);
  for (size_t i = 0; i < n; ++i)
    for (int a = 0; a < 3; ++a) b.centroids[3 * i + a] = 0.5 * (tboxes[6 * i + a] + tboxes[6 * i + 3 + a]);
  // Force at least one internal node even when n <= max_leaf, so the root is a node.
  BuildOptions forced = o;
  if ((int)n <= o.max_leaf) forced.max_leaf = (int)n - 1;
  Builder bf(boxes, tboxes, forced, nodes, *order);
  bf.centroids.swap(b.centroids);
  Box root_box;
  int32_t root = bf.build(0, (uint32_t)n, &root_box, depth);
  if (sah_cost) *sah_cost += bf.sah;
  return root;
}

// bvh.rs:14-83, node for node (see flat_scene.hpp).  The reference clones the object vector at
// every node and sorts the [start, end) slice of its clone; sorting disjoint slices of one index
// vector in place yields the same tree.
namespace {
struct RefBuilder {
  const std::vector<double>& boxes;
  const std::vector<double>& sort_boxes;
  std::vector<rt::FlatNode>* nodes;
  std::vector<uint32_t>& idx;
  rt::HostRng rng;
  int32_t build(uint32_t start, uint32_t end, Box* out, int32_t* depth) {
    int axis = (int)rt::host_rng_below(rng, 2);  // bvh.rs:24: gen_range(0..2), z is never chosen
    auto less = [&](uint32_t a, uint32_t b) { return sort_boxes[6 * (size_t)a + axis] < sort_boxes[6 * (size_t)b + axis]; };
    uint32_t span = end - start;
    size_t me = nodes->size();
    nodes->emplace_back();
    Box b0, b1;
    int32_t c0, c1, d0 = 0, d1 = 0;
    auto leaf = [&](uint32_t slot, Box* bx) { bx->reset(); bx->grow(&boxes[6 * (size_t)idx[slot]]); return rt::make_leaf(slot, 1); };
    if (span == 1) {          // bvh.rs:53-55: left = right = the same object
      c0 = leaf(start, &b0); c1 = leaf(start, &b1);
    } else if (span == 2) {   // bvh.rs:56-63
      if (!less(idx[start], idx[start + 1])) std::swap(idx[start], idx[start + 1]);
      c0 = leaf(start, &b0); c1 = leaf(start + 1, &b1);
    } else {                  // bvh.rs:64-68
      std::stable_sort(idx.begin() + start, idx.begin() + end, less);
      uint32_t mid = start + span / 2;
      c0 = build(start, mid, &b0, &d0);
      c1 = build(mid, end, &b1, &d1);
    }
    rt::FlatNode& nd = (*nodes)[me];
    for (int a = 0; a < 3; ++a) {
      nd.bmin[0][a] = b0.mn[a]; nd.bmax[0][a] = b0.mx[a];
      nd.bmin[1][a] = b1.mn[a]; nd.bmax[1][a] = b1.mx[a];
    }
    nd.child[0] = c0; nd.child[1] = c1;
    nd.pad[0] = axis; nd.pad[1] = 0;
    *out = b0; out->grow(b1);
    *depth = 1 + std::max(d0, d1);
    return (int32_t)me;
  }
};
}  // namespace

int32_t build_bvh_reference(const std::vector<double>& boxes, const std::vector<double>& sort_boxes,
                            uint64_t seed, std::vector<rt::FlatNode>* nodes, std::vector<uint32_t>* order,
                            int32_t* depth) {
  size_t n = boxes.size() / 6;
  order->resize(n);
  for (size_t i = 0; i < n; ++i) (*order)[i] = (uint32_t)i;
  *depth = 0;
  if (n < 1) return -1;
  RefBuilder rb{boxes, sort_boxes, nodes, *order, rt::HostRng{seed}};
  Box root;
  return rb.build(0, (uint32_t)n, &root, depth);
}

}  // namespace rtx
