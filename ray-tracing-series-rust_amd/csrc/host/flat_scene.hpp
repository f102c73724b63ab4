// Host-owned flattened scene (`rtx_flat` behind the C ABI): the arrays of
// core/flat_types.hpp in std::vectors, plus build statistics.
#pragma once
#include <string>
#include <vector>
#include "../core/flat_types.hpp"
#include "scene_graph.hpp"

namespace rtx {

struct FlatScene {
  std::vector<rt::FlatSphere> spheres;
  std::vector<rt::FlatMovingSphere> moving_spheres;
  std::vector<rt::FlatRect> rects;
  std::vector<rt::FlatTriangle> triangles;
  std::vector<rt::FlatNode> nodes;
  std::vector<rt::FlatNode32> nodes32;  // same tree, boxes rounded outward to f32 (core/cull32.hpp)
  std::vector<rt::FlatMotion32> motion32;  // time-aware boxes (empty unless a BVH holds MovingSpheres), parallel to nodes
  std::vector<rt::PrimRef> refs;
  std::vector<rt::FlatEntry> entries;
  std::vector<int32_t> top_level;
  std::vector<rt::FlatMaterial> materials;
  std::vector<rt::FlatTexture> textures;
  std::vector<rt::FlatPerlin> perlins;
  std::vector<rt::FlatImage> images;
  std::vector<rt::real> texels;
  std::vector<float> top_box32;  // 6 per top-level slot (SceneView::top_box32)
  std::vector<rt::FlatGravitySphere> gravity_spheres;
  std::vector<rt::real> gravity_y;
  int32_t max_stack = 0;      // deepest BVH (number of stacked far children a walk can hold)
  int32_t n_bvh = 0;
  uint32_t features = 0;      // rt::Feature bits reachable in this scene
  double sah_cost = 0.0;      // summed SAH cost of all BVHs (diagnostic)
  double bvh_build_ms = 0.0;  // wall time of all BVH builds (host or GPU builder)
  double bvh_device_ms = 0.0; // GPU builder only: device time (HIP events), uploads and downloads included

  // pointers into the vectors above (host memory)
  rt::SceneView view() const {
    rt::SceneView v;
    v.spheres = spheres.data();
    v.moving_spheres = moving_spheres.data();
    v.rects = rects.data();
    v.triangles = triangles.data();
    v.nodes = nodes.data();
    v.nodes32 = nodes32.data();
    v.motion32 = motion32.empty() ? nullptr : motion32.data();
    v.refs = refs.data();
    v.entries = entries.data();
    v.top_level = top_level.data();
    v.materials = materials.data();
    v.textures = textures.data();
    v.perlins = perlins.data();
    v.images = images.data();
    v.texels = texels.data();
    v.top_box32 = top_box32.empty() ? nullptr : top_box32.data();
    v.gravity_spheres = gravity_spheres.data();
    v.gravity_y = gravity_y.data();
    v.n_top_level = (int32_t)top_level.size();
    v.max_stack = max_stack;
    v.features = features;
    v.pad = 0;
    return v;
  }
  size_t total_bytes() const;
};

struct BuildOptions {
  int max_leaf = 0;   // primitives per BVH leaf (1..8); 0 = per BVH: 2 when it holds triangles, else 1 (measured)
  int sah_bins = 32;
  // 1: build every BVH with the REFERENCE's rule instead of SAH (bvh.rs:14-83: random axis in {x, y},
  // stable sort by the time-(0,0) box minimum, median split, one object per leaf, a span of one stored
  // twice).  Same image, different traversal statistics -- an A/B switch.
  int reference_bvh = 0;
  uint64_t bvh_seed = 1;
  // 1: BVHs of >= 1024 primitives are built on the current GPU (csrc/hip/lbvh.hip: Morton clusters + radix tree + refit)
  // instead of by the host SAH builder: same image, a tree of lower quality, built in milliseconds.
  int gpu_builder = 0;
  // 1 (default): a single-primitive leaf whose box is at least as big as its sibling's is visited first by every ray
  // (bvh_build.cpp); 0 restores the plain near-child-by-split-axis order for A/B (RTX_LEAF_FIRST=0).
  int leaf_first = 1;
  // 1 (default): a BVH that holds MovingSpheres is partitioned by its primitives' boxes at the middle of its time interval
  // instead of by their boxes over the interval (flatten.cpp); RTX_MOTION_TOPOLOGY=0 for A/B.
  int motion_topology = 1;
};

// Flatten `world` (any Hittable handle of `g`).  Returns false and sets *err when the graph
// uses a nesting the kernels do not implement ("scene shapes").
bool flatten_scene(const SceneGraph& g, int32_t world, const BuildOptions& opt, FlatScene* out,
                   std::string* err);

// Binned-SAH BVH over axis-aligned boxes.  boxes = 6 doubles (min xyz, max xyz) per
// primitive.  Emits nodes (appended to *nodes, child indices absolute) and the
// permutation `order` (slot -> input primitive).  Returns the root node index, or -1 if
// n < 2 (the caller emits a GROUP instead).  *depth = the stack depth a walk needs.
// topology_boxes (optional, same count): the boxes the splits are CHOSEN by, when they differ from the boxes the nodes store --
// a BVH of moving spheres stores the reference's boxes over its whole time interval but is better partitioned by where the
// spheres are at one instant (flatten.cpp).
int32_t build_bvh(const std::vector<double>& boxes, const BuildOptions& opt,
                  std::vector<rt::FlatNode>* nodes, std::vector<uint32_t>* order, int32_t* depth,
                  double* sah_cost, const std::vector<double>* topology_boxes = nullptr);
// The GPU builder (csrc/hip/lbvh.hip).  Same contract; -1 with *err set when it cannot run (no GPU, allocation failure).
int32_t build_bvh_gpu(const std::vector<double>& boxes, int max_leaf, std::vector<rt::FlatNode>* nodes,
                      std::vector<uint32_t>* order, int32_t* depth, double* device_ms, std::string* err);
// The reference's builder.  `sort_boxes` are the bounding_box(0.0, 0.0) boxes its comparator uses
// (bvh.rs:27-28), `boxes` the (time0, time1) boxes the node bounds are made of (bvh.rs:71-78).
int32_t build_bvh_reference(const std::vector<double>& boxes, const std::vector<double>& sort_boxes,
                            uint64_t seed, std::vector<rt::FlatNode>* nodes, std::vector<uint32_t>* order,
                            int32_t* depth);

}  // namespace rtx
