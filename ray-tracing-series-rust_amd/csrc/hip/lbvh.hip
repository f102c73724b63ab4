// GPU BVH builder: Morton-ordered clusters + Karras' binary radix tree + bottom-up refit, on gfx950.
//
// Replaces BvhNode::new (/root/reference/src/bvh.rs:14-83: clones the object vector at every node, O(n^2)) and, at
// mesh scale, this library's own host SAH builder (csrc/host/bvh_build.cpp: ~1 s for 871 200 triangles, longer than
// the frame it is built for).  Only topology comes from here; every box is the exact f64 union of the primitives'
// reference boxes (min / max are exact), so a tree from this builder gives bit-identical images to a tree from any
// other builder: closest hits do not depend on the culling structure (tests/test_gpu_lbvh.py).
//
//   1. k_centroid_bounds   block-reduced min / max of the box centroids                      (n threads)
//   2. k_morton            63-bit Morton code of every centroid (21 bits per axis)           (n threads)
//   3. hipcub radix sort   (code, primitive) pairs                                           (rocPRIM)
//   4. k_radix_tree        leaf k = `cluster` consecutive primitives of the sorted order; one thread per
//                          internal node finds its key range and split (Karras 2012, sec. 3)  (m - 1 threads)
//   5. k_leaf_boxes + k_refit   leaf boxes, then each leaf climbs: the second child to arrive at a node unions the
//                          two child boxes and goes on (one atomic counter per node)          (m threads)
//   6. k_emit_nodes        FlatNode i = boxes and codes of the two children of internal node i, split axis = axis
//                          on which the children's centres are farthest apart; k_depth: deepest leaf
// HBM traffic: a few passes over n x 48 B of boxes and n x 12 B of keys: ~0.3 GB for 871 200 triangles.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <chrono>
#include <string>
#include <vector>
#include "../core/flat_types.hpp"
#include "../host/flat_scene.hpp"

namespace rtx {

namespace {

#define LBVH_TRY(expr)                                                                   \
  do {                                                                                   \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess) { *err = std::string(#expr) + ": " + hipGetErrorString(e_); ok = false; goto done; } \
  } while (0)

struct Box6 { double v[6]; };  // min xyz, max xyz

__device__ __forceinline__ double atomic_min_f64(double* addr, double val) {
  unsigned long long* a = (unsigned long long*)addr;
  unsigned long long old = *a;
  while (__longlong_as_double((long long)old) > val) {
    const unsigned long long assumed = old;
    old = atomicCAS(a, assumed, (unsigned long long)__double_as_longlong(val));
    if (old == assumed) break;
  }
  return __longlong_as_double((long long)old);
}
__device__ __forceinline__ double atomic_max_f64(double* addr, double val) {
  unsigned long long* a = (unsigned long long*)addr;
  unsigned long long old = *a;
  while (__longlong_as_double((long long)old) < val) {
    const unsigned long long assumed = old;
    old = atomicCAS(a, assumed, (unsigned long long)__double_as_longlong(val));
    if (old == assumed) break;
  }
  return __longlong_as_double((long long)old);
}

// bounds[0..2] = min, bounds[3..5] = max of the centroids (bounds pre-set to +inf / -inf)
__global__ __launch_bounds__(256) void k_centroid_bounds(const Box6* __restrict__ boxes, uint32_t n, double* bounds) {
  __shared__ double red[6][256];
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  double c[3] = {0.0, 0.0, 0.0};
  const bool valid = i < n;
  if (valid) for (int a = 0; a < 3; ++a) c[a] = 0.5 * (boxes[i].v[a] + boxes[i].v[3 + a]);
  for (int a = 0; a < 3; ++a) {
    red[a][threadIdx.x] = valid ? c[a] : __builtin_huge_val();
    red[3 + a][threadIdx.x] = valid ? c[a] : -__builtin_huge_val();
  }
  __syncthreads();
  for (uint32_t s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s)
      for (int a = 0; a < 3; ++a) {
        red[a][threadIdx.x] = fmin(red[a][threadIdx.x], red[a][threadIdx.x + s]);
        red[3 + a][threadIdx.x] = fmax(red[3 + a][threadIdx.x], red[3 + a][threadIdx.x + s]);
      }
    __syncthreads();
  }
  if (threadIdx.x < 3) atomic_min_f64(&bounds[threadIdx.x], red[threadIdx.x][0]);
  else if (threadIdx.x < 6) atomic_max_f64(&bounds[threadIdx.x], red[threadIdx.x][0]);
}

__device__ __forceinline__ unsigned long long spread21(unsigned long long x) {  // bit k of x -> bit 3k
  x &= 0x1fffffull;
  x = (x | x << 32) & 0x1f00000000ffffull;
  x = (x | x << 16) & 0x1f0000ff0000ffull;
  x = (x | x << 8) & 0x100f00f00f00f00full;
  x = (x | x << 4) & 0x10c30c30c30c30c3ull;
  x = (x | x << 2) & 0x1249249249249249ull;
  return x;
}
__global__ __launch_bounds__(256) void k_morton(const Box6* __restrict__ boxes, uint32_t n, const double* __restrict__ bounds,
                                               unsigned long long* __restrict__ keys, uint32_t* __restrict__ prims) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  unsigned long long code = 0;
  for (int a = 0; a < 3; ++a) {
    const double lo = bounds[a], hi = bounds[3 + a];
    const double c = 0.5 * (boxes[i].v[a] + boxes[i].v[3 + a]);
    double u = hi > lo ? (c - lo) / (hi - lo) : 0.0;
    u = u < 0.0 ? 0.0 : (u > 1.0 ? 1.0 : u);  // NaN -> 0
    if (!(u == u)) u = 0.0;
    unsigned long long q = (unsigned long long)(u * 2097151.0);
    code |= spread21(q) << (2 - a);
  }
  keys[i] = code;
  prims[i] = i;
}

// Key of leaf k = Morton code of its first primitive; ties are broken by the leaf index (Karras 2012, sec. 4).
__device__ __forceinline__ int lbvh_delta(const unsigned long long* __restrict__ keys, uint32_t cluster, int m, int i, int j) {
  if (j < 0 || j >= m) return -1;
  const unsigned long long a = keys[(size_t)i * cluster], b = keys[(size_t)j * cluster];
  if (a == b) return 64 + __clz((unsigned)(i ^ j));
  return __clzll((long long)(a ^ b));
}
// child code: >= 0 internal node index, < 0: ~leaf index
__global__ __launch_bounds__(256) void k_radix_tree(const unsigned long long* __restrict__ keys, uint32_t cluster, int m,
                                                   int2* __restrict__ children, int* __restrict__ parent_of_node,
                                                   int* __restrict__ parent_of_leaf) {
  const int i = (int)(blockIdx.x * 256u + threadIdx.x);
  if (i >= m - 1) return;
  const int d = lbvh_delta(keys, cluster, m, i, i + 1) - lbvh_delta(keys, cluster, m, i, i - 1) >= 0 ? 1 : -1;
  const int dmin = lbvh_delta(keys, cluster, m, i, i - d);
  int lmax = 2;
  while (lbvh_delta(keys, cluster, m, i, i + lmax * d) > dmin) lmax <<= 1;
  int l = 0;
  for (int t = lmax >> 1; t >= 1; t >>= 1)
    if (lbvh_delta(keys, cluster, m, i, i + (l + t) * d) > dmin) l += t;
  const int j = i + l * d;
  const int dnode = lbvh_delta(keys, cluster, m, i, j);
  int s = 0;
  for (int t = (l + 1) >> 1;; t = (t + 1) >> 1) {
    if (lbvh_delta(keys, cluster, m, i, i + (s + t) * d) > dnode) s += t;
    if (t == 1) break;
  }
  const int gamma = i + s * d + (d < 0 ? -1 : 0);
  const int lo = i < j ? i : j, hi = i < j ? j : i;
  const int left = lo == gamma ? ~gamma : gamma;
  const int right = hi == gamma + 1 ? ~(gamma + 1) : gamma + 1;
  children[i] = make_int2(left, right);
  if (left >= 0) parent_of_node[left] = i; else parent_of_leaf[~left] = i;
  if (right >= 0) parent_of_node[right] = i; else parent_of_leaf[~right] = i;
  if (i == 0) parent_of_node[0] = -1;
}

__global__ __launch_bounds__(256) void k_leaf_boxes(const Box6* __restrict__ boxes, const uint32_t* __restrict__ prims, uint32_t n,
                                                   uint32_t cluster, int m, Box6* __restrict__ leaf_box) {
  const int k = (int)(blockIdx.x * 256u + threadIdx.x);
  if (k >= m) return;
  Box6 b;
  for (int a = 0; a < 3; ++a) { b.v[a] = __builtin_huge_val(); b.v[3 + a] = -__builtin_huge_val(); }
  const uint32_t first = (uint32_t)k * cluster, last = first + cluster < n ? first + cluster : n;
  for (uint32_t p = first; p < last; ++p) {
    const Box6 q = boxes[prims[p]];
    for (int a = 0; a < 3; ++a) { b.v[a] = fmin(b.v[a], q.v[a]); b.v[3 + a] = fmax(b.v[3 + a], q.v[3 + a]); }
  }
  leaf_box[k] = b;
}

__global__ __launch_bounds__(256) void k_refit(const int2* __restrict__ children, const int* __restrict__ parent_of_node,
                                              const int* __restrict__ parent_of_leaf, const Box6* __restrict__ leaf_box, int m,
                                              Box6* node_box, unsigned int* arrivals) {
  const int k = (int)(blockIdx.x * 256u + threadIdx.x);
  if (k >= m) return;
  int node = parent_of_leaf[k];
  while (node >= 0) {
    __threadfence();  // this thread's box stores (previous iteration) before its arrival
    if (atomicAdd(&arrivals[node], 1u) == 0u) return;  // first child to arrive: the sibling's climber finishes the node
    __threadfence();
    const int2 ch = children[node];
    // A sibling's box was written by another CU, possibly on another XCD: agent-scope atomic loads and stores go past
    // the (non-coherent) L1 and write through L2 (MI355X_MICROARCH.md, inter-workgroup visibility: "8-B agent atomics
    // both sides").  Leaf boxes were written by the previous kernel: plain loads.
    Box6 a, b, u;
    for (int c = 0; c < 6; ++c) {
      a.v[c] = ch.x >= 0 ? __hip_atomic_load(&node_box[ch.x].v[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : leaf_box[~ch.x].v[c];
      b.v[c] = ch.y >= 0 ? __hip_atomic_load(&node_box[ch.y].v[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : leaf_box[~ch.y].v[c];
    }
    for (int c = 0; c < 3; ++c) { u.v[c] = fmin(a.v[c], b.v[c]); u.v[3 + c] = fmax(a.v[3 + c], b.v[3 + c]); }
    for (int c = 0; c < 6; ++c) __hip_atomic_store(&node_box[node].v[c], u.v[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    node = parent_of_node[node];
  }
}

__global__ __launch_bounds__(256) void k_emit_nodes(const int2* __restrict__ children, const Box6* __restrict__ node_box,
                                                   const Box6* __restrict__ leaf_box, uint32_t n, uint32_t cluster, int m,
                                                   int32_t base, rt::FlatNode* __restrict__ out) {
  const int i = (int)(blockIdx.x * 256u + threadIdx.x);
  if (i >= m - 1) return;
  const int2 ch = children[i];
  const int c[2] = {ch.x, ch.y};
  rt::FlatNode nd;
  double centre[2][3];
  for (int s = 0; s < 2; ++s) {
    const Box6 b = c[s] >= 0 ? node_box[c[s]] : leaf_box[~c[s]];
    for (int a = 0; a < 3; ++a) { nd.bmin[s][a] = b.v[a]; nd.bmax[s][a] = b.v[3 + a]; centre[s][a] = 0.5 * (b.v[a] + b.v[3 + a]); }
    if (c[s] >= 0) nd.child[s] = base + c[s];
    else {
      const uint32_t first = (uint32_t)(~c[s]) * cluster;
      const uint32_t count = first + cluster < n ? cluster : n - first;
      nd.child[s] = rt::make_leaf(first, count);
    }
  }
  int axis = 0;
  double sep = -1.0;
  for (int a = 0; a < 3; ++a) { const double d = fabs(centre[0][a] - centre[1][a]); if (d > sep) { sep = d; axis = a; } }
  nd.pad[0] = axis;
  nd.pad[1] = 0;
  out[i] = nd;
}

__global__ __launch_bounds__(256) void k_depth(const int* __restrict__ parent_of_node, const int* __restrict__ parent_of_leaf, int m,
                                              int* max_depth) {
  const int k = (int)(blockIdx.x * 256u + threadIdx.x);
  if (k >= m) return;
  int d = 0;
  for (int node = parent_of_leaf[k]; node >= 0; node = parent_of_node[node]) ++d;
  atomicMax(max_depth, d);
}

}  // namespace

// Appends the tree to *nodes (indices are absolute positions in that vector, like the host builders).  Returns the
// root's index, or -1 with *err set.  *order = primitive indices in leaf order; *depth = deepest leaf (an upper bound
// of the stack a walk needs); *device_ms = time on the device (HIP events: upload of the boxes to download of the nodes).
int32_t build_bvh_gpu(const std::vector<double>& boxes, int max_leaf, std::vector<rt::FlatNode>* nodes,
                      std::vector<uint32_t>* order, int32_t* depth, double* device_ms, std::string* err) {
  const size_t n = boxes.size() / 6;
  *depth = 0;
  if (n < 2 || n >= (1u << 28)) { *err = "build_bvh_gpu: primitive count out of range"; return -1; }
  uint32_t cluster = (uint32_t)(max_leaf < 1 ? 1 : (max_leaf > 8 ? 8 : max_leaf));
  if (n <= cluster) cluster = (uint32_t)(n - 1);  // at least two leaves: the root must be a node
  const int m = (int)((n + cluster - 1) / cluster);
  bool ok = true;
  Box6 *d_boxes = nullptr, *d_leaf_box = nullptr, *d_node_box = nullptr;
  double* d_bounds = nullptr;
  unsigned long long *d_keys = nullptr, *d_keys_sorted = nullptr;
  uint32_t *d_prims = nullptr, *d_prims_sorted = nullptr;
  int2* d_children = nullptr;
  int *d_parent_node = nullptr, *d_parent_leaf = nullptr, *d_depth = nullptr;
  unsigned int* d_arrivals = nullptr;
  rt::FlatNode* d_nodes = nullptr;
  void* d_temp = nullptr;
  size_t temp_bytes = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  int32_t root = -1;
  const uint32_t gn = (uint32_t)((n + 255) / 256), gm = (uint32_t)((m + 255) / 256);
  const int32_t base = (int32_t)nodes->size();
  {
    LBVH_TRY(hipEventCreate(&ev0));
    LBVH_TRY(hipEventCreate(&ev1));
    LBVH_TRY(hipMalloc((void**)&d_boxes, n * sizeof(Box6)));
    LBVH_TRY(hipMalloc((void**)&d_bounds, 6 * sizeof(double)));
    LBVH_TRY(hipMalloc((void**)&d_keys, n * 8));
    LBVH_TRY(hipMalloc((void**)&d_keys_sorted, n * 8));
    LBVH_TRY(hipMalloc((void**)&d_prims, n * 4));
    LBVH_TRY(hipMalloc((void**)&d_prims_sorted, n * 4));
    LBVH_TRY(hipMalloc((void**)&d_leaf_box, (size_t)m * sizeof(Box6)));
    LBVH_TRY(hipMalloc((void**)&d_node_box, (size_t)m * sizeof(Box6)));
    LBVH_TRY(hipMalloc((void**)&d_children, (size_t)m * sizeof(int2)));
    LBVH_TRY(hipMalloc((void**)&d_parent_node, (size_t)m * 4));
    LBVH_TRY(hipMalloc((void**)&d_parent_leaf, (size_t)m * 4));
    LBVH_TRY(hipMalloc((void**)&d_arrivals, (size_t)m * 4));
    LBVH_TRY(hipMalloc((void**)&d_depth, 4));
    LBVH_TRY(hipMalloc((void**)&d_nodes, (size_t)(m - 1) * sizeof(rt::FlatNode)));
    LBVH_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, d_keys, d_keys_sorted, d_prims, d_prims_sorted, (int)n, 0, 63));
    LBVH_TRY(hipMalloc(&d_temp, temp_bytes));
    const double init[6] = {__builtin_huge_val(), __builtin_huge_val(), __builtin_huge_val(), -__builtin_huge_val(), -__builtin_huge_val(), -__builtin_huge_val()};
    LBVH_TRY(hipEventRecord(ev0, 0));
    LBVH_TRY(hipMemcpyAsync(d_boxes, boxes.data(), n * sizeof(Box6), hipMemcpyHostToDevice, 0));
    LBVH_TRY(hipMemcpyAsync(d_bounds, init, sizeof(init), hipMemcpyHostToDevice, 0));
    LBVH_TRY(hipMemsetAsync(d_arrivals, 0, (size_t)m * 4, 0));
    LBVH_TRY(hipMemsetAsync(d_depth, 0, 4, 0));
    hipLaunchKernelGGL(k_centroid_bounds, dim3(gn), dim3(256), 0, 0, d_boxes, (uint32_t)n, d_bounds);
    hipLaunchKernelGGL(k_morton, dim3(gn), dim3(256), 0, 0, d_boxes, (uint32_t)n, d_bounds, d_keys, d_prims);
    LBVH_TRY(hipGetLastError());
    LBVH_TRY(hipcub::DeviceRadixSort::SortPairs(d_temp, temp_bytes, d_keys, d_keys_sorted, d_prims, d_prims_sorted, (int)n, 0, 63));
    hipLaunchKernelGGL(k_radix_tree, dim3(gm), dim3(256), 0, 0, d_keys_sorted, cluster, m, d_children, d_parent_node, d_parent_leaf);
    hipLaunchKernelGGL(k_leaf_boxes, dim3(gm), dim3(256), 0, 0, d_boxes, d_prims_sorted, (uint32_t)n, cluster, m, d_leaf_box);
    hipLaunchKernelGGL(k_refit, dim3(gm), dim3(256), 0, 0, d_children, d_parent_node, d_parent_leaf, d_leaf_box, m, d_node_box, d_arrivals);
    hipLaunchKernelGGL(k_emit_nodes, dim3(gm), dim3(256), 0, 0, d_children, d_node_box, d_leaf_box, (uint32_t)n, cluster, m, base, d_nodes);
    hipLaunchKernelGGL(k_depth, dim3(gm), dim3(256), 0, 0, d_parent_node, d_parent_leaf, m, d_depth);
    LBVH_TRY(hipGetLastError());
    nodes->resize((size_t)base + (size_t)(m - 1));
    order->resize(n);
    LBVH_TRY(hipMemcpyAsync(nodes->data() + base, d_nodes, (size_t)(m - 1) * sizeof(rt::FlatNode), hipMemcpyDeviceToHost, 0));
    LBVH_TRY(hipMemcpyAsync(order->data(), d_prims_sorted, n * 4, hipMemcpyDeviceToHost, 0));
    int h_depth = 0;
    LBVH_TRY(hipMemcpyAsync(&h_depth, d_depth, 4, hipMemcpyDeviceToHost, 0));
    LBVH_TRY(hipEventRecord(ev1, 0));
    LBVH_TRY(hipEventSynchronize(ev1));
    float ms = 0.f;
    LBVH_TRY(hipEventElapsedTime(&ms, ev0, ev1));
    if (device_ms) *device_ms += (double)ms;
    *depth = h_depth;
    root = base;  // Karras' root is internal node 0
  }
done:
  if (!ok) nodes->resize((size_t)base);
  for (void* p : {(void*)d_boxes, (void*)d_leaf_box, (void*)d_node_box, (void*)d_bounds, (void*)d_keys, (void*)d_keys_sorted,
                  (void*)d_prims, (void*)d_prims_sorted, (void*)d_children, (void*)d_parent_node, (void*)d_parent_leaf,
                  (void*)d_depth, (void*)d_arrivals, (void*)d_nodes, d_temp})
    if (p) (void)hipFree(p);
  if (ev0) (void)hipEventDestroy(ev0);
  if (ev1) (void)hipEventDestroy(ev1);
  return ok ? root : -1;
}

}  // namespace rtx
