// HIP kernels and launcher for the per-pixel x per-sample loop of render_scene
// (/root/reference/src/world.rs:1207-1226) on gfx950.
//
//   k_trace_*        one camera path per (pixel, sample): path_begin + path_step loop
//                    (core/integrator.hpp), radiance written to the pass's sample buffer
//   k_reduce_samples per pixel, adds the pass's samples IN SAMPLE ORDER onto the accumulator
//                    (the reference's `pixel += ray_color(..)` order, world.rs:1215, so sums
//                    are bit-identical to a sequential CPU loop whatever the scheduling was)
//   k_tonemap        get_normalized_color (vec3.rs:89-107) -> RGB8
//
// No CPU fallback exists in this library: without a GPU every render entry point returns
// RTX_EHIP.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <new>
#include <type_traits>
#include <utility>
#include <string>
#include <vector>
#include "../../../include/rtx_abi.h"
#include "../core/cull32.hpp"
#include "../core/integrator.hpp"
#include "../host/flat_scene.hpp"
#include "f32_bridge.hpp"
#ifndef RTX_F32_TU
#include "abi_internal.hpp"
#else
// The f32 compilation of this file (render_f32.hip) sees none of the C ABI's handle types: it exports the four
// functions of f32_bridge.hpp and render.hip proper owns the handles.
namespace rtx { void set_error(const std::string& msg); }
#endif

namespace rtx {

// ------------------------------------------------------------------ device scene
#define TRACE_CHUNK_DEFAULT 512u
struct FlatNode4;
typedef const FlatNode4 FlatNode4Dev;
struct WorldDesc;

struct LdsSceneDims {  // what k_trace_lds (trace_lds.inc) copies into LDS
  uint32_t n_nodes, n_refs, n_spheres, n_moving;  // n_refs = leaf slots; one record per slot, in slot order: n_spheres or n_moving = n_refs
  uint32_t node_dwords;  // LDSK_NODE_DWORDS, or LDSK_MOTION_NODE_DWORDS for the time-aware instantiation
  uint32_t n_uni;        // scenes with moving spheres: how many of the n_moving records are the scene's STATIC spheres, kept as moving spheres that stand still (n_spheres = 0 then; informational: the kernel decides per slot)
};

// The plain primitive entries beside the BVH in the world list (the dragon room's seven rectangles), as a kernel argument:
// list position, primitive reference and -- when all of them are rectangles -- the records themselves, for up to 8 entries.
// Read from the world list they cost three DEPENDENT scalar loads each, per ray (top_level[k] -> entries[..].a -> the
// primitive record); from the kernarg segment a record is one load at base + i * stride.  n < 0: not applicable, the kernel
// walks the list.  (Measured: the same loop fully unrolled over the 8 slots is 8 % SLOWER than the list walk -- eight copies
// of the three-axis rectangle test do not fit the instruction cache next to the rest of the kernel.)
struct VoteTop {
  int32_t n;
  int32_t all_rects;
  int32_t k[8];
  uint32_t ref[8];
  rt::FlatRect rect[8];
};

struct DeviceScene {
  int device = -1;
  std::vector<void*> allocations;
  rt::SceneView view;   // device pointers
  size_t scene_bytes = 0;
  int32_t n_nodes = 0;
  // workspace (grown on demand by render calls)
  double* samples = nullptr;
  size_t samples_bytes = 0;
  double* accum = nullptr;
  size_t accum_bytes = 0;
  rt::TraceCounters* counters = nullptr;
  unsigned int* work_counter = nullptr;
  hipEvent_t ev[2] = {nullptr, nullptr};
  int n_cu = 256;
  int blocks_per_cu[3] = {1, 1, 1};  // resident 256-thread blocks per CU for each preset's persistent kernel
  bool force_simple = false;          // RTX_TRACE_KERNEL=simple
  bool force_persistent = false;      // RTX_TRACE_KERNEL=persistent
  bool force_stream = false;          // RTX_TRACE_KERNEL=stream
  bool force_vote = false;            // RTX_TRACE_KERNEL=vote
  bool vote_diag = false;             // RTX_TRACE_KERNEL=vote_diag: occupancy counters on stderr (never timed)
  unsigned long long* diag = nullptr;
  int vote_blocks_per_cu[2] = {1, 1};
  bool vote_ring[2] = {false, false}; // k_trace_vote keeps a ring of ready primary rays in LDS (when it costs no occupancy)
  bool single_bvh = false;            // world == one BVH entry -> k_trace_lds / k_trace_stream / k_trace_wq apply
  FlatNode4Dev* nodes4 = nullptr;     // 4-wide culling tree of the BVH entry (big triangle meshes), built at upload
  int wide_levels = 0, wide_blocks_per_cu = 1;
  int wide_pers_blocks_per_cu[3] = {1, 1, 1};
  bool vote_ok = false;               // world == one BVH entry + plain primitive entries -> k_trace_vote applies
  int32_t vote_bvh_pos = 0;           // position of the BVH entry in the top-level list
  int32_t vote_tri_base = -1;         // >= 0: that BVH is a pure triangle mesh whose slot s is triangle vote_tri_base + s
  int stream_blocks_per_cu[2] = {1, 1};
  uint32_t walk_threshold = 18;       // RTX_WALK_THRESHOLD (1 = never carry a walk over); 18 measured best on C2 (12..22 within 1 %)
  uint32_t regen_min = 1;             // wide k_trace_vote: lanes that must be waiting before the wave regenerates (RTX_REGEN_MIN)
  bool single_leaf = false;           // every BVH leaf holds one primitive (k_trace_lds tests it without a loop)
  uint32_t leaf_weight = 3;           // RTX_LEAF_WEIGHT: node lanes x weight >= leaf lanes -> node step (default: leaf size + 1... see upload)
  // the same plan for the time-aware instantiation (bigger node records: its own ring size); chosen per render, when the camera's
  // shutter lies inside the BVH's time interval [motion_t0, motion_t1] (RTX_MOTION=0 turns it off)
  // ... and for the 4-wide collapse of the tree (static worlds; RTX_LDS_WIDE=0 turns it off): its node image, ring and stack depth
  bool w4_ok = false, w4_ring = false;
  uint32_t w4_ring_cap = 0, w4_levels = 0;
  LdsSceneDims w4_dims = {0, 0, 0, 0, 0, 0};
  const uint32_t* w4_image = nullptr;
  bool motion_ok = false, motion_ring = false;
  uint32_t motion_ring_cap = 0;
  LdsSceneDims motion_dims = {0, 0, 0, 0, 0, 0};
  int motion_axis = -1;               // 0 / 1 / 2: every slope of the time-aware boxes is zero except along this axis (-1: no such axis)
  double motion_t0 = 0.0, motion_t1 = 0.0;
  // Passes of one render pipelined two deep (render_impl): odd passes run on an internal stream with their own half of the sample
  // buffer and their own work counter, so the tail, the reduction of pass k and the start of pass k + 1 overlap.  RTX_PASS_PIPELINE=0: off.
  hipStream_t aux_stream = nullptr;
  hipEvent_t ev_pass[3] = {nullptr, nullptr, nullptr};  // reduction of an even / odd pass done; start of the render
  bool pass_pipeline = true;
  bool mv_common = false;             // every MovingSphere of the scene has the same (time0, time1) = (mv_t0, mv_t1): k_trace_lds divides once per bounce (RTX_MV_COMMON=0: off)
  double mv_t0 = 0.0, mv_t1 = 1.0;
  bool lds_ok = false;                // scene geometry fits in LDS -> k_trace_lds (RTX_SCENE_LDS=0 turns it off)
  bool lds_ring = false;
  uint32_t lds_ring_cap = 64;         // entries per wave ring: 64, or 48 / 32 when the scene leaves less LDS
  uint32_t lds_chunk = TRACE_CHUNK_DEFAULT;  // sample indices a wave reserves per grab (RTX_CHUNK)
  LdsSceneDims lds_dims = {0, 0, 0, 0, 0, 0};
  bool force_wq = false;              // RTX_TRACE_KERNEL=wq: workgroup-queue kernel (trace_wq.inc)
  bool wq_diag = false;               // RTX_TRACE_KERNEL=wq_diag: stage occupancy counters on stderr (never timed)
  bool wq_ok = false;                 // world fits k_trace_wq's 16-bit work items and LDS budget
  uint32_t wq_paths = 0, wq_levels = 0, wq_walkers = 12, wq_batch_min = 48;
  unsigned int* error_word = nullptr;
  bool force_world = false;           // RTX_TRACE_KERNEL=world: k_trace_world even where a more special kernel applies (A/B)
  bool world_diag = false;            // RTX_TRACE_KERNEL=world_diag: region counters of k_trace_world on stderr (never timed)
  uint32_t world_threshold = 8;       // k_trace_world: walk steps have priority while this many lanes walk (RTX_WORLD_THRESHOLD; 0 = plain majority vote)
  int world_blocks_per_cu[4][2] = {{1, 1}, {1, 1}, {1, 1}, {1, 1}};  // [book2 preset / any / all incl. gravity spheres / no sphere media][binary / wide]
  const struct WorldDesc* world_desc = nullptr;       // per-slot records of the world list for k_trace_world
  VoteTop vote_top;                   // the plain entries beside the BVH, for k_trace_vote's kernarg (n = -1: not applicable)
  double gravity_time_limit = 1e300;  // scenes with GravitySpheres: the largest shutter time a render accepts (set at upload)
  uint32_t vote_tables = 0;           // wide k_trace_vote: materials | textures << 16 to keep in LDS (0: none; RTX_MAT_LDS=0)
  size_t vote_tables_bytes = 0;
  uint32_t world_mat_lds = 0, world_tex_lds = 0;  // material / texture records k_trace_world copies into LDS (RTX_MAT_LDS=0: none)
  uint32_t world_perlin_lds = 0;      // Perlin tables k_trace_world copies into LDS (RTX_PERLIN_LDS=0: none)
  // wavefront integrator (trace_wave.inc): path pool in HBM, grown on demand by render calls
  bool wave_ok = false;               // world == one BVH + plain primitive entries, sphere / mesh preset
  bool force_wave = false;            // RTX_TRACE_KERNEL=wavefront
  bool wave_default = false;          // the launcher prefers it for this scene (set at upload)
  uint32_t wave_paths = 1u << 22;     // RTX_WF_PATHS: path slots P
  uint32_t wave_refill = 16;          // RTX_WF_REFILL: free lanes a wave waits for before it takes new slots
  uint32_t wave_check = 8;            // RTX_WF_CHECK: iterations between two looks at the counters
  void* wave_mem = nullptr;
  size_t wave_bytes = 0;
  uint32_t* wave_host_ctrl = nullptr; // pinned
  int wave_blocks_per_cu = 1;
  int wave_iterations = 0;            // of the last render (diagnostics)
  int wave_occ = 4;                   // RTX_WF_OCC: 256-thread blocks per CU k_wf_trace is compiled for (4 / 5 / 6; mesh room)
  bool wave_verbose = false;          // RTX_WF_VERBOSE
};

#define HIP_TRY(expr)                                                                      \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess) {                                                                \
      set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                        \
      return RTX_EHIP;                                                                     \
    }                                                                                      \
  } while (0)

template <class T>
static rtx_status upload_array(DeviceScene* ds, const std::vector<T>& v, const T** out) {
  *out = nullptr;
  if (v.empty()) return RTX_OK;
  void* p = nullptr;
  size_t bytes = v.size() * sizeof(T);
  HIP_TRY(hipMalloc(&p, bytes));
  ds->allocations.push_back(p);
  HIP_TRY(hipMemcpy(p, v.data(), bytes, hipMemcpyHostToDevice));
  ds->scene_bytes += bytes;
  *out = (const T*)p;
  return RTX_OK;
}

static void free_device_scene(DeviceScene* ds) {
  if (!ds) return;
  for (void* p : ds->allocations) (void)hipFree(p);
  if (ds->samples) (void)hipFree(ds->samples);
  if (ds->accum) (void)hipFree(ds->accum);
  if (ds->counters) (void)hipFree(ds->counters);
  if (ds->work_counter) (void)hipFree(ds->work_counter);
  if (ds->diag) (void)hipFree(ds->diag);
  if (ds->error_word) (void)hipFree(ds->error_word);
  if (ds->wave_mem) (void)hipFree(ds->wave_mem);
  if (ds->wave_host_ctrl) (void)hipHostFree(ds->wave_host_ctrl);
  for (int i = 0; i < 2; ++i)
    if (ds->ev[i]) (void)hipEventDestroy(ds->ev[i]);
  for (int i = 0; i < 3; ++i)
    if (ds->ev_pass[i]) (void)hipEventDestroy(ds->ev_pass[i]);
  if (ds->aux_stream) (void)hipStreamDestroy(ds->aux_stream);
  delete ds;
}

// Load through the constant address space: the address is wave-uniform, the compiler may use a scalar load.
template <class T>
__device__ __forceinline__ T dev_load_uniform(const T* p) { return *(const __attribute__((address_space(4))) T*)(p); }

// 64-bit lane mask of a predicate, straight from the compare (HIP's __ballot(int) first materialises 0/1 in a VGPR).
__device__ __forceinline__ unsigned long long wave_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// ------------------------------------------------------------------ pixel enumeration
// Local pixel lp of a shard -> image (column i, row j).  Rows of a shard are the rows j with
// (j / block_rows) % shard_count == shard_index, compacted in ascending j.
struct ShardMap {
  int32_t width, block_rows, shard_index, shard_count;
};
__device__ __forceinline__ void shard_pixel(const ShardMap& m, uint32_t lp, uint32_t* i, uint32_t* j) {
  uint32_t lr = lp / (uint32_t)m.width;
  *i = lp - lr * (uint32_t)m.width;
  uint32_t k = lr / (uint32_t)m.block_rows;
  uint32_t within = lr - k * (uint32_t)m.block_rows;
  *j = (k * (uint32_t)m.shard_count + (uint32_t)m.shard_index) * (uint32_t)m.block_rows + within;
}

// ------------------------------------------------------------------ LDS traversal stack
// Slot (level, lane) lives at base[level * TRACE_BLOCK]: lanes of a wave touch consecutive
// dwords -> conflict-free ds_write_b32 / ds_read_b32.
#define TRACE_BLOCK 256
struct LdsStack {
  int32_t* base;
  int n;
  static constexpr bool kBottom = false;
  __device__ __forceinline__ void reset() { n = 0; }
  __device__ __forceinline__ void push(int32_t v) { base[n * TRACE_BLOCK] = v; ++n; }
  __device__ __forceinline__ int32_t pop() { --n; return base[n * TRACE_BLOCK]; }
  __device__ __forceinline__ bool empty() const { return n == 0; }
};
// The same stack with a bottom: slot 0 of every thread holds "walk done" (0x7fffffff = WALK_DONE, trace_vote.inc) while a walk
// is on and the walk's own entries start at slot 1, so popping an "empty" stack returns "done" like any other item -- no
// emptiness test, no branch in the step functions (k_trace_vote, k_trace_world; k_trace_lds has its 16-bit twin).  The spare
// level every launcher already sizes the stacks with (tree height + 1, wide: peak + 1) is this slot.  reset() writes it anew
// for every walk: the wide step's last act on an exhausted stack is to dump its four missed children there (walk_node_step4).
struct LdsStackB {
  int32_t* base;
  int n;
  static constexpr bool kBottom = true;
  __device__ __forceinline__ void reset() { base[0] = 0x7fffffff; n = 1; }
  __device__ __forceinline__ void init() { reset(); }
  __device__ __forceinline__ void push(int32_t v) { base[n * TRACE_BLOCK] = v; ++n; }
  __device__ __forceinline__ int32_t pop() { --n; return base[n * TRACE_BLOCK]; }
  __device__ __forceinline__ int32_t top() const { return base[(n - 1) * TRACE_BLOCK]; }
  __device__ __forceinline__ bool empty() const { return n <= 1; }
};

__device__ __forceinline__ void flush_counters(const rt::TraceCounters& c, rt::TraceCounters* g) {
  atomicAdd(&g->box_tests, c.box_tests);
  atomicAdd(&g->sphere_tests, c.sphere_tests);
  atomicAdd(&g->moving_sphere_tests, c.moving_sphere_tests);
  atomicAdd(&g->rect_tests, c.rect_tests);
  atomicAdd(&g->triangle_tests, c.triangle_tests);
  atomicAdd(&g->scatters, c.scatters);
  atomicAdd(&g->texels, c.texels);
  atomicAdd(&g->perlin_calls, c.perlin_calls);
  atomicAdd(&g->rays, c.rays);
  atomicAdd(&g->samples, c.samples);
}

// ------------------------------------------------------------------ kernels
#include "trace_basic.inc"   // k_trace_simple, k_trace_persistent, k_trace_stream
#include "trace_vote.inc"    // voting walk, 4-wide tree, k_trace_vote
#include "trace_world.inc"   // k_trace_world: any world, per-lane scan of the world list with carried-over walks
#ifdef RTX_EXPERIMENTAL_KERNELS
#include "trace_wq.inc"      // k_trace_wq: measured dead end kept for A/B (build with -DRTX_EXPERIMENTAL_KERNELS)
#endif
#include "trace_lds.inc"     // k_trace_lds (the headline kernel)
#include "trace_wave.inc"    // k_wf_generate / k_wf_trace / k_wf_shade: the split-kernel integrator (path state in HBM)
#include "post_kernels.inc"  // k_reduce_samples, k_tonemap, device self tests

// ------------------------------------------------------------------ launcher
static int shard_row_count(int32_t height, const RtxShard& sh, int32_t row_limit) {
  int n = 0;
  for (int32_t j = 0; j < height && j < row_limit; ++j)
    if ((j / sh.block_rows) % sh.shard_count == sh.shard_index) ++n;
  return n;
}

static rtx_status validate(const void* s, const RtxCamera* cam, const RtxConfig* cfg,
                           const RtxShard* shard, RtxShard* sh_out) {
  if (!s || !cam || !cfg) { set_error("render: NULL scene, camera or config"); return RTX_EINVAL; }
  if (cfg->threads <= 0 || cfg->image_width <= 0 || cfg->samples_per_pixel <= 0 || cfg->max_depth <= 0) {
    set_error("render: Config::new asserts threads, image_width, samples_per_pixel, max_depth > 0 (world.rs:36-40)");
    return RTX_EINVAL;
  }
  if (rtx_image_height(cfg) <= 0) { set_error("render: image height <= 0 (Screen::new asserts, screen.rs:14)"); return RTX_EINVAL; }
  if (!(cam->time1 < cam->time2)) { set_error("render: camera time1 >= time2 (gen_range panics on an empty range, camera.rs:69)"); return RTX_EINVAL; }
  RtxShard sh = {0, 1, 1, 0};
  if (shard) sh = *shard;
  if (sh.shard_count <= 0 || sh.shard_index < 0 || sh.shard_index >= sh.shard_count || sh.block_rows <= 0) {
    set_error("render: bad shard");
    return RTX_EINVAL;
  }
  *sh_out = sh;
  return RTX_OK;
}

static rt::RenderParams make_params(const RtxCamera* cam, const RtxConfig* cfg) {
  rt::RenderParams rp;
  static_assert(sizeof(RtxCamera) == 24 * sizeof(double) && sizeof(rt::FlatCamera) == 24 * sizeof(rt::real), "camera layout");
  for (int k = 0; k < 24; ++k) ((rt::real*)&rp.cam)[k] = (rt::real)((const double*)cam)[k];
  rp.background = rt::v3(cfg->background[0], cfg->background[1], cfg->background[2]);
  rp.image_width = cfg->image_width;
  rp.image_height = rtx_image_height(cfg);
  rp.samples_per_pixel = cfg->samples_per_pixel;
  rp.max_depth = cfg->max_depth;
  rp.seed = cfg->seed;
  return rp;
}

// One pass (s_count samples of every pixel of the shard) through the wavefront integrator: iterations of
// generate -> trace -> shade over the P path slots until every sample of the pass has been written.
static rtx_status wave_pass(DeviceScene* ds, const rt::RenderParams& rp, const ShardMap& sm, uint32_t s_begin, uint32_t total,
                            uint32_t npix, hipStream_t stream, int preset, uint32_t feat) {
  uint32_t P = ds->wave_paths;
  if ((uint64_t)P > (uint64_t)total) P = total;
  P = (P + WF_SEG - 1u) & ~(WF_SEG - 1u);
  const uint32_t n_seg = P / WF_SEG;
  const size_t R = sizeof(rt::real);
  const size_t bytes = 64 + (size_t)P * (16 + 7 * R + 3 * R + 3 * R + R + 4 + 4 + 4 + 4) + (size_t)n_seg * 8;
  if (bytes > ds->wave_bytes) {
    if (ds->wave_mem) { HIP_TRY(hipStreamSynchronize(stream)); HIP_TRY(hipFree(ds->wave_mem)); ds->wave_mem = nullptr; ds->wave_bytes = 0; }
    HIP_TRY(hipMalloc(&ds->wave_mem, bytes));
    ds->wave_bytes = bytes;
  }
  if (!ds->wave_host_ctrl) HIP_TRY(hipHostMalloc((void**)&ds->wave_host_ctrl, 4 * sizeof(uint32_t), hipHostMallocDefault));
  WavePool pool;
  {
    unsigned char* m = (unsigned char*)ds->wave_mem;
    pool.ctrl = (uint32_t*)m; m += 64;
    pool.rng = (unsigned long long*)m; m += (size_t)P * 16;
    pool.ray = (rt::real*)m; m += (size_t)P * 7 * R;
    pool.product = (rt::real*)m; m += (size_t)P * 3 * R;
    pool.output = (rt::real*)m; m += (size_t)P * 3 * R;
    pool.hit_t = (rt::real*)m; m += (size_t)P * R;
    pool.depth = (int32_t*)m; m += (size_t)P * 4;
    pool.g = (uint32_t*)m; m += (size_t)P * 4;
    pool.hit_ref = (uint32_t*)m; m += (size_t)P * 4;
    pool.free_list = (uint32_t*)m; m += (size_t)P * 4;
    pool.n_free = (uint32_t*)m; m += (size_t)n_seg * 4;
    pool.cursor = (uint32_t*)m; m += (size_t)n_seg * 4;
    pool.P = P;
  }
  const bool wide = preset == 1 && ds->nodes4 != nullptr;
  const uint32_t levels = (uint32_t)(wide ? ds->wide_levels : ds->view.max_stack + 1);
  const size_t lds = (size_t)levels * TRACE_BLOCK * sizeof(int32_t);
  if (lds > 64 * 1024) { set_error("render: BVH too deep for the LDS traversal stack"); return RTX_EUNSUPPORTED; }
  const bool room = (feat & ~P_MESH_ROOM) == 0;
  const int occ = ds->wave_occ;
  const uint32_t leaf_weight = ds->leaf_weight, refill = ds->wave_refill, bvh_pos = (uint32_t)ds->vote_bvh_pos;
  // every combination the launcher can ask for, once: (trace kernel, shade kernel) by preset / tree / occupancy target
#define WF_CASES(X)                                                                                                          \
  if (preset == 0) { X(P_SPHERES, false, 4); }                                                                               \
  else if (!wide) { X(P_MESH, false, 4); }                                                                                   \
  else if (room && occ == 5) { X(P_MESH_ROOM, true, 5); }                                                                    \
  else if (room && occ == 6) { X(P_MESH_ROOM, true, 6); }                                                                    \
  else if (room) { X(P_MESH_ROOM, true, 4); }                                                                                \
  else { X(P_MESH, true, 4); }
  int nb = 0;
  hipError_t oe = hipSuccess;
#define WF_OCC(FEAT, WIDEF, MB) oe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_wf_trace<FEAT, WIDEF, MB>, TRACE_BLOCK, lds)
  WF_CASES(WF_OCC)
#undef WF_OCC
  if (oe != hipSuccess || nb <= 0) { (void)hipGetLastError(); nb = 1; }
  ds->wave_blocks_per_cu = nb;
  const uint64_t want = ((uint64_t)P + TRACE_BLOCK - 1) / TRACE_BLOCK;
  const uint64_t resident = (uint64_t)ds->n_cu * (uint64_t)nb;
  const uint32_t tgrid = (uint32_t)(want < resident ? want : resident);
  hipLaunchKernelGGL(k_wf_init, dim3((P + 255u) / 256u), dim3(256), 0, stream, pool);
  uint32_t check_every = ds->wave_check;
  int it = 0;
  for (;; ++it) {
    hipLaunchKernelGGL(k_wf_generate, dim3(n_seg), dim3(WF_SEG), 0, stream, pool, rp, sm, s_begin, total, npix);
    if ((uint32_t)(it + 1) % check_every == 0u) {
      HIP_TRY(hipMemsetAsync(pool.ctrl, 0, 16, stream));
      hipLaunchKernelGGL(k_wf_count, dim3(256), dim3(256), 0, stream, pool);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipMemcpyAsync(ds->wave_host_ctrl, pool.ctrl, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
      HIP_TRY(hipStreamSynchronize(stream));
      // no path alive after a generation: every segment's share of the pass is used up and every path has ended
      if (ds->wave_host_ctrl[0] == 0u) break;
      if (ds->wave_host_ctrl[0] < P / 2u) check_every = 2;  // the tail: the pool drains within max_depth iterations
      if (it > 100000000) { set_error("render: wavefront integrator did not converge"); return RTX_EHIP; }
    }
#define WF_LAUNCH(FEAT, WIDEF, MB)                                                                                           \
  do {                                                                                                                       \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_wf_trace<FEAT, WIDEF, MB>), dim3(tgrid), dim3(TRACE_BLOCK), lds, stream, ds->view, pool, \
                       leaf_weight, refill, bvh_pos, ds->nodes4, ds->vote_tri_base);                                          \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_wf_shade<FEAT>), dim3(n_seg), dim3(WF_SEG), 0, stream, ds->view, pool, rp, ds->samples, bvh_pos); \
  } while (0)
    WF_CASES(WF_LAUNCH)
#undef WF_LAUNCH
  }
#undef WF_CASES
  ds->wave_iterations = it + 1;
  if (ds->wave_verbose) fprintf(stderr, "[rtx] wavefront: %u slots, %d iterations, %d blocks per CU (%u stack levels)\n", P, it + 1, nb, levels);
  return RTX_OK;
}

template <bool COUNT>
static rtx_status render_impl(DeviceScene* ds, const RtxCamera* cam, const RtxConfig* cfg,
                              const RtxShard* shard, double* d_accum_out, uint8_t* d_rgb8_out,
                              hipStream_t stream, RtxRenderStats* stats) {
  RtxShard sh;
  rtx_status st = validate(ds, cam, cfg, shard, &sh);
  if (st != RTX_OK) return st;
  int cur = -1;
  HIP_TRY(hipGetDevice(&cur));
  if (cur != ds->device) { set_error("render: scene was uploaded to a different device than the current one"); return RTX_EINVAL; }
  if ((ds->view.features & rt::F_GRAVITY_SPHERE) && !(cam->time2 <= ds->gravity_time_limit)) {
    set_error("render: shutter time " + std::to_string(cam->time2) + " is beyond the GravitySpheres' stored trajectory (limit " +
              std::to_string(ds->gravity_time_limit) + " s): get_center's fallback loop (hit.rs:380-390) would run unbounded on the GPU");
    return RTX_EINVAL;
  }

  const rt::RenderParams rp = make_params(cam, cfg);
  const int32_t w = rp.image_width, h = rp.image_height;
  // world.rs:1198-1202: chunk_size = h / threads; rows >= threads * chunk_size are never rendered.
  int32_t row_limit = h;
  if (cfg->row_chunk_compat) row_limit = (h / cfg->threads) * cfg->threads;
  const int rows_all = shard_row_count(h, sh, h);
  const int rows_active = shard_row_count(h, sh, row_limit);
  const uint64_t npix_all = (uint64_t)rows_all * w, npix = (uint64_t)rows_active * w;
  if (npix_all >= (1ull << 31)) { set_error("render: shard larger than 2^31 pixels"); return RTX_EINVAL; }
  if (stats) memset(stats, 0, sizeof(*stats));
  if (npix_all == 0) return RTX_OK;

  // ---- workspace
  // Default budget: 24 GiB (of 288 GB: C5 takes 16 passes instead of 67, C3 10 instead of 40), but never more than a third of the
  // HBM that is free right now plus what this handle already holds -- other scene handles, a second frame in flight, torch's caching
  // allocator or a smaller part shrink it, and a frame then takes more passes instead of failing.  An explicit
  // cfg->sample_buffer_bytes is taken as given.
  uint64_t budget = cfg->sample_buffer_bytes;
  if (budget == 0) {
    budget = 24ull << 30;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
      const uint64_t avail = ((uint64_t)free_b + ds->samples_bytes) / 3;
      if (avail < budget) budget = avail;
    } else {
      (void)hipGetLastError();
    }
    if (budget < ds->samples_bytes) budget = ds->samples_bytes;  // what is already there can be used
  }
  uint64_t per_sample_plane = npix * 24ull;
  uint32_t spp = (uint32_t)cfg->samples_per_pixel;
  uint32_t spp_pass = spp;
  // Two passes in flight (below) when the render needs several passes anyway (C5: 16, C3: 10): each pass then gets half of the
  // buffer.  A render that fits one pass stays one launch -- cut in two it gains the overlapped half of its reduction and loses as
  // much to the second launch (C2: 6550 against 6574 Msamples/s; with two FRAMES in flight on top, 6644 against 6722), whereas
  // C5 gains 1.7 %.  Not for the counting / timing entry points (stats != NULL synchronises per pass).
  const bool pipeline = ds->pass_pipeline && !COUNT && stats == nullptr && !ds->force_simple && per_sample_plane > 0 &&
                        (uint64_t)spp * per_sample_plane > budget && budget / 2 >= per_sample_plane;
  const uint64_t pass_budget = pipeline ? budget / 2 : budget;
  if (per_sample_plane > 0 && (uint64_t)spp_pass * per_sample_plane > pass_budget) {
    spp_pass = (uint32_t)(pass_budget / per_sample_plane);
    if (spp_pass < 1) spp_pass = 1;
  }
  // a pass's (sample, pixel) index space is addressed with 32-bit indices
  while (spp_pass > 1 && (uint64_t)spp_pass * npix >= 0xFFFF0000ull) --spp_pass;
  if ((uint64_t)spp_pass * npix >= 0xFFFF0000ull) { set_error("render: shard too large for one pass"); return RTX_EINVAL; }
  size_t need_samples = (size_t)spp_pass * per_sample_plane * (pipeline ? 2 : 1);
  if (need_samples > ds->samples_bytes) {
    if (ds->samples) { HIP_TRY(hipStreamSynchronize(stream)); HIP_TRY(hipFree(ds->samples)); ds->samples = nullptr; ds->samples_bytes = 0; }
    // out of memory: halve the pass until the buffer fits (down to one sample per pass) before giving up
    for (;;) {
      hipError_t me = hipMalloc((void**)&ds->samples, need_samples);
      if (me == hipSuccess) break;
      (void)hipGetLastError();
      ds->samples = nullptr;
      if (me != hipErrorOutOfMemory || spp_pass <= 1) {
        set_error(std::string("render: sample buffer of ") + std::to_string(need_samples) + " bytes: " + hipGetErrorString(me));
        return me == hipErrorOutOfMemory ? RTX_ENOMEM : RTX_EHIP;
      }
      spp_pass = (spp_pass + 1) / 2;
      need_samples = (size_t)spp_pass * per_sample_plane * (pipeline ? 2 : 1);
    }
    ds->samples_bytes = need_samples;
  }
  double* accum = d_accum_out;
  if (!accum) {
    size_t need = (size_t)npix_all * 24;
    if (need > ds->accum_bytes) {
      if (ds->accum) { HIP_TRY(hipStreamSynchronize(stream)); HIP_TRY(hipFree(ds->accum)); ds->accum = nullptr; ds->accum_bytes = 0; }
      HIP_TRY(hipMalloc((void**)&ds->accum, need));
      ds->accum_bytes = need;
    }
    accum = ds->accum;
  }
  if (!ds->counters) HIP_TRY(hipMalloc((void**)&ds->counters, sizeof(rt::TraceCounters)));
  if (!ds->work_counter) HIP_TRY(hipMalloc((void**)&ds->work_counter, 2 * sizeof(unsigned int)));
  if (pipeline && !ds->aux_stream) {
    HIP_TRY(hipStreamCreateWithFlags(&ds->aux_stream, hipStreamNonBlocking));
    for (int i = 0; i < 3; ++i) HIP_TRY(hipEventCreateWithFlags(&ds->ev_pass[i], hipEventDisableTiming));
  }
  if (!ds->ev[0]) { HIP_TRY(hipEventCreate(&ds->ev[0])); HIP_TRY(hipEventCreate(&ds->ev[1])); }
  if (COUNT) HIP_TRY(hipMemsetAsync(ds->counters, 0, sizeof(rt::TraceCounters), stream));

  // rows skipped by row_chunk_compat stay (0,0,0) as in the reference's Screen::new
  if (npix < npix_all) {
    HIP_TRY(hipMemsetAsync(accum + 3 * npix, 0, (size_t)(npix_all - npix) * 24, stream));
    if (d_rgb8_out) HIP_TRY(hipMemsetAsync(d_rgb8_out + 3 * npix, 0, (size_t)(npix_all - npix) * 3, stream));
  }

  ShardMap sm = {w, sh.block_rows, sh.shard_index, sh.shard_count};
  int stack_levels = ds->view.max_stack + 1;
  size_t lds_bytes = (size_t)stack_levels * TRACE_BLOCK * sizeof(int32_t);
  if (lds_bytes > 64 * 1024) { set_error("render: BVH too deep for the LDS traversal stack"); return RTX_EUNSUPPORTED; }
  const uint32_t feat = ds->view.features;
  const int preset = ((feat & ~P_SPHERES) == 0) ? 0 : (((feat & ~P_MESH) == 0) ? 1 : 2);
  const bool use_simple = COUNT || ds->force_simple;

  float trace_ms = 0.f;
  int passes = 0;
  int32_t kernel_used = RTX_KERNEL_SIMPLE;
  if (npix > 0) {
    // Passes two deep: even passes on the caller's stream, odd ones on ds->aux_stream, each with its own half of the sample
    // buffer and its own work counter.  Within a stream: trace(k), reduce(k), trace(k + 2), ... -- so a half is not overwritten
    // before it has been summed; across streams reduce(k) waits for reduce(k - 1) -- so every pixel's samples are still added in
    // ascending order, bit for bit what one stream does.  What overlaps: the last waves of trace(k) (a few long paths in
    // otherwise idle CUs), reduce(k) and the first waves of trace(k + 1).
    double* const samples_base = ds->samples;
    unsigned int* const counter_base = ds->work_counter;
    struct RestoreScene {
      DeviceScene* d; double* s; unsigned int* w;
      ~RestoreScene() { d->samples = s; d->work_counter = w; }
    } restore_scene{ds, samples_base, counter_base};
    hipStream_t const caller_stream = stream;
    if (pipeline) {
      HIP_TRY(hipEventRecord(ds->ev_pass[2], caller_stream));
      HIP_TRY(hipStreamWaitEvent(ds->aux_stream, ds->ev_pass[2], 0));
    }
    for (uint32_t s_begin = 0; s_begin < spp; s_begin += spp_pass, ++passes) {
      const int half = pipeline ? (passes & 1) : 0;
      hipStream_t stream = half ? ds->aux_stream : caller_stream;  // (shadows the parameter: every launch below goes to this pass's stream)
      ds->samples = samples_base + (size_t)half * (size_t)spp_pass * (size_t)npix * 3u;
      ds->work_counter = counter_base + half;
      uint32_t s_count = spp - s_begin < spp_pass ? spp - s_begin : spp_pass;
      uint32_t total = (uint32_t)((uint64_t)s_count * npix);
      if (stats) HIP_TRY(hipEventRecord(ds->ev[0], stream));
      if (use_simple) {
        kernel_used = RTX_KERNEL_SIMPLE;
        uint64_t want = ((uint64_t)total + TRACE_BLOCK - 1) / TRACE_BLOCK;
        uint32_t grid = (uint32_t)(want < (uint64_t)ds->n_cu * 8 ? want : (uint64_t)ds->n_cu * 8);
#define LAUNCH_SIMPLE(FEAT)                                                                          \
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_trace_simple<FEAT, COUNT>), dim3(grid), dim3(TRACE_BLOCK),     \
                     lds_bytes, stream, ds->view, rp, sm, s_begin, total, (uint32_t)npix, ds->samples, \
                     ds->counters)
        if constexpr (COUNT) { LAUNCH_SIMPLE(P_ALL); }
        else {
          if (preset == 0) { LAUNCH_SIMPLE(P_SPHERES); }
          else if (preset == 1) { LAUNCH_SIMPLE(P_MESH); }
          else { LAUNCH_SIMPLE(P_ALL); }
        }
#undef LAUNCH_SIMPLE
      } else if (ds->lds_ok && preset == 0 && !ds->force_wq && !ds->force_vote && !ds->force_persistent && !ds->force_stream && !ds->force_world && !ds->force_wave) {
        kernel_used = RTX_KERNEL_LDS;
        HIP_TRY(hipMemsetAsync(ds->work_counter, 0, sizeof(unsigned int), stream));
        // time-aware boxes when the scene has them and every ray's time lies inside the BVH's interval (camera.rs:69: [time1, time2))
        const bool motion = ds->motion_ok && (double)cam->time1 >= ds->motion_t0 && (double)cam->time2 <= ds->motion_t1;
        const bool w4 = !motion && ds->w4_ok;
        const bool ring = motion ? ds->motion_ring : (w4 ? ds->w4_ring : ds->lds_ring);
        const uint32_t ring_cap = motion ? ds->motion_ring_cap : (w4 ? ds->w4_ring_cap : ds->lds_ring_cap);
        const LdsSceneDims dims = motion ? ds->motion_dims : (w4 ? ds->w4_dims : ds->lds_dims);
        const uint32_t lds_levels = w4 ? ds->w4_levels : (uint32_t)stack_levels;
        const rt::real m_t0 = (rt::real)ds->motion_t0, m_inv = (rt::real)(1.0 / (ds->motion_t1 - ds->motion_t0));
        const LdsKernelLayout L = ldsk_layout(lds_levels, ring ? ring_cap : 0u, dims);
        uint64_t want = ((uint64_t)total + LDSK_BLOCK - 1) / LDSK_BLOCK;
        uint32_t grid = (uint32_t)(want < (uint64_t)ds->n_cu ? want : (uint64_t)ds->n_cu);
#define LAUNCH_LDS2(FEAT, RINGF, MOTIONF, W4F)                                                                        \
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_trace_lds<FEAT, RINGF, MOTIONF, W4F>), dim3(grid), dim3(LDSK_BLOCK), L.total, stream, \
                     ds->view, rp, sm, s_begin, total, (uint32_t)npix, ds->samples, ds->work_counter,      \
                     ds->leaf_weight, ds->walk_threshold | (ds->single_leaf ? 0x100u : 0u), ds->lds_chunk, ring_cap, lds_levels, dims, m_t0, m_inv, ds->w4_image, \
                     ds->mv_common ? 1u : 0u, (rt::real)ds->mv_t0, (rt::real)ds->mv_t1)
#define LAUNCH_LDS(FEAT, MOTIONF, W4F) do { if (ring) { LAUNCH_LDS2(FEAT, true, MOTIONF, W4F); } else { LAUNCH_LDS2(FEAT, false, MOTIONF, W4F); } } while (0)
        // static spheres without checker textures (the Book-1 final scene): the leaner instantiation
        if ((feat & ~P_STATIC_SPHERES) == 0) { if (w4) { LAUNCH_LDS(P_STATIC_SPHERES, 0u, true); } else { LAUNCH_LDS(P_STATIC_SPHERES, 0u, false); } }
        else if (motion && ds->motion_axis == 1) { LAUNCH_LDS(P_SPHERES, 3u, false); }  // slopes along y only (Book-1 at HEAD)
        else if (motion) { LAUNCH_LDS(P_SPHERES, 1u, false); }
        else if (w4) { LAUNCH_LDS(P_SPHERES, 0u, true); }
        else { LAUNCH_LDS(P_SPHERES, 0u, false); }
#undef LAUNCH_LDS
#undef LAUNCH_LDS2
#ifdef RTX_EXPERIMENTAL_KERNELS
      } else if (ds->force_wq && ds->wq_ok && preset == 0) {
        kernel_used = RTX_KERNEL_WQ;
        HIP_TRY(hipMemsetAsync(ds->work_counter, 0, sizeof(unsigned int), stream));
        if (!ds->error_word) HIP_TRY(hipMalloc((void**)&ds->error_word, sizeof(unsigned int)));
        HIP_TRY(hipMemsetAsync(ds->error_word, 0, sizeof(unsigned int), stream));
        const WqLayout L = wq_layout(ds->wq_paths, ds->wq_levels);
        uint64_t want = ((uint64_t)total + ds->wq_paths - 1) / ds->wq_paths;
        uint32_t grid = (uint32_t)(want < (uint64_t)ds->n_cu ? want : (uint64_t)ds->n_cu);
        if (ds->wq_diag) {
          if (!ds->diag) HIP_TRY(hipMalloc((void**)&ds->diag, 24 * sizeof(unsigned long long)));
          HIP_TRY(hipMemsetAsync(ds->diag, 0, 24 * sizeof(unsigned long long), stream));
          hipLaunchKernelGGL(HIP_KERNEL_NAME(k_trace_wq<P_SPHERES, true>), dim3(grid), dim3(WQ_THREADS), L.total, stream,
                             ds->view, rp, sm, s_begin, total, (uint32_t)npix, ds->samples, ds->work_counter,
                             ds->error_word, ds->wq_paths, ds->wq_levels, ds->wq_walkers, ds->wq_batch_min, ds->diag);
          HIP_TRY(hipStreamSynchronize(stream));
          unsigned long long h[24];
          HIP_TRY(hipMemcpy(h, ds->diag, sizeof(h), hipMemcpyDeviceToHost));
          const char* tnames[6] = {"gen", "walk(total)", "walk:refill", "shade", "idle", "kernel"};
          for (int k = 0; k < 6; ++k)
            fprintf(stderr, "[wq_diag] wave-time %-12s %6.2f %%\n", tnames[k], h[16 + 5] ? 100.0 * (double)h[16 + k] / (double)h[16 + 5] : 0.0);
          const char* names[8] = {"node_step", "leaf_step", "refill", "shade", "gen", "idle_poll", "walk_loop", "short_refill"};
          for (int k = 0; k < 8; ++k)
            fprintf(stderr, "[wq_diag] %-12s executions %llu lanes %llu mean %.2f\n", names[k], h[2 * k], h[2 * k + 1],
                    h[2 * k] ? (double)h[2 * k + 1] / (double)h[2 * k] : 0.0);
        } else {
          hipLaunchKernelGGL(HIP_KERNEL_NAME(k_trace_wq<P_SPHERES, false>), dim3(grid), dim3(WQ_THREADS), L.total, stream,
                             ds->view, rp, sm, s_begin, total, (uint32_t)npix, ds->samples, ds->work_counter,
                             ds->error_word, ds->wq_paths, ds->wq_levels, ds->wq_walkers, ds->wq_batch_min, (unsigned long long*)nullptr);
        }
        HIP_TRY(hipGetLastError());
        unsigned int err = 0;
        HIP_TRY(hipMemcpyAsync(&err, ds->error_word, sizeof(err), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        if (err != 0) { set_error("render: k_trace_wq aborted (bounded wait tripped, code " + std::to_string(err) + ")"); return RTX_EHIP; }
#endif
      } else if (ds->wave_ok && preset < 2 && (ds->force_wave || (ds->wave_default && !ds->force_vote && !ds->force_persistent && !ds->force_world && !ds->force_stream))) {
        kernel_used = RTX_KERNEL_WAVEFRONT;
        st = wave_pass(ds, rp, sm, s_begin, total, (uint32_t)npix, stream, preset, feat);
        if (st != RTX_OK) return st;
      } else if (ds->vote_ok && preset < 2 && !ds->force_persistent && !ds->force_world && !(ds->force_stream && ds->single_bvh)) {
        kernel_used = RTX_KERNEL_VOTE;
        HIP_TRY(hipMemsetAsync(ds->work_counter, 0, sizeof(unsigned int), stream));
        uint64_t want = ((uint64_t)total + TRACE_BLOCK - 1) / TRACE_BLOCK;
        uint64_t resident = (uint64_t)ds->n_cu * (uint64_t)ds->vote_blocks_per_cu[preset];
        uint32_t grid = 0;
        const bool ring = ds->vote_ring[preset];
        const size_t vote_lds = lds_bytes + (ring ? (TRACE_BLOCK / 64) * RING_BYTES_PER_WAVE : 0);
        grid = (uint32_t)(want < resident ? want : resident);
#define LAUNCH_VOTE(FEAT, DIAGF, RINGF, DIAGP)                                                        \
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_trace_vote<FEAT, DIAGF, RINGF, false>), dim3(grid), dim3(TRACE_BLOCK), vote_lds, \
                     stream, ds->view, rp, sm, s_begin, total, (uint32_t)npix, ds->samples,            \
                     ds->work_counter, DIAGP, ds->leaf_weight, ds->walk_threshold, (uint32_t)stack_levels, (uint32_t)ds->vote_bvh_pos, \
                     (const FlatNode4*)nullptr, ds->vote_tri_base, 0u, ds->vote_top)
        if (ds->vote_diag && preset == 0) {
          if (!ds->diag) HIP_TRY(hipMalloc((void**)&ds->diag, 24 * sizeof(unsigned long long)));
          HIP_TRY(hipMemsetAsync(ds->diag, 0, 12 * sizeof(unsigned long long), stream));
          if (ring) { LAUNCH_VOTE(P_SPHERES, true, true, ds->diag); } else { LAUNCH_VOTE(P_SPHERES, true, false, ds->diag); }
          HIP_TRY(hipStreamSynchronize(stream));
          unsigned long long h[12];
          HIP_TRY(hipMemcpy(h, ds->diag, sizeof(h), hipMemcpyDeviceToHost));
          const char* names[6] = {"outer", "regen", "node_step", "leaf_step", "shade_hit", "walking"};
          for (int k = 0; k < 6; ++k)
            fprintf(stderr, "[vote_diag] %-10s executions %llu lanes %llu mean lanes %.2f\n", names[k], h[2 * k], h[2 * k + 1],
                    h[2 * k] ? (double)h[2 * k + 1] / (double)h[2 * k] : 0.0);
        }
        else if (preset == 0) { if (ring) { LAUNCH_VOTE(P_SPHERES, false, true, (unsigned long long*)nullptr); } else { LAUNCH_VOTE(P_SPHERES, false, false, (unsigned long long*)nullptr); } }
        else if (ds->nodes4) {
          // material / texture records behind the stacks when that costs no resident block
          uint32_t lds_tables = 0;
          size_t wide_lds = (size_t)ds->wide_levels * TRACE_BLOCK * sizeof(int32_t);
          if (ds->vote_tables_bytes > 0 && (wide_lds + ds->vote_tables_bytes) * (size_t)ds->wide_blocks_per_cu <= 160 * 1024) {
            lds_tables = ds->vote_tables;
            wide_lds += ds->vote_tables_bytes;
          }
          uint64_t res4 = (uint64_t)ds->n_cu * (uint64_t)ds->wide_blocks_per_cu;
          grid = (uint32_t)(want < res4 ? want : res4);
#define LAUNCH_VOTE_WIDE(FEAT)                                                                         \
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_trace_vote<FEAT, false, false, true>), dim3(grid), dim3(TRACE_BLOCK), wide_lds, \
                     stream, ds->view, rp, sm, s_begin, total, (uint32_t)npix, ds->samples, ds->work_counter,  \
                     (unsigned long long*)nullptr, ds->leaf_weight, ds->walk_threshold | (ds->regen_min << 16), (uint32_t)ds->wide_levels, \
                     (uint32_t)ds->vote_bvh_pos, ds->nodes4, ds->vote_tri_base, lds_tables, ds->vote_top)
          // a triangle mesh in a room of rectangles, no spheres / lists / glass (the dragon room): the leaner instantiation
          if (ds->vote_diag && (feat & ~P_MESH_ROOM) == 0) {
            if (!ds->diag) HIP_TRY(hipMalloc((void**)&ds->diag, 24 * sizeof(unsigned long long)));
            HIP_TRY(hipMemsetAsync(ds->diag, 0, 24 * sizeof(unsigned long long), stream));
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_trace_vote<P_MESH_ROOM, true, false, true>), dim3(grid), dim3(TRACE_BLOCK), wide_lds,
                               stream, ds->view, rp, sm, s_begin, total, (uint32_t)npix, ds->samples, ds->work_counter,
                               ds->diag, ds->leaf_weight, ds->walk_threshold, (uint32_t)ds->wide_levels,
                               (uint32_t)ds->vote_bvh_pos, ds->nodes4, ds->vote_tri_base, lds_tables, ds->vote_top);
            HIP_TRY(hipStreamSynchronize(stream));
            unsigned long long h[24];
            HIP_TRY(hipMemcpy(h, ds->diag, sizeof(h), hipMemcpyDeviceToHost));
            const char* names[6] = {"outer", "regen", "node_step", "leaf_step", "shade_hit", "shade_all"};
            for (int k = 0; k < 6; ++k)
              fprintf(stderr, "[vote_diag] %-10s executions %llu lanes %llu mean lanes %.2f\n", names[k], h[2 * k], h[2 * k + 1],
                      h[2 * k] ? (double)h[2 * k + 1] / (double)h[2 * k] : 0.0);
            if (h[12]) {
              float v[12];
              for (int k = 0; k < 6; ++k) { uint32_t lo = (uint32_t)h[13 + k], hi = (uint32_t)(h[13 + k] >> 32); memcpy(&v[2 * k], &lo, 4); memcpy(&v[2 * k + 1], &hi, 4); }
              fprintf(stderr, "[vote_diag] %llu walks of >= 50000 node steps; the first: origin (%g %g %g) direction (%g %g %g) 1/d (%g %g %g) err2 %g t_min %g t_max %g depth left %llu\n",
                      h[12], v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8], v[9], v[10], v[11], h[19]);
            }
          }
          else if ((feat & ~P_MESH_ROOM) == 0) { LAUNCH_VOTE_WIDE(P_MESH_ROOM); } else { LAUNCH_VOTE_WIDE(P_MESH); }
#undef LAUNCH_VOTE_WIDE
        }
        else { if (ring) { LAUNCH_VOTE(P_MESH, false, true, (unsigned long long*)nullptr); } else { LAUNCH_VOTE(P_MESH, false, false, (unsigned long long*)nullptr); } }
#undef LAUNCH_VOTE
#ifdef RTX_EXPERIMENTAL_KERNELS
      } else if (ds->single_bvh && preset < 2 && ds->force_stream) {
        kernel_used = RTX_KERNEL_STREAM;
        HIP_TRY(hipMemsetAsync(ds->work_counter, 0, sizeof(unsigned int), stream));
        uint64_t want = ((uint64_t)total + TRACE_BLOCK - 1) / TRACE_BLOCK;
        uint64_t resident = (uint64_t)ds->n_cu * (uint64_t)ds->stream_blocks_per_cu[preset];
        uint32_t grid = (uint32_t)(want < resident ? want : resident);
#define LAUNCH_STREAM(FEAT)                                                                           \
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_trace_stream<FEAT>), dim3(grid), dim3(TRACE_BLOCK), lds_bytes,  \
                     stream, ds->view, rp, sm, s_begin, total, (uint32_t)npix, ds->samples,            \
                     ds->work_counter, ds->walk_threshold)
        if (preset == 0) { LAUNCH_STREAM(P_SPHERES); }
        else { LAUNCH_STREAM(P_MESH); }
#undef LAUNCH_STREAM
#endif
      } else if (!ds->force_persistent || (feat & rt::F_GRAVITY_SPHERE)) {
        kernel_used = RTX_KERNEL_WORLD;
        HIP_TRY(hipMemsetAsync(ds->work_counter, 0, sizeof(unsigned int), stream));
        const bool wide = ds->nodes4 != nullptr;
        const bool book2 = (feat & ~P_BOOK2) == 0;
        const bool has_gravity = (feat & rt::F_GRAVITY_SPHERE) != 0;
        const bool no_sphere_media = (feat & rt::F_MEDIUM_SPHERE) == 0;
        const uint32_t levels = (uint32_t)(wide ? ds->wide_levels : stack_levels);
        const size_t world_lds = (size_t)levels * TRACE_BLOCK * sizeof(int32_t) + (size_t)WORLD_SLOT_F64 * TRACE_BLOCK * sizeof(rt::real) +
                                 (size_t)ds->world_perlin_lds * sizeof(rt::FlatPerlin) +
                                 (size_t)ds->world_mat_lds * sizeof(rt::FlatMaterial) + (size_t)ds->world_tex_lds * sizeof(rt::FlatTexture);
        uint64_t want = ((uint64_t)total + TRACE_BLOCK - 1) / TRACE_BLOCK;
        uint64_t resident = (uint64_t)ds->n_cu * (uint64_t)ds->world_blocks_per_cu[has_gravity ? 2 : (book2 ? 0 : (no_sphere_media ? 3 : 1))][wide ? 1 : 0];
        uint32_t grid = (uint32_t)(want < resident ? want : resident);
#define LAUNCH_WORLD(FEAT, WIDEF, WPS)                                                                   \
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_trace_world<FEAT, WIDEF, WPS>), dim3(grid), dim3(TRACE_BLOCK), world_lds, stream, \
                     ds->view, rp, sm, s_begin, total, (uint32_t)npix, ds->samples, ds->work_counter, ds->view.entries, \
                     ds->view.top_level, ds->view.spheres, ds->view.moving_spheres, ds->view.rects, ds->view.triangles, \
                     ds->view.materials, ds->view.textures, ds->view.refs, ds->nodes4, ds->world_desc, ds->leaf_weight, ds->world_threshold, levels, ds->world_perlin_lds, ds->world_mat_lds, ds->world_tex_lds)
        if (ds->world_diag && book2 && wide) {
          if (!ds->diag) HIP_TRY(hipMalloc((void**)&ds->diag, 24 * sizeof(unsigned long long)));
          HIP_TRY(hipMemsetAsync(ds->diag, 0, 24 * sizeof(unsigned long long), stream));
          hipLaunchKernelGGL(HIP_KERNEL_NAME(k_trace_world<P_BOOK2, true, WORLD_WPS, true>), dim3(grid), dim3(TRACE_BLOCK), world_lds, stream,
                             ds->view, rp, sm, s_begin, total, (uint32_t)npix, ds->samples, ds->work_counter, ds->view.entries,
                             ds->view.top_level, ds->view.spheres, ds->view.moving_spheres, ds->view.rects, ds->view.triangles,
                             ds->view.materials, ds->view.textures, ds->view.refs, ds->nodes4, ds->world_desc, ds->leaf_weight, ds->world_threshold, levels, ds->world_perlin_lds, ds->world_mat_lds, ds->world_tex_lds, ds->diag);
          HIP_TRY(hipStreamSynchronize(stream));
          unsigned long long hd[16];
          HIP_TRY(hipMemcpy(hd, ds->diag, sizeof(hd), hipMemcpyDeviceToHost));
          const char* names[8] = {"node_step", "leaf_step", "sweep", "shade(lean)", "regen", "direct_entry(all)", "direct_entry(run)", "shade(rare)"};
          for (int k = 0; k < 8; ++k)
            fprintf(stderr, "[world_diag] %-18s executions %llu lanes %llu mean lanes %.2f\n", names[k], hd[2 * k], hd[2 * k + 1],
                    hd[2 * k] ? (double)hd[2 * k + 1] / (double)hd[2 * k] : 0.0);
        } else if (has_gravity) {  // the bouncing-ball scene: the instantiation that carries GravitySphere code
          if (wide) { LAUNCH_WORLD(P_ALL, true, WORLD_WPS); } else { LAUNCH_WORLD(P_ALL, false, WORLD_WPS); }
        } else if (book2) {
          if (wide) { LAUNCH_WORLD(P_BOOK2, true, WORLD_WPS); } else { LAUNCH_WORLD(P_BOOK2, false, WORLD_WPS); }
        } else if (no_sphere_media) {
          if (wide) { LAUNCH_WORLD(P_NO_SPHERE_MEDIA, true, WORLD_WPS); } else { LAUNCH_WORLD(P_NO_SPHERE_MEDIA, false, WORLD_WPS); }
        } else {
          if (wide) { LAUNCH_WORLD(P_ANY, true, WORLD_WPS); } else { LAUNCH_WORLD(P_ANY, false, WORLD_WPS); }
        }
#undef LAUNCH_WORLD
      } else {
        kernel_used = RTX_KERNEL_PERSISTENT;
        HIP_TRY(hipMemsetAsync(ds->work_counter, 0, sizeof(unsigned int), stream));
        uint64_t want = ((uint64_t)total + TRACE_BLOCK - 1) / TRACE_BLOCK;
        uint64_t resident = (uint64_t)ds->n_cu * (uint64_t)ds->blocks_per_cu[preset];
        uint32_t grid = (uint32_t)(want < resident ? want : resident);
#define LAUNCH_PERSISTENT(FEAT, WIDEF, VIEW, LDSB)                                                     \
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_trace_persistent<FEAT, WIDEF>), dim3(grid), dim3(TRACE_BLOCK),  \
                     LDSB, stream, VIEW, rp, sm, s_begin, total, (uint32_t)npix, ds->samples,            \
                     ds->work_counter, ds->view.entries, ds->view.top_level, ds->view.spheres,             \
                     ds->view.moving_spheres, ds->view.rects, ds->view.triangles, ds->view.materials,      \
                     ds->view.textures, ds->view.refs)
        rt::SceneView pv = ds->view;
        pv.pad = ds->leaf_weight;
        if (ds->nodes4 && preset >= 1) {
          rt::SceneView wv = pv;
          wv.nodes = (const rt::FlatNode*)ds->nodes4;  // the wide tree rides in the slot of the (unused) f64 tree
          const size_t wide_lds = (size_t)ds->wide_levels * TRACE_BLOCK * sizeof(int32_t);
          resident = (uint64_t)ds->n_cu * (uint64_t)ds->wide_pers_blocks_per_cu[preset];
          grid = (uint32_t)(want < resident ? want : resident);
          if (preset == 1) { LAUNCH_PERSISTENT(P_MESH, true, wv, wide_lds); }
          else { LAUNCH_PERSISTENT(P_ANY, true, wv, wide_lds); }
        }
        else if (preset == 0) { LAUNCH_PERSISTENT(P_SPHERES, false, pv, lds_bytes); }
        else if (preset == 1) { LAUNCH_PERSISTENT(P_MESH, false, pv, lds_bytes); }
        else { LAUNCH_PERSISTENT(P_ANY, false, pv, lds_bytes); }
#undef LAUNCH_PERSISTENT
      }
      HIP_TRY(hipGetLastError());
      if (stats) {
        HIP_TRY(hipEventRecord(ds->ev[1], stream));
        HIP_TRY(hipEventSynchronize(ds->ev[1]));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, ds->ev[0], ds->ev[1]));
        trace_ms += ms;
      }
      uint32_t pgrid = (uint32_t)((npix + 255) / 256);
      if (pipeline && passes > 0) HIP_TRY(hipStreamWaitEvent(stream, ds->ev_pass[1 - half], 0));  // the previous pass's sums are in
      hipLaunchKernelGGL(k_reduce_samples, dim3(pgrid), dim3(256), 0, stream, ds->samples, accum,
                         (uint32_t)npix, s_count, s_begin == 0 ? 1 : 0);
      HIP_TRY(hipGetLastError());
      if (pipeline) HIP_TRY(hipEventRecord(ds->ev_pass[half], stream));
    }
    if (pipeline && passes > 0 && ((passes - 1) & 1)) HIP_TRY(hipStreamWaitEvent(caller_stream, ds->ev_pass[1], 0));
    if (d_rgb8_out) {
      uint32_t pgrid = (uint32_t)((npix + 255) / 256);
      hipLaunchKernelGGL(k_tonemap, dim3(pgrid), dim3(256), 0, stream, accum, d_rgb8_out, (uint32_t)npix, spp);
      HIP_TRY(hipGetLastError());
    }
  }
  if (stats) {
    HIP_TRY(hipStreamSynchronize(stream));
    stats->trace_ms = trace_ms;
    stats->trace_launches = passes;
    stats->passes = passes;
    stats->trace_kernel = kernel_used;
    stats->sample_buffer_bytes = need_samples;
    stats->samples = (uint64_t)spp * npix;  // (pixel, sample) paths this call traced; the work counters below only in count mode
    if (COUNT) {
      rt::TraceCounters c;
      HIP_TRY(hipMemcpy(&c, ds->counters, sizeof(c), hipMemcpyDeviceToHost));
      stats->samples = c.samples; stats->rays = c.rays; stats->box_tests = c.box_tests;
      stats->sphere_tests = c.sphere_tests; stats->moving_sphere_tests = c.moving_sphere_tests;
      stats->rect_tests = c.rect_tests; stats->triangle_tests = c.triangle_tests;
      stats->scatters = c.scatters; stats->texels = c.texels; stats->perlin_calls = c.perlin_calls;
    }
  }
  return RTX_OK;
}

// Upload one flattened scene to the current device and size the launches of every kernel for it.
static rtx_status scene_upload_impl(const FlatScene& fs, DeviceScene** out) {
  *out = nullptr;
  DeviceScene* ds = new (std::nothrow) DeviceScene();
  if (!ds) { set_error("out of memory"); return RTX_ENOMEM; }
  memset(&ds->vote_top, 0, sizeof(ds->vote_top));
  ds->vote_top.n = -1;
  hipError_t e = hipGetDevice(&ds->device);
  if (e != hipSuccess) {
    set_error(std::string("hipGetDevice: ") + hipGetErrorString(e) + " (this library has no CPU render path)");
    delete ds;
    return RTX_EHIP;
  }
  rt::SceneView& v = ds->view;
  memset(&v, 0, sizeof(v));
  rtx_status st;
#define UP(field, vec) if ((st = upload_array(ds, fs.vec, &v.field)) != RTX_OK) { free_device_scene(ds); return st; }
  UP(spheres, spheres) UP(moving_spheres, moving_spheres) UP(rects, rects) UP(triangles, triangles)
  UP(nodes, nodes) UP(nodes32, nodes32) UP(motion32, motion32) UP(refs, refs) UP(entries, entries) UP(top_level, top_level)
  UP(materials, materials) UP(textures, textures) UP(perlins, perlins) UP(images, images) UP(texels, texels)
  UP(top_box32, top_box32) UP(gravity_spheres, gravity_spheres) UP(gravity_y, gravity_y)
#undef UP
  v.n_top_level = (int32_t)fs.top_level.size();
  v.max_stack = fs.max_stack;
  v.features = fs.features;
  ds->n_nodes = (int32_t)fs.nodes.size();
  // GravitySphere::get_center (hit.rs:369-391) leaves its stored trajectory for a brute-force loop of (time - time0) / 0.001
  // steps, per sphere test, per ray: a shutter far beyond the table is an effectively unbounded kernel.  Renders are limited to
  // RTX_GRAVITY_SLACK_S seconds past the shortest table (10 000 loop steps); rtx_render* return RTX_EINVAL beyond.
  ds->gravity_time_limit = 1e300;
  for (const rt::FlatGravitySphere& g : fs.gravity_spheres)
    ds->gravity_time_limit = std::min(ds->gravity_time_limit, (double)g.table_len * 0.001 + 10.0);
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, ds->device) == hipSuccess && prop.multiProcessorCount > 0)
      ds->n_cu = prop.multiProcessorCount;
    size_t lds = (size_t)(fs.max_stack + 1) * TRACE_BLOCK * sizeof(int32_t);
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_trace_persistent<P_SPHERES, false>, TRACE_BLOCK, lds) == hipSuccess && nb > 0) ds->blocks_per_cu[0] = nb;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_trace_persistent<P_MESH, false>, TRACE_BLOCK, lds) == hipSuccess && nb > 0) ds->blocks_per_cu[1] = nb;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_trace_persistent<P_ANY, false>, TRACE_BLOCK, lds) == hipSuccess && nb > 0) ds->blocks_per_cu[2] = nb;
#ifdef RTX_EXPERIMENTAL_KERNELS
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_trace_stream<P_SPHERES>, TRACE_BLOCK, lds) == hipSuccess && nb > 0) ds->stream_blocks_per_cu[0] = nb;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_trace_stream<P_MESH>, TRACE_BLOCK, lds) == hipSuccess && nb > 0) ds->stream_blocks_per_cu[1] = nb;
#endif
    const char* k = getenv("RTX_TRACE_KERNEL");
    ds->force_simple = (k && strcmp(k, "simple") == 0);
    ds->force_persistent = (k && strcmp(k, "persistent") == 0);
#ifdef RTX_EXPERIMENTAL_KERNELS
    ds->force_stream = (k && strcmp(k, "stream") == 0);
#endif
    ds->vote_diag = (k && strcmp(k, "vote_diag") == 0);
    ds->world_diag = (k && strcmp(k, "world_diag") == 0);
    ds->force_world = ds->world_diag || (k && strcmp(k, "world") == 0);
    ds->force_vote = ds->vote_diag || (k && strcmp(k, "vote") == 0);
    ds->force_wave = (k && strcmp(k, "wavefront") == 0);
    {
      const char* wp = getenv("RTX_WF_PATHS");
      if (wp && atol(wp) >= 256 && atol(wp) <= (1l << 27)) ds->wave_paths = (uint32_t)atol(wp);
      const char* wr = getenv("RTX_WF_REFILL");
      if (wr && atoi(wr) >= 1 && atoi(wr) <= 64) ds->wave_refill = (uint32_t)atoi(wr);
      const char* wo = getenv("RTX_WF_OCC");
      if (wo && atoi(wo) >= 4 && atoi(wo) <= 6) ds->wave_occ = atoi(wo);
      ds->wave_verbose = getenv("RTX_WF_VERBOSE") != nullptr;
      const char* wc = getenv("RTX_WF_CHECK");
      if (wc && atoi(wc) >= 1 && atoi(wc) <= 4096) ds->wave_check = (uint32_t)atoi(wc);
    }
    {
      const size_t lds_ring = lds + (TRACE_BLOCK / 64) * RING_BYTES_PER_WAVE;
      int nb0 = 0, nb1 = 0;
      const char* rg = getenv("RTX_RING");
      // preset 0
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb0, k_trace_vote<P_SPHERES, false, false, false>, TRACE_BLOCK, lds) != hipSuccess) nb0 = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb1, k_trace_vote<P_SPHERES, false, true, false>, TRACE_BLOCK, lds_ring) != hipSuccess) nb1 = 0;
      ds->vote_ring[0] = nb1 > 0 && nb1 >= nb0 && lds_ring <= 64 * 1024;
      if (rg && nb1 > 0 && lds_ring <= 64 * 1024) ds->vote_ring[0] = atoi(rg) != 0;
      nb = ds->vote_ring[0] ? nb1 : nb0;
      if (nb > 0) ds->vote_blocks_per_cu[0] = nb;
      // preset 1
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb0, k_trace_vote<P_MESH, false, false, false>, TRACE_BLOCK, lds) != hipSuccess) nb0 = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb1, k_trace_vote<P_MESH, false, true, false>, TRACE_BLOCK, lds_ring) != hipSuccess) nb1 = 0;
      ds->vote_ring[1] = false;  // measured on the mesh room: no gain, and the ring's live state spills 50 dwords there
      if (rg && nb1 > 0 && lds_ring <= 64 * 1024) ds->vote_ring[1] = atoi(rg) != 0;
      nb = ds->vote_ring[1] ? nb1 : nb0;
      if (nb > 0) ds->vote_blocks_per_cu[1] = nb;
    }
    ds->single_bvh = fs.top_level.size() == 1 && fs.entries[fs.top_level[0]].kind == rt::ENTRY_BVH;
    {
      int n_bvh = 0, n_other = 0;
      for (size_t k = 0; k < fs.top_level.size(); ++k) {
        const int32_t kind = fs.entries[fs.top_level[k]].kind;
        if (kind == rt::ENTRY_BVH) { ++n_bvh; ds->vote_bvh_pos = (int32_t)k; }
        else if (kind != rt::ENTRY_PRIM) ++n_other;
      }
      ds->vote_ok = n_bvh == 1 && n_other == 0;
      if (ds->vote_ok && fs.top_level.size() <= 9 && !(getenv("RTX_VOTE_TOP") && atoi(getenv("RTX_VOTE_TOP")) == 0)) {
        ds->vote_top.n = 0;
        ds->vote_top.all_rects = 1;
        for (size_t k = 0; k < fs.top_level.size(); ++k) {
          if ((int32_t)k == ds->vote_bvh_pos) continue;
          const rt::PrimRef ref = (rt::PrimRef)fs.entries[fs.top_level[k]].a;
          ds->vote_top.k[ds->vote_top.n] = (int32_t)k;
          ds->vote_top.ref[ds->vote_top.n] = (uint32_t)ref;
          if (rt::primref_type(ref) == rt::PRIM_RECT) ds->vote_top.rect[ds->vote_top.n] = fs.rects[rt::primref_index(ref)];
          else ds->vote_top.all_rects = 0;
          ds->vote_top.n += 1;
        }
        if (!ds->vote_top.all_rects) ds->vote_top.n = -1;
      }
      ds->wave_ok = ds->vote_ok;
      if (ds->vote_ok) {
        const rt::FlatEntry& be = fs.entries[fs.top_level[ds->vote_bvh_pos]];
        bool pure = be.c > 0 && rt::primref_type(fs.refs[be.b]) == rt::PRIM_TRIANGLE;
        const uint32_t t0 = pure ? rt::primref_index(fs.refs[be.b]) : 0u;
        for (int32_t k = 0; pure && k < be.c; ++k)
          pure = fs.refs[be.b + k] == rt::make_primref(rt::PRIM_TRIANGLE, t0 + (uint32_t)k);
        const char* tb = getenv("RTX_TRI_DIRECT");
        if (pure && !(tb && atoi(tb) == 0)) ds->vote_tri_base = (int32_t)t0;
      }
    }
    // 4-wide culling tree (see FlatNode4) for every BVH of the scene: big triangle meshes under k_trace_vote, and any
    // world that takes k_trace_persistent (Book-2: two BVHs walked per bounce, each step a dependent L2 fetch).
    // RTX_WIDE=0/1 overrides the size tests.
    {
      const bool spheres_preset = (fs.features & ~P_SPHERES) == 0;
      const bool mesh_preset = !spheres_preset && (fs.features & ~P_MESH) == 0;
      const int preset_i = spheres_preset ? 0 : (mesh_preset ? 1 : 2);
      const char* wd = getenv("RTX_WIDE");
      const bool for_vote = ds->vote_ok && mesh_preset;
      const bool for_pers = !(ds->vote_ok && preset_i < 2) && preset_i >= 1;
      bool want_wide = (for_vote && fs.nodes.size() >= 4096) || (for_pers && fs.nodes.size() >= 256);
      if (wd) want_wide = (for_vote || for_pers) && atoi(wd) != 0 && !fs.nodes.empty();
      if (want_wide) {
        std::vector<FlatNode4> wide(fs.nodes.size());
        memset(wide.data(), 0, wide.size() * sizeof(FlatNode4));
        int peak = 0;
        for (const rt::FlatEntry& e : fs.entries)
          if (e.kind == rt::ENTRY_BVH) peak = std::max(peak, build_wide_nodes(fs.nodes, e.a, &wide));
        ds->wide_levels = peak + 1;  // (the spare level is the bottom slot of LdsStackB; walk_node_step4 needs none of its own)
        const size_t wide_lds = (size_t)ds->wide_levels * TRACE_BLOCK * sizeof(int32_t);
        int nbw = 0;
        bool ok = wide_lds <= 64 * 1024;
        if (ok && for_vote) ok = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nbw, k_trace_vote<P_MESH, false, false, true>, TRACE_BLOCK, wide_lds) == hipSuccess && nbw > 0;
        if (ok && for_vote) ds->wide_blocks_per_cu = nbw;
        if (ok) {
          int n1 = 0, n2 = 0;
          if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n1, k_trace_persistent<P_MESH, true>, TRACE_BLOCK, wide_lds) == hipSuccess && n1 > 0) ds->wide_pers_blocks_per_cu[1] = n1;
          if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n2, k_trace_persistent<P_ANY, true>, TRACE_BLOCK, wide_lds) == hipSuccess && n2 > 0) ds->wide_pers_blocks_per_cu[2] = n2;
          ok = n1 > 0 && n2 > 0;
        }
        if (ok) {
          const FlatNode4* dptr = nullptr;
          if ((st = upload_array(ds, wide, &dptr)) != RTX_OK) { free_device_scene(ds); return st; }
          ds->nodes4 = dptr;
        }
        if (wd) fprintf(stderr, "[rtx] RTX_WIDE: 4-wide tree %s (%d stack levels)\n", ds->nodes4 ? "on" : "off", ds->wide_levels);
        if (getenv("RTX_VALIDATE")) {  // walk every wide tree on the host: codes in range, stack use within wide_levels
          for (const rt::FlatEntry& e : fs.entries) {
            if (e.kind != rt::ENTRY_BVH) continue;
            std::vector<std::pair<int32_t, int>> todo;  // (code, stack entries below it)
            todo.push_back({e.a, 0});
            size_t visited = 0, bad = 0; int deepest = 0;
            while (!todo.empty()) {
              auto [code, below] = todo.back(); todo.pop_back();
              if (code < 0) { if (rt::leaf_first(code) + rt::leaf_count(code) > (uint32_t)e.c) ++bad; continue; }
              if ((size_t)code >= wide.size()) { ++bad; continue; }
              ++visited;
              int nk = 0;
              for (int k = 0; k < 4; ++k) if (wide[code].child[k] != 0x7fffffff) ++nk;
              deepest = std::max(deepest, below + nk);
              for (int k = 0; k < 4; ++k) if (wide[code].child[k] != 0x7fffffff) todo.push_back({wide[code].child[k], below + nk - 1});
            }
            fprintf(stderr, "[rtx] RTX_VALIDATE: BVH root %d refs %d: %zu wide nodes walked, %zu bad codes, deepest stack %d of %d levels, sizeof(real) %zu\n",
                    e.a, e.c, visited, bad, deepest, ds->wide_levels, sizeof(rt::real));
          }
        }
      }
    }
    {
      const char* ml = getenv("RTX_MAT_LDS");
      if (!fs.materials.empty() && !fs.textures.empty() && fs.materials.size() <= 16 && fs.textures.size() <= 16 && !(ml && atoi(ml) == 0)) {
        ds->vote_tables = (uint32_t)fs.materials.size() | ((uint32_t)fs.textures.size() << 16);
        ds->vote_tables_bytes = fs.materials.size() * sizeof(rt::FlatMaterial) + fs.textures.size() * sizeof(rt::FlatTexture);
      }
    }
    {
      const std::vector<WorldDesc> wd = build_world_desc(fs);
      const WorldDesc* dptr = nullptr;
      if ((st = upload_array(ds, wd, &dptr)) != RTX_OK) { free_device_scene(ds); return st; }
      ds->world_desc = dptr;
    }
    {
      const char* wth = getenv("RTX_WORLD_THRESHOLD");
      if (wth && atoi(wth) >= 0 && atoi(wth) <= 64) ds->world_threshold = (uint32_t)atoi(wth);
      for (int wd = 0; wd < 2; ++wd) {
        const uint32_t levels = (uint32_t)(wd ? ds->wide_levels : fs.max_stack + 1);
        size_t wl = (size_t)levels * TRACE_BLOCK * sizeof(int32_t) + (size_t)WORLD_SLOT_F64 * TRACE_BLOCK * sizeof(rt::real);
        if (wl > 64 * 1024) continue;
        // Perlin tables in LDS when that costs no resident block (3 per CU at 168 VGPRs: up to 53 KB each)
        if ((wd != 0) == (ds->nodes4 != nullptr)) {
          const char* pl = getenv("RTX_PERLIN_LDS");
          const size_t n_p = fs.perlins.size();
          if (n_p >= 1 && n_p <= 2 && !(pl && atoi(pl) == 0) && wl + n_p * sizeof(rt::FlatPerlin) <= 52 * 1024) {
            ds->world_perlin_lds = (uint32_t)n_p;
            wl += n_p * sizeof(rt::FlatPerlin);
          }
          const char* ml = getenv("RTX_MAT_LDS");
          const size_t mt = fs.materials.size() * sizeof(rt::FlatMaterial) + fs.textures.size() * sizeof(rt::FlatTexture);
          if (!fs.materials.empty() && !fs.textures.empty() && fs.materials.size() <= 64 && fs.textures.size() <= 64 && !(ml && atoi(ml) == 0) &&
              wl + mt <= 52 * 1024) {
            ds->world_mat_lds = (uint32_t)fs.materials.size();
            ds->world_tex_lds = (uint32_t)fs.textures.size();
            wl += mt;
          }
        }
        int n = 0;
#define WORLD_OCC(FEAT, WIDEF, WPS, OUT) if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_trace_world<FEAT, WIDEF, WPS>, TRACE_BLOCK, wl) == hipSuccess && n > 0) OUT = n
        if (wd) { WORLD_OCC(P_BOOK2, true, WORLD_WPS, ds->world_blocks_per_cu[0][1]); WORLD_OCC(P_ANY, true, WORLD_WPS, ds->world_blocks_per_cu[1][1]); WORLD_OCC(P_ALL, true, WORLD_WPS, ds->world_blocks_per_cu[2][1]); WORLD_OCC(P_NO_SPHERE_MEDIA, true, WORLD_WPS, ds->world_blocks_per_cu[3][1]); }
        else { WORLD_OCC(P_BOOK2, false, WORLD_WPS, ds->world_blocks_per_cu[0][0]); WORLD_OCC(P_ANY, false, WORLD_WPS, ds->world_blocks_per_cu[1][0]); WORLD_OCC(P_ALL, false, WORLD_WPS, ds->world_blocks_per_cu[2][0]); WORLD_OCC(P_NO_SPHERE_MEDIA, false, WORLD_WPS, ds->world_blocks_per_cu[3][0]); }
#undef WORLD_OCC
      }
    }
    if (ds->single_bvh && (fs.features & ~P_SPHERES) == 0 && fs.nodes32.size() <= LDSK_MAX_NODES) {
      uint32_t max_count = 0, max_end = 0;
      for (const rt::FlatNode& nd : fs.nodes)
        for (int ch = 0; ch < 2; ++ch)
          if (nd.child[ch] < 0) {
            max_count = std::max(max_count, rt::leaf_count(nd.child[ch]));
            max_end = std::max(max_end, rt::leaf_first(nd.child[ch]) + rt::leaf_count(nd.child[ch]));
          }
      int lds_max = 0;
      (void)hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, ds->device);
      if (lds_max > 160 * 1024) lds_max = 160 * 1024;
      // primitive records in LDS: one per leaf slot, in slot order (trace_lds.inc) -- spheres, or moving spheres when the scene has any
      ds->lds_dims = {(uint32_t)fs.nodes32.size(), max_end, max_end, 0u, LDSK_NODE_DWORDS, 0u};
      if (!fs.moving_spheres.empty()) {
        const char* mc = getenv("RTX_MV_COMMON");
        bool same = !(mc && atoi(mc) == 0);
        for (const rt::FlatMovingSphere& ms : fs.moving_spheres)
          same = same && ms.time0 == fs.moving_spheres[0].time0 && ms.time1 == fs.moving_spheres[0].time1;
        if (same) { ds->mv_common = true; ds->mv_t0 = (double)fs.moving_spheres[0].time0; ds->mv_t1 = (double)fs.moving_spheres[0].time1; }
      }
      if (fs.features & rt::F_MOVING_SPHERE) {  // k_trace_lds<P_SPHERES>: one kind of primitive in LDS (trace_lds.inc: UNI)
        ds->lds_dims.n_uni = (uint32_t)fs.spheres.size();
        ds->lds_dims.n_moving = max_end;
        ds->lds_dims.n_spheres = 0u;
      }
      ds->motion_dims = ds->lds_dims;
      ds->motion_dims.node_dwords = LDSK_MOTION_NODE_DWORDS;
      {
        // slopes along one axis only?  (a scene whose spheres all move the same way; the y-only instantiation exists: Book-1 at HEAD)
        bool moves[3] = {false, false, false};
        for (const rt::FlatMotion32& m : fs.motion32)
          for (int ch = 0; ch < 2; ++ch)
            for (int a = 0; a < 3; ++a) moves[a] = moves[a] || m.dlo[ch][a] != 0.0f || m.dhi[ch][a] != 0.0f;
        const char* ma = getenv("RTX_MOTION_AXIS");
        if (!fs.motion32.empty() && moves[1] && !moves[0] && !moves[2] && !(ma && atoi(ma) == 0)) {
          ds->motion_axis = 1;
          ds->motion_dims.node_dwords = LDSK_MOTION1_NODE_DWORDS;
        }
      }
      const uint32_t levels = (uint32_t)fs.max_stack + 1u;
      const char* ck = getenv("RTX_CHUNK");
      if (ck && atoi(ck) >= 64 && atoi(ck) <= 65536) ds->lds_chunk = (uint32_t)atoi(ck);
      const char* sl = getenv("RTX_SCENE_LDS");
      const char* rg = getenv("RTX_RING");
      const bool want_ring = !(rg && atoi(rg) == 0);
      if (max_count <= 4 && max_end <= LDSK_MAX_SLOTS && !(sl && atoi(sl) == 0) && lds_max > 0) {
        for (uint32_t cap : {64u, 48u}) {
          if (want_ring && !ds->lds_ok && ldsk_layout(levels, cap, ds->lds_dims).total <= (uint32_t)lds_max) {
            ds->lds_ok = true; ds->lds_ring = true; ds->lds_ring_cap = cap;
          }
        }
        if (!ds->lds_ok && ldsk_layout(levels, 0u, ds->lds_dims).total <= (uint32_t)lds_max) { ds->lds_ok = true; ds->lds_ring = false; }
        const rt::FlatEntry& be = fs.entries[fs.top_level[0]];
        // the 4-wide collapse of the tree
        // Measured on C2 and NOT the default: 236 wide nodes, 5.05 steps per ray against 10.6, bit-identical -- and 2.4 % slower
        // (5997 against 6143 Msamples/s): sorting four children by entry distance and four conditional stack writes make a wide
        // step ~2.3 x a binary one, whose near / far order comes for free out of the address.  Kept as the A/B partner (RTX_LDS_WIDE=1).
        const char* lw = getenv("RTX_LDS_WIDE");
        if (ds->lds_ok && lw && atoi(lw) != 0) {
          std::vector<uint32_t> image;
          uint32_t n_wide = 0;
          const uint32_t wl = build_lds_wide_image(fs.nodes, be.a, &image, &n_wide);
          if (wl > 0u) {
            ds->w4_dims = ds->lds_dims;
            ds->w4_dims.n_nodes = n_wide;
            ds->w4_dims.node_dwords = LDSK_WIDE_NODE_DWORDS;
            ds->w4_levels = wl;
            for (uint32_t cap : {64u, 48u}) {
              if (want_ring && !ds->w4_ok && ldsk_layout(wl, cap, ds->w4_dims).total <= (uint32_t)lds_max) {
                ds->w4_ok = true; ds->w4_ring = true; ds->w4_ring_cap = cap;
              }
            }
            if (!ds->w4_ok && ldsk_layout(wl, 0u, ds->w4_dims).total <= (uint32_t)lds_max) { ds->w4_ok = true; ds->w4_ring = false; }
            if (ds->w4_ok) {
              const uint32_t* dptr = nullptr;
              if ((st = upload_array(ds, image, &dptr)) != RTX_OK) { free_device_scene(ds); return st; }
              ds->w4_image = dptr;
            }
          }
        }
        // the time-aware instantiation: the world's one BVH holds moving spheres and came with an interval
        const char* mo = getenv("RTX_MOTION");
        if (ds->lds_ok && !fs.motion32.empty() && (fs.features & rt::F_MOVING_SPHERE) && (double)be.f[0] < (double)be.f[1] && !(mo && atoi(mo) == 0)) {
          ds->motion_t0 = (double)be.f[0]; ds->motion_t1 = (double)be.f[1];
          // (a small ring is worse than none: its refills run with that few lanes -- HEAD Book-1, ring of 16: 3137 Msamples/s against
          // 3774 without; ring of 32 on the final build of round 3: 4963 against 5196.  Rings are 64 or 48 entries, or absent.)
          for (uint32_t cap : {64u, 48u}) {
            if (want_ring && !ds->motion_ok && ldsk_layout(levels, cap, ds->motion_dims).total <= (uint32_t)lds_max) {
              ds->motion_ok = true; ds->motion_ring = true; ds->motion_ring_cap = cap;
            }
          }
          if (!ds->motion_ok && ldsk_layout(levels, 0u, ds->motion_dims).total <= (uint32_t)lds_max) { ds->motion_ok = true; ds->motion_ring = false; }
        }
      }
      if (ds->lds_ok) {
        // the limit is a property of the function, not of this scene: raise it to the device maximum once, so that
        // scenes uploaded earlier (with other LDS sizes) keep launching
        const int bytes = lds_max;
        hipError_t ae = hipSuccess;
#define LDS_ATTR(FEAT, RINGF, MOTIONF, W4F) if (ae == hipSuccess) ae = hipFuncSetAttribute((const void*)k_trace_lds<FEAT, RINGF, MOTIONF, W4F>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes)
        LDS_ATTR(P_SPHERES, true, 0u, false); LDS_ATTR(P_SPHERES, false, 0u, false); LDS_ATTR(P_STATIC_SPHERES, true, 0u, false); LDS_ATTR(P_STATIC_SPHERES, false, 0u, false);
        LDS_ATTR(P_SPHERES, true, 1u, false); LDS_ATTR(P_SPHERES, false, 1u, false); LDS_ATTR(P_SPHERES, true, 3u, false); LDS_ATTR(P_SPHERES, false, 3u, false);
        LDS_ATTR(P_SPHERES, true, 0u, true); LDS_ATTR(P_SPHERES, false, 0u, true); LDS_ATTR(P_STATIC_SPHERES, true, 0u, true); LDS_ATTR(P_STATIC_SPHERES, false, 0u, true);
#undef LDS_ATTR
        if (ae != hipSuccess) { (void)hipGetLastError(); ds->lds_ok = false; }
      }
      if (sl) fprintf(stderr, "[rtx] RTX_SCENE_LDS: 4-wide tree %s (%u nodes, %u levels, ring of %u, %u B)\n", ds->w4_ok ? "on" : "off", ds->w4_dims.n_nodes, ds->w4_levels,
                      ds->w4_ring ? ds->w4_ring_cap : 0u, ds->w4_ok ? ldsk_layout(ds->w4_levels, ds->w4_ring ? ds->w4_ring_cap : 0u, ds->w4_dims).total : 0u);
      if (sl) fprintf(stderr, "[rtx] RTX_SCENE_LDS: k_trace_lds %s (ring of %u, %u B of LDS); time-aware boxes %s (ring of %u, %u B)\n", ds->lds_ok ? "on" : "off",
                      ds->lds_ring ? ds->lds_ring_cap : 0u, ldsk_layout(levels, ds->lds_ring ? ds->lds_ring_cap : 0u, ds->lds_dims).total,
                      ds->motion_ok ? "on" : "off", ds->motion_ring ? ds->motion_ring_cap : 0u,
                      ds->motion_ok ? ldsk_layout(levels, ds->motion_ring ? ds->motion_ring_cap : 0u, ds->motion_dims).total : 0u);
    }
#ifdef RTX_EXPERIMENTAL_KERNELS
    ds->wq_diag = (k && strcmp(k, "wq_diag") == 0);
    ds->force_wq = ds->wq_diag || (k && strcmp(k, "wq") == 0);
    if (ds->single_bvh && (fs.features & ~P_SPHERES) == 0 && fs.nodes.size() <= WQ_MAX_NODES) {
      const rt::FlatEntry& be = fs.entries[fs.top_level[0]];
      uint32_t max_count = 0, max_end = 0;
      for (const rt::FlatNode& nd : fs.nodes)
        for (int ch = 0; ch < 2; ++ch)
          if (nd.child[ch] < 0) {
            max_count = std::max(max_count, rt::leaf_count(nd.child[ch]));
            max_end = std::max(max_end, rt::leaf_first(nd.child[ch]) + rt::leaf_count(nd.child[ch]));
          }
      int lds_max = 0;
      (void)hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, ds->device);
      if (lds_max > 160 * 1024) lds_max = 160 * 1024;
      ds->wq_levels = (uint32_t)fs.max_stack + 1u;
      ds->wq_paths = lds_max > 0 ? wq_paths_for(ds->wq_levels, (uint32_t)lds_max) : 0u;
      const char* wp = getenv("RTX_WQ_PATHS");
      if (wp && atoi(wp) >= 64 && (uint32_t)atoi(wp) <= ds->wq_paths) ds->wq_paths = (uint32_t)atoi(wp) & ~63u;
      const char* ww = getenv("RTX_WQ_WALKERS");
      ds->wq_walkers = ds->wq_paths >= 192 ? (ds->wq_paths - 128) / 64 : 1;
      if (ww && atoi(ww) >= 1 && atoi(ww) <= 16) ds->wq_walkers = (uint32_t)atoi(ww);
      (void)be;
      const char* wb = getenv("RTX_WQ_BATCH");
      if (wb && atoi(wb) >= 1 && atoi(wb) <= 64) ds->wq_batch_min = (uint32_t)atoi(wb);
      ds->wq_ok = ds->wq_paths > 0 && max_count <= 4 && max_end <= WQ_MAX_SLOTS;
      if (ds->wq_ok && ds->force_wq) {
        const WqLayout L = wq_layout(ds->wq_paths, ds->wq_levels);
        (void)L;
        if (hipFuncSetAttribute((const void*)k_trace_wq<P_SPHERES, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max) != hipSuccess ||
            hipFuncSetAttribute((const void*)k_trace_wq<P_SPHERES, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max) != hipSuccess) {
          (void)hipGetLastError();
          ds->wq_ok = false;
        }
      }
    }
    if (ds->force_wq)
      fprintf(stderr, "[rtx] RTX_TRACE_KERNEL=wq: %s (paths %u, stack levels %u, walkers %u)\n",
              ds->wq_ok ? "applies" : "does NOT apply to this world, default kernel runs", ds->wq_paths, ds->wq_levels, ds->wq_walkers);
#endif
    // a leaf step costs about (primitives per leaf) x 1.3 node steps for spheres: vote weight 1 for single-primitive
    // leaves, 3 otherwise (measured on C2 / HEAD / C4)
    {
      uint32_t max_count = 1;
      for (const rt::FlatNode& nd : fs.nodes)
        for (int ch = 0; ch < 2; ++ch)
          if (nd.child[ch] < 0) max_count = std::max(max_count, rt::leaf_count(nd.child[ch]));
      ds->leaf_weight = max_count <= 1 ? 1u : 3u;
      ds->single_leaf = max_count <= 1;
      // latency-bound wide walks: measured best on the dragon room (639 vs 575 Msamples/s)
      if (ds->nodes4) { ds->leaf_weight = 1u; ds->walk_threshold = 24u; ds->regen_min = 8u; }  // regeneration waits for 8 lanes: 803 -> 824 on C4 (4: 817, 16: 801)
      // k_trace_lds since the ground is asked first and the node step got shorter (round 3): C2 10 / 12 / 14 / 16 / 18 -> 6153 / 6188 /
      // 6185 / 6170 / 6140, HEAD Book-1 3859 / 3850 / 3830 / 3808 / 3750 Msamples/s
      else if (ds->lds_ok) ds->walk_threshold = 12u;
    }
    const char* rm = getenv("RTX_REGEN_MIN");
    if (rm && atoi(rm) >= 1 && atoi(rm) <= 64) ds->regen_min = (uint32_t)atoi(rm);
    const char* lw = getenv("RTX_LEAF_WEIGHT");
    if (lw && atoi(lw) >= 1 && atoi(lw) <= 64) ds->leaf_weight = (uint32_t)atoi(lw);
    const char* pp = getenv("RTX_PASS_PIPELINE");
    if (pp) ds->pass_pipeline = atoi(pp) != 0;
    const char* sg = getenv("RTX_SINGLE_LEAF");
    if (sg && atoi(sg) == 0) ds->single_leaf = false;  // A/B: the general leaf loop on a tree of one-primitive leaves
    const char* wt = getenv("RTX_WALK_THRESHOLD");
    if (wt && atoi(wt) >= 1 && atoi(wt) <= 64) ds->walk_threshold = (uint32_t)atoi(wt);
  }
  *out = ds;
  return RTX_OK;
}

static rtx_status scene_trim_impl(DeviceScene* ds) {
  int cur = -1;
  HIP_TRY(hipGetDevice(&cur));
  if (cur != ds->device) { set_error("rtx_scene_trim: scene lives on a different device than the current one"); return RTX_EINVAL; }
  HIP_TRY(hipDeviceSynchronize());
  if (ds->samples) { HIP_TRY(hipFree(ds->samples)); ds->samples = nullptr; ds->samples_bytes = 0; }
  if (ds->accum) { HIP_TRY(hipFree(ds->accum)); ds->accum = nullptr; ds->accum_bytes = 0; }
  return RTX_OK;
}

}  // namespace rtx

using namespace rtx;

#ifdef RTX_F32_TU
#include "f32_entry.inc"   // the f32 compilation's side of f32_bridge.hpp
#else
#include "f32_convert.inc" // f64 flat arrays -> the f32 compilation's layouts

// Every render entry point funnels through here: the scene handle says which compilation owns the device scene.
template <bool COUNT>
static rtx_status render_any(const rtx_scene* s, const RtxCamera* cam, const RtxConfig* cfg, const RtxShard* shard,
                             double* d_accum_rgb, uint8_t* d_rgb8, hipStream_t stream, RtxRenderStats* stats) {
  if (!s) { set_error("render: NULL scene, camera or config"); return RTX_EINVAL; }
  if (s->f32) {
    if (COUNT) { set_error("rtx_render_count: the work counters belong to the f64 path (upload the scene with rtx_scene_upload)"); return RTX_EUNSUPPORTED; }
    return rtx_f32_render(s->device_scene, cam, cfg, shard, d_accum_rgb, d_rgb8, (void*)stream, stats);
  }
  return render_impl<COUNT>(scene_device(s), cam, cfg, shard, d_accum_rgb, d_rgb8, stream, stats);
}

extern "C" {

rtx_status rtx_scene_upload(const rtx_flat* f, rtx_scene** out) {
  if (!f || !out) { set_error("rtx_scene_upload: NULL argument"); return RTX_EINVAL; }
  *out = nullptr;
  DeviceScene* ds = nullptr;
  rtx_status st = scene_upload_impl(*flat_of(f), &ds);
  if (st != RTX_OK) return st;
  *out = make_scene_handle(ds, 0);
  return RTX_OK;
}

rtx_status rtx_scene_upload_f32(const rtx_flat* f, rtx_scene** out) {
  if (!f || !out) { set_error("rtx_scene_upload_f32: NULL argument"); return RTX_EINVAL; }
  *out = nullptr;
  void* ds32 = nullptr;
  rtx_status st = upload_as_f32(*flat_of(f), &ds32);
  if (st != RTX_OK) return st;
  *out = make_scene_handle((DeviceScene*)ds32, 1);
  return RTX_OK;
}

int32_t rtx_scene_is_f32(const rtx_scene* s) { return s && s->f32 ? 1 : 0; }

rtx_status rtx_scene_trim(rtx_scene* s) {
  if (!s) { set_error("rtx_scene_trim: NULL scene"); return RTX_EINVAL; }
  if (s->f32) return rtx_f32_trim(s->device_scene);
  return scene_trim_impl(scene_device(s));
}

void rtx_scene_destroy(rtx_scene* s) {
  if (!s) return;
  if (s->f32) rtx_f32_destroy(s->device_scene);
  else free_device_scene(scene_device(s));
  free_scene_handle(s);
}

rtx_status rtx_render_device(const rtx_scene* s, const RtxCamera* cam, const RtxConfig* cfg,
                             const RtxShard* shard, double* d_accum_rgb, uint8_t* d_rgb8,
                             void* hip_stream, RtxRenderStats* stats) {
  return render_any<false>(s, cam, cfg, shard, d_accum_rgb, d_rgb8, (hipStream_t)hip_stream, stats);
}

rtx_status rtx_render_count(const rtx_scene* s, const RtxCamera* cam, const RtxConfig* cfg,
                            const RtxShard* shard, RtxRenderStats* stats) {
  if (!stats) { set_error("rtx_render_count: stats is NULL"); return RTX_EINVAL; }
  return render_any<true>(s, cam, cfg, shard, nullptr, nullptr, (hipStream_t) nullptr, stats);
}

rtx_status rtx_device_math(int32_t fn, const double* x, const double* y, int64_t n, double* out) {
  if (!x || !y || !out || n < 0) { set_error("rtx_device_math: bad argument"); return RTX_EINVAL; }
  if (n == 0) return RTX_OK;
  double *dx = nullptr, *dy = nullptr, *dout = nullptr;
  size_t bytes = (size_t)n * sizeof(double);
  rtx_status st = RTX_OK;
  auto fail = [&](const char* what, hipError_t e) { set_error(std::string(what) + ": " + hipGetErrorString(e)); st = RTX_EHIP; };
  hipError_t e;
  if ((e = hipMalloc((void**)&dx, bytes)) != hipSuccess) fail("hipMalloc", e);
  if (st == RTX_OK && (e = hipMalloc((void**)&dy, bytes)) != hipSuccess) fail("hipMalloc", e);
  if (st == RTX_OK && (e = hipMalloc((void**)&dout, bytes)) != hipSuccess) fail("hipMalloc", e);
  if (st == RTX_OK && (e = hipMemcpy(dx, x, bytes, hipMemcpyHostToDevice)) != hipSuccess) fail("hipMemcpy", e);
  if (st == RTX_OK && (e = hipMemcpy(dy, y, bytes, hipMemcpyHostToDevice)) != hipSuccess) fail("hipMemcpy", e);
  if (st == RTX_OK) {
    hipLaunchKernelGGL(k_device_math, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (int)fn, dx, dy, (long long)n, dout);
    if ((e = hipGetLastError()) != hipSuccess) fail("k_device_math", e);
  }
  if (st == RTX_OK && (e = hipMemcpy(out, dout, bytes, hipMemcpyDeviceToHost)) != hipSuccess) fail("hipMemcpy", e);
  if (dx) (void)hipFree(dx);
  if (dy) (void)hipFree(dy);
  if (dout) (void)hipFree(dout);
  return st;
}

rtx_status rtx_device_stream(uint64_t seed, uint64_t pixel, uint32_t sample, int32_t n, double* out) {
  if (!out || n < 0) { set_error("rtx_device_stream: bad argument"); return RTX_EINVAL; }
  if (n == 0) return RTX_OK;
  double* d = nullptr;
  HIP_TRY(hipMalloc((void**)&d, (size_t)n * sizeof(double)));
  hipLaunchKernelGGL(k_device_stream, dim3(1), dim3(64), 0, 0, (unsigned long long)seed, (unsigned long long)pixel, (unsigned int)sample, (int)n, d);
  hipError_t e = hipMemcpy(out, d, (size_t)n * sizeof(double), hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (e != hipSuccess) { set_error(std::string("rtx_device_stream: ") + hipGetErrorString(e)); return RTX_EHIP; }
  return RTX_OK;
}

rtx_status rtx_render(const rtx_scene* s, const RtxCamera* cam, const RtxConfig* cfg, RtxFrame* out) {
  if (!out) { set_error("rtx_render: NULL frame"); return RTX_EINVAL; }
  RtxShard sh;
  rtx_status st = validate(s, cam, cfg, nullptr, &sh);
  if (st != RTX_OK) return st;
  size_t npix = (size_t)cfg->image_width * (size_t)rtx_image_height(cfg);
  double* d_accum = nullptr;
  uint8_t* d_rgb = nullptr;
  if (hipMalloc((void**)&d_accum, npix * 24) != hipSuccess || hipMalloc((void**)&d_rgb, npix * 3) != hipSuccess) {
    if (d_accum) (void)hipFree(d_accum);
    set_error("rtx_render: hipMalloc of the frame failed");
    return RTX_EHIP;
  }
  st = render_any<false>(s, cam, cfg, nullptr, d_accum, d_rgb, (hipStream_t) nullptr, nullptr);
  if (st == RTX_OK) {
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess && out->accum_rgb) e = hipMemcpy(out->accum_rgb, d_accum, npix * 24, hipMemcpyDeviceToHost);
    if (e == hipSuccess && out->rgb8) e = hipMemcpy(out->rgb8, d_rgb, npix * 3, hipMemcpyDeviceToHost);
    if (e != hipSuccess) { set_error(std::string("rtx_render: ") + hipGetErrorString(e)); st = RTX_EHIP; }
  }
  (void)hipFree(d_accum);
  (void)hipFree(d_rgb);
  return st;
}

}  // extern "C"

// render_scene_with_time (world.rs:1249-1330): one frame of the video experiment on a scene that stays resident.
extern "C" rtx_status rtx_render_scene_with_time(const rtx_scene* s, double t0, double t1, const char* path,
                                                 int32_t row_chunk_compat, const RtxConfig* overrides) {
  using namespace rtx;
  if (!s || !path) { set_error("rtx_render_scene_with_time: NULL argument"); return RTX_EINVAL; }
  // world.rs:1252-1275: everything about the frame is a constant there
  const double lookfrom[3] = {13, 2, 3}, lookat[3] = {0, 0, 0}, vup[3] = {0, 1, 0};
  RtxCamera cam;
  rtx_status st = rtx_camera_new(lookfrom, lookat, vup, 20.0, 1.0, 0.1, 10.0, t0, t1, &cam);
  if (st != RTX_OK) return st;  // t0 >= t1: gen_range(t0..t1) panics in the reference
  RtxConfig cfg;
  st = rtx_config_new(1.0, 500, 500, 50, 11, &cfg);  // THREADS = 11 (world.rs:18)
  if (st != RTX_OK) return st;
  cfg.background[0] = 0.7; cfg.background[1] = 0.8; cfg.background[2] = 1.0;
  cfg.row_chunk_compat = row_chunk_compat ? 1 : 0;
  if (overrides) {
    if (overrides->image_width > 0) cfg.image_width = overrides->image_width;
    if (overrides->samples_per_pixel > 0) cfg.samples_per_pixel = overrides->samples_per_pixel;
    if (overrides->max_depth > 0) cfg.max_depth = overrides->max_depth;
    cfg.seed = overrides->seed;
    cfg.sample_buffer_bytes = overrides->sample_buffer_bytes;
  }
  const size_t npix = (size_t)cfg.image_width * (size_t)rtx_image_height(&cfg);
  std::vector<uint8_t> rgb(npix * 3);
  RtxFrame frame = {nullptr, rgb.data()};
  st = rtx_render(s, &cam, &cfg, &frame);
  if (st != RTX_OK) return st;
  return rtx_write_ppm(path, cfg.image_width, rtx_image_height(&cfg), rgb.data());  // screen.write_to_ppm_file(path)
}

#include "multi.inc"  // rtx_multi_*: one process, several GPUs, one RCCL gather
#endif  // !RTX_F32_TU
