// ORACLE O2 -- TEST INFRASTRUCTURE, NOT A RENDER PATH OF THE PRODUCT.
//
// The product's shared host/device core (ray-tracing-series-rust_amd/csrc/core/*.hpp: the
// exact functions the HIP kernels inline) compiled for the host with g++ -ffp-contract=off and
// driven by a plain nested loop in the order of render_scene (/root/reference/src/world.rs:
// 1207-1226: rows, pixels, samples; sequential per-pixel sum; get_normalized_color).  It walks
// the product's FLATTENED scene (rtx_flat), so it checks two things independently of the GPU:
//   * O2 == O1 : the flattener, the SAH BVH and the iterative traversal reproduce the literal
//                object-graph semantics (oracle/o1_literal.cpp);
//   * GPU == O2: the kernels compute what the same source computes on a CPU, bit for bit.
// It also yields the exact work counters behind the algorithmic-bytes model.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call this.
// Parity status vs the reference's OUTPUT: statistical pins against its three images, bit-level parity unpinned
// (see o1_literal.cpp header).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <atomic>
#include <thread>
#include <vector>
#include "../ray-tracing-series-rust_amd/csrc/core/integrator.hpp"
#include "../ray-tracing-series-rust_amd/csrc/host/flat_scene.hpp"
#include "oracle_abi.h"

namespace {

rt::RenderParams make_params(const OracleCamera* cam, const OracleConfig* cfg) {
  rt::RenderParams rp;
  static_assert(sizeof(OracleCamera) == sizeof(rt::FlatCamera), "camera layout");
  memcpy(&rp.cam, cam, sizeof(rt::FlatCamera));
  rp.background = rt::v3(cfg->background[0], cfg->background[1], cfg->background[2]);
  rp.image_width = cfg->image_width;
  rp.image_height = cfg->image_height;
  rp.samples_per_pixel = cfg->samples_per_pixel;
  rp.max_depth = cfg->max_depth;
  rp.seed = cfg->seed;
  return rp;
}

void add_counters(OracleCounters* dst, const rt::TraceCounters& c) {
  dst->box_tests += c.box_tests; dst->sphere_tests += c.sphere_tests;
  dst->moving_sphere_tests += c.moving_sphere_tests; dst->rect_tests += c.rect_tests;
  dst->triangle_tests += c.triangle_tests; dst->scatters += c.scatters; dst->texels += c.texels;
  dst->perlin_calls += c.perlin_calls; dst->rays += c.rays; dst->samples += c.samples;
}

}  // namespace

extern "C" {

int oracle_o2_render(const void* flat, const OracleCamera* cam, const OracleConfig* cfg,
                     int32_t shard_index, int32_t shard_count, int32_t block_rows,
                     double* accum_rgb, uint8_t* rgb8, OracleCounters* counters) {
  if (!flat || !cam || !cfg || shard_count <= 0 || block_rows <= 0) return 1;
  const rtx::FlatScene& fs = *(const rtx::FlatScene*)flat;
  const rt::SceneView sv = fs.view();
  const rt::RenderParams rp = make_params(cam, cfg);
  const int32_t w = rp.image_width, h = rp.image_height;
  std::vector<int32_t> rows;
  for (int32_t j = 0; j < h; ++j)
    if ((j / block_rows) % shard_count == shard_index) rows.push_back(j);
  int32_t row_limit = h;
  if (cfg->row_chunk_compat && cfg->threads > 0) row_limit = (h / cfg->threads) * cfg->threads;
  if (accum_rgb) memset(accum_rgb, 0, sizeof(double) * 3 * rows.size() * (size_t)w);
  if (rgb8) memset(rgb8, 0, 3 * rows.size() * (size_t)w);
  if (counters) memset(counters, 0, sizeof(*counters));
  int threads = cfg->threads > 0 ? cfg->threads : 1;
  std::vector<rt::TraceCounters> tc((size_t)threads);
  for (auto& c : tc) memset(&c, 0, sizeof(c));
  // Work is dealt in 8-pixel pieces of the shard's pixel list (row-major) from one atomic counter: a pixel's value does not depend
  // on who computes it, and a shard of a few rows at a high sample count (the full-spp spot rows of tests/test_gpu_full_size.py)
  // keeps every thread busy.
  const size_t n_pix = rows.size() * (size_t)w, piece = 8;
  std::atomic<size_t> next{0};
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; ++t) {
    pool.emplace_back([&, t]() {
      rt::LocalStack<128> stack;
      stack.n = 0;
      for (;;) {
        const size_t p0 = next.fetch_add(piece);
        if (p0 >= n_pix) break;
        const size_t p1 = p0 + piece < n_pix ? p0 + piece : n_pix;
        for (size_t p = p0; p < p1; ++p) {
          const size_t lr = p / (size_t)w;
          const int32_t i = (int32_t)(p - lr * (size_t)w), j = rows[lr];
          if (j >= row_limit) continue;
          rt::Color pixel = rt::v3(0, 0, 0);
          for (int32_t s = 0; s < rp.samples_per_pixel; ++s) {
            rt::Color c = counters ? rt::trace_sample<rt::F_ALL, true>(sv, rp, (uint32_t)i, (uint32_t)j, (uint32_t)s, stack, &tc[t])
                                   : rt::trace_sample<rt::F_ALL, false>(sv, rp, (uint32_t)i, (uint32_t)j, (uint32_t)s, stack, nullptr);
            pixel += c;
          }
          size_t o = 3 * p;
          if (accum_rgb) { accum_rgb[o] = pixel.x; accum_rgb[o + 1] = pixel.y; accum_rgb[o + 2] = pixel.z; }
          if (rgb8) {
            int32_t c[3];
            rt::tone_map(pixel, (uint32_t)rp.samples_per_pixel, c);
            rgb8[o] = (uint8_t)c[0]; rgb8[o + 1] = (uint8_t)c[1]; rgb8[o + 2] = (uint8_t)c[2];
          }
        }
      }
    });
  }
  for (std::thread& th : pool) th.join();
  if (counters)
    for (const auto& c : tc) add_counters(counters, c);
  return 0;
}

int oracle_o2_sample(const void* flat, const OracleCamera* cam, const OracleConfig* cfg, int32_t i,
                     int32_t j, int32_t sample, double rgb[3]) {
  if (!flat || !cam || !cfg) return 1;
  const rtx::FlatScene& fs = *(const rtx::FlatScene*)flat;
  const rt::SceneView sv = fs.view();
  const rt::RenderParams rp = make_params(cam, cfg);
  rt::LocalStack<128> stack;
  stack.n = 0;
  rt::Color c = rt::trace_sample<rt::F_ALL, false>(sv, rp, (uint32_t)i, (uint32_t)j, (uint32_t)sample, stack, nullptr);
  rgb[0] = c.x; rgb[1] = c.y; rgb[2] = c.z;
  return 0;
}

// ---- probes into the product's shared core (same KATs as the O1 probes) ---------------------
void oracle_core_vec3_ops(const double a[3], const double b[3], double t, double out[24]) {
  rt::Vec3 A = rt::v3(a[0], a[1], a[2]), B = rt::v3(b[0], b[1], b[2]);
  int k = 0;
  auto put = [&](rt::Vec3 v) { out[k++] = v.x; out[k++] = v.y; out[k++] = v.z; };
  put(A + B); put(A - B); put(A * B); put(A * t); put(A / t); put(-A); put(rt::cross(A, B));
  out[k++] = rt::dot(A, B); out[k++] = rt::length_squared(A); out[k++] = rt::length(A);
}
void oracle_core_tone_map(const double sum[3], uint32_t spp, int32_t out[3]) {
  rt::tone_map(rt::v3(sum[0], sum[1], sum[2]), spp, out);
}
void oracle_core_sphere_uv(const double p[3], double uv[2]) { rt::get_sphere_uv(rt::v3(p[0], p[1], p[2]), &uv[0], &uv[1]); }
double oracle_core_reflectance(double cosine, double ref_idx) { return rt::reflectance(cosine, ref_idx); }
void oracle_core_refract(const double uv[3], const double n[3], double ratio, double out[3]) {
  rt::Vec3 r = rt::refract(rt::v3(uv[0], uv[1], uv[2]), rt::v3(n[0], n[1], n[2]), ratio);
  out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void oracle_core_reflect(const double v[3], const double n[3], double out[3]) {
  rt::Vec3 r = rt::reflect(rt::v3(v[0], v[1], v[2]), rt::v3(n[0], n[1], n[2]));
  out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
int oracle_core_aabb_hit(const double mn[3], const double mx[3], const double o[3], const double d[3], double t_min, double t_max) {
  rt::Vec3 inv = rt::v3(1.0 / d[0], 1.0 / d[1], 1.0 / d[2]);
  return rt::aabb_hit(mn, mx, rt::v3(o[0], o[1], o[2]), inv, t_min, t_max) ? 1 : 0;
}
// world_hit over a flattened scene with one ray: out = {t, p.xyz, n.xyz, u, v, front_face}
int oracle_core_world_hit(const void* flat, const double o[3], const double d[3], double time, double t_min,
                          double t_max, uint64_t rng_seed, double out[10]) {
  if (!flat) return -1;
  const rtx::FlatScene& fs = *(const rtx::FlatScene*)flat;
  const rt::SceneView sv = fs.view();
  rt::LocalStack<128> stack;
  stack.n = 0;
  rt::Rng rng = rt::rng_for_sample(rng_seed, 0, 0);
  rt::HitRecord rec;
  rt::Ray r = rt::make_ray(rt::v3(o[0], o[1], o[2]), rt::v3(d[0], d[1], d[2]), time);
  if (!rt::world_hit<rt::F_ALL, false>(sv, r, t_min, t_max, &rec, rng, stack, nullptr)) return 0;
  out[0] = rec.t; out[1] = rec.p.x; out[2] = rec.p.y; out[3] = rec.p.z;
  out[4] = rec.normal.x; out[5] = rec.normal.y; out[6] = rec.normal.z;
  out[7] = rec.u; out[8] = rec.v; out[9] = rec.front_face ? 1.0 : 0.0;
  return 1;
}
// Structural audit of a flattened scene (used by tests/test_host_logic.py): checks that every BVH
// node box encloses its subtree's primitive boxes and every primitive sits in exactly one leaf.
// Returns 0 when consistent, else a positive error code.
static int audit_node(const rtx::FlatScene& fs, const rt::FlatEntry& e, int32_t child, const double* mn, const double* mx,
                      std::vector<int>& seen, int depth, int* max_depth) {
  if (depth > *max_depth) *max_depth = depth;
  if (rt::node_child_is_leaf(child)) {
    uint32_t f = rt::leaf_first(child), k = rt::leaf_count(child);
    if (f + k > (uint32_t)e.c) return 2;
    for (uint32_t i = 0; i < k; ++i) seen[f + i]++;
    return 0;
  }
  if (child < 0 || (size_t)child >= fs.nodes.size()) return 3;
  const rt::FlatNode& n = fs.nodes[child];
  for (int c = 0; c < 2; ++c) {
    for (int a = 0; a < 3; ++a)
      if (mn && (n.bmin[c][a] < mn[a] || n.bmax[c][a] > mx[a])) return 4;
    int rc = audit_node(fs, e, n.child[c], n.bmin[c], n.bmax[c], seen, depth + 1, max_depth);
    if (rc) return rc;
  }
  return 0;
}
int oracle_audit_flat(const void* flat, int32_t* max_depth_out) {
  if (!flat) return -1;
  const rtx::FlatScene& fs = *(const rtx::FlatScene*)flat;
  int max_depth = 0;
  for (const rt::FlatEntry& e : fs.entries) {
    if (e.kind != rt::ENTRY_BVH) continue;
    std::vector<int> seen((size_t)e.c, 0);
    int d = 0;
    int rc = audit_node(fs, e, e.a, nullptr, nullptr, seen, 1, &d);
    if (rc) return rc;
    for (int v : seen)
      if (v != 1) return 5;
    if (d > max_depth) max_depth = d;
  }
  if (max_depth_out) *max_depth_out = max_depth;
  return max_depth <= fs.max_stack + 1 ? 0 : 6;
}

// Audit of the time-aware culling boxes (FlatMotion32): for every BVH that carries them and `n_times` instants of its interval
// (both ends included), every child box -- evaluated in f32 exactly as k_trace_lds does: s32 = fl((t - t0) * inv_dt) clamped to
// [0, 1], plane = fmaf(s32, slope, plane0) -- must contain the f64 box of every sphere below that child at that instant.
// Returns 0 when it does, else a positive code; *checked = (box, primitive, instant) triples looked at.
static void motion_collect(const rtx::FlatScene& fs, const rt::FlatEntry& e, int32_t child, std::vector<rt::PrimRef>* out) {
  if (rt::node_child_is_leaf(child)) {
    for (uint32_t i = 0; i < rt::leaf_count(child); ++i) out->push_back(fs.refs[(size_t)e.b + rt::leaf_first(child) + i]);
    return;
  }
  motion_collect(fs, e, fs.nodes[(size_t)child].child[0], out);
  motion_collect(fs, e, fs.nodes[(size_t)child].child[1], out);
}
int oracle_audit_motion(const void* flat, int32_t n_times, int64_t* checked) {
  if (!flat || n_times < 2) return -1;
  const rtx::FlatScene& fs = *(const rtx::FlatScene*)flat;
  int64_t n_checked = 0;
  if (checked) *checked = 0;
  if (fs.motion32.empty()) return 0;
  if (fs.motion32.size() != fs.nodes.size()) return 1;
  for (const rt::FlatEntry& e : fs.entries) {
    if (e.kind != rt::ENTRY_BVH || e.a < 0 || !(e.f[0] < e.f[1])) continue;
    std::vector<int32_t> todo{e.a};
    while (!todo.empty()) {
      const int32_t n = todo.back();
      todo.pop_back();
      const rt::FlatMotion32& m = fs.motion32[(size_t)n];
      for (int c = 0; c < 2; ++c) {
        const int32_t child = fs.nodes[(size_t)n].child[c];
        if (!rt::node_child_is_leaf(child)) todo.push_back(child);
        std::vector<rt::PrimRef> prims;
        motion_collect(fs, e, child, &prims);
        for (int32_t k = 0; k < n_times; ++k) {
          // instants: the two ends, then a low-discrepancy sweep of the interior
          const double u = k == 0 ? 0.0 : (k == 1 ? 1.0 : std::fmod(0.6180339887498949 * (double)k, 1.0));
          const double t = e.f[0] + u * (e.f[1] - e.f[0]);
          const double inv_dt = 1.0 / (e.f[1] - e.f[0]);
          const float s32 = std::fmin(std::fmax((float)((t - e.f[0]) * inv_dt), 0.f), 1.f);
          for (rt::PrimRef ref : prims) {
            const uint32_t idx = rt::primref_index(ref), ty = rt::primref_type(ref);
            double c3[3], r;
            if (ty == rt::PRIM_SPHERE) { const rt::FlatSphere& sp = fs.spheres[idx]; c3[0] = sp.cx; c3[1] = sp.cy; c3[2] = sp.cz; r = sp.radius; }
            else if (ty == rt::PRIM_MOVING_SPHERE) {
              const rt::FlatMovingSphere& sp = fs.moving_spheres[idx];
              const rt::Vec3 cc = rt::moving_sphere_center(sp, t);
              c3[0] = cc.x; c3[1] = cc.y; c3[2] = cc.z; r = sp.radius;
            } else return 2;  // motion boxes are only built for BVHs of spheres and moving spheres... and whatever else is static
            for (int a = 0; a < 3; ++a) {
              const float lo = std::fmaf(s32, m.dlo[c][a], m.lo0[c][a]), hi = std::fmaf(s32, m.dhi[c][a], m.hi0[c][a]);
              if (!((double)lo <= c3[a] - std::fabs(r)) || !((double)hi >= c3[a] + std::fabs(r))) return 3;
              ++n_checked;
            }
          }
        }
      }
    }
  }
  if (checked) *checked = n_checked;
  return 0;
}
// Mean half-area of the leaf boxes of the time-aware tree at the middle of the interval over that of the static (union) boxes:
// what the time-aware test can cull that the static one cannot (diagnostic for tests / DESIGN).
double oracle_motion_leaf_area_ratio(const void* flat) {
  if (!flat) return -1.0;
  const rtx::FlatScene& fs = *(const rtx::FlatScene*)flat;
  if (fs.motion32.empty()) return 1.0;
  double a_static = 0.0, a_motion = 0.0;
  for (size_t n = 0; n < fs.nodes.size(); ++n)
    for (int c = 0; c < 2; ++c) {
      if (!rt::node_child_is_leaf(fs.nodes[n].child[c])) continue;
      const rt::FlatNode32& q = fs.nodes32[n];
      const rt::FlatMotion32& m = fs.motion32[n];
      double d[3], e[3];
      for (int a = 0; a < 3; ++a) {
        d[a] = (double)q.hi[c][a] - (double)q.lo[c][a];
        e[a] = ((double)m.hi0[c][a] + 0.5 * (double)m.dhi[c][a]) - ((double)m.lo0[c][a] + 0.5 * (double)m.dlo[c][a]);
      }
      a_static += d[0] * d[1] + d[1] * d[2] + d[2] * d[0];
      a_motion += e[0] * e[1] + e[1] * e[2] + e[2] * e[0];
    }
  return a_static > 0.0 ? a_motion / a_static : 1.0;
}

void oracle_philox4x32_10(uint32_t ctr[4], uint32_t k0, uint32_t k1) { rt::philox4x32_10(ctr, k0, k1); }
uint64_t oracle_splitmix64_next(uint64_t* state) {
  rt::HostRng h{*state};
  uint64_t r = rt::host_rng_next_u64(h);
  *state = h.state;
  return r;
}
void oracle_sample_stream(uint64_t seed, uint64_t pixel, uint32_t sample, int32_t n, double* out) {
  rt::Rng g = rt::rng_for_sample(seed, pixel, sample);
  for (int32_t k = 0; k < n; ++k) out[k] = rt::rng_f64(g);
}
// fn: 0 sin, 1 cos, 2 log, 3 acos, 4 atan2(x, y), 5 tan, 6 sqrt
void oracle_rt_math(int32_t fn, const double* x, const double* y, int64_t n, double* out) {
  for (int64_t k = 0; k < n; ++k) {
    switch (fn) {
      case 0: out[k] = rt::rt_sin(x[k]); break;
      case 1: out[k] = rt::rt_cos(x[k]); break;
      case 2: out[k] = rt::rt_log(x[k]); break;
      case 3: out[k] = rt::rt_acos(x[k]); break;
      case 4: out[k] = rt::rt_atan2(x[k], y[k]); break;
      case 5: out[k] = rt::rt_tan(x[k]); break;
      case 10: out[k] = (double)rt::rt_sin_sign(x[k]); break;
      default: out[k] = rt::rt_sqrt(x[k]); break;
    }
  }
}

}  // extern "C"
// The walk k_trace_lds does, on the CPU, for worlds that are ONE BVH of spheres: the f32 culling tree (static boxes, or the
// time-aware boxes when use_motion != 0) visited near child first, every hit decided by the f64 primitive tests, one path at a
// time.  Two uses: (1) its frame must equal oracle_o2_render's bit for bit -- the culling structure cannot be seen; (2) its
// counters say what the time-aware boxes save: counts[0] = rays, [1] = node visits (two box tests each), [2] = primitive tests.
extern "C" int oracle_lds_walk_render(const void* flat, const OracleCamera* cam, const OracleConfig* cfg, int32_t use_motion,
                                      int32_t row_stride, double* accum_rgb, uint64_t counts[3]) {
  if (!flat || !cam || !cfg || row_stride <= 0) return 1;
  const rtx::FlatScene& fs = *(const rtx::FlatScene*)flat;
  if (fs.top_level.size() != 1 || fs.entries[fs.top_level[0]].kind != rt::ENTRY_BVH) return 2;
  if (use_motion && fs.motion32.empty()) return 3;
  const rt::FlatEntry& be = fs.entries[fs.top_level[0]];
  const rt::SceneView sv = fs.view();
  const rt::RenderParams rp = make_params(cam, cfg);
  const int32_t w = rp.image_width, h = rp.image_height;
  std::vector<int32_t> rows;
  for (int32_t j = 0; j < h; j += row_stride) rows.push_back(j);
  const int threads = cfg->threads > 0 ? cfg->threads : 1;
  std::vector<uint64_t> tc((size_t)threads * 3, 0);
  std::atomic<size_t> next{0};
  const double m_t0 = be.f[0], m_inv = use_motion ? 1.0 / (be.f[1] - be.f[0]) : 0.0;
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; ++t) {
    pool.emplace_back([&, t]() {
      std::vector<int32_t> stack(256);
      for (;;) {
        const size_t p = next.fetch_add(1);
        if (p >= rows.size() * (size_t)w) break;
        const int32_t j = rows[p / (size_t)w], i = (int32_t)(p % (size_t)w);
        rt::Color pixel = rt::v3(0, 0, 0);
        for (int32_t s = 0; s < rp.samples_per_pixel; ++s) {
          rt::PathState ps;
          rt::path_begin(rp, (uint32_t)i, (uint32_t)j, (uint32_t)s, &ps);
          for (;;) {
            if (rt::path_bounce_begin(&ps)) break;
            tc[3 * t]++;
            const rt::Ray32 q = rt::make_ray32(ps.ray, rt::ray_t_min(ps.ray));
            const uint32_t dir_neg = rt::ray_dir_neg(ps.ray);
            const float s32 = std::fmin(std::fmax((float)((ps.ray.time - m_t0) * m_inv), 0.f), 1.f);
            float t_max32 = __builtin_huge_valf();
            rt::Closest best; best.t = RT_INFINITY; best.hit = false; best.ref = 0; best.order = 0;
            int n_stack = 0;
            int32_t cur = be.a;
            for (;;) {
              if (cur >= 0) {
                tc[3 * t + 1]++;
                const rt::FlatNode32& nd = fs.nodes32[(size_t)cur];
                float lo[2][3], hi[2][3];
                for (int c = 0; c < 2; ++c)
                  for (int a = 0; a < 3; ++a) {
                    if (use_motion) {
                      const rt::FlatMotion32& m = fs.motion32[(size_t)cur];
                      lo[c][a] = std::fmaf(s32, m.dlo[c][a], m.lo0[c][a]); hi[c][a] = std::fmaf(s32, m.dhi[c][a], m.hi0[c][a]);
                    } else { lo[c][a] = nd.lo[c][a]; hi[c][a] = nd.hi[c][a]; }
                  }
                const int first = (int)((dir_neg >> (uint32_t)nd.axis) & 1u);
                const bool hf = rt::cull32_may_hit(lo[first], hi[first], q, t_max32), hs = rt::cull32_may_hit(lo[1 - first], hi[1 - first], q, t_max32);
                const int32_t cf = nd.child[first], cs = nd.child[1 - first];
                // leaves are kept as their (negative) codes, -1 never occurs as a code (0x80000000 | ...), INT32_MIN + x < -1
                if (hf) { cur = cf; if (hs) stack[(size_t)n_stack++] = cs; }
                else if (hs) cur = cs;
                else if (n_stack > 0) cur = stack[(size_t)--n_stack];
                else break;
              } else {
                const uint32_t f = rt::leaf_first(cur), k = rt::leaf_count(cur);
                for (uint32_t x = 0; x < k; ++x) {
                  tc[3 * t + 2]++;
                  rt::offer_prim<rt::F_ALL, false>(sv, fs.refs[(size_t)be.b + f + x], f + x, ps.ray, rt::ray_t_min(ps.ray), &best, nullptr);
                }
                t_max32 = rt::cull_round_up(best.t);
                if (n_stack > 0) cur = stack[(size_t)--n_stack];
                else break;
              }
            }
            rt::HitRecord rec;
            if (best.hit) rt::prim_finalize<rt::F_ALL>(sv, best.ref, ps.ray, best.t, &rec);
            if (rt::path_bounce_end<rt::F_ALL, false>(sv, rp, &ps, best.hit, rec, nullptr)) break;
          }
          pixel += ps.output;
        }
        if (accum_rgb) { accum_rgb[3 * p] = pixel.x; accum_rgb[3 * p + 1] = pixel.y; accum_rgb[3 * p + 2] = pixel.z; }
      }
    });
  }
  for (std::thread& th : pool) th.join();
  if (counts) {
    counts[0] = counts[1] = counts[2] = 0;
    for (int t = 0; t < threads; ++t) for (int k = 0; k < 3; ++k) counts[k] += tc[3 * (size_t)t + k];
  }
  return 0;
}

// Diagnostic: how many steps a 4-wide collapse of the world's one BVH would take (children sorted by entry distance, nearest
// first, as walk_node_step4 does), for the same paths as oracle_lds_walk_render: counts[0] = rays, [1] = wide node steps,
// [2] = primitive tests, [3] = child boxes tested.  The collapse opens the child with the largest area until four slots are used.
extern "C" int oracle_wide_walk_stats(const void* flat, const OracleCamera* cam, const OracleConfig* cfg, int32_t row_stride, uint64_t counts[4]) {
  if (!flat || !cam || !cfg || row_stride <= 0) return 1;
  const rtx::FlatScene& fs = *(const rtx::FlatScene*)flat;
  if (fs.top_level.size() != 1 || fs.entries[fs.top_level[0]].kind != rt::ENTRY_BVH) return 2;
  const rt::FlatEntry& be = fs.entries[fs.top_level[0]];
  struct W { float lo[4][3], hi[4][3]; int32_t child[4]; int n; };
  std::vector<W> wide(fs.nodes.size());
  {
    std::vector<int32_t> todo{be.a};
    while (!todo.empty()) {
      const int32_t n = todo.back(); todo.pop_back();
      struct S { int32_t code; const float *lo, *hi; double area; };
      auto slot = [&](int32_t node, int c) {
        const rt::FlatNode32& q = fs.nodes32[(size_t)node];
        const double dx = (double)q.hi[c][0] - q.lo[c][0], dy = (double)q.hi[c][1] - q.lo[c][1], dz = (double)q.hi[c][2] - q.lo[c][2];
        return S{q.child[c], q.lo[c], q.hi[c], dx * dy + dy * dz + dz * dx};
      };
      S sl[4] = {slot(n, 0), slot(n, 1), {}, {}};
      int ns = 2;
      while (ns < 4) {
        int bi = -1; double ba = -1.0;
        for (int k = 0; k < ns; ++k) if (sl[k].code >= 0 && sl[k].area > ba) { ba = sl[k].area; bi = k; }
        if (bi < 0) break;
        const int32_t open = sl[bi].code;
        sl[bi] = slot(open, 0); sl[ns++] = slot(open, 1);
      }
      W& w = wide[(size_t)n];
      w.n = ns;
      for (int k = 0; k < ns; ++k) {
        for (int a = 0; a < 3; ++a) { w.lo[k][a] = sl[k].lo[a]; w.hi[k][a] = sl[k].hi[a]; }
        w.child[k] = sl[k].code;
        if (sl[k].code >= 0) todo.push_back(sl[k].code);
      }
    }
  }
  const rt::SceneView sv = fs.view();
  const rt::RenderParams rp = make_params(cam, cfg);
  const int32_t w = rp.image_width, h = rp.image_height;
  uint64_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;
  std::vector<int32_t> stack(256);
  for (int32_t j = 0; j < h; j += row_stride)
    for (int32_t i = 0; i < w; ++i)
      for (int32_t s = 0; s < rp.samples_per_pixel; ++s) {
        rt::PathState ps;
        rt::path_begin(rp, (uint32_t)i, (uint32_t)j, (uint32_t)s, &ps);
        for (;;) {
          if (rt::path_bounce_begin(&ps)) break;
          ++c0;
          const rt::Ray32 q = rt::make_ray32(ps.ray, rt::ray_t_min(ps.ray));
          float t_max32 = __builtin_huge_valf();
          rt::Closest best; best.t = RT_INFINITY; best.hit = false; best.ref = 0; best.order = 0;
          int n_stack = 0;
          int32_t cur = be.a;
          for (;;) {
            if (cur >= 0) {
              ++c1;
              const W& nd = wide[(size_t)cur];
              struct H { float key; int32_t code; } hits[4];
              int nh = 0;
              for (int k = 0; k < nd.n; ++k) {
                ++c3;
                if (!rt::cull32_may_hit(nd.lo[k], nd.hi[k], q, t_max32)) continue;
                // entry distance as the sort key
                float tn = q.t_min;
                const float a0 = std::fmin(std::fmaf(nd.lo[k][0], q.ix, -q.oix), std::fmaf(nd.hi[k][0], q.ix, -q.oix));
                const float a1 = std::fmin(std::fmaf(nd.lo[k][1], q.iy, -q.oiy), std::fmaf(nd.hi[k][1], q.iy, -q.oiy));
                const float a2 = std::fmin(std::fmaf(nd.lo[k][2], q.iz, -q.oiz), std::fmaf(nd.hi[k][2], q.iz, -q.oiz));
                tn = std::fmax(tn, std::fmax(a0, std::fmax(a1, a2)));
                hits[nh++] = H{tn, nd.child[k]};
              }
              std::stable_sort(hits, hits + nh, [](const H& x, const H& y) { return x.key > y.key; });  // farthest first: nearest ends on top
              for (int k = 0; k < nh; ++k) stack[(size_t)n_stack++] = hits[k].code;
              if (n_stack > 0) cur = stack[(size_t)--n_stack]; else break;
            } else {
              const uint32_t f = rt::leaf_first(cur), k = rt::leaf_count(cur);
              for (uint32_t x = 0; x < k; ++x) { ++c2; rt::offer_prim<rt::F_ALL, false>(sv, fs.refs[(size_t)be.b + f + x], f + x, ps.ray, rt::ray_t_min(ps.ray), &best, nullptr); }
              t_max32 = rt::cull_round_up(best.t);
              if (n_stack > 0) cur = stack[(size_t)--n_stack]; else break;
            }
          }
          rt::HitRecord rec;
          if (best.hit) rt::prim_finalize<rt::F_ALL>(sv, best.ref, ps.ray, best.t, &rec);
          if (rt::path_bounce_end<rt::F_ALL, false>(sv, rp, &ps, best.hit, rec, nullptr)) break;
        }
      }
  counts[0] = c0; counts[1] = c1; counts[2] = c2; counts[3] = c3;
  return 0;
}
