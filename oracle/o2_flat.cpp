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
// Parity status vs the reference's OUTPUT: unpinned (see o1_literal.cpp header).
#include <algorithm>
#include <cstring>
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
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; ++t) {
    pool.emplace_back([&, t]() {
      rt::LocalStack<128> stack;
      stack.n = 0;
      for (size_t lr = (size_t)t; lr < rows.size(); lr += (size_t)threads) {
        int32_t j = rows[lr];
        if (j >= row_limit) continue;
        for (int32_t i = 0; i < w; ++i) {
          rt::Color pixel = rt::v3(0, 0, 0);
          for (int32_t s = 0; s < rp.samples_per_pixel; ++s) {
            rt::Color c = counters ? rt::trace_sample<rt::F_ALL, true>(sv, rp, (uint32_t)i, (uint32_t)j, (uint32_t)s, stack, &tc[t])
                                   : rt::trace_sample<rt::F_ALL, false>(sv, rp, (uint32_t)i, (uint32_t)j, (uint32_t)s, stack, nullptr);
            pixel += c;
          }
          size_t o = 3 * (lr * (size_t)w + (size_t)i);
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
      default: out[k] = rt::rt_sqrt(x[k]); break;
    }
  }
}

}  // extern "C"
