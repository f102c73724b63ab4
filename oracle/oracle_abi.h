/* ORACLE -- TEST INFRASTRUCTURE.  C entry points of liboracle.so (CPU checkers).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (ray-tracing-series-rust_amd/) never does. */
#ifndef ORACLE_ABI_H
#define ORACLE_ABI_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct OracleCamera { /* same layout as RtxCamera (include/rtx_abi.h) */
  double origin[3], lower_left_corner[3], horizontal[3], vertical[3], u[3], v[3], w[3];
  double lens_radius, time1, time2;
} OracleCamera;

typedef struct OracleConfig {
  int32_t image_width, image_height;
  int32_t samples_per_pixel, max_depth;
  int32_t threads;          /* CPU threads (row bands, world.rs:1198-1227) */
  int32_t row_chunk_compat; /* 1: drop rows >= threads*floor(h/threads) like the reference */
  uint64_t seed;            /* render seed */
  uint64_t bvh_seed;        /* O1 only: stream for the reference BVH's random split axis */
  double background[3];
} OracleConfig;

typedef struct OracleCounters {
  uint64_t box_tests, sphere_tests, moving_sphere_tests, rect_tests, triangle_tests;
  uint64_t scatters, texels, perlin_calls, rays, samples;
} OracleCounters;

/* O1: literal object-graph restatement (oracle/o1_literal.cpp).  graph = rtx_builder_graph(). */
int oracle_o1_render(const void* graph, int32_t world_handle, const OracleCamera* cam,
                     const OracleConfig* cfg, double* accum_rgb, uint8_t* rgb8);
int oracle_o1_render_window(const void* graph, int32_t world_handle, const OracleCamera* cam, const OracleConfig* cfg,
                            int32_t j0, int32_t j1, int32_t i0, int32_t i1, double* accum_rgb);
/* O2: CPU loop over the product's FLATTENED arrays through the product's shared core headers
 * (oracle/o2_flat.cpp).  flat = rtx_flat_arrays().  Renders the shard
 * { j : (j / block_rows) % shard_count == shard_index } compacted, like rtx_render_device.
 * counters may be NULL. */
int oracle_o2_render(const void* flat, const OracleCamera* cam, const OracleConfig* cfg,
                     int32_t shard_index, int32_t shard_count, int32_t block_rows,
                     double* accum_rgb, uint8_t* rgb8, OracleCounters* counters);
/* One sample's radiance through O2 (debugging aid for parity failures). */
int oracle_o2_sample(const void* flat, const OracleCamera* cam, const OracleConfig* cfg, int32_t i,
                     int32_t j, int32_t sample, double rgb[3]);

/* Known-answer probes into O1's restated functions. */
void oracle_o1_vec3_ops(const double a[3], const double b[3], double t, double out[24]);
void oracle_o1_tone_map(const double sum[3], uint32_t spp, int32_t out[3]);
void oracle_o1_sphere_uv(const double p[3], double uv[2]);
double oracle_o1_reflectance(double cosine, double ref_idx);
void oracle_o1_refract(const double uv[3], const double n[3], double ratio, double out[3]);
void oracle_o1_reflect(const double v[3], const double n[3], double out[3]);
int oracle_o1_aabb_hit(const double mn[3], const double mx[3], const double o[3], const double d[3],
                       double t_min, double t_max);
int oracle_o1_hit(const void* graph, int32_t handle, const double o[3], const double d[3], double time,
                  double t_min, double t_max, uint64_t rng_seed, double out[10]);

/* The same probes into the product's shared core (ray-tracing-series-rust_amd/csrc/core). */
void oracle_core_vec3_ops(const double a[3], const double b[3], double t, double out[24]);
void oracle_core_tone_map(const double sum[3], uint32_t spp, int32_t out[3]);
void oracle_core_sphere_uv(const double p[3], double uv[2]);
double oracle_core_reflectance(double cosine, double ref_idx);
void oracle_core_refract(const double uv[3], const double n[3], double ratio, double out[3]);
void oracle_core_reflect(const double v[3], const double n[3], double out[3]);
int oracle_core_aabb_hit(const double mn[3], const double mx[3], const double o[3], const double d[3],
                         double t_min, double t_max);
int oracle_core_world_hit(const void* flat, const double o[3], const double d[3], double time, double t_min,
                          double t_max, uint64_t rng_seed, double out[10]);
int oracle_audit_flat(const void* flat, int32_t* max_depth_out);
int oracle_audit_motion(const void* flat, int32_t n_times, int64_t* checked);
double oracle_motion_leaf_area_ratio(const void* flat);
int oracle_lds_walk_render(const void* flat, const OracleCamera* cam, const OracleConfig* cfg, int32_t use_motion,
                           int32_t row_stride, double* accum_rgb, uint64_t counts[3]);

/* Shared-core probes (product headers compiled for the host): RNG and rt_math. */
void oracle_philox4x32_10(uint32_t ctr[4], uint32_t k0, uint32_t k1);
uint64_t oracle_splitmix64_next(uint64_t* state);
void oracle_sample_stream(uint64_t seed, uint64_t pixel, uint32_t sample, int32_t n, double* out);
void oracle_rt_math(int32_t fn, const double* x, const double* y, int64_t n, double* out);

#ifdef __cplusplus
}
#endif
#endif
