// One camera path = one (pixel, sample) of render_scene's inner loop.
//
// Reference: /root/reference/src/world.rs:1211-1215 (jitter, primary ray,
// accumulate) and world.rs:52-93 (ray_color's iterative bounce loop).  The path
// is exposed as begin/step so the device can interleave many paths per lane
// (regeneration) while the CPU checker simply loops step() to completion; both
// execute the same arithmetic in the same order.
#pragma once
#include "shading.hpp"

namespace rt {

struct PathState {
  Ray ray;
  Color product;  // world.rs:59
  Color output;   // world.rs:60
  int32_t depth;  // world.rs:56 (counts down)
  Rng rng;
};

// world.rs:1212-1214: u = (i + rand)/(w-1), v = (j + rand)/(h-1), cam.get_ray(u, v).
// (i, j) = (column, row) with j = 0 the BOTTOM row, as the reference's Screen stores it.
RT_HD void path_begin(const RenderParams& rp, uint32_t i, uint32_t j, uint32_t sample, PathState* ps) {
  uint64_t pixel_index = (uint64_t)j * (uint64_t)rp.image_width + (uint64_t)i;
  ps->rng = rng_for_sample(rp.seed, pixel_index, sample);
  real ru = rng_f64(ps->rng);
  real u = ((real)i + ru) / (real)(rp.image_width - 1);
  real rv = rng_f64(ps->rng);
  real v = ((real)j + rv) / (real)(rp.image_height - 1);
  ps->ray = camera_get_ray(rp.cam, u, v, ps->rng);
  ps->product = v3(1, 1, 1);
  ps->output = v3(0, 0, 0);
  ps->depth = rp.max_depth;
}

// One iteration of ray_color's loop, in two halves around the world.hit() query so the device
// can run the query as a resumable walk:
//   path_bounce_begin   world.rs:64-67  depth -= 1; exhausted paths end with what they gathered
//   path_bounce_end     world.rs:68-90  miss -> background; hit -> scatter (draws), emitted, update
// Both return true when the path has ended and ps->output holds the sample's radiance.
RT_HD bool path_bounce_begin(PathState* ps) {
  ps->depth -= 1;
#if defined(RT_F32)
  // A ray that is not finite (a plane hit at t = inf behind a direction component of exactly 0 -- single-precision rays
  // have those -- leaves a NaN hit point) can hit nothing, yet no box test can rule it out: it would visit every node and
  // every primitive of every BVH, alone in its wave.  The fast mode ends such a path with what it has gathered.
  const real mag = rt_fabs(ps->ray.origin.x) + rt_fabs(ps->ray.origin.y) + rt_fabs(ps->ray.origin.z) +
                   rt_fabs(ps->ray.direction.x) + rt_fabs(ps->ray.direction.y) + rt_fabs(ps->ray.direction.z);
  if (!(mag < RT_INFINITY)) return true;
#endif
  return ps->depth < 0;
}

template <uint32_t F, bool COUNT>
RT_HD bool path_bounce_end(const SceneView& sv, const RenderParams& rp, PathState* ps, bool hit,
                           const HitRecord& rec, TraceCounters* cnt) {
  if (!hit) {
    ps->output += ps->product * rp.background;  // world.rs:86-89
    return true;
  }
  const FlatMaterial& m = sv.materials[rec.mat];
  Ray scattered;
  Color attenuation;
  // world.rs:69-84: scatter first (it draws from the stream), then emitted.
  bool did_scatter = material_scatter<F, COUNT>(sv, m, ps->ray, rec, ps->rng, &scattered, &attenuation, cnt);
  Color emitted = material_emitted<F, COUNT>(sv, m, rec, cnt);
  ps->output += emitted * ps->product;
  if (!did_scatter) return true;
  ps->product *= attenuation;
  ps->ray = scattered;
  return false;
}

template <uint32_t F, bool COUNT, class STACK, class WALK = SerialWalk>
RT_HD bool path_step(const SceneView& sv, const RenderParams& rp, PathState* ps, STACK& stack,
                     TraceCounters* cnt) {
  if (path_bounce_begin(ps)) return true;
  HitRecord rec;
  bool hit = world_hit<F, COUNT, STACK, WALK>(sv, ps->ray, ray_t_min(ps->ray), RT_INFINITY, &rec, ps->rng, stack, cnt);
  return path_bounce_end<F, COUNT>(sv, rp, ps, hit, rec, cnt);
}

// Whole sample on one thread (CPU checker; also the device's simplest kernel).
template <uint32_t F, bool COUNT, class STACK>
RT_HD Color trace_sample(const SceneView& sv, const RenderParams& rp, uint32_t i, uint32_t j,
                         uint32_t sample, STACK& stack, TraceCounters* cnt) {
  PathState ps;
  path_begin(rp, i, j, sample, &ps);
  if (COUNT) cnt->samples++;
  while (!path_step<F, COUNT>(sv, rp, &ps, stack, cnt)) {
  }
  return ps.output;
}

}  // namespace rt
