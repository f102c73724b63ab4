// Camera::new and the scene catalogue (get_world_cam) -- host side, L5 of the reference.
#pragma once
#include <string>
#include "scene_graph.hpp"

namespace rtx {

// Camera::new(lookfrom, lookat, vup, vfov_deg, aspect_ratio, aperture, focus_dist, time1, time2)
// /root/reference/src/camera.rs:20-57.
rt::FlatCamera camera_new(const double lookfrom[3], const double lookat[3], const double vup[3],
                          double vfov, double aspect_ratio, double aperture, double focus_dist,
                          double time1, double time2);

// Scene ids: 0..12 and the default arm are those of get_world_cam
// (/root/reference/src/world.rs:876-1179).  Ids >= RTX_SCENE_BOOK1_CANONICAL are additions of
// this build, documented in DESIGN.md.
enum : int32_t {
  RTX_SCENE_CHECKERED_SPHERES = 0,
  RTX_SCENE_TWO_PERLIN = 1,
  RTX_SCENE_EARTH = 2,
  RTX_SCENE_SIMPLE_LIGHT = 3,
  RTX_SCENE_CORNELL_BOX = 4,
  RTX_SCENE_CORNELL_SMOKE = 5,
  RTX_SCENE_BOOK2_FINAL = 6,
  RTX_SCENE_MOVING_TEST = 7,
  RTX_SCENE_RANDOM_MOVING = 8,  // gen_random_scene_moving: the GravitySphere video scene (world.rs:169-244)
  RTX_SCENE_BENCHMARK_TEST = 9,
  RTX_SCENE_TRIANGLE_TEST = 10,
  RTX_SCENE_STANFORD_DRAGON = 11,
  RTX_SCENE_TRIANGULAR_PRISM = 12,
  RTX_SCENE_BOOK1_HEAD = 13,        // the `_` arm: gen_random_scene at HEAD (checker ground, moving spheres)
  RTX_SCENE_BOOK1_CANONICAL = 100,  // Book-1 final scene as README.md:12-23 benchmarked it
  RTX_SCENE_EMPTY = 101,            // empty HittableList (edge case)
};

struct SceneOptions {
  double camera_aspect = 0.0;       // <= 0: the reference's hard-coded value for that scene
  const char* earth_ppm = nullptr;  // "earthshit.ppm" stand-in; nullptr/missing -> procedural texture
  const char* dragon_ply = nullptr; // dragon_vrip*.ply; nullptr/missing -> procedural mesh
  int64_t mesh_triangles = 871200;  // size of the procedural mesh
  int32_t book2_boxes_per_side = 20;
  int32_t book2_spheres = 1000;
};

struct WorldCam {
  int32_t world = -1;
  rt::FlatCamera cam;
  double background[3] = {0, 0, 0};
  double camera_aspect = 0.0;
};

bool get_world_cam(SceneGraph& g, int32_t scene_id, const SceneOptions& opt, WorldCam* out,
                   std::string* err);

// Procedural stand-ins for the two assets the reference does not ship.
void procedural_earth(int32_t w, int32_t h, std::vector<double>* texels);
void procedural_mesh(int64_t target_triangles, std::vector<double>* vertices,
                     std::vector<int64_t>* faces);

}  // namespace rtx
