// Stand-in for the reference binary (/root/reference/src/main.rs:1-16): pick a scene from the
// catalogue, render it, print P3 PPM on stdout and "Time taken" on stderr.  The reference hard-codes
// THREADS = 11, SCENE_ID = 11 and Config::new(1.6, 600, 1000, 50, THREADS); here the same values are the
// defaults and every one of them can be overridden on the command line.
//
//   rtx_render [--scene ID] [--aspect A] [--width W] [--spp S] [--depth D] [--threads T] [--seed N]
//              [--scene-seed N] [--out FILE.ppm] [--camera-aspect A] [--ply FILE] [--earth FILE.ppm]
//              [--row-chunk-compat]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "../csrc/host/world.hpp"

static const size_t THREADS = 11;  // main.rs:4
static const int SCENE_ID = 11;    // main.rs:5

int main(int argc, char** argv) {
  auto start = std::chrono::steady_clock::now();  // main.rs:8: the timer covers scene build + render + PPM
  int scene_id = SCENE_ID, width = 600, spp = 1000, depth = 50;
  size_t threads = THREADS;
  double aspect = 1.6, camera_aspect = 0.0;
  uint64_t seed = 1, scene_seed = 1;
  const char* out = nullptr;
  const char* ply = nullptr;
  const char* earth = nullptr;
  bool compat = false;
  for (int i = 1; i < argc; ++i) {
    auto need = [&](const char* flag) -> const char* {
      if (i + 1 >= argc) { fprintf(stderr, "%s needs a value\n", flag); exit(2); }
      return argv[++i];
    };
    if (!strcmp(argv[i], "--scene")) scene_id = atoi(need("--scene"));
    else if (!strcmp(argv[i], "--aspect")) aspect = atof(need("--aspect"));
    else if (!strcmp(argv[i], "--camera-aspect")) camera_aspect = atof(need("--camera-aspect"));
    else if (!strcmp(argv[i], "--width")) width = atoi(need("--width"));
    else if (!strcmp(argv[i], "--spp")) spp = atoi(need("--spp"));
    else if (!strcmp(argv[i], "--depth")) depth = atoi(need("--depth"));
    else if (!strcmp(argv[i], "--threads")) threads = (size_t)atoi(need("--threads"));
    else if (!strcmp(argv[i], "--seed")) seed = strtoull(need("--seed"), nullptr, 10);
    else if (!strcmp(argv[i], "--scene-seed")) scene_seed = strtoull(need("--scene-seed"), nullptr, 10);
    else if (!strcmp(argv[i], "--out")) out = need("--out");
    else if (!strcmp(argv[i], "--ply")) ply = need("--ply");
    else if (!strcmp(argv[i], "--earth")) earth = need("--earth");
    else if (!strcmp(argv[i], "--row-chunk-compat")) compat = true;
    else { fprintf(stderr, "unknown argument %s\n", argv[i]); return 2; }
  }
  try {
    rtsr::Scene scene(scene_seed);
    RtxSceneOptions opt;
    memset(&opt, 0, sizeof(opt));
    opt.camera_aspect = camera_aspect;
    opt.dragon_ply = ply ? ply : "./models/dragon_recon/dragon_vrip_res2.ply";  // world.rs:684
    opt.earth_ppm = earth ? earth : "earthshit.ppm";                              // world.rs:290,580
    rtsr::WorldCam wc = rtsr::get_world_cam(scene, scene_id, &opt);               // main.rs:10
    rtsr::Config config = rtsr::Config::new_(aspect, width, spp, depth, threads);  // main.rs:11
    config.c.seed = seed;
    config.c.row_chunk_compat = compat ? 1 : 0;
    rtsr::Screen screen = rtsr::render_scene(scene, wc.world, wc.cam, wc.background, config);  // main.rs:13
    if (out) screen.write_to_ppm_file(out);
    else screen.write_to_ppm();
  } catch (const rtsr::Error& e) {
    fprintf(stderr, "rtx_render: %s (status %d)\n", e.what(), (int)e.status);
    return 1;
  }
  double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count();
  fprintf(stderr, "Time taken: %.3fs\n", secs);  // main.rs:15
  return 0;
}
