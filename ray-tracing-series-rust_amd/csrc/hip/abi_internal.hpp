// Glue shared by abi.cpp (host-only entry points) and render.hip (device entry points).
#pragma once
#include <string>
#include "../../../include/rtx_abi.h"
#include "../host/flat_scene.hpp"
#include "../host/scene_graph.hpp"

struct rtx_builder {
  rtx::SceneGraph graph;
  explicit rtx_builder(uint64_t seed) : graph(seed) {}
};
struct rtx_flat {
  rtx::FlatScene scene;
};
struct rtx_scene {
  void* device_scene;  // rtx::DeviceScene of render.hip, or (f32 != 0) the device scene of its f32 compilation render_f32.hip
  int32_t f32;
};

namespace rtx {

struct DeviceScene;

void set_error(const std::string& msg);

inline const FlatScene* flat_of(const rtx_flat* f) { return &f->scene; }
inline DeviceScene* scene_device(const rtx_scene* s) { return (DeviceScene*)s->device_scene; }
inline rtx_scene* make_scene_handle(DeviceScene* ds, int32_t f32) {
  rtx_scene* s = new rtx_scene;
  s->device_scene = ds;
  s->f32 = f32;
  return s;
}
inline void free_scene_handle(rtx_scene* s) { delete s; }

}  // namespace rtx
