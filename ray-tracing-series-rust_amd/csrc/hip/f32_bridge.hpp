// The seam between the two compilations of the device path.
//
// render.hip is compiled twice: as itself (arithmetic type rt::real = double: the bit-exact path every parity test
// checks) and through render_f32.hip (rt::real = float, namespaces renamed to rt32 / rtx32: the statistical fast mode of
// SURVEY.md 8f-4).  The f64 compilation owns the C ABI and its handle types; it hands the f32 compilation byte images of
// the flat arrays already converted to the f32 layouts (f32_convert.inc) and calls it through the four functions below.
// Nothing here mentions a type of either namespace, so both compilations see the same declarations.
#pragma once
#include <cstddef>
#include <cstdint>
#include "../../../include/rtx_abi.h"

enum RtxF32Array : int {
  RTX32_SPHERES = 0, RTX32_MOVING_SPHERES, RTX32_RECTS, RTX32_TRIANGLES, RTX32_NODES, RTX32_NODES32, RTX32_REFS,
  RTX32_ENTRIES, RTX32_TOP_LEVEL, RTX32_MATERIALS, RTX32_TEXTURES, RTX32_PERLINS, RTX32_IMAGES, RTX32_TEXELS,
  RTX32_TOP_BOX32, RTX32_GRAVITY_SPHERES, RTX32_GRAVITY_Y, RTX32_MOTION32, RTX32_N_ARRAYS
};
struct RtxF32Blobs {
  const void* data[RTX32_N_ARRAYS];
  size_t bytes[RTX32_N_ARRAYS];
  size_t elem_bytes[RTX32_N_ARRAYS];  // what the converter believes one element occupies; checked against sizeof on the other side
  int32_t max_stack, n_bvh;
  uint32_t features;
};

rtx_status rtx_f32_upload(const RtxF32Blobs* blobs, void** device_scene);
rtx_status rtx_f32_render(void* device_scene, const RtxCamera* cam, const RtxConfig* cfg, const RtxShard* shard,
                          double* d_accum_rgb, uint8_t* d_rgb8, void* hip_stream, RtxRenderStats* stats);
rtx_status rtx_f32_trim(void* device_scene);
void rtx_f32_destroy(void* device_scene);
void rtx_f32_set_error(const char* msg);  // defined by the f64 compilation: both report through rtx_last_error
