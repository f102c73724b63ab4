// The device path a second time, in single precision (the statistical fast mode).
//
// This file adds no code of its own: it compiles render.hip -- kernels, launcher, upload -- with the arithmetic type
// rt::real set to float and every namespace renamed, so that the two compilations share one set of sources and cannot
// drift apart.  What changes with the type: vectors, rays, hit records and path state shrink to half the registers,
// adds / multiplies / FMAs issue in 2 clocks instead of 4 (issue-cost table) and sqrt / sin / cos / log in
// 8 instead of a double-precision expansion.  What does not: the RNG stream (the same counter-based 64-bit generator;
// a unit float is its top 24 bits), the order samples are added in (k_reduce_samples still sums doubles in sample
// order), and the scene: the same flatten, converted field by field (f32_convert.inc).
//
// Images of this path are NOT bit-comparable with the reference's: tests/test_gpu_f32.py holds it to a statistical
// bar against the f64 path instead.  Everything the C ABI promises about bit-exactness is about rtx_scene_upload.
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
#include <cstddef>
#include <cstdint>
#include "../../../include/rtx_abi.h"
#include "f32_bridge.hpp"

#define RTX_F32_TU 1
#define RT_F32 1
#define RT_REAL float
#define rt rt32
#define rtx rtx32
#include "render.hip"
