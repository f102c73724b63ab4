// Micro-benchmark: is one v_pk_fma_f32 cheaper than two v_fma_f32 at full occupancy on gfx950?  (tools, not product)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(1024) void k_fma(float* out, int iters, float m, float c) {
  float a[16];
  for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0.001f + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(1024) void k_pk(float* out, int iters, float m, float c) {
  f2 a[8];
  for (int i = 0; i < 8; ++i) a[i] = f2{threadIdx.x * 0.001f + i, threadIdx.x * 0.002f + i};
  f2 mm{m, m}, cc{c, c};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(mm), "v"(cc));
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// broadcast through op_sel: both halves of the result use the LOW half of src1 / src2
__global__ __launch_bounds__(1024) void k_pk_bcast(float* out, int iters, float m, float c) {
  f2 a[8];
  for (int i = 0; i < 8; ++i) a[i] = f2{threadIdx.x * 0.001f + i, threadIdx.x * 0.002f + i};
  f2 mm{m, 0.f}, cc{c, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i)
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,0]" : "+v"(a[i]) : "v"(mm), "v"(cc));
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(1024) void k_f64(double* out, int iters, double m, double c) {
  double a[16];
  for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0.001 + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
  }
  double s = 0;
  for (int i = 0; i < 16; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class K, class T> float run(K k, T* out, int blocks, int iters, T m, T c) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(1024), 0, 0, out, 16, m, c);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(1024), 0, 0, out, iters, m, c);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  const int blocks = 256, iters = 20000;
  void* out; hipMalloc(&out, (size_t)blocks * 1024 * 8);
  for (int rep = 0; rep < 2; ++rep) {
    float t0 = run(k_fma, (float*)out, blocks, iters, 0.999f, 0.001f);
    float t1 = run(k_pk, (float*)out, blocks, iters, 0.999f, 0.001f);
    float t2 = run(k_pk_bcast, (float*)out, blocks, iters, 0.999f, 0.001f);
    float t3 = run(k_f64, (double*)out, blocks, iters, 0.999, 0.001);
    // per wave-instruction clocks at 2.4 GHz, 4 waves per SIMD: ms * 2.4e6 / (iters * 64 instr) / 4 waves
    printf("64 v_fma_f32: %.3f ms   32 v_pk_fma_f32: %.3f ms   32 v_pk_fma_f32 op_sel bcast: %.3f ms   64 v_fma_f64: %.3f ms\n", t0, t1, t2, t3);
    printf("  clocks per wave-instruction (4 waves/SIMD, 2.4 GHz): fma %.2f  pk %.2f  pk_bcast %.2f  f64 %.2f\n",
           t0 * 2.4e6 / (iters * 64.0) / 4, t1 * 2.4e6 / (iters * 32.0) / 4, t2 * 2.4e6 / (iters * 32.0) / 4, t3 * 2.4e6 / (iters * 64.0) / 4);
  }
  return 0;
}
