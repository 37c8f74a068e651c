// Throughput of v_fma_f32 vs v_pk_fma_f32 vs v_pk_mul/max on gfx950 (one or two waves per SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(512) void k(float* out, int iters, int mode, int waves) {
  if ((int)(threadIdx.x >> 6) >= waves) return;
  float a = threadIdx.x * 0.001f, b = 1.0001f;
  if (mode == 0) {
    float x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3, x4 = a + 4, x5 = a + 5, x6 = a + 6, x7 = a + 7;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        x0 = fmaf(x0, b, a); x1 = fmaf(x1, b, a); x2 = fmaf(x2, b, a); x3 = fmaf(x3, b, a);
        x4 = fmaf(x4, b, a); x5 = fmaf(x5, b, a); x6 = fmaf(x6, b, a); x7 = fmaf(x7, b, a);
      }
    }
    out[threadIdx.x + blockIdx.x * 512] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  } else if (mode == 1) {
    f2 x0 = {a, a + 1}, x1 = {a + 2, a + 3}, x2 = {a + 4, a + 5}, x3 = {a + 6, a + 7}, x4 = {a + 8, a + 9}, x5 = {a, a - 1}, x6 = {a, a - 2}, x7 = {a, a - 3};
    const f2 bb = {b, b}, aa = {a, a};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        x0 = __builtin_elementwise_fma(x0, bb, aa); x1 = __builtin_elementwise_fma(x1, bb, aa);
        x2 = __builtin_elementwise_fma(x2, bb, aa); x3 = __builtin_elementwise_fma(x3, bb, aa);
        x4 = __builtin_elementwise_fma(x4, bb, aa); x5 = __builtin_elementwise_fma(x5, bb, aa);
        x6 = __builtin_elementwise_fma(x6, bb, aa); x7 = __builtin_elementwise_fma(x7, bb, aa);
      }
    }
    f2 s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    out[threadIdx.x + blockIdx.x * 512] = s[0] + s[1];
  } else {
    float x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3, x4 = a + 4, x5 = a + 5, x6 = a + 6, x7 = a + 7;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        x0 = fmaxf(x0 * b, a); x1 = fmaxf(x1 * b, a); x2 = fmaxf(x2 * b, a); x3 = fmaxf(x3 * b, a);
        x4 = fmaxf(x4 * b, a); x5 = fmaxf(x5 * b, a); x6 = fmaxf(x6 * b, a); x7 = fmaxf(x7 * b, a);
      }
    }
    out[threadIdx.x + blockIdx.x * 512] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  }
}
int main() {
  float* d; hipMalloc(&d, 256 * 512 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int waves : {4, 8})
    for (int mode : {0, 1, 2}) {
      hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, iters, mode, waves); hipDeviceSynchronize();
      hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, iters, mode, waves); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const char* what = mode == 0 ? "32 x v_fma_f32" : mode == 1 ? "32 x v_pk_fma_f32 (64 fma)" : "32 x (v_mul + v_max)";
      printf("%d wave(s)/SIMD  %-28s %.3f ms -> %.1f cycles/iter @2.4GHz\n", waves / 4, what, ms, ms * 1e-3 * 2.4e9 / iters);
    }
  return 0;
}
