#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f16v __attribute__((ext_vector_type(16)));
template <int CH>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f16v a[CH];
  for (int c = 0; c < CH; ++c) for (int r = 0; r < 16; ++r) a[c][r] = 0.f;
  const float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-3f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8 / CH; ++u)
#pragma unroll
      for (int c = 0; c < CH; ++c) a[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a[c], 0, 0, 0);
  }
  float s = 0.f;
  for (int c = 0; c < CH; ++c) for (int r = 0; r < 16; ++r) s += a[c][r];
  if (s == 12345.f) out[0] = s;
}
template <int CH> void run(float* out, int wgs) {
  const int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<CH><<<wgs, 256>>>(out, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0); k<CH><<<wgs, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mf = (double)wgs * 4 * iters * 8;
  printf("chains %d, waves/SIMD %d: %.1f cycles per MFMA per SIMD @2.4GHz (%.1f TF/s)\n", CH, wgs / 256, ms * 1e-3 * 2.4e9 / (mf / 1024), mf * 4096 / (ms * 1e-3) / 1e12);
}
int main() {
  float* out; hipMalloc(&out, 4096);
  run<1>(out, 256); run<2>(out, 256); run<4>(out, 256);
  run<1>(out, 512); run<2>(out, 512); run<4>(out, 512);
  run<1>(out, 1024); run<2>(out, 1024);
  return 0;
}
