// Issue cost of v_mfma_f32_4x4x1_16B_f32 (16 independent 4x4 blocks, K = 1: 512 FLOP per instruction) next to the
// v_mfma_f32_16x16x4_f32 the triplet kernel uses (2 048 FLOP): cycles per instruction per SIMD with 1 / 2 / 4 waves per SIMD and
// 4 or 8 independent accumulator chains per wave.  The round-3 review's question: could the multi-block form serve the padded
// tail tile of a triplet segment?  It pays only if it issues in ~8 cycles.     hipcc --offload-arch=gfx950 -O3 mfma_4x4x1.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int CH, bool SMALL>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f4 a[CH];
  for (int c = 0; c < CH; ++c) a[c] = (f4){0.f, 0.f, 0.f, 0.f};
  const float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-3f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16 / CH; ++u)
#pragma unroll
      for (int c = 0; c < CH; ++c)
        a[c] = SMALL ? __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a[c], 0, 0, 0) : __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a[c], 0, 0, 0);
  }
  float s = 0.f;
  for (int c = 0; c < CH; ++c) for (int r = 0; r < 4; ++r) s += a[c][r];
  if (s == 12345.f) out[0] = s;
}
template <int CH, bool SMALL> void run(float* out, int wgs) {
  const int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<CH, SMALL><<<wgs, 256>>>(out, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0); k<CH, SMALL><<<wgs, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double per_simd = (double)(wgs / 256) * iters * 16;            // instructions issued per SIMD (one wave per SIMD per 256 workgroups)
  const double flop = (double)wgs * 4 * iters * 16 * (SMALL ? 512 : 2048);
  printf("%-24s chains %d, waves/SIMD %d: %6.1f cycles per instruction per SIMD @2.4GHz  (%.1f TF/s)\n",
         SMALL ? "v_mfma_f32_4x4x1_16B_f32" : "v_mfma_f32_16x16x4_f32", CH, wgs / 256, ms * 1e-3 * 2.4e9 / per_simd, flop / (ms * 1e-3) / 1e12);
}
int main() {
  float* out; hipMalloc(&out, 4096);
  run<4, false>(out, 256); run<8, false>(out, 256); run<8, false>(out, 512); run<8, false>(out, 1024);
  run<4, true>(out, 256); run<8, true>(out, 256); run<16, true>(out, 256); run<8, true>(out, 512); run<8, true>(out, 1024);
  return 0;
}
