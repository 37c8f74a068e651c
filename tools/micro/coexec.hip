// Does a VALU-only wave run beside an MFMA-only wave on the SAME SIMD (f32 16x16x4 MFMA)?
// 512-thread workgroup: waves w and w+4 share a SIMD.  mode bit0: waves 0-3 run MFMA loop, bit1: waves 4-7 run VALU loop.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void k(float* out, int iters, int mode) {
  const int wave = threadIdx.x >> 6;
  float a = threadIdx.x * 0.001f, b = 1.0001f;
  if (wave < 4) {
    if (mode & 1) {
      f4 c0 = {0,0,0,0}, c1 = {0,0,0,0}, c2 = {0,0,0,0}, c3 = {0,0,0,0};
      for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
      }
      out[threadIdx.x + blockIdx.x * 512] = c0[0] + c1[1] + c2[2] + c3[3];
    }
  } else if (mode & 2) {
    float x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3, x4 = a + 4, x5 = a + 5, x6 = a + 6, x7 = a + 7;
    for (int i = 0; i < iters; ++i) {   // 8 independent fma chains x 4 = 32 VALU per iteration
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        x0 = fmaf(x0, b, a); x1 = fmaf(x1, b, a); x2 = fmaf(x2, b, a); x3 = fmaf(x3, b, a);
        x4 = fmaf(x4, b, a); x5 = fmaf(x5, b, a); x6 = fmaf(x6, b, a); x7 = fmaf(x7, b, a);
      }
    }
    out[threadIdx.x + blockIdx.x * 512] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  }
  // mode bit2: every wave interleaves 1 MFMA with 4 VALU in ONE stream
  if (mode & 4) {
    f4 c0 = {0,0,0,0}, c1 = {0,0,0,0};
    float x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3;
    for (int i = 0; i < iters; ++i) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
      x0 = fmaf(x0, b, a); x1 = fmaf(x1, b, a); x2 = fmaf(x2, b, a); x3 = fmaf(x3, b, a);
      c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
      x0 = fmaf(x0, b, a); x1 = fmaf(x1, b, a); x2 = fmaf(x2, b, a); x3 = fmaf(x3, b, a);
    }
    out[threadIdx.x + blockIdx.x * 512] = c0[0] + c1[1] + x0 + x1 + x2 + x3;
  }
}
int main() {
  float* d; hipMalloc(&d, 256 * 512 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int mode : {1, 2, 3, 4}) {
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, iters, mode); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, iters, mode); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const char* what = mode == 1 ? "MFMA only (4/iter)" : mode == 2 ? "VALU only (32/iter)" : mode == 3 ? "MFMA wave + VALU wave on one SIMD" : "one stream: 2 MFMA + 8 VALU per iter, 2 waves/SIMD";
    printf("mode %d %-52s %.3f ms  -> %.1f cycles/iter @2.4GHz\n", mode, what, ms, ms * 1e-3 * 2.4e9 / iters);
  }
  return 0;
}
