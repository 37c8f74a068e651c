// Measurement aid, not part of the sampler: the fp32 matrix rate this GPU actually sustains, measured inside the run that quotes it.
// bench.py divides the triplet kernel's executed FLOPs by the NOMINAL fp32 MFMA peak (157.3 TF/s, MI355X_MICROARCH.md) and, beside
// it, by what this kernel measures on the same box in the same process ("verify with micro-benchmarks and use the
// measured peaks as denominators").  Every wave issues `iters` x 16 v_mfma_f32_16x16x4_f32 -- the instruction of the attention kernels
// -- in 8 independent accumulator chains, 4 waves per SIMD at the default grid: nothing but MFMA issue.
#include "common.h"
#include "../../include/phoregen_hip.h"

namespace pg {

__global__ __launch_bounds__(256) void micro_mfma_f32_kernel(float* sink, int iters) {
  f4 a[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) a[c] = (f4){0.f, 0.f, 0.f, 0.f};
  const float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-3f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int c = 0; c < 8; ++c) a[c] = mfma16(x, y, a[c]);
  }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) s += a[c][r];
  if (s == 12345.f) sink[0] = s;          // (never true: keeps the chains alive)
}

}  // namespace pg

extern "C" int pg_micro_mfma_f32(int workgroups, int iters, float* sink, double* flops, void* stream) {
  if (workgroups <= 0 || iters <= 0 || !sink) { pg::set_error("pg_micro_mfma_f32: bad arguments"); return PG_ERR_ARG; }
  hipLaunchKernelGGL(pg::micro_mfma_f32_kernel, dim3(workgroups), dim3(256), 0, (hipStream_t)stream, sink, iters);
  if (flops) *flops = (double)workgroups * 4.0 * (double)iters * 16.0 * 2048.0;      // 4 waves x iters x 16 MFMA x 2 x 16 x 16 x 4 FLOP
  return pg::check_launch("pg_micro_mfma_f32");
}
