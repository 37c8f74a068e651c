// Do a full-rate fp32 MFMA stream and an HBM copy stream run at their own speeds when they share the chip?
// Three timings: MFMA kernel alone, copy kernel alone, both on two streams.  Sized like one bond-row GEMM
// (6.7 GFLOP of 32x32x2 fp32 MFMA; 104 MB read + 104 MB written).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_hbm.hip -o tools/micro/mfma_hbm && tools/micro/mfma_hbm
#include <hip/hip_runtime.h>
#include <stdio.h>

constexpr int REPS = 8;     // both kernels do REPS x the work of one bond-row GEMM, so launch overhead is small

typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void stamp(unsigned long long* ts, int which) {   // [0] earliest start, [1] latest end (100 MHz)
  if (threadIdx.x == 0) {
    const unsigned long long now = wall_clock64();
    if (which == 0) atomicMin(ts, now); else atomicMax(ts + 1, now);
  }
}

__global__ __launch_bounds__(256) void mfma_only(float* out, int iters, unsigned long long* ts) {
  stamp(ts, 0);
  f16v a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
  const float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-3f;
  for (int i = 0; i < iters; ++i) {
    a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
  }
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
  if (s == 12345.f) out[0] = s;
  stamp(ts, 1);
}

__global__ __launch_bounds__(256) void copy_only(const f4* __restrict__ src, f4* __restrict__ dst, size_t n4, int reps,
                                                 unsigned long long* ts, int prio) {
  if (prio) __builtin_amdgcn_s_setprio(3);      // the copy waves' address arithmetic wins the VALU / MFMA issue port when ready
  stamp(ts, 0);
  for (int r = 0; r < reps; ++r)
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256)
      __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
  stamp(ts, 1);
}

// the same copy with NO vector-ALU instruction in its loop: buffer addressing = SGPR descriptor + fixed per-lane offset + SGPR
// tile offset, loop control on the scalar unit
typedef int i4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void copy_sgpr(const float* src, float* dst, unsigned bytes, int reps, unsigned long long* ts) {
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)dst, 0, bytes, 0x00020000);
  const unsigned voff = threadIdx.x * 16;
  const unsigned step = gridDim.x * 4096;
  stamp(ts, 0);
  for (int r = 0; r < reps; ++r)
    for (unsigned so = blockIdx.x * 4096; so < bytes; so += 4 * step) {     // out-of-range offsets: loads return 0, stores are dropped
      const i4 v0 = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, so, 0);
      const i4 v1 = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, so + step, 0);
      const i4 v2 = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, so + 2 * step, 0);
      const i4 v3 = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, so + 3 * step, 0);
      __builtin_amdgcn_raw_buffer_store_b128(v0, rd, voff, so, 0);
      __builtin_amdgcn_raw_buffer_store_b128(v1, rd, voff, so + step, 0);
      __builtin_amdgcn_raw_buffer_store_b128(v2, rd, voff, so + 2 * step, 0);
      __builtin_amdgcn_raw_buffer_store_b128(v3, rd, voff, so + 3 * step, 0);
    }
  stamp(ts, 1);
}

static unsigned long long* g_ts;     // device: {mfma start, mfma end, copy start, copy end}
static unsigned long long g_host[4];
static int g_prio = 0;

static float timed(hipStream_t s0, hipStream_t s1, int mode, float* out, const f4* src, f4* dst, size_t n4, int iters,
                   int mfma_wgs, int copy_wgs) {
  hipEvent_t e0, e1, f1;
  hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&f1);
  float best = 1e9f;
  for (int rep = 0; rep < 8; ++rep) {
    const unsigned long long init[4] = {~0ull, 0ull, ~0ull, 0ull};
    hipMemcpy(g_ts, init, sizeof(init), hipMemcpyHostToDevice);
    hipDeviceSynchronize();
    hipEventRecord(e0, s0);
    hipStreamWaitEvent(s1, e0, 0);
    if (g_prio == 2) {       // scalar-addressed copy, launched FIRST so that its waves are resident before the MFMA kernel arrives
      if (mode & 2) copy_sgpr<<<copy_wgs, 256, 0, s1>>>((const float*)src, (float*)dst, (unsigned)(n4 * 16), REPS, g_ts + 2);
      if (mode & 1) mfma_only<<<mfma_wgs, 256, 0, s0>>>(out, iters, g_ts);
    } else {
      if (mode & 1) mfma_only<<<mfma_wgs, 256, 0, s0>>>(out, iters, g_ts);
      if (mode & 2) copy_only<<<copy_wgs, 256, 0, s1>>>(src, dst, n4, REPS, g_ts + 2, g_prio);
    }
    hipEventRecord(f1, s1);
    hipStreamWaitEvent(s0, f1, 0);
    hipEventRecord(e1, s0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep >= 2 && ms < best) { best = ms; hipMemcpy(g_host, g_ts, sizeof(g_host), hipMemcpyDeviceToHost); }
  }
  return best;
}

int main() {
  const size_t bytes = 203720ull * 128 * 4;       // one [E_bond,128] fp32 matrix
  const size_t n4 = bytes / 16;
  f4 *src, *dst; float* out;
  hipMalloc(&src, bytes); hipMalloc(&dst, bytes); hipMalloc(&out, 4096); hipMalloc(&g_ts, 64);
  hipMemset(src, 0, bytes);
  hipStream_t s0, s1; hipStreamCreate(&s0); hipStreamCreate(&s1);
  // 6.68 GFLOP = 203720*128*128*2; one 32x32x2 MFMA = 4096 FLOP; grid of G workgroups x 4 waves x 4 MFMA per iteration
  for (g_prio = 0; g_prio < 3; g_prio += 2)
  for (int wgs_per_cu = 1; wgs_per_cu <= 2; ++wgs_per_cu) {
    const int mfma_wgs = 256 * wgs_per_cu;
    const int iters = (int)(203720.0 * 128 * 128 * 2 / 4096 / (mfma_wgs * 4.0 * 4.0)) * REPS;
    for (int copy_wgs = 1024; copy_wgs <= 2048; copy_wgs *= 2) {
      const float a = timed(s0, s1, 1, out, src, dst, n4, iters, mfma_wgs, copy_wgs);
      const float b = timed(s0, s1, 2, out, src, dst, n4, iters, mfma_wgs, copy_wgs);
      const float c = timed(s0, s1, 3, out, src, dst, n4, iters, mfma_wgs, copy_wgs);
      printf("copy variant %d, mfma wgs/CU %d (iters %d), copy wgs %4d:  mfma alone %.1f us (%.1f TF/s)  copy alone %.1f us (%.2f TB/s)  together %.1f us\n",
             g_prio, wgs_per_cu, iters, copy_wgs, a * 1e3, REPS * 203720.0 * 128 * 128 * 2 / (a * 1e-3) / 1e12, b * 1e3,
             REPS * 2.0 * bytes / (b * 1e-3) / 1e12, c * 1e3);
      const double t0 = (double)(g_host[0] < g_host[2] ? g_host[0] : g_host[2]);
      printf("      together, device clock (us from the first start):  mfma %.1f .. %.1f   copy %.1f .. %.1f\n",
             (g_host[0] - t0) / 100.0, (g_host[1] - t0) / 100.0, (g_host[2] - t0) / 100.0, (g_host[3] - t0) / 100.0);
    }
  }
  return 0;
}
