// Shared device/host helpers for the PhoreGen gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define PG_OK 0
#define PG_ERR_ARG 1
#define PG_ERR_HIP 2

namespace pg {

void set_error(const char* fmt, ...);

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE property of a kernel: remembers (kernel, device) pairs, so a
// second model on cuda:1 in the same process gets its own reservation (defined in graph_ops.hip)
int reserve_lds(const void* kernel, size_t bytes, const char* what);


// timing-only ablation switches exist only in -DPG_ABLATE builds (tools/); the product kernels carry none
#ifdef PG_ABLATE
#define PG_ABL(bit) ((pg_ablate_mask & (bit)) != 0)
#define PG_ABL_PARAM , int pg_ablate_mask
#define PG_ABL_ARG(env) , pg::ablate_from_env(env)
#else
#define PG_ABL(bit) (false)
#define PG_ABL_PARAM
#define PG_ABL_ARG(env)
#endif

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return PG_ERR_HIP;
  }
  return PG_OK;
}

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

// v_mfma_f32_16x16x4_f32: lane l supplies A[row=l&15][k=l>>4], B[k=l>>4][col=l&15];
// D reg r of lane l is D[row=4*(l>>4)+r][col=l&15].
__device__ __forceinline__ f4 mfma16(float a, float b, f4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
// v_mfma_f32_32x32x2_f32: lane l supplies A[row=l&31][k=l>>5], B[k=l>>5][col=l&31];
// D reg r of lane l is D[row=(r&3)+8*(r>>2)+4*(l>>5)][col=l&31].
__device__ __forceinline__ f16v mfma32(float a, float b, f16v c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// 20 fixed Gaussian offsets of models/common.py:18 (coeff = -0.5, common.py:23)
__device__ __constant__ const float kSmearOff[20] = {0.f, 1.f, 1.25f, 1.5f, 1.75f, 2.f, 2.25f, 2.5f, 2.75f, 3.f,
                                                     3.5f, 4.f, 4.5f, 5.f, 5.5f, 6.f, 7.f, 8.f, 9.f, 10.f};

__device__ __forceinline__ float smear(float d, int i) {
  float t = d - kSmearOff[i];
  return expf(-0.5f * t * t);
}

__device__ __forceinline__ float ssp(float v) {  // shifted softplus (models/common.py:58-64): softplus(v) - ln 2, torch threshold 20
  // log(1 + e^v) on the hardware exp2 / log2 (1 ulp each): the sum 1 + e^v carries an ABSOLUTE error of <= 6e-8, which is what the
  // subtraction of ln 2 leaves of any softplus anyway; the libm log1pf(expf()) pair cost ~140 instructions per element
  const float e = __builtin_amdgcn_exp2f(v * 1.44269504088896340736f);
  const float sp = v > 20.f ? v : __builtin_amdgcn_logf(1.0f + e) * 0.69314718055994530942f;
  return sp - 0.69314718055994530942f;
}

// compute units of the CURRENT device (256 on an MI355X in SPX mode; a partitioned device reports its share): the persistent grids and
// the grid caps of the launches are sized from it.  Queried once per device (graph_ops.hip).
int num_cu();
#define kNumCU (::pg::num_cu())

#ifdef PG_ABLATE
inline int ablate_from_env(const char* name) {
  const char* e = getenv(name);
  return e ? atoi(e) : 0;
}
#endif

}  // namespace pg
