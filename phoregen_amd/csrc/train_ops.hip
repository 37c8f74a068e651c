// Training-path kernels that are not attention: weight gradients of the dense layers, the stand-alone
// LayerNorm+ReLU of the query MLPs with its adjoint, and the weight gradient of the folded second key/value layers.
// (PhoreDiff.compute_loss, models/diffusion.py:249-352; MLP = models/common.py:99-119.)
#include "common.h"
#include "../../include/phoregen_hip.h"

namespace pg {

// ------------------------------------------------------------------------------------------------
// gW[n,k] += sum_r dY[r,n] * X[r,k]       (rows are the contraction index; M is huge, N and K are small)
// Workgroup = 4 waves in 2x2, output tile 64(n) x 64(k), one 32x32x2 accumulator per wave; blockIdx.z strides over
// row chunks of 64 (float4 loads) and the partial tile is added with atomics.  Both operands are read from LDS with the lane
// running along the row-major fast dimension: A[i=n][kk=row] = sdY[row][n], B[kk=row][j=k] = sX[row][k].
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gemm_wgrad_kernel(const float* dY, int ldy, const float* X, int ldx, int M, int N,
                                                         int K, float* gW, int ldgw, float* gb) {
  constexpr int R = 64;                                  // rows per staged chunk
  __shared__ __attribute__((aligned(16))) float sdY[R][64 + 4];
  __shared__ __attribute__((aligned(16))) float sX[R][64 + 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = blockIdx.x * 64, k0 = blockIdx.y * 64;
  const int wn = (wave & 1) * 32, wk = (wave >> 1) * 32;
  f16v acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float colsum = 0.f;
  const int lr = tid >> 2, lc = (tid & 3) * 16;          // loader: row lr, 16 consecutive columns from lc (4 x float4)
  const bool vecY = (ldy & 3) == 0 && ((size_t)dY & 15) == 0;      // 16-byte loads wherever a whole float4 lies inside the matrix
  const bool vecX = (ldx & 3) == 0 && ((size_t)X & 15) == 0;
  // software pipeline: the next chunk's rows are requested before the products of the current one are issued (the loads of a chunk
  // are a full HBM round trip; with one chunk in flight the matrix pipe idled for most of it: 45-57 TF/s -> see tools/bench_wgrad.py)
  f4 vy[4], vx[4];
  auto fetch = [&](int row0) {
    const int row = row0 + lr;
    const bool in = row < M;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = lc + 4 * q;
      vy[q] = (f4){0.f, 0.f, 0.f, 0.f};
      vx[q] = (f4){0.f, 0.f, 0.f, 0.f};
      if (in) {
        if (vecY && n0 + c + 4 <= N) vy[q] = *reinterpret_cast<const f4*>(dY + (size_t)row * ldy + n0 + c);
        else {
#pragma unroll
          for (int j = 0; j < 4; ++j) if (n0 + c + j < N) vy[q][j] = dY[(size_t)row * ldy + n0 + c + j];
        }
        if (vecX && k0 + c + 4 <= K) vx[q] = *reinterpret_cast<const f4*>(X + (size_t)row * ldx + k0 + c);
        else {
#pragma unroll
          for (int j = 0; j < 4; ++j) if (k0 + c + j < K) vx[q][j] = X[(size_t)row * ldx + k0 + c + j];
        }
      }
    }
  };
  const int step = gridDim.z * R;
  if ((int)(blockIdx.z * R) < M) fetch(blockIdx.z * R);
  for (int row0 = blockIdx.z * R; row0 < M; row0 += step) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      *reinterpret_cast<f4*>(&sdY[lr][lc + 4 * q]) = vy[q];
      *reinterpret_cast<f4*>(&sX[lr][lc + 4 * q]) = vx[q];
    }
    __syncthreads();
    if (row0 + step < M) fetch(row0 + step);
    if (gb && blockIdx.y == 0 && tid < 64) {
#pragma unroll 8
      for (int r = 0; r < R; ++r) colsum += sdY[r][tid];
    }
#pragma unroll 8
    for (int kk = 0; kk < R; kk += 2) {
      const int rr = kk + (lane >> 5);
      acc = mfma32(sdY[rr][wn + (lane & 31)], sX[rr][wk + (lane & 31)], acc);
    }
    __syncthreads();
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int n = n0 + wn + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), k = k0 + wk + (lane & 31);
    if (n < N && k < K && acc[r] != 0.f) atomicAdd(gW + (size_t)n * ldgw + k, acc[r]);
  }
  if (gb && blockIdx.y == 0 && tid < 64 && n0 + tid < N) atomicAdd(gb + n0 + tid, colsum);
}

// ------------------------------------------------------------------------------------------------
// Y = ReLU(LN(X) * gamma + beta) over 128 channels, one wave per row (2 channels per lane), and its adjoint
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ln_relu_kernel(const float* X, int ldx, const float* gamma, const float* beta, int M,
                                                      float* Y, int ldy) {
  const int lane = threadIdx.x & 63;
  const float g0 = gamma[lane], g1 = gamma[64 + lane], b0 = beta[lane], b1 = beta[64 + lane];
  for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < M; row += gridDim.x * 4) {
    const float x0 = X[(size_t)row * ldx + lane], x1 = X[(size_t)row * ldx + 64 + lane];
    const float mu = wave_sum(x0 + x1) * (1.f / 128.f);
    const float d0 = x0 - mu, d1 = x1 - mu;
    const float var = wave_sum(d0 * d0 + d1 * d1) * (1.f / 128.f);
    const float rs = 1.0f / sqrtf(var + 1e-5f);
    Y[(size_t)row * ldy + lane] = fmaxf(d0 * rs * g0 + b0, 0.f);
    Y[(size_t)row * ldy + 64 + lane] = fmaxf(d1 * rs * g1 + b1, 0.f);
  }
}

__global__ __launch_bounds__(256) void ln_relu_bwd_kernel(const float* X, int ldx, const float* gamma, const float* beta,
                                                          const float* gY, int ldgy, int M, float* gX, int ldgx,
                                                          float* ggamma, float* gbeta) {
  const int lane = threadIdx.x & 63;
  const float g0 = gamma[lane], g1 = gamma[64 + lane], b0 = beta[lane], b1 = beta[64 + lane];
  float gg0 = 0.f, gg1 = 0.f, gb0 = 0.f, gb1 = 0.f;
  for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < M; row += gridDim.x * 4) {
    const float x0 = X[(size_t)row * ldx + lane], x1 = X[(size_t)row * ldx + 64 + lane];
    const float mu = wave_sum(x0 + x1) * (1.f / 128.f);
    const float d0 = x0 - mu, d1 = x1 - mu;
    const float var = wave_sum(d0 * d0 + d1 * d1) * (1.f / 128.f);
    const float rs = 1.0f / sqrtf(var + 1e-5f);
    const float h0 = d0 * rs, h1 = d1 * rs;
    float y0 = gY[(size_t)row * ldgy + lane], y1 = gY[(size_t)row * ldgy + 64 + lane];
    y0 = (h0 * g0 + b0) > 0.f ? y0 : 0.f;
    y1 = (h1 * g1 + b1) > 0.f ? y1 : 0.f;
    gg0 += y0 * h0; gg1 += y1 * h1; gb0 += y0; gb1 += y1;
    const float a0 = y0 * g0, a1 = y1 * g1;                  // d x_hat
    const float m1 = wave_sum(a0 + a1) * (1.f / 128.f);
    const float m2 = wave_sum(a0 * h0 + a1 * h1) * (1.f / 128.f);
    gX[(size_t)row * ldgx + lane] = rs * (a0 - m1 - h0 * m2);
    gX[(size_t)row * ldgx + 64 + lane] = rs * (a1 - m1 - h1 * m2);
  }
  __shared__ float red[4][256];
  const int w = threadIdx.x >> 6;
  red[w][lane] = gg0; red[w][64 + lane] = gg1; red[w][128 + lane] = gb0; red[w][192 + lane] = gb1;
  __syncthreads();
  const float v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
  if (threadIdx.x < 128) atomicAdd(ggamma + threadIdx.x, v);
  else atomicAdd(gbeta + threadIdx.x - 128, v);
}

// ------------------------------------------------------------------------------------------------
// gW2_l[(2i)*64 + lane][j]     += sum_s X[s, 8m + j]     * T[s][i*64 + lane]
// gW2_l[(2i + 1)*64 + lane][j] += sum_s X[s, 8m + 4 + j] * T[s][i*64 + lane]       (lane = (g, m), i = 0..31)
// blockIdx.y selects 4 consecutive i; one wave per segment stride.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fold_wgrad_kernel(const float* X, int ldx, const float* T, int n, const int* ids,
                                                         float* gW2_l) {
  const int lane = threadIdx.x & 63, m = lane & 15;
  const int i0 = blockIdx.y * 4;
  float acc[4][8];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int d = 0; d < 8; ++d) acc[a][d] = 0.f;
  // 4 segments per iteration: 4 x (2 float4 of X + 4 dwords of T) loads in flight per lane (the kernel is pure streaming)
  const int stride = gridDim.x * 4;
  for (int si0 = blockIdx.x * 4 + (threadIdx.x >> 6); si0 < n; si0 += 4 * stride) {
    f4 xa[4], xb[4];
    float tv[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int si = si0 + u * stride;
      const bool ok = si < n;
      const int s = ok ? (ids ? ids[si] : si) : 0;
      const float* xp = X + (size_t)s * ldx + 8 * m;
      xa[u] = ok ? *reinterpret_cast<const f4*>(xp) : (f4){0.f, 0.f, 0.f, 0.f};
      xb[u] = ok ? *reinterpret_cast<const f4*>(xp + 4) : (f4){0.f, 0.f, 0.f, 0.f};
      const float* tp = T + (size_t)s * 2048 + lane;
#pragma unroll
      for (int a = 0; a < 4; ++a) tv[u][a] = ok ? tp[(i0 + a) * 64] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[a][j] = fmaf(xa[u][j], tv[u][a], acc[a][j]);
          acc[a][4 + j] = fmaf(xb[u][j], tv[u][a], acc[a][4 + j]);
        }
  }
  // the 4 waves of the workgroup hold partial sums of the same 64 x 32 values: reduce them in LDS, one global atomic per
  // value per workgroup (the flush, not the streaming, dominated this kernel when every wave flushed on its own)
  __shared__ float red[4][32][64];
  const int w = threadIdx.x >> 6;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int d = 0; d < 8; ++d) red[w][a * 8 + d][lane] = acc[a][d];
  __syncthreads();
  for (int i = threadIdx.x; i < 32 * 64; i += 256) {
    const int e = i >> 6, l = i & 63, a = e >> 3, d = e & 7;
    const float v = red[0][e][l] + red[1][e][l] + red[2][e][l] + red[3][e][l];
    atomicAdd(gW2_l + ((size_t)(2 * (i0 + a) + (d >> 2)) * 64 + l) * 4 + (d & 3), v);
  }
}

// ------------------------------------------------------------------------------------------------
// out[ctx(a)][c] = sum of Y[e][c] over the bond rows e that leave (by_src) or reach (by dst) ligand atom a: the adjoint of a
// gathered operand of pg_gemm (Y[e] += A[bond_src[e]] / A[bond_dst[e]]).  One wave per atom walks its n - 1 rows through the
// graph's edge-id table (8 rows in flight, 16 bytes per lane): every row is read once, nothing is atomic -- torch's index_add_
// on the same 167 MB took 141 us per call, 30 calls per training step.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bond_rows_sum_kernel(PgTopo t, const float* Y, int ldy, int ncol, int by_src, float* out,
                                                            int ldo) {
  const int lane = threadIdx.x & 63;
  for (int a = blockIdx.x * 4 + (threadIdx.x >> 6); a < t.n_lig; a += gridDim.x * 4) {
    const int ctx = t.lig2ctx[a];
    const int gi = t.ctx_graph[ctx];
    const int n = t.g_nlig[gi], la = ctx - (t.g_ctx_off[gi] + t.g_nph[gi]);
    const int* eid_g = t.eid + t.g_eid_off[gi];
    for (int c = 4 * lane; c < ncol; c += 256) {
      f4 acc = {0.f, 0.f, 0.f, 0.f};
      for (int b0 = 0; b0 < n; b0 += 8) {
        f4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int b = b0 + u;
          const bool ok = b < n && b != la;
          const int e = ok ? (by_src ? eid_g[la * n + b] : eid_g[b * n + la]) : 0;
          v[u] = ok ? *reinterpret_cast<const f4*>(Y + (size_t)e * ldy + c) : (f4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
      }
      *reinterpret_cast<f4*>(out + (size_t)ctx * ldo + c) = acc;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// the bias side of the value unfold's adjoint (out[s, 8h+d] = ... + b2v[8h+d] * swn[s, h]):
//   gswn[s, h] (=) sum_d gout[s, 8h+d] * b2v[8h+d]          gb2v[c] (+=) sum_s gout[s, c] * swn[s, c >> 3]
// over the rows `ids` (all n rows if NULL).  One wave per row, lane = channels lane and lane + 64; d b2v is summed per
// workgroup before it is added.  (Five elementwise / reduction passes over [n, 128] tensors in the tensor-op form.)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void unfold_bias_grad_kernel(const float* gout, int ldg, const float* swn, const float* b2v, int n,
                                                               const int* ids, float* gswn, float* gb2v) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const float b0 = b2v[lane], b1 = b2v[64 + lane];
  float a0 = 0.f, a1 = 0.f;
  for (int si = blockIdx.x * 4 + w; si < n; si += gridDim.x * 4) {
    const int s = ids ? ids[si] : si;
    const float v0 = gout[(size_t)s * ldg + lane], v1 = gout[(size_t)s * ldg + 64 + lane];
    const float sw0 = swn[(size_t)s * 16 + (lane >> 3)], sw1 = swn[(size_t)s * 16 + 8 + (lane >> 3)];
    a0 = fmaf(v0, sw0, a0);
    a1 = fmaf(v1, sw1, a1);
    float p0 = v0 * b0, p1 = v1 * b1;
#pragma unroll
    for (int o = 1; o <= 4; o <<= 1) { p0 += __shfl_xor(p0, o); p1 += __shfl_xor(p1, o); }
    if ((lane & 7) == 0) {
      gswn[(size_t)s * 16 + (lane >> 3)] = p0;
      gswn[(size_t)s * 16 + 8 + (lane >> 3)] = p1;
    }
  }
  __shared__ float red[4][128];
  red[w][lane] = a0;
  red[w][64 + lane] = a1;
  __syncthreads();
  if (threadIdx.x < 128) atomicAdd(gb2v + threadIdx.x, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

}  // namespace pg

using namespace pg;

extern "C" int pg_attn_unfold_bias_grad(const float* gout, int ldg, const float* swn, const float* b2v, int n, const int* ids,
                                        float* gswn, float* gb2v, void* stream) {
  if (n <= 0) return PG_OK;
  if (!gout || !swn || !b2v || !gswn || !gb2v) { set_error("pg_attn_unfold_bias_grad: null argument"); return PG_ERR_ARG; }
  int blocks = (n + 3) / 4;
  if (blocks > 4 * kNumCU) blocks = 4 * kNumCU;
  hipLaunchKernelGGL(unfold_bias_grad_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, gout, ldg, swn, b2v, n, ids, gswn, gb2v);
  return check_launch("pg_attn_unfold_bias_grad");
}

extern "C" int pg_bond_rows_sum(const PgTopo* t, const float* Y, int ldy, int ncol, int by_src, float* out, int ldo, void* stream) {
  if (!t || !Y || !out) { set_error("pg_bond_rows_sum: null argument"); return PG_ERR_ARG; }
  if ((ncol & 3) || (ldy & 3) || (ldo & 3) || ((size_t)Y & 15) || ((size_t)out & 15)) {
    set_error("pg_bond_rows_sum: columns and leading dimensions must be multiples of 4 floats, 16-byte aligned rows");
    return PG_ERR_ARG;
  }
  if (t->n_lig <= 0 || ncol <= 0) return PG_OK;
  int blocks = (t->n_lig + 3) / 4;
  if (blocks > 8 * kNumCU) blocks = 8 * kNumCU;
  hipLaunchKernelGGL(bond_rows_sum_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, *t, Y, ldy, ncol, by_src, out, ldo);
  return check_launch("pg_bond_rows_sum");
}

extern "C" int pg_gemm_wgrad(const float* dY, int ldy, const float* X, int ldx, int M, int N, int K, float* gW, int ldgw,
                             float* gb, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0) return PG_OK;
  const int bn = (N + 63) / 64, bk = (K + 63) / 64;
  int split = (8 * kNumCU + bn * bk - 1) / (bn * bk);
  const int chunks = (M + 63) / 64;
  if (split > chunks) split = chunks;
  if (split < 1) split = 1;
  hipLaunchKernelGGL(gemm_wgrad_kernel, dim3(bn, bk, split), dim3(256), 0, (hipStream_t)stream, dY, ldy, X, ldx, M, N, K,
                     gW, ldgw, gb);
  return check_launch("pg_gemm_wgrad");
}

extern "C" int pg_ln_relu(const float* X, int ldx, const float* gamma, const float* beta, int M, float* Y, int ldy,
                          void* stream) {
  if (M <= 0) return PG_OK;
  int blocks = (M + 3) / 4;
  if (blocks > 8 * kNumCU) blocks = 8 * kNumCU;
  hipLaunchKernelGGL(ln_relu_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, X, ldx, gamma, beta, M, Y, ldy);
  return check_launch("pg_ln_relu");
}

extern "C" int pg_ln_relu_bwd(const float* X, int ldx, const float* gamma, const float* beta, const float* gY, int ldgy,
                              int M, float* gX, int ldgx, float* ggamma, float* gbeta, void* stream) {
  if (M <= 0) return PG_OK;
  int blocks = (M + 3) / 4;
  if (blocks > 4 * kNumCU) blocks = 4 * kNumCU;
  hipLaunchKernelGGL(ln_relu_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, X, ldx, gamma, beta, gY, ldgy, M,
                     gX, ldgx, ggamma, gbeta);
  return check_launch("pg_ln_relu_bwd");
}

extern "C" int pg_attn_fold_wgrad(const float* X, int ldx, const float* T, int n, const int* ids, float* gW2_l,
                                  void* stream) {
  if (n <= 0) return PG_OK;
  int blocks = (n + 3) / 4;
  if (blocks > kNumCU / 2) blocks = kNumCU / 2;     // x 8 column groups = 1024 workgroups; few flushes, long streams
  hipLaunchKernelGGL(fold_wgrad_kernel, dim3(blocks, 8), dim3(256), 0, (hipStream_t)stream, X, ldx, T, n, ids, gW2_l);
  return check_launch("pg_attn_fold_wgrad");
}
