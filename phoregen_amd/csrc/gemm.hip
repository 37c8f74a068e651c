// fp32 MFMA GEMM for the dense node / edge linear layers:  Y = act(Xcat . W^T + bias + gathered adds).
// 128x128 block tile, 4 waves x (64x64) of v_mfma_f32_32x32x2_f32, K streamed through LDS in 32-wide
// chunks (row stride 33 floats: conflict-free ds_read_b32 for both operands).  Optional
// LayerNorm(128)+ReLU folded into the A-tile load (second half of models/common.py:99-119 MLPs).
#include <stdlib.h>

#include "common.h"
#include "../../include/phoregen_hip.h"

namespace pg {

constexpr int BM = 128, BN = 128, BK = 32, LDT = BK + 1;


__global__ __launch_bounds__(256) void gemm_kernel(PgGemm p PG_ABL_PARAM) {
  __shared__ __attribute__((aligned(16))) float smem[(BM + BN) * LDT];   // staging tiles; reused by the epilogue
  float* const As = smem;
  float* const Bs = smem + BM * LDT;
  __shared__ float rstat[BM * 2];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // column tiles of one row block are adjacent in dispatch order: the A rows they share are read from HBM once, then from L2.
  // Row blocks sit on grid.y (<= 65535): a taller problem (M > ~8.39 M rows) walks them with a stride
  const int col0 = blockIdx.x * BN;
  const int K = p.K1 + p.K2;
  for (int rb = blockIdx.y; rb * BM < p.M; rb += gridDim.y) {
  const int row0 = rb * BM;
  const int wr = (wave >> 1) * 64, wc = (wave & 1) * 64;  // wave sub-tile origin
  const bool ln = p.ln_gamma != nullptr;

  if (ln) {  // per-row mean / rstd over K1 = 128 columns: 8 lanes per row, each 4 x float4 (coalesced 128-B pieces)
    const int sub = tid & 7;
    for (int r = tid >> 3; r < BM; r += 32) {
      const int grow = row0 + r;
      const int prow = (p.rows && grow < p.M) ? p.rows[grow] : grow;
      f4 v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        v[i] = (f4){0.f, 0.f, 0.f, 0.f};
        if (grow < p.M) v[i] = *reinterpret_cast<const f4*>(p.X + (size_t)prow * p.ldx + (i * 8 + sub) * 4);
      }
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
      s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
      const float mu = s * (1.f / 128.f);
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float d = v[i][j] - mu; q = fmaf(d, d, q); }
      q += __shfl_xor(q, 1); q += __shfl_xor(q, 2); q += __shfl_xor(q, 4);
      if (sub == 0) {
        rstat[r * 2] = mu;
        rstat[r * 2 + 1] = 1.0f / sqrtf(q * (1.f / 128.f) + 1e-5f);
      }
    }
    __syncthreads();
  }

  f16v acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const bool fastX = (p.ldx & 3) == 0 && (p.K1 & 3) == 0 && ((size_t)p.X & 15) == 0;
  const bool fastW = (p.ldw & 3) == 0 && (K & 3) == 0 && ((size_t)p.W & 15) == 0;

  for (int k0 = 0; k0 < K; k0 += BK) {
    // ---- stage A (X rows) and B (W rows) chunks: each thread 4 x (4 consecutive k) ----
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = (tid >> 3) + 32 * i, kq = (tid & 7) * 4, kk = k0 + kq;
      const int grow = row0 + r;
      const int prow = (p.rows && grow < p.M) ? p.rows[grow] : grow;
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (grow < p.M && !PG_ABL(1)) {
        if (fastX && kk + 3 < p.K1) {
          const float4 t = *reinterpret_cast<const float4*>(p.X + (size_t)prow * p.ldx + kk);
          v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int k = kk + j;
            if (k < p.K1) v[j] = p.X[(size_t)prow * p.ldx + k];
            else if (k < K) v[j] = p.X2[(size_t)prow * p.ldx2 + (k - p.K1)];
          }
        }
        if (ln) {
          const float mu = rstat[r * 2], rs = rstat[r * 2 + 1];
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (kk + j < p.K1) v[j] = fmaxf((v[j] - mu) * rs * p.ln_gamma[kk + j] + p.ln_beta[kk + j], 0.f);
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) As[r * LDT + kq + j] = v[j];

      const int gcol = col0 + r;
      float w[4] = {0.f, 0.f, 0.f, 0.f};
      if (gcol < p.N && !PG_ABL(1)) {
        if (fastW && kk + 3 < K) {
          const float4 t = *reinterpret_cast<const float4*>(p.W + (size_t)gcol * p.ldw + kk);
          w[0] = t.x; w[1] = t.y; w[2] = t.z; w[3] = t.w;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (kk + j < K) w[j] = p.W[(size_t)gcol * p.ldw + kk + j];
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) Bs[r * LDT + kq + j] = w[j];
    }
    __syncthreads();
    // ---- 16 k-steps of 2 ----
    const int l31 = lane & 31, kh = lane >> 5;
#pragma unroll 4
    for (int ks = 0; ks < (PG_ABL(2) ? 1 : BK / 2); ++ks) {
      const int k = ks * 2 + kh;
      const float a0 = As[(wr + l31) * LDT + k], a1 = As[(wr + 32 + l31) * LDT + k];
      const float b0 = Bs[(wc + l31) * LDT + k], b1 = Bs[(wc + 32 + l31) * LDT + k];
      acc[0][0] = mfma32(a0, b0, acc[0][0]);
      acc[0][1] = mfma32(a0, b1, acc[0][1]);
      acc[1][0] = mfma32(a1, b0, acc[1][0]);
      acc[1][1] = mfma32(a1, b1, acc[1][1]);
    }
    __syncthreads();
  }

  // ---- epilogue: accumulators -> LDS (one 64-row half at a time, reusing the staging buffers) -> row-wise float4
  // pieces: bias + gathered adds + activation, 16-byte loads/stores when the operands allow it ----
  float* const Cs = smem;                     // 64 x 132 floats = 33792 B = sizeof(smem)
  static_assert(64 * (BN + 4) <= (BM + BN) * LDT, "epilogue tile must fit the staging buffer");
  constexpr int LDC = BN + 4;
  const int l31 = lane & 31, lh = lane >> 5;
  const bool vec_ok = (p.ldy & 3) == 0 && ((size_t)p.Y & 15) == 0 && (p.N & 3) == 0 &&
                      (!p.add1 || ((p.ld_add1 & 3) == 0 && ((size_t)p.add1 & 15) == 0)) &&
                      (!p.add2 || ((p.ld_add2 & 3) == 0 && ((size_t)p.add2 & 15) == 0));
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    __syncthreads();
    if ((wave >> 1) == half) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            Cs[(32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh) * LDC + wc + 32 * j + l31] = acc[i][j][r];
    }
    __syncthreads();
    // 64 rows x 32 float4 = 2048 pieces over 256 threads
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int piece = it * 256 + tid;
      const int r = piece >> 5, c4 = (piece & 31) * 4;
      const int grow = row0 + half * 64 + r, gcol = col0 + c4;
      if (grow >= p.M || gcol >= p.N || (PG_ABL(4) && r != 0)) continue;
      f4 v = *reinterpret_cast<const f4*>(Cs + r * LDC + c4);
      const int prow = p.rows ? p.rows[grow] : grow;
      const int a1 = p.add1 ? (p.idx1 ? p.idx1[grow] : prow) : 0;
      const int a2 = p.add2 ? (p.idx2 ? p.idx2[grow] : prow) : 0;
      if (vec_ok) {
        if (p.bias) v += *reinterpret_cast<const f4*>(p.bias + gcol);
        if (p.add1) v += *reinterpret_cast<const f4*>(p.add1 + (size_t)a1 * p.ld_add1 + gcol);
        if (p.add2) v += *reinterpret_cast<const f4*>(p.add2 + (size_t)a2 * p.ld_add2 + gcol);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (p.act == 1) v[j] = ssp(v[j]);
          else if (p.act == 2) v[j] = fmaxf(v[j], 0.f);
          v[j] *= p.out_scale;
        }
        *reinterpret_cast<f4*>(p.Y + (size_t)prow * p.ldy + gcol) = v;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (gcol + j >= p.N) break;
          float x = v[j];
          if (p.bias) x += p.bias[gcol + j];
          if (p.add1) x += p.add1[(size_t)a1 * p.ld_add1 + gcol + j];
          if (p.add2) x += p.add2[(size_t)a2 * p.ld_add2 + gcol + j];
          if (p.act == 1) x = ssp(x);
          else if (p.act == 2) x = fmaxf(x, 0.f);
          p.Y[(size_t)prow * p.ldy + gcol + j] = x * p.out_scale;
        }
      }
    }
  }
  __syncthreads();      // the next row block restages the LDS tiles this one's epilogue has just read
  }
}

// ---- small per-row linear (n_out <= 16): one wave per row ----------------------------------------
__global__ __launch_bounds__(256) void rows_linear_kernel(const float* X, int ldx, int K, const float* W, const float* b,
                                                          int n_out, int M, const int* rows, float* Y, int ldy) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= M) return;
  const int src = rows ? rows[r] : r;
  float x[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) x[i] = (lane + 64 * i) < K ? X[(size_t)src * ldx + lane + 64 * i] : 0.f;
  for (int o = 0; o < n_out; ++o) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (lane + 64 * i < K) s += x[i] * W[o * K + lane + 64 * i];
    s = wave_sum(s);
    if (lane == 0) Y[(size_t)r * ldy + o] = s + (b ? b[o] : 0.f);
  }
}

// K = 128 form of the same product on the matrix pipe: 64-row tiles staged in LDS by coalesced 16-byte loads (rows 132 floats
// apart), one 16-row block per wave, v_mfma_f32_16x16x4_f32 with A = X (lane (row m, k g) from LDS), B = W (lane (k g, output m),
// 32 registers for the whole kernel), D lane (g, m) = rows 4g.., output m.  The one-wave-per-row kernel above reads a 104 MB
// operand at 1.35 TB/s (77 us); this one is bound by that read.
constexpr int RL_LD = 132;
__global__ __launch_bounds__(256) void rows_linear_mfma_kernel(const float* X, int ldx, const float* W, const float* b, int n_out,
                                                               int M, const int* rows, float* Y, int ldy) {
  __shared__ __attribute__((aligned(16))) float xs[64 * RL_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, m = lane & 15;
  float wr[32];
#pragma unroll
  for (int s_ = 0; s_ < 32; ++s_) wr[s_] = m < n_out ? W[m * 128 + 4 * s_ + g] : 0.f;
  const float bias = (b && m < n_out) ? b[m] : 0.f;
  const int n_tiles = (M + 63) >> 6;
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int row0 = tile * 64;
    __syncthreads();                                   // the previous tile is no longer read
#pragma unroll
    for (int i = 0; i < 8; ++i) {                      // 64 rows x 32 float4: thread = (row tid >> 5 + 8 i, piece tid & 31)
      const int r = (tid >> 5) + 8 * i, c4 = tid & 31;
      int grow = row0 + r;
      grow = grow < M ? grow : M - 1;
      const int src = rows ? rows[grow] : grow;
      *reinterpret_cast<f4*>(xs + r * RL_LD + 4 * c4) = __builtin_nontemporal_load(reinterpret_cast<const f4*>(X + (size_t)src * ldx) + c4);
    }
    __syncthreads();
    f4 acc = {bias, bias, bias, bias};
    const float* xa = xs + (16 * wave + m) * RL_LD + g;
#pragma unroll
    for (int s_ = 0; s_ < 32; ++s_) acc = mfma16(xa[4 * s_], wr[s_], acc);
    if (m < n_out) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int grow = row0 + 16 * wave + 4 * g + r;
        if (grow < M) Y[(size_t)grow * ldy + m] = acc[r];
      }
    }
  }
}

}  // namespace pg

namespace pg {   // gemm_stream.hip
bool gemm_stream_eligible(const PgGemm* p);
int launch_gemm_stream(const PgGemm* p, hipStream_t st);
}

static int g_gemm_stream = 1;      // 0: tiled kernel only (tests: the two kernels against each other)
extern "C" int pg_debug_gemm_streaming(int on) { const int old = g_gemm_stream; g_gemm_stream = on; return old; }

extern "C" int pg_gemm(const PgGemm* p, void* stream) {
  if (!p || !p->X || !p->W || !p->Y || p->M < 0 || p->N <= 0) { pg::set_error("pg_gemm: bad arguments"); return PG_ERR_ARG; }
  if (p->M == 0) return PG_OK;
  if (p->K2 > 0 && !p->X2) { pg::set_error("pg_gemm: K2 > 0 without X2"); return PG_ERR_ARG; }
  if (p->ln_gamma && (p->K2 != 0 || p->K1 != 128 || (p->ldx & 3) || ((size_t)p->X & 15))) { pg::set_error("pg_gemm: LayerNorm-on-load needs K1 == 128, K2 == 0, 16-byte aligned rows"); return PG_ERR_ARG; }
  // K = 128 (+ 20) / K = 20 products with a plain epilogue or LayerNorm-on-load: the streaming kernel (LDS-DMA tiles, no vector-ALU
  // work on the memory path; gemm_stream.hip); everything else (row subsets, odd K, two gathered operands): the tiled kernel
  if (g_gemm_stream && pg::gemm_stream_eligible(p)) return pg::launch_gemm_stream(p, (hipStream_t)stream);
  const int row_blocks = (p->M + pg::BM - 1) / pg::BM;
  dim3 grid((p->N + pg::BN - 1) / pg::BN, row_blocks < 65535 ? row_blocks : 65535);
  hipLaunchKernelGGL(pg::gemm_kernel, grid, dim3(256), 0, (hipStream_t)stream, *p PG_ABL_ARG("PG_GEMM_ABLATE"));
  return pg::check_launch("pg_gemm");
}

extern "C" int pg_rows_linear(const float* X, int ldx, int K, const float* W, const float* b, int n_out, int M,
                              const int* rows, float* Y, int ldy, void* stream) {
  if (n_out > 16 || n_out <= 0 || K <= 0 || K > 256) { pg::set_error("pg_rows_linear: n_out must be 1..16, K 1..256"); return PG_ERR_ARG; }
  if (M == 0) return PG_OK;
  if (K == 128 && (ldx & 3) == 0 && ((size_t)X & 15) == 0) {
    int blocks = (M + 63) / 64;
    if (blocks > 8 * kNumCU) blocks = 8 * kNumCU;
    hipLaunchKernelGGL(pg::rows_linear_mfma_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, X, ldx, W, b, n_out, M, rows, Y, ldy);
    return pg::check_launch("pg_rows_linear");
  }
  hipLaunchKernelGGL(pg::rows_linear_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, X, ldx, K, W, b,
                     n_out, M, rows, Y, ldy);
  return pg::check_launch("pg_rows_linear");
}
