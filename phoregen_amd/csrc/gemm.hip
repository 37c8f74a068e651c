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
  // column tiles of one row block are adjacent in dispatch order: the A rows they share are read from HBM once, then from L2
  const int row0 = blockIdx.y * BM, col0 = blockIdx.x * BN;
  const int K = p.K1 + p.K2;
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
}

// ------------------------------------------------------------------------------------------------------------
// W-stationary persistent variant for the tall bond-row GEMMs (M ~ 2e5, K <= 148, 128 output columns per block):
// the 128 x K weight block stays in LDS for the whole kernel, 64-row A tiles stream through a double buffer
// (next tile prefetched into registers during the MFMAs), the finished tile is staged through the consumed A buffer
// for float4 bias / gathered-row / activation epilogues.  8 waves: 2 row blocks x 4 column blocks of 32x32.
// ------------------------------------------------------------------------------------------------------------
constexpr int WS_THREADS = 512, WS_BM = 64, WS_LDC = 132;

template <int KP /* LDS row stride of the K dimension, odd */>
__global__ __launch_bounds__(WS_THREADS) void gemm_ws_kernel(PgGemm p, int tiles_per_col) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  constexpr int ABUF = WS_BM * (KP > WS_LDC ? KP : WS_LDC);
  float* const Ws = sm;                      // [128][KP]
  float* const Ab = sm + 128 * KP;           // [2][ABUF]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = p.K1 + p.K2;
  const int col0 = blockIdx.y * BN;
  const int rb = wave >> 2, cb = wave & 3;
  const int l31 = lane & 31, kh = lane >> 5;
  const bool ln = p.ln_gamma != nullptr;

  for (int i = tid; i < 128 * (K >> 2); i += WS_THREADS) {          // W rows, float4 pieces (K % 4 == 0)
    const int r = i / (K >> 2), kq = (i % (K >> 2)) * 4;
    f4 w = {0.f, 0.f, 0.f, 0.f};
    if (col0 + r < p.N) w = *reinterpret_cast<const f4*>(p.W + (size_t)(col0 + r) * p.ldw + kq);
#pragma unroll
    for (int j = 0; j < 4; ++j) Ws[r * KP + kq + j] = w[j];
  }

  // fetch mapping: 8 threads per row, thread `sub` owns float4 pieces sub, sub+8, ... of the row
  const int fr = tid >> 3, sub = tid & 7;
  f4 ra[5];
  auto fetch = [&](int tile) {
    const int grow = tile * WS_BM + fr;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      ra[j] = (f4){0.f, 0.f, 0.f, 0.f};
      const int kq = (j * 8 + sub) * 4;
      if (grow < p.M) {
        if (kq < p.K1) ra[j] = *reinterpret_cast<const f4*>(p.X + (size_t)grow * p.ldx + kq);
        else if (kq < K) ra[j] = *reinterpret_cast<const f4*>(p.X2 + (size_t)grow * p.ldx2 + (kq - p.K1));
      }
    }
  };
  auto stage = [&](float* dst) {
    if (ln) {     // LayerNorm(128)+ReLU of the fetched row (K1 == 128, K2 == 0): 8 lanes hold one row
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) s += (ra[j][0] + ra[j][1]) + (ra[j][2] + ra[j][3]);
      s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
      const float mu = s * (1.f / 128.f);
      float q = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) { ra[j][e] -= mu; q = fmaf(ra[j][e], ra[j][e], q); }
      q += __shfl_xor(q, 1); q += __shfl_xor(q, 2); q += __shfl_xor(q, 4);
      const float rs = 1.0f / sqrtf(q * (1.f / 128.f) + 1e-5f);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int kq = (j * 8 + sub) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) ra[j][e] = fmaxf(ra[j][e] * rs * p.ln_gamma[kq + e] + p.ln_beta[kq + e], 0.f);
      }
    }
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int kq = (j * 8 + sub) * 4;
      if (kq < K) {
#pragma unroll
        for (int e = 0; e < 4; ++e) dst[fr * KP + kq + e] = ra[j][e];
      }
    }
  };

  const int n_tiles = (p.M + WS_BM - 1) / WS_BM;
  const int t_begin = blockIdx.x * tiles_per_col, t_end = min(n_tiles, t_begin + tiles_per_col);
  const bool vec_ok = (p.ldy & 3) == 0 && ((size_t)p.Y & 15) == 0 && (p.N & 3) == 0 &&
                      (!p.add1 || ((p.ld_add1 & 3) == 0 && ((size_t)p.add1 & 15) == 0)) &&
                      (!p.add2 || ((p.ld_add2 & 3) == 0 && ((size_t)p.add2 & 15) == 0));
  if (t_begin < t_end) {
    fetch(t_begin);
    stage(Ab);
  }
  __syncthreads();
  int cur = 0;
  for (int tile = t_begin; tile < t_end; ++tile) {
    float* const A = Ab + cur * ABUF;
    if (tile + 1 < t_end) fetch(tile + 1);
    f16v acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float* ap = A + (rb * 32 + l31) * KP + kh;
    const float* wp = Ws + (cb * 32 + l31) * KP + kh;
#pragma unroll 8
    for (int ks = 0; ks < (K >> 1); ++ks) acc = mfma32(ap[2 * ks], wp[2 * ks], acc);
    __syncthreads();                                   // all waves are done reading A
#pragma unroll
    for (int r = 0; r < 16; ++r)                        // finished 64 x 128 tile -> the consumed A buffer
      A[(rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh) * WS_LDC + cb * 32 + l31] = acc[r];
    if (tile + 1 < t_end) stage(Ab + (1 - cur) * ABUF);
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 4; ++it) {                    // 64 rows x 32 float4 pieces over 512 threads
      const int piece = it * WS_THREADS + tid;
      const int r = piece >> 5, c4 = (piece & 31) * 4;
      const int grow = tile * WS_BM + r, gcol = col0 + c4;
      if (grow >= p.M || gcol >= p.N) continue;
      f4 v = *reinterpret_cast<const f4*>(A + r * WS_LDC + c4);
      const int a1 = p.add1 ? (p.idx1 ? p.idx1[grow] : grow) : 0;
      const int a2 = p.add2 ? (p.idx2 ? p.idx2[grow] : grow) : 0;
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
        *reinterpret_cast<f4*>(p.Y + (size_t)grow * p.ldy + gcol) = v;
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
          p.Y[(size_t)grow * p.ldy + gcol + j] = x * p.out_scale;
        }
      }
    }
    __syncthreads();                                   // epilogue reads of A are done before it is restaged
    cur = 1 - cur;
  }
}

template <int KP>
static int launch_ws(const PgGemm* p, hipStream_t st) {
  constexpr int ABUF = WS_BM * (KP > WS_LDC ? KP : WS_LDC);
  const size_t lds = (128 * KP + 2 * ABUF) * sizeof(float);
  if (int rc = reserve_lds(reinterpret_cast<const void*>(gemm_ws_kernel<KP>), lds, "pg_gemm(ws)")) return rc;
  const int n_tiles = (p->M + WS_BM - 1) / WS_BM, n_col = (p->N + BN - 1) / BN;
  int row_groups = kNumCU / n_col;
  if (row_groups < 1) row_groups = 1;
  const int tiles_per = (n_tiles + row_groups - 1) / row_groups;
  row_groups = (n_tiles + tiles_per - 1) / tiles_per;
  hipLaunchKernelGGL(gemm_ws_kernel<KP>, dim3(row_groups, n_col), dim3(WS_THREADS), lds, st, *p, tiles_per);
  return check_launch("pg_gemm(ws)");
}

// ------------------------------------------------------------------------------------------------------------
// Wave-specialised variant for the tall bond-row products (M ~ 2e5, K = 128 or 128+20, N a multiple of 128, plain epilogue:
// bias + up to two gathered row adds).  The tiled kernel above runs its load, MFMA and store phases in lockstep on all
// co-resident workgroups, so their times add (26 + 45 + 28 us at N = 128).  Here a persistent 8-wave workgroup per CU splits
// the roles:
//   * waves 4-7 ("memory"): fetch the NEXT 64-row A tile (and the tile's gather indices) into the other LDS buffer;
//   * waves 0-3 ("compute"): gathered epilogue operands -> registers (not touched until the MFMAs are done: waves issue in
//     order), 64 x 32 output block per wave on v_mfma_f32_32x32x2_f32 with the 128 x K weight block resident in LDS,
//     then bias / adds and stores straight from the accumulators;
//   * one workgroup barrier per tile hands the buffers over.  HBM reads, MFMAs and stores of neighbouring tiles overlap by
//     construction instead of by luck.
// ------------------------------------------------------------------------------------------------------------
constexpr int SP_THREADS = 512, SP_BM = 64;

template <int KP /* odd LDS row stride >= K */>
__global__ __launch_bounds__(SP_THREADS) void gemm_sp_kernel(PgGemm p PG_ABL_PARAM) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* const Ws = sm;                               // [128][KP]
  float* const Ab = Ws + 128 * KP;                    // [2][64][KP]
  int* const Ix = reinterpret_cast<int*>(Ab + 2 * SP_BM * KP);   // [2 buffers][2 index arrays][64]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool mem_role = wave >= 4;
  constexpr int K = KP - 1, K4 = K >> 2;            // 128 or 148: compile-time, so the piece -> (row, k) maps cost no divisions
  const int l31 = lane & 31, lh = lane >> 5;
  const int n_tiles = (p.M + SP_BM - 1) / SP_BM;
  const int n_cb = (p.N + 127) / 128;

  // memory-role mapping: 256 lanes, 64 rows x K4 float4 pieces
  const int mt = tid - 256;
  auto fetch_store = [&](int tile, int buf) {
    f4 ra[10];
    constexpr int total = SP_BM * K4;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      int e = mt + i * 256;
      e = e < total ? e : total - 1;
      const int r = e / K4, k4 = e - r * K4;
      int grow = tile * SP_BM + r;
      grow = grow < p.M ? grow : p.M - 1;
      const int kq = 4 * k4;
      ra[i] = kq < p.K1 ? *reinterpret_cast<const f4*>(p.X + (size_t)grow * p.ldx + kq)
                        : *reinterpret_cast<const f4*>(p.X2 + (size_t)grow * p.ldx2 + (kq - p.K1));
    }
    int iv = 0;
    if (mt < 128) {
      const int* src = mt < 64 ? p.idx1 : p.idx2;
      int grow = tile * SP_BM + (mt & 63);
      grow = grow < p.M ? grow : p.M - 1;
      iv = src ? src[grow] : grow;
    }
    float* dst = Ab + buf * SP_BM * KP;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      const int e = mt + i * 256;
      if (e < total) {
        const int r = e / K4, k4 = e - r * K4;
        float* d = dst + r * KP + 4 * k4;
        d[0] = ra[i][0]; d[1] = ra[i][1]; d[2] = ra[i][2]; d[3] = ra[i][3];
      }
    }
    if (mt < 128) Ix[buf * 128 + mt] = iv;
  };

  for (int cb = 0; cb < n_cb; ++cb) {
    const int col0 = cb * 128;
    __syncthreads();                                   // previous column block: nobody reads Ws / Ab any more
    for (int i = tid; i < 128 * K4; i += SP_THREADS) { // the column block's weight rows
      const int r = i / K4, kq = (i - r * K4) * 4;
      f4 w = {0.f, 0.f, 0.f, 0.f};
      if (col0 + r < p.N) w = *reinterpret_cast<const f4*>(p.W + (size_t)(col0 + r) * p.ldw + kq);
      float* d = Ws + r * KP + kq;
      d[0] = w[0]; d[1] = w[1]; d[2] = w[2]; d[3] = w[3];
    }
    if (mem_role && (int)blockIdx.x < n_tiles) fetch_store(blockIdx.x, 0);
    __syncthreads();
    int it = 0;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x, ++it) {
      const int buf = it & 1;
      if (mem_role) {
        if (tile + (int)gridDim.x < n_tiles && !PG_ABL(1)) fetch_store(tile + gridDim.x, buf ^ 1);
      } else {
        const int gcol = col0 + 32 * wave + l31;
        const bool col_ok = gcol < p.N;
        const int row0 = tile * SP_BM;
        // epilogue operands first (raw, consumed after the MFMAs)
        f16v l1[2], l2[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) { l1[i][r] = 0.f; l2[i][r] = 0.f; }
        const float bsv = (p.bias && col_ok) ? p.bias[gcol] : 0.f;
        if (col_ok) {
          const int* ix = Ix + buf * 128;
          if (p.add1) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int r = 0; r < 16; ++r)
                l1[i][r] = p.add1[(size_t)ix[32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh] * p.ld_add1 + gcol];
          }
          if (p.add2) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int r = 0; r < 16; ++r)
                l2[i][r] = p.add2[(size_t)ix[64 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh] * p.ld_add2 + gcol];
          }
        }
        f16v acc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        const float* a0 = Ab + buf * SP_BM * KP + l31 * KP + lh;
        const float* a1 = a0 + 32 * KP;
        const float* bw = Ws + (32 * wave + l31) * KP + lh;
        // fully unrolled (K is a compile-time constant): the scheduler spreads the 3 LDS operand reads of a k-step far ahead of
        // their MFMAs; a 4-step rolled loop put a wait in front of every MFMA and ran at half the rate
        constexpr int ksteps = K >> 1;
#pragma unroll
        for (int ks = 0; ks < ksteps; ++ks) {
          if (PG_ABL(2) && ks > 0) break;
          const float b = bw[2 * ks];
          acc[0] = mfma32(a0[2 * ks], b, acc[0]);
          acc[1] = mfma32(a1[2 * ks], b, acc[1]);
        }
        if (col_ok && !PG_ABL(4)) {
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int grow = row0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
              if (grow < p.M) p.Y[(size_t)grow * p.ldy + gcol] = ((acc[i][r] + bsv) + l1[i][r]) + l2[i][r];
            }
        }
      }
      __syncthreads();                                 // next tile landed, this tile's buffer is free
    }
  }
}

template <int KP>
static int launch_sp(const PgGemm* p, hipStream_t st) {
  const size_t lds = ((size_t)128 * KP + 2 * SP_BM * KP + 256) * sizeof(float);
  if (int rc = reserve_lds(reinterpret_cast<const void*>(gemm_sp_kernel<KP>), lds, "pg_gemm(sp)")) return rc;
  const int n_tiles = (p->M + SP_BM - 1) / SP_BM;
  hipLaunchKernelGGL(gemm_sp_kernel<KP>, dim3(n_tiles < kNumCU ? n_tiles : kNumCU), dim3(SP_THREADS), lds, st, *p PG_ABL_ARG("PG_GEMM_ABLATE"));
  return check_launch("pg_gemm(sp)");
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

static int g_gemm_sp = 1;      // 0: tiled kernel only; 1: default; 2: wave-specialised kernel also at K = 128; 4: no streaming kernel
extern "C" int pg_debug_gemm_specialised(int on) { const int old = g_gemm_sp; g_gemm_sp = on; return old; }

extern "C" int pg_gemm(const PgGemm* p, void* stream) {
  if (!p || !p->X || !p->W || !p->Y || p->M < 0 || p->N <= 0) { pg::set_error("pg_gemm: bad arguments"); return PG_ERR_ARG; }
  if (p->M == 0) return PG_OK;
  if (p->K2 > 0 && !p->X2) { pg::set_error("pg_gemm: K2 > 0 without X2"); return PG_ERR_ARG; }
  if (p->ln_gamma && (p->K2 != 0 || p->K1 != 128 || (p->ldx & 3) || ((size_t)p->X & 15))) { pg::set_error("pg_gemm: LayerNorm-on-load needs K1 == 128, K2 == 0, 16-byte aligned rows"); return PG_ERR_ARG; }
  // tall, thin-K problems: W-stationary persistent kernel (needs float4-able operands)
  const int K = p->K1 + p->K2;
  const bool al = (p->ldx & 3) == 0 && ((size_t)p->X & 15) == 0 && (p->ldw & 3) == 0 && ((size_t)p->W & 15) == 0 &&
                  (p->K1 & 3) == 0 && (p->K2 & 3) == 0 && (!p->K2 || ((p->ldx2 & 3) == 0 && ((size_t)p->X2 & 15) == 0));
  // K = 128 (+ 20) / K = 20 products with a plain epilogue or LayerNorm-on-load: the streaming kernel (LDS-DMA tiles, no vector-ALU
  // work on the memory path; gemm_stream.hip)
  if ((g_gemm_sp & 1) && pg::gemm_stream_eligible(p)) return pg::launch_gemm_stream(p, (hipStream_t)stream);
  // (measured, tools/bench_gemm.py: a clear win for the LayerNorm-on-load form, a wash or slightly worse otherwise)
  if (p->M >= 32768 && al && p->N <= 256 && p->ln_gamma && !p->rows) {
    if (K == 128) return pg::launch_ws<129>(p, (hipStream_t)stream);
    if (K == 148) return pg::launch_ws<149>(p, (hipStream_t)stream);
  }
  // plain-epilogue bond-row products: wave-specialised persistent kernel (loads / MFMAs / stores of neighbouring tiles overlap)
  // Measured (tools/bench_gemm.py, M = 203 720): K = 148 with two gathered adds 221 vs 295 us on the tiled kernel; at K = 128
  // the specialised kernel is 5-10 % slower than the tiled one (118 vs 108 us: its per-tile barrier hand-over, not the MFMAs,
  // sets the pace), so only the [h_bond | G] product takes this path unless pg_debug_gemm_specialised(2) forces it
  if (g_gemm_sp && p->M >= 32768 && al && !p->ln_gamma && !p->rows && p->act == 0 && p->out_scale == 1.0f && p->K1 == 128 &&
      (p->K2 == 20 || (p->K2 == 0 && g_gemm_sp == 2)) && (p->N & 127) == 0 && (!p->add1 || p->idx1) && (!p->add2 || p->idx2)) {
    return p->K2 ? pg::launch_sp<149>(p, (hipStream_t)stream) : pg::launch_sp<129>(p, (hipStream_t)stream);
  }
  dim3 grid((p->N + pg::BN - 1) / pg::BN, (p->M + pg::BM - 1) / pg::BM);
  hipLaunchKernelGGL(pg::gemm_kernel, grid, dim3(256), 0, (hipStream_t)stream, *p PG_ABL_ARG("PG_GEMM_ABLATE"));
  return pg::check_launch("pg_gemm");
}

extern "C" int pg_rows_linear(const float* X, int ldx, int K, const float* W, const float* b, int n_out, int M,
                              const int* rows, float* Y, int ldy, void* stream) {
  if (n_out > 16 || n_out <= 0 || K <= 0 || K > 256) { pg::set_error("pg_rows_linear: n_out must be 1..16, K 1..256"); return PG_ERR_ARG; }
  if (M == 0) return PG_OK;
  if (K == 128 && (ldx & 3) == 0 && ((size_t)X & 15) == 0) {
    int blocks = (M + 63) / 64;
    if (blocks > 8 * pg::kNumCU) blocks = 8 * pg::kNumCU;
    hipLaunchKernelGGL(pg::rows_linear_mfma_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, X, ldx, W, b, n_out, M, rows, Y, ldy);
    return pg::check_launch("pg_rows_linear");
  }
  hipLaunchKernelGGL(pg::rows_linear_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, X, ldx, K, W, b,
                     n_out, M, rows, Y, ldy);
  return pg::check_launch("pg_rows_linear");
}
