// Streaming fp32 GEMM for the tall bond-row products (M ~ 2e5 rows, K = 128, N = 128 or 256; models/common.py:99-119 first
// layers over h_bond, call sites uni_denoiser.py:43-59,141-155):  Y = X . W^T + bias + add1[idx1] + add2[idx2].
//
// Why another kernel: on a gfx950 SIMD a wave that issues MFMAs back to back leaves no issue slot for the VECTOR-ALU
// instructions of the other waves (tools/micro/mfma_hbm.hip: an MFMA kernel and a copy kernel whose loop computes its addresses
// on the vector ALU take the SUM of their times on two streams, 356 + 243 -> 600 us; the same copy addressed through a buffer
// descriptor + scalar offsets overlaps, 385 us).  The tiled kernel (gemm.hip) stages its tiles through registers and the
// vector ALU, so its load, MFMA and store phases add up (26 + 45 + 28 us at N = 128) no matter how many workgroups are
// resident.  Here the steady state of the memory path has NO vector-ALU instruction:
//   * A tiles (64 rows x 128 k) go HBM -> LDS by `buffer_load_dwordx4 ... lds` (LDS-DMA): descriptor + one lane-fixed offset
//     register + a scalar tile offset, the LDS destination in M0; two stages, the next tile in flight during the MFMAs;
//   * the LDS image is what the DMA writes (lane-linear 1 KB pieces of 8 rows x 128 B) with the XOR swizzle applied to the
//     SOURCE address: piece (i, j) = rows 8i.., k 32j..; row r of a piece at r*128 B, its 16-byte slot p holds
//     k-group p ^ ((row >> 1) & 7) -> the MFMA operand reads (one ds_read_b128 per lane = a whole slot) are conflict-free in the
//     16-lane groups the LDS serves a b128 read in;
//   * the wave's 32 columns of W stay in 64 registers for the whole kernel (lane = (column, k half)): one LDS operand per MFMA;
//   * results leave straight from the accumulators by buffer stores (descriptor + lane-fixed offset + scalar row offset;
//     every tile is a full tile: the last one starts at row M - 64);
//   * the gathered epilogue operands of tile t+1 are fetched during tile t and become the MFMA accumulator init of tile t+1.
// One raw s_barrier per tile; the LDS-DMA is retired by a counted s_waitcnt (the compiler does not count inline-asm loads):
// the only younger vector-memory operations of a wave at that point are the tile's 32 stores (more would be harmless, fewer
// would let the wait pass early; tests/test_host_cpu.py cross-compiles this file and checks 32 buffer_store_dword per tile
// body and no scratch in every variant).
// v_mfma_f32_32x32x2_f32: A lane (l31 = row, kh = k), B lane (l31 = column, kh = k), D reg r = row 8(r>>2) + 4 kh + (r&3).
// The k order inside a chunk of eight is free: step s (0..3) of chunk c uses k = 8c + 4 kh + s on both operands, so a lane's
// four values of a chunk are one 16-byte slot.
#include <type_traits>

#include "common.h"
#include "../../include/phoregen_hip.h"

namespace pg {

typedef int i4v __attribute__((ext_vector_type(4)));
typedef int i2v __attribute__((ext_vector_type(2)));

constexpr int ST_BM = 64;                       // rows per tile
constexpr int ST_STAGE = ST_BM * 128 * 4;       // 32 KB

__device__ __forceinline__ i4v st_desc(const void* base, unsigned bytes) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(base);
  i4v d;
  d[0] = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu));
  d[1] = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu));      // stride 0: raw buffer, byte offsets
  d[2] = __builtin_amdgcn_readfirstlane((int)bytes);
  d[3] = 0x00020000;
  return d;
}

// one 1 KB piece: 64 lanes x 16 bytes from base + voff + soff to LDS byte address lds_dst + 16 * lane
__device__ __forceinline__ void st_dma(unsigned lds_dst, unsigned voff, i4v desc, unsigned soff) {
  asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :: "s"(lds_dst), "v"(voff), "s"(desc), "s"(soff) : "memory");
}

template <int NW /* waves: 32 output columns each */, int NADD /* 0 | 1: rows add1[idx1[r]] | 2: rows add1[r] */, int K1 /* 128 | 0 */,
          int K2 /* 0 | 20 */, bool LN /* LayerNorm(128) + ReLU on the X rows (K1 = 128, NW = 4) */,
          bool SSP = false /* shifted softplus on the result (the heads' first layers) */>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void gemm_stream_kernel(PgGemm p, int n_tiles) {
  extern __shared__ __attribute__((aligned(1024))) char st_lds[];     // the ONLY LDS object: stage s at byte s * 32 KB
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l31 = lane & 31, kh = lane >> 5;
  const int colw = blockIdx.y * (NW * 32) + 32 * wave;                 // the wave's first output column
  const int col = colw + l31;

  // ---- the wave's W slice: 64 registers for the whole kernel ----
  float Wr[K1 ? 64 : 1];
  if constexpr (K1 > 0) {
    const float* wrow = p.W + (size_t)col * p.ldw + 4 * kh;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const f4 w = *reinterpret_cast<const f4*>(wrow + 8 * c);
#pragma unroll
      for (int e = 0; e < 4; ++e) Wr[4 * c + e] = w[e];
    }
  }
  // second operand [X | X2] (K2 = 20, the Gaussian smearing of the bond length): no LDS, the lane's ten values of a row
  // (k = K1 + 10 kh + s) come straight from HBM one tile ahead; same split of k for the weights
  float W2r[K2 ? 10 : 1];
  if constexpr (K2 > 0) {
#pragma unroll
    for (int s2 = 0; s2 < 10; ++s2) W2r[s2] = p.W[(size_t)col * p.ldw + K1 + 10 * kh + s2];
  }
  const unsigned ldx2b = (unsigned)p.ldx2 * 4u;
  const __amdgpu_buffer_rsrc_t descX2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(K2 ? p.X2 : nullptr), 0, (unsigned)p.M * ldx2b, 0x00020000);
  const unsigned voffX2 = (unsigned)l31 * ldx2b + 40u * kh;
  float xa[2][10], xb[2][10];               // this tile's / the next tile's X2 values, roles swap with the LDS stage
  auto load_x2 = [&](unsigned row0, float (&x)[2][10]) {
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const unsigned so = (row0 + 32u * b) * ldx2b;
      const i4v v0 = __builtin_amdgcn_raw_buffer_load_b128(descX2, voffX2, so, 0);
      const i4v v1 = __builtin_amdgcn_raw_buffer_load_b128(descX2, voffX2 + 16u, so, 0);
      const i2v v2 = __builtin_amdgcn_raw_buffer_load_b64(descX2, voffX2 + 32u, so, 0);
#pragma unroll
      for (int e = 0; e < 4; ++e) { x[b][e] = __builtin_bit_cast(float, (int)v0[e]); x[b][4 + e] = __builtin_bit_cast(float, (int)v1[e]); }
      x[b][8] = __builtin_bit_cast(float, (int)v2[0]);
      x[b][9] = __builtin_bit_cast(float, (int)v2[1]);
    }
  };
  const float bias = p.bias ? p.bias[col] : 0.f;
  f16v biasv;
#pragma unroll
  for (int r = 0; r < 16; ++r) biasv[r] = bias;

  // LayerNorm-on-load: gamma | beta behind the two stages (reached by ds instructions only, never by the DMA)
  const f4* const lnt = reinterpret_cast<const f4*>(st_lds + 2 * ST_STAGE);
  if constexpr (LN) {
    float* t = reinterpret_cast<float*>(st_lds + 2 * ST_STAGE);
    if (threadIdx.x < 128) { t[threadIdx.x] = p.ln_gamma[threadIdx.x]; t[128 + threadIdx.x] = p.ln_beta[threadIdx.x]; }
    __syncthreads();                             // (no DMA in flight yet)
  }

  // ---- lane-fixed offsets ----
  const unsigned ldxb = (unsigned)p.ldx * 4u, ldyb = (unsigned)p.ldy * 4u;
  const i4v descX = st_desc(p.X, (unsigned)p.M * ldxb);
  const __amdgpu_buffer_rsrc_t descY = __builtin_amdgcn_make_buffer_rsrc(p.Y, 0, (unsigned)p.M * ldyb, 0x00020000);
  // DMA source of lane (row_sub, slot pp) in a piece: the slot holds k-group pp ^ ((row >> 1) & 7), row = 8 i + row_sub
  const unsigned row_sub = lane >> 3, pp = lane & 7;
  const unsigned voff_even = row_sub * ldxb + ((pp ^ ((row_sub >> 1) & 7u)) << 4);
  const unsigned voff_odd = row_sub * ldxb + ((pp ^ ((4u + (row_sub >> 1)) & 7u)) << 4);
  const unsigned voff_wave = (wave & 1) ? voff_odd : voff_even;        // NW == 8: wave w fetches piece row i = w
  // operand reads: row R = 32 b + l31, chunk c (k = 8c .. 8c+7): the lane's k-group is 2c + kh = 8 j + 2 q + kh (j = c >> 2,
  // q = c & 3) -> piece (4 b + (l31 >> 3), j), row l31 & 7, slot (2 q + kh) ^ ((l31 >> 1) & 7)
  unsigned rd[4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
    rd[q] = (unsigned)(l31 >> 3) * 4096u + (unsigned)(l31 & 7) * 128u + (((unsigned)(2 * q + kh) ^ ((unsigned)(l31 >> 1) & 7u)) << 4);
  unsigned voffY[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) voffY[t] = (unsigned)(4 * kh + t) * ldyb + (unsigned)l31 * 4u;

  // gathered epilogue operands of the NEXT tile (registers carried round the loop): they initialise its accumulators
  f16v g1[2];
  const __amdgpu_buffer_rsrc_t descI1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(NADD ? p.idx1 : nullptr), 0, (unsigned)p.M * 4u, 0x00020000);
  const __amdgpu_buffer_rsrc_t descA1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(NADD ? p.add1 : nullptr), 0, 0xfffffff0u, 0x00020000);
  const unsigned ld1b = (unsigned)p.ld_add1 * 4u;
  const unsigned colb = (unsigned)col * 4u;
  // every tile is a full tile: the last one starts at row M - 64 and recomputes a few rows of its neighbour (same values)
  const unsigned last_row0 = (unsigned)p.M - ST_BM;
  auto tile_row0 = [&](unsigned tl) { const unsigned r = tl * ST_BM; return r < last_row0 ? r : last_row0; };
  // indices of the lane's 16 rows of block b: rows 32 b + 8 q + 4 kh + t -> one 16-byte load per (b, q)
  auto load_idx = [&](__amdgpu_buffer_rsrc_t dI, unsigned row0, i4v (&ix)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) ix[e] = __builtin_amdgcn_raw_buffer_load_b128(dI, 16u * kh, (row0 + 8u * e) * 4u, 0);
  };
  auto gather = [&](__amdgpu_buffer_rsrc_t dA, unsigned ldb, const i4v (&ix)[8], f16v (&g)[2]) {
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int t = 0; t < 4; ++t)
          g[b][4 * q + t] = __builtin_bit_cast(
              float, __builtin_amdgcn_raw_buffer_load_b32(dA, (unsigned)ix[4 * b + q][t] * ldb + colb, 0, 0));
  };

  // NADD == 2: the added rows are the output rows themselves (add1[r]): same addressing as the stores, no indices
  unsigned voffA[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) voffA[t] = (unsigned)(4 * kh + t) * ld1b + colb;
  auto gather_plain = [&](unsigned row0, f16v (&g)[2]) {
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        g[b][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(descA1, voffA[r & 3], (row0 + 32u * b + 8u * (r >> 2)) * ld1b, 0));
  };

  const unsigned tile_step = gridDim.x;
  unsigned tile = blockIdx.x;

  auto dma_tile = [&](unsigned tl, unsigned stage) {
    const unsigned row0b = tile_row0(tl) * ldxb;
    if constexpr (NW == 4) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const unsigned i = 2u * wave + (q >> 2), j = q & 3;
        st_dma(stage * ST_STAGE + (i * 4u + j) * 1024u, (q >> 2) ? voff_odd : voff_even, descX, row0b + i * 8u * ldxb + j * 128u);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        st_dma(stage * ST_STAGE + ((unsigned)wave * 4u + j) * 1024u, voff_wave, descX, row0b + (unsigned)wave * 8u * ldxb + j * 128u);
    }
  };

  // ---- prologue: first tile's DMA and gathers ----
  if constexpr (K2 > 0) load_x2(tile_row0(tile), xa);
  if constexpr (K1 > 0) dma_tile(tile, 0);
  if constexpr (NADD == 1) { i4v ix[8]; load_idx(descI1, tile_row0(tile), ix); gather(descA1, ld1b, ix, g1); }
  if constexpr (NADD == 2) gather_plain(tile_row0(tile), g1);

  auto body = [&](auto stage_c, auto first_c) {
    constexpr unsigned stage = decltype(stage_c)::value;
    constexpr bool first = decltype(first_c)::value;
    // this tile's DMA has landed (own pieces: counted wait; everybody's: barrier); the other stage is no longer read
    if constexpr (K1 > 0) {
      if constexpr (first) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(32)\n\ts_barrier" ::: "memory");
    }
    const unsigned next = tile + tile_step;
    float (&xc)[2][10] = stage ? xb : xa;
    float (&xn)[2][10] = stage ? xa : xb;
    // VMEM order matters (one in-order counter): the index loads go out BEFORE the DMA and are consumed late in the tile, so
    // the wait the compiler puts in front of their use (it cannot see the DMA) finds the DMA long since landed
    i4v ix1[8];
    if constexpr (NADD == 1) load_idx(descI1, tile_row0(next), ix1);
    if constexpr (K2 > 0) load_x2(tile_row0(next), xn);
    if constexpr (K1 > 0) dma_tile(next, stage ^ 1u);   // past the last tile: the clamped last tile once more, never consumed

    if constexpr (LN) {
      // normalise the landed tile in place: 4 threads per row, each its 128-byte piece row, read in k order (slot q ^ sw holds
      // k-group q): a row's sums must not depend on where in a tile the row sits (rows of the anchored last tile are computed
      // twice, and a graph alone must give the bits it gives inside a batch).  Same arithmetic as the tiled kernel's
      // LayerNorm-on-load (two-pass variance, (x - mu) * rstd * gamma + beta, ReLU)
      const unsigned row = threadIdx.x >> 2, jj = threadIdx.x & 3;
      char* const base = st_lds + stage * ST_STAGE + ((row >> 3) * 4u + jj) * 1024u + (row & 7u) * 128u;
      const unsigned sw = (row >> 1) & 7u;
      f4 v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = *reinterpret_cast<const f4*>(base + 16 * (q ^ sw));   // v[q] = k-group 8 jj + q
      float sm = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) sm += (v[q][0] + v[q][1]) + (v[q][2] + v[q][3]);
      sm += __shfl_xor(sm, 1);
      sm += __shfl_xor(sm, 2);
      const float mu = sm * (1.f / 128.f);
      float qs = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[q][e] - mu; qs = fmaf(d, d, qs); }
      qs += __shfl_xor(qs, 1);
      qs += __shfl_xor(qs, 2);
      const float rs = 1.0f / sqrtf(qs * (1.f / 128.f) + 1e-5f);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const unsigned k4 = 8u * jj + (unsigned)q;
        const f4 ga = lnt[k4], be = lnt[32 + k4];
        f4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = fmaxf((v[q][e] - mu) * rs * ga[e] + be[e], 0.f);
        *reinterpret_cast<f4*>(base + 16 * (q ^ sw)) = o;
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // (not __syncthreads: its vmcnt(0) would drain the DMA)
    }

    // accumulator init: the gathered operand (+ bias) of this tile, fetched during the previous one; without a gathered operand
    // the bias vector is the C input of the tile's first MFMAs (no vector-ALU instruction at all in the steady state)
    f16v acc[2];
    if constexpr (NADD >= 1) {
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = g1[b][r] + bias;
    } else {
      acc[0] = biasv;
      acc[1] = biasv;
    }

    if constexpr (K2 > 0) {
#pragma unroll
      for (int s2 = 0; s2 < 10; ++s2) {
        acc[0] = mfma32(xc[0][s2], W2r[s2], acc[0]);
        acc[1] = mfma32(xc[1][s2], W2r[s2], acc[1]);
      }
    }
    if constexpr (K1 > 0) {
      // operands one k-group ahead of their MFMAs (the compiler otherwise reads them right in front of the use)
      auto frag = [&](int c, int b) {
        return *reinterpret_cast<const f4*>(st_lds + rd[c & 3] + (stage * ST_STAGE + b * 16384u + (c >> 2) * 1024u));
      };
      f4 aA0 = frag(0, 0), aA1 = frag(0, 1), aB0, aB1;
#pragma unroll
      for (int c = 0; c < 16; c += 2) {
        if (c == 12) {                            // last quarter: the next tile's gathered operand goes out
          if constexpr (NADD == 1) gather(descA1, ld1b, ix1, g1);
          if constexpr (NADD == 2) gather_plain(tile_row0(next), g1);
        }
        aB0 = frag(c + 1, 0); aB1 = frag(c + 1, 1);
        __builtin_amdgcn_sched_barrier(0);        // the reads of chunk c+1 stay in front of the MFMAs of chunk c
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc[0] = mfma32(aA0[e], Wr[4 * c + e], acc[0]);
          acc[1] = mfma32(aA1[e], Wr[4 * c + e], acc[1]);
        }
        if (c + 2 < 16) { aA0 = frag(c + 2, 0); aA1 = frag(c + 2, 1); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc[0] = mfma32(aB0[e], Wr[4 * c + 4 + e], acc[0]);
          acc[1] = mfma32(aB1[e], Wr[4 * c + 4 + e], acc[1]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    } else {
      if constexpr (NADD == 1) gather(descA1, ld1b, ix1, g1);
      if constexpr (NADD == 2) gather_plain(tile_row0(next), g1);
    }

    const unsigned row0 = tile_row0(tile);
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const unsigned soff = (row0 + 32u * b + 8u * (r >> 2)) * ldyb + (unsigned)colw * 4u;
        float v = acc[b][r];              // (a named float: __builtin_bit_cast straight from the vector element read element 0)
        if constexpr (LN) v *= p.out_scale;
        if constexpr (SSP) v = ssp(v);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), descY, voffY[r & 3], soff, 0);
      }
    tile = next;
  };

  body(std::integral_constant<unsigned, 0>{}, std::true_type{});
  while ((int)tile < n_tiles) {
    body(std::integral_constant<unsigned, 1>{}, std::false_type{});
    if ((int)tile >= n_tiles) break;
    body(std::integral_constant<unsigned, 0>{}, std::false_type{});
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the trailing (out-of-range) DMA must not outlive the workgroup's LDS
}

template <int NW, int NADD, int K1, int K2, bool LN = false, bool SSP = false>
static int launch_stream_t(const PgGemm* p, hipStream_t st) {
  const void* k = reinterpret_cast<const void*>(gemm_stream_kernel<NW, NADD, K1, K2, LN, SSP>);
  const size_t lds = K1 ? 2 * ST_STAGE + (LN ? 1024 : 0) : 0;
  if (lds) if (int rc = reserve_lds(k, lds, "pg_gemm(stream)")) return rc;
  const int n_tiles = (p->M + ST_BM - 1) / ST_BM;
  const int n_cb = p->N / (NW * 32);
  int per_cb = (NW == 4 ? 2 : 1) * kNumCU / n_cb;
  if (per_cb < 1) per_cb = 1;
  if (per_cb > n_tiles) per_cb = n_tiles;
  hipLaunchKernelGGL((gemm_stream_kernel<NW, NADD, K1, K2, LN, SSP>), dim3(per_cb, n_cb), dim3(NW * 64), lds, st, *p, n_tiles);
  return check_launch("pg_gemm(stream)");
}

// eligible: K = 128 from X (optionally + 20 from X2), or K = 20 alone; no row subset / activation; N a multiple of 128; at most
// one added operand (rows add1[idx1[r]] with the operand's row count, or rows add1[r]); 16-byte aligned rows; everything
// addressable with 32-bit byte offsets; at least one full tile of rows.  LayerNorm-on-load: K = 128, N = 128, no added operand
// (the second layer of the query MLPs); out_scale only there.  Shifted softplus: K = 128, N = 128, bias only (the heads)
bool gemm_stream_eligible(const PgGemm* p) {
  const bool k128 = p->K1 == 128 && (p->K2 == 0 || p->K2 == 20), k20 = p->K1 == 20 && p->K2 == 0;
  if (!(k128 || k20) || p->rows || (p->N & 127) || p->M < ST_BM) return false;
  if (p->act == 1) { if (p->K1 != 128 || p->K2 || p->N != 128 || p->ln_gamma || p->add1 || p->add2 || p->out_scale != 1.0f) return false; }
  else if (p->act != 0) return false;
  if (p->ln_gamma) { if (p->K1 != 128 || p->K2 || p->N != 128 || p->add1 || p->add2) return false; }
  else if (p->out_scale != 1.0f) return false;
  if ((p->ldx & 3) || ((size_t)p->X & 15) || (p->ldw & 3) || ((size_t)p->W & 15)) return false;     // 16-byte loads of X rows and W rows
  if (p->K2 && ((p->ldx2 & 1) || ((size_t)p->X2 & 7))) return false;
  if (k20 && (p->ldx & 1)) return false;
  if (p->add2) return false;
  if (p->add1 && p->idx1 && (p->add_rows <= 0 || (size_t)p->add_rows * p->ld_add1 * 4 >= 0xfffff000ull)) return false;
  if (p->add1 && !p->idx1 && ((size_t)p->M * p->ld_add1 * 4 >= 0xfffff000ull || p->add1 == p->Y)) return false;   // (in place: the last tile recomputes rows)
  if ((size_t)p->M * p->ldx * 4 >= 0xfffff000ull || (size_t)p->M * p->ldy * 4 >= 0xfffff000ull) return false;
  return true;
}

template <int K1, int K2>
static int launch_stream_k(const PgGemm* p, hipStream_t st) {
  const bool wide = (p->N & 255) == 0;           // 8 waves share one A tile for 256 columns
  const int nadd = p->add1 ? (p->idx1 ? 1 : 2) : 0;
  if (wide) {
    if (nadd == 0) return launch_stream_t<8, 0, K1, K2>(p, st);
    if (nadd == 1) return launch_stream_t<8, 1, K1, K2>(p, st);
    return launch_stream_t<8, 2, K1, K2>(p, st);
  }
  if (nadd == 0) return launch_stream_t<4, 0, K1, K2>(p, st);
  if (nadd == 1) return launch_stream_t<4, 1, K1, K2>(p, st);
  return launch_stream_t<4, 2, K1, K2>(p, st);
}

int launch_gemm_stream(const PgGemm* p, hipStream_t st) {
  if (p->ln_gamma) return launch_stream_t<4, 0, 128, 0, true>(p, st);
  if (p->act == 1) return launch_stream_t<4, 0, 128, 0, false, true>(p, st);
  if (p->K1 == 20) {                             // K = 20 alone: the X operand takes the X2 (register) path
    PgGemm q = *p;
    q.X2 = p->X; q.ldx2 = p->ldx; q.K2 = 20; q.K1 = 0;
    return launch_stream_k<0, 20>(&q, st);
  }
  return p->K2 ? launch_stream_k<128, 20>(p, st) : launch_stream_k<128, 0>(p, st);
}

}  // namespace pg
