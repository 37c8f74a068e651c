// fp32 GEMM for the tall K = 128 products on the bf16 matrix pipe:  Y = X . W^T + bias + add1[idx1]  (same contract as
// gemm_stream.hip, which this kernel replaces for K1 = 128).
//
// Why.  v_mfma_f32_32x32x2_f32 runs at the fp32 VECTOR rate (64 FLOP/clk/SIMD, MI355X_MICROARCH.md "Matrix cores") and on the
// vector datapath, so an fp32-MFMA GEMM is compute-bound at 157 TF/s and starves every other wave's vector instructions.  The
// bf16 matrix pipe is 16 x faster and separate.  Every fp32 operand splits EXACTLY into three bf16 numbers
//     x = x1 + x2 + x3,   x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2)      (8 + 8 + 8 significand bits, RNE)
// and a product x . w into nine bf16 x bf16 partial products, each exact in fp32.  The kernel accumulates the six leading ones
//     x1w1 + x1w2 + x2w1 + x2w2 + x1w3 + x3w1
// in fp32 MFMA accumulators; the three dropped ones (x2w3, x3w2, x3w3) are together below 2^-25 |x w|, i.e. below the
// rounding error of ONE fp32 multiply (2^-24 |x w|): the result is an fp32 GEMM in every respect but the summation order
// (tests: against float64 at the fp32 GEMM's tolerance, against the fp32 MFMA kernels, and the whole parity suite unchanged).
// 6 MFMAs at 1/16 the cost = 0.375 of the fp32 MFMA time; the tall products become HBM-bound.
//
// Structure (one workgroup = 4 waves = a 64-row x 128-column tile stream, persistent, two workgroups per CU):
//   * the fp32 A tile (64 x 128) arrives by LDS-DMA (buffer_load_dwordx4 ... lds) in a row-major 32 KB stage, one tile ahead;
//   * SPLIT phase: thread t converts the 16-byte quads t, t + 256, ... of the stage (conflict-free reads; optional
//     LayerNorm + ReLU on the rows first, a row = 32 consecutive lanes) and writes three bf16 planes (48 KB) whose 16-byte
//     slots are XOR-swizzled by the row so that the MFMA operand reads (ds_read_b128 = 8 consecutive k) are conflict-free;
//   * MFMA phase: v_mfma_f32_32x32x16_bf16, the wave's 32 columns of W pre-split in 96 registers for the whole kernel;
//     the bond-length smearing columns of the [h_bond | G] product (K2 = 20) stay on the fp32 MFMA from registers;
//   * the gathered operand of a tile is fetched during its MFMAs and added in the epilogue; stores from the accumulators;
//     every tile is a full tile (the last one is anchored at row M - 64), a row's bits do not depend on its position.
// v_mfma_f32_32x32x16_bf16: lane (r = l & 31, h = l >> 5) holds A[row r][k = 8h + j], B[k = 8h + j][col r], j = 0..7;
// D reg e = row 8 (e >> 2) + 4 h + (e & 3), col r.
#include <type_traits>

#include "common.h"
#include "../../include/phoregen_hip.h"

namespace pg {

typedef int e_i4 __attribute__((ext_vector_type(4)));
typedef int e_i2 __attribute__((ext_vector_type(2)));
typedef __bf16 e_bf8 __attribute__((ext_vector_type(8)));

constexpr int EM_BM = 64;
constexpr unsigned EM_STAGE = EM_BM * 128 * 4;      // 32 KB fp32 stage
constexpr unsigned EM_PLANE = EM_BM * 128 * 2;      // 16 KB per bf16 plane
constexpr unsigned EM_LDS = EM_STAGE + 3 * EM_PLANE;

__device__ __forceinline__ e_i4 em_desc(const void* base, unsigned bytes) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(base);
  e_i4 d;
  d[0] = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu));
  d[1] = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu));
  d[2] = __builtin_amdgcn_readfirstlane((int)bytes);
  d[3] = 0x00020000;
  return d;
}
__device__ __forceinline__ void em_dma(unsigned lds_dst, unsigned voff, e_i4 desc, unsigned soff) {
  asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :: "s"(lds_dst), "v"(voff), "s"(desc), "s"(soff) : "memory");
}
__device__ __forceinline__ unsigned em_cvt_pk(float lo, float hi) {       // {bf16(lo), bf16(hi)}, round to nearest even
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
// exact three-way split of two floats: p1 + p2 + p3 == (a, b) as {bf16(a), bf16(b)} pairs
__device__ __forceinline__ void em_split2(float a, float b, unsigned& p1, unsigned& p2, unsigned& p3) {
  p1 = em_cvt_pk(a, b);
  const float ra = a - __builtin_bit_cast(float, p1 << 16), rb = b - __builtin_bit_cast(float, p1 & 0xffff0000u);
  p2 = em_cvt_pk(ra, rb);
  const float sa = ra - __builtin_bit_cast(float, p2 << 16), sb = rb - __builtin_bit_cast(float, p2 & 0xffff0000u);
  p3 = em_cvt_pk(sa, sb);
}
__device__ __forceinline__ f16v em_mfma(e_i4 a, e_i4 b, f16v c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(e_bf8, a), __builtin_bit_cast(e_bf8, b), c, 0, 0, 0);
}

template <int NADD /* 0 | 1: rows add1[idx1[r]] | 2: rows add1[r] */, int K2 /* 0 | 20 */, bool LN, bool SSP>
__global__ __launch_bounds__(256, 2) void gemm_emu_kernel(PgGemm p, int n_tiles) {
  extern __shared__ __attribute__((aligned(1024))) char em_lds[];     // the ONLY LDS object: fp32 stage | plane 1 | 2 | 3
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, kh = lane >> 5;
  const int colw = blockIdx.y * 128 + 32 * wave;                       // the wave's first output column
  const int col = colw + l31;

  // ---- the wave's W slice, split once: chunk c (k = 16c .. 16c+15), lane half kh -> k = 16c + 8kh + j ----
  e_i4 W1[8], W2[8], W3[8];
  {
    const float* wrow = p.W + (size_t)col * p.ldw + 8 * kh;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const f4 a = *reinterpret_cast<const f4*>(wrow + 16 * c), b = *reinterpret_cast<const f4*>(wrow + 16 * c + 4);
      unsigned q1, q2, q3;
      em_split2(a[0], a[1], q1, q2, q3); W1[c][0] = (int)q1; W2[c][0] = (int)q2; W3[c][0] = (int)q3;
      em_split2(a[2], a[3], q1, q2, q3); W1[c][1] = (int)q1; W2[c][1] = (int)q2; W3[c][1] = (int)q3;
      em_split2(b[0], b[1], q1, q2, q3); W1[c][2] = (int)q1; W2[c][2] = (int)q2; W3[c][2] = (int)q3;
      em_split2(b[2], b[3], q1, q2, q3); W1[c][3] = (int)q1; W2[c][3] = (int)q2; W3[c][3] = (int)q3;
    }
  }
  // second operand [X | X2] (K2 = 20, the Gaussian smearing of the bond length): fp32 MFMA from registers, as in gemm_stream.hip
  // (v_mfma_f32_32x32x2_f32: lane (l31, kh) supplies k = K1 + 10 kh + s)
  float W2r[K2 ? 10 : 1];
  if constexpr (K2 > 0) {
#pragma unroll
    for (int s2 = 0; s2 < 10; ++s2) W2r[s2] = p.W[(size_t)col * p.ldw + 128 + 10 * kh + s2];
  }
  const unsigned ldx2b = (unsigned)p.ldx2 * 4u;
  const __amdgpu_buffer_rsrc_t descX2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(K2 ? p.X2 : nullptr), 0, (unsigned)p.M * ldx2b, 0x00020000);
  const unsigned voffX2 = (unsigned)l31 * ldx2b + 40u * kh;
  float xa[2][10];
  auto load_x2 = [&](unsigned row0) {
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const unsigned so = (row0 + 32u * b) * ldx2b;
      const e_i4 v0 = __builtin_amdgcn_raw_buffer_load_b128(descX2, voffX2, so, 0);
      const e_i4 v1 = __builtin_amdgcn_raw_buffer_load_b128(descX2, voffX2 + 16u, so, 0);
      const e_i2 v2 = __builtin_amdgcn_raw_buffer_load_b64(descX2, voffX2 + 32u, so, 0);
#pragma unroll
      for (int e = 0; e < 4; ++e) { xa[b][e] = __builtin_bit_cast(float, (int)v0[e]); xa[b][4 + e] = __builtin_bit_cast(float, (int)v1[e]); }
      xa[b][8] = __builtin_bit_cast(float, (int)v2[0]);
      xa[b][9] = __builtin_bit_cast(float, (int)v2[1]);
    }
  };
  const float bias = p.bias ? p.bias[col] : 0.f;

  // ---- lane-fixed offsets ----
  const unsigned ldxb = (unsigned)p.ldx * 4u, ldyb = (unsigned)p.ldy * 4u;
  const e_i4 descX = em_desc(p.X, (unsigned)p.M * ldxb);
  const __amdgpu_buffer_rsrc_t descY = __builtin_amdgcn_make_buffer_rsrc(p.Y, 0, (unsigned)p.M * ldyb, 0x00020000);
  const unsigned voff_dma = (unsigned)kh * ldxb + (unsigned)l31 * 16u;       // a DMA piece = two rows of 512 B
  // SPLIT phase: quad s = tid + 256 i -> row (tid >> 5) + 8 i, 16-byte fp32 slot tid & 31 (k = 4 (tid & 31) ..), bf16 slot
  // (tid & 31) >> 1 half tid & 1; plane slot XOR-swizzled by row & 15 = (tid >> 5) + 8 (i & 1)
  const unsigned srow = (unsigned)tid >> 5, sslot = (unsigned)tid & 31u;
  unsigned pw[2];
#pragma unroll
  for (int o = 0; o < 2; ++o) pw[o] = ((((sslot >> 1) ^ (srow + 8u * o)) & 15u) << 4) + (sslot & 1u) * 8u;
  f4 ln_g = {1.f, 1.f, 1.f, 1.f}, ln_b = {0.f, 0.f, 0.f, 0.f};
  if constexpr (LN) {
    ln_g = *reinterpret_cast<const f4*>(p.ln_gamma + 4 * sslot);
    ln_b = *reinterpret_cast<const f4*>(p.ln_beta + 4 * sslot);
  }
  // MFMA phase: operand (row 32 b + l31, chunk c, lane half kh) = bf16 slot (2c + kh) ^ (l31 & 15) of the plane row
  unsigned rd[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) rd[c] = EM_STAGE + (unsigned)l31 * 256u + ((((unsigned)(2 * c + kh)) ^ ((unsigned)l31 & 15u)) << 4);
  unsigned voffY[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) voffY[t] = (unsigned)(4 * kh + t) * ldyb + (unsigned)l31 * 4u;

  const __amdgpu_buffer_rsrc_t descI1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(NADD ? p.idx1 : nullptr), 0, (unsigned)p.M * 4u, 0x00020000);
  const __amdgpu_buffer_rsrc_t descA1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(NADD ? p.add1 : nullptr), 0, 0xfffffff0u, 0x00020000);
  const unsigned ld1b = (unsigned)p.ld_add1 * 4u;
  const unsigned colb = (unsigned)col * 4u;
  unsigned voffA[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) voffA[t] = (unsigned)(4 * kh + t) * ld1b + colb;
  const unsigned last_row0 = (unsigned)p.M - EM_BM;
  auto tile_row0 = [&](unsigned tl) { const unsigned r = tl * EM_BM; return r < last_row0 ? r : last_row0; };

  auto dma_tile = [&](unsigned tl) {              // 32 pieces of two rows; wave w issues pieces 8w .. 8w+7
    const unsigned row0b = tile_row0(tl) * ldxb;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const unsigned piece = 8u * wave + q;
      em_dma(piece * 1024u, voff_dma, descX, row0b + 2u * piece * ldxb);
    }
  };

  const unsigned tile_step = gridDim.x;
  unsigned tile = blockIdx.x;
  dma_tile(tile);
  bool first = true;
  while ((int)tile < n_tiles) {
    const unsigned row0 = tile_row0(tile);
    const unsigned next = tile + tile_step;
    // this tile's DMA has landed (own pieces: counted wait -- the only younger vector-memory operations of the wave are the
    // previous tile's 32 stores; everybody's: barrier), and everybody has left the previous tile's MFMA phase (the planes are free)
    if (first) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(32)\n\ts_barrier" ::: "memory");
    first = false;

    // ---------------- SPLIT phase ----------------
    {
      f4 v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const f4*>(em_lds + ((unsigned)tid + 256u * i) * 16u);
      if constexpr (LN) {
        // LayerNorm(128) + ReLU on the rows (the second layer of the query MLPs): row (tid >> 5) + 8 i lives in the 32 lanes
        // of this half-wave; two-pass variance, fixed butterfly order: a row's bits do not depend on where it sits
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          float sm = (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
#pragma unroll
          for (int o = 1; o < 32; o <<= 1) sm += __shfl_xor(sm, o);
          const float mu = sm * (1.f / 128.f);
          float qs = 0.f;
#pragma unroll
          for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mu; qs = fmaf(d, d, qs); }
#pragma unroll
          for (int o = 1; o < 32; o <<= 1) qs += __shfl_xor(qs, o);
          const float rs = 1.0f / sqrtf(qs * (1.f / 128.f) + 1e-5f);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[i][e] = fmaxf((v[i][e] - mu) * rs * ln_g[e] + ln_b[e], 0.f);
        }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        unsigned a1, a2, a3, b1, b2, b3;
        em_split2(v[i][0], v[i][1], a1, a2, a3);
        em_split2(v[i][2], v[i][3], b1, b2, b3);
        char* const dst = em_lds + EM_STAGE + (srow + 8u * i) * 256u + pw[i & 1];
        *reinterpret_cast<e_i2*>(dst) = (e_i2){(int)a1, (int)b1};
        *reinterpret_cast<e_i2*>(dst + EM_PLANE) = (e_i2){(int)a2, (int)b2};
        *reinterpret_cast<e_i2*>(dst + 2 * EM_PLANE) = (e_i2){(int)a3, (int)b3};
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // planes complete, the fp32 stage is free
    // next tile into the stage (past the last tile: the clamped last tile once more, never consumed), this tile's side operands
    dma_tile(next);
    if constexpr (K2 > 0) load_x2(row0);
    f16v g1[2];
    if constexpr (NADD == 1) {
      // indices of the lane's 16 rows of block b: rows 32 b + 8 q + 4 kh + t -> one 16-byte load per (b, q); then the gathered values
      e_i4 ix[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) ix[e] = __builtin_amdgcn_raw_buffer_load_b128(descI1, 16u * kh, (row0 + 8u * e) * 4u, 0);
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int t = 0; t < 4; ++t)
            g1[b][4 * q + t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(descA1, (unsigned)ix[4 * b + q][t] * ld1b + colb, 0, 0));
    }
    if constexpr (NADD == 2) {
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          g1[b][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(descA1, voffA[r & 3], (row0 + 32u * b + 8u * (r >> 2)) * ld1b, 0));
    }

    // ---------------- MFMA phase ----------------
    f16v acc[2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
    auto frag = [&](int c, int b, int pl) {
      return *reinterpret_cast<const e_i4*>(em_lds + rd[c] + (unsigned)(b * 8192 + pl * (int)EM_PLANE));
    };
#pragma unroll
    for (int c = 0; c < 8; ++c) {
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const e_i4 a1 = frag(c, b, 0), a2 = frag(c, b, 1), a3 = frag(c, b, 2);
        // smallest partial products first
        acc[b] = em_mfma(a3, W1[c], acc[b]);
        acc[b] = em_mfma(a1, W3[c], acc[b]);
        acc[b] = em_mfma(a2, W2[c], acc[b]);
        acc[b] = em_mfma(a2, W1[c], acc[b]);
        acc[b] = em_mfma(a1, W2[c], acc[b]);
        acc[b] = em_mfma(a1, W1[c], acc[b]);
      }
    }
    if constexpr (K2 > 0) {
#pragma unroll
      for (int s2 = 0; s2 < 10; ++s2) {
        acc[0] = mfma32(xa[0][s2], W2r[s2], acc[0]);
        acc[1] = mfma32(xa[1][s2], W2r[s2], acc[1]);
      }
    }

#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const unsigned soff = (row0 + 32u * b + 8u * (r >> 2)) * ldyb + (unsigned)colw * 4u;
        float v = acc[b][r] + bias;
        if constexpr (NADD >= 1) v += g1[b][r];
        if constexpr (LN) v *= p.out_scale;
        if constexpr (SSP) v = ssp(v);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), descY, voffY[r & 3], soff, 0);
      }
    tile = next;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the trailing (never consumed) DMA must not outlive the workgroup's LDS
}

template <int NADD, int K2, bool LN = false, bool SSP = false>
static int launch_emu_t(const PgGemm* p, hipStream_t st) {
  const void* k = reinterpret_cast<const void*>(gemm_emu_kernel<NADD, K2, LN, SSP>);
  if (int rc = reserve_lds(k, EM_LDS, "pg_gemm(bf16x6)")) return rc;
  const int n_tiles = (p->M + EM_BM - 1) / EM_BM;
  const int n_cb = p->N / 128;
  int per_cb = 2 * kNumCU / n_cb;
  if (per_cb < 1) per_cb = 1;
  if (per_cb > n_tiles) per_cb = n_tiles;
  hipLaunchKernelGGL((gemm_emu_kernel<NADD, K2, LN, SSP>), dim3(per_cb, n_cb), dim3(256), EM_LDS, st, *p, n_tiles);
  return check_launch("pg_gemm(bf16x6)");
}

// K1 = 128 shapes of gemm_stream_eligible (the caller has checked that predicate): everything but K = 20 alone
bool gemm_emu_eligible(const PgGemm* p) { return p->K1 == 128; }

int launch_gemm_emu(const PgGemm* p, hipStream_t st) {
  if (p->ln_gamma) return launch_emu_t<0, 0, true>(p, st);
  if (p->act == 1) return launch_emu_t<0, 0, false, true>(p, st);
  const int nadd = p->add1 ? (p->idx1 ? 1 : 2) : 0;
  if (p->K2) {
    if (nadd == 0) return launch_emu_t<0, 20>(p, st);
    if (nadd == 1) return launch_emu_t<1, 20>(p, st);
    return launch_emu_t<2, 20>(p, st);
  }
  if (nadd == 0) return launch_emu_t<0, 0>(p, st);
  if (nadd == 1) return launch_emu_t<1, 0>(p, st);
  return launch_emu_t<2, 0>(p, st);
}

}  // namespace pg
