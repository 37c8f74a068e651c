// Round-5 bounded question (micro-benchmark only, no product change): the two K = 128 products of ONE 16-row tile of the triplet
// kernel (csrc/triplet2.hip) -- pass A  logits[row, head] = sum_c z[row, c] U[c, head]  and pass B  S^T[c, head] = sum_row zv[row, c] a[row, head]
// -- as
//   MODE 0  64 x v_mfma_f32_16x16x4_f32 (the product kernel's form; fp32 MFMAs hold the SIMD's vector issue for all of their 32 cycles)
//   MODE 1  error-compensated 2-term f16 split on v_mfma_f32_16x16x32_f16: x = hi + lo (hi = rtz f16 of x, lo = f16 of x - hi), products
//           hi.hi + hi.lo + lo.hi accumulated in fp32 (lo.lo ~ 2^-22 dropped): 12 MFMAs per product
//   MODE 2  3-term bf16 split on v_mfma_f32_16x16x32_bf16: hi + mid + lo (8 bits each), products hh + hm + mh + hl + lh + mm: 24 MFMAs per product
// WITH the split's vector-ALU work and in the register layout the kernel's own first-layer MFMAs leave the activations in (no cross-lane
// re-layout is needed: a lane holds 32 channels of ONE row after the feature MFMAs, and the contraction order is free), at 3 waves per SIMD
// (768-thread workgroups, one per CU) like the product kernel.  The activations are re-split every tile (opaque register barrier), the
// per-segment operand U every third tile (a 3-tile segment).  Pass B's K = 32 instruction contracts the rows of TWO tiles.
// Reported per mode: wave cycles per tile (s_memtime), wall time per tile, the clock held (s_memtime / s_memrealtime), and -- `check` --
// the max error of the tile product on LayerNorm-ed random rows against a float64 product, relative to max |result|.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/split_tile.hip -o tools/micro/split_tile && tools/micro/split_tile
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

#define OPAQUE(x) asm volatile("" : "+v"(x))

__device__ __forceinline__ f4 mfma_f32(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f4 mfma_f16(h8 a, h8 b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f4 mfma_bf16(b8 a, b8 b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

// ---- 2-term f16 split of 8 floats (one K-group of an MFMA operand): hi = rtz(x), lo = f16(x - hi) ----
typedef float f2 __attribute__((ext_vector_type(2)));
template <bool RTN = false>
__device__ __forceinline__ void split_f16(const float* x, h8& hi, h8& lo) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const h2 h = __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(x[2 * p], x[2 * p + 1]));
    // residual x - hi, exact in fp32, one instruction each: the mixed-precision FMA reads hi's half straight from the packed register
    float r0, r1;
    const unsigned hp = __builtin_bit_cast(unsigned, h);
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hp), "v"(x[2 * p]));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hp), "v"(x[2 * p + 1]));
    // (RTN: the residual rounded to nearest instead of truncated -- v_cvt_pk_f16_f32 on gfx950 -- halves its error)
    const h2 l = RTN ? __builtin_convertvector((f2){r0, r1}, h2) : __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(r0, r1));
    hi[2 * p] = h[0]; hi[2 * p + 1] = h[1];
    lo[2 * p] = l[0]; lo[2 * p + 1] = l[1];
  }
}

// ---- robust 2-term f16 split: lo' = (x - hi) * 2^11 (the MFMA flushes f16 subnormals: an unscaled residual of |x| < 2^-3 would be lost);
//      products with ONE lo' factor are accumulated apart and folded in with 2^-11 ----
__device__ __forceinline__ void split_f16s(const float* x, h8& hi, h8& lo) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const h2 h = __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(x[2 * p], x[2 * p + 1]));
    float r0, r1;
    const unsigned hp = __builtin_bit_cast(unsigned, h);
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hp), "v"(x[2 * p]));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hp), "v"(x[2 * p + 1]));
    const h2 l = __builtin_convertvector((f2){r0 * 2048.0f, r1 * 2048.0f}, h2);
    hi[2 * p] = h[0]; hi[2 * p + 1] = h[1];
    lo[2 * p] = l[0]; lo[2 * p + 1] = l[1];
  }
}

// ---- 3-term bf16 split of 8 floats: hi = top 16 bits of x, mid = top 16 bits of (x - hi), lo = top 16 bits of (x - hi - mid) ----
__device__ __forceinline__ unsigned pack_top16(float a, float b) {      // (a's top half in the low 16 bits, b's in the high): one v_perm_b32
  return __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, b), __builtin_bit_cast(unsigned, a), 0x07060302u);
}
__device__ __forceinline__ void split_bf16(const float* x, b8& hi, b8& mid, b8& lo) {
  u4 H, M, L;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float a = x[2 * p], b = x[2 * p + 1];
    const float ah = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, a) & 0xffff0000u);
    const float bh = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, b) & 0xffff0000u);
    const float a1 = a - ah, b1 = b - bh;
    const float am = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, a1) & 0xffff0000u);
    const float bm = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, b1) & 0xffff0000u);
    const float a2 = a1 - am, b2 = b1 - bm;
    H[p] = pack_top16(a, b); M[p] = pack_top16(a1, b1); L[p] = pack_top16(a2, b2);
  }
  hi = __builtin_bit_cast(b8, H); mid = __builtin_bit_cast(b8, M); lo = __builtin_bit_cast(b8, L);
}

// One wave = one stream of tiles.  Register-resident operands: z[32] (this tile's activations -- pass A: row m, channels 16 tq + 4 g + r;
// pass B: channel 16 tq + m, rows 4 g + r -- regenerated behind an opaque barrier for each pass, as the kernel's LayerNorm / ReLU would
// leave them), U[32] (query-folded key weights of the segment), aw[4] (softmax weights of the lane's four rows).
// Pass B contracts over the 16 rows of the tile: the K = 32 instruction takes the lane's 4 rows in k-slots 0..3 and zeros in 4..7 (half of
// its K is padding; pairing two tiles would need both tiles' activations live, 64 more registers in a kernel at its cap).
template <int MODE>
__global__ __launch_bounds__(768) void tile_kernel(float* out, unsigned long long* clk, int tiles) {
  const int lane = threadIdx.x & 63;
  float z[32], U[32], aw[4], useed[8];
  unsigned s = 1234567u + threadIdx.x * 2654435761u + blockIdx.x * 40503u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) * (1.0f / 16777216.0f) - 0.5f) * 0.25f; };
#pragma unroll
  for (int i = 0; i < 32; ++i) z[i] = fmaxf(rnd() * 8.f, 0.f);
#pragma unroll
  for (int i = 0; i < 8; ++i) useed[i] = rnd();
#pragma unroll
  for (int i = 0; i < 4; ++i) aw[i] = rnd() + 0.2f;
  f4 lg = {0.f, 0.f, 0.f, 0.f};
  f4 sT[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) sT[t] = (f4){0.f, 0.f, 0.f, 0.f};
  h8 Uh[4], Ul[4];
  b8 Ubh[4], Ubm[4], Ubl[4];
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < tiles; ++it) {
    if ((it % 3) == 0) {                                     // a new 3-tile segment: the segment's U arrives (one multiply per value stands
#pragma unroll                                               // for the query fold, the same in every mode) and is split
      for (int i = 0; i < 8; ++i) OPAQUE(useed[i]);
#pragma unroll
      for (int i = 0; i < 32; ++i) U[i] = useed[i & 7] * (1.0f + 0.03125f * (float)(i >> 3));
      if constexpr (MODE == 1 || MODE == 3) {
#pragma unroll
        for (int t = 0; t < 4; ++t) split_f16<MODE == 3>(U + 8 * t, Uh[t], Ul[t]);
      }
      if constexpr (MODE == 4) {
#pragma unroll
        for (int t = 0; t < 4; ++t) split_f16s(U + 8 * t, Uh[t], Ul[t]);
      }
      if constexpr (MODE == 2) {
#pragma unroll
        for (int t = 0; t < 4; ++t) split_bf16(U + 8 * t, Ubh[t], Ubm[t], Ubl[t]);
      }
    }
    // ---------------- pass A ----------------
#pragma unroll
    for (int i = 0; i < 32; ++i) OPAQUE(z[i]);
    if constexpr (MODE == 0) {
      f4 a4[4] = {lg, (f4){0.f, 0.f, 0.f, 0.f}, (f4){0.f, 0.f, 0.f, 0.f}, (f4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int tq = 0; tq < 8; ++tq)
#pragma unroll
        for (int r = 0; r < 4; ++r) a4[r] = mfma_f32(z[4 * tq + r], U[4 * tq + r], a4[r]);
      lg = (a4[0] + a4[1]) + (a4[2] + a4[3]);
    } else if constexpr (MODE == 1 || MODE == 3) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        h8 zh, zl;
        split_f16<MODE == 3>(z + 8 * t, zh, zl);
        if constexpr (MODE == 3) lg = mfma_f16(zl, Ul[t], lg);
        lg = mfma_f16(zh, Ul[t], lg);
        lg = mfma_f16(zl, Uh[t], lg);
        lg = mfma_f16(zh, Uh[t], lg);
      }
    } else if constexpr (MODE == 4) {
      f4 x4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        h8 zh, zl;
        split_f16s(z + 8 * t, zh, zl);
        x4 = mfma_f16(zh, Ul[t], x4);
        x4 = mfma_f16(zl, Uh[t], x4);
        lg = mfma_f16(zh, Uh[t], lg);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) lg[r] = __builtin_fmaf(x4[r], 1.0f / 2048.0f, lg[r]);
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        b8 zh, zm, zl;
        split_bf16(z + 8 * t, zh, zm, zl);
        lg = mfma_bf16(zh, Ubh[t], lg);
        lg = mfma_bf16(zh, Ubm[t], lg); lg = mfma_bf16(zm, Ubh[t], lg);
        lg = mfma_bf16(zh, Ubl[t], lg); lg = mfma_bf16(zl, Ubh[t], lg); lg = mfma_bf16(zm, Ubm[t], lg);
      }
    }
    // ---------------- pass B ----------------
#pragma unroll
    for (int i = 0; i < 32; ++i) OPAQUE(z[i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) OPAQUE(aw[i]);
    if constexpr (MODE == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int tq = 0; tq < 8; ++tq) sT[tq] = mfma_f32(z[4 * tq + r], aw[r], sT[tq]);
    } else if constexpr (MODE == 1 || MODE == 3) {
      const float aw8[8] = {aw[0], aw[1], aw[2], aw[3], 0.f, 0.f, 0.f, 0.f};
      h8 ah, al;
      split_f16<MODE == 3>(aw8, ah, al);
#pragma unroll
      for (int q = 0; q < 4; ++q) {              // 8 values = channel blocks 2q, 2q+1 x the lane's 4 rows
        h8 a, b;
        split_f16<MODE == 3>(z + 8 * q, a, b);
#pragma unroll
        for (int o = 0; o < 2; ++o) {
          h8 xh = {a[4 * o], a[4 * o + 1], a[4 * o + 2], a[4 * o + 3], 0, 0, 0, 0};
          h8 xl = {b[4 * o], b[4 * o + 1], b[4 * o + 2], b[4 * o + 3], 0, 0, 0, 0};
          if constexpr (MODE == 3) sT[2 * q + o] = mfma_f16(xl, al, sT[2 * q + o]);
          sT[2 * q + o] = mfma_f16(xh, al, sT[2 * q + o]);
          sT[2 * q + o] = mfma_f16(xl, ah, sT[2 * q + o]);
          sT[2 * q + o] = mfma_f16(xh, ah, sT[2 * q + o]);
        }
      }
    } else if constexpr (MODE == 4) {
      const float aw8[8] = {aw[0], aw[1], aw[2], aw[3], 0.f, 0.f, 0.f, 0.f};
      h8 ah, al;
      split_f16s(aw8, ah, al);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        h8 a, b;
        split_f16s(z + 8 * q, a, b);
#pragma unroll
        for (int o = 0; o < 2; ++o) {
          h8 xh = {a[4 * o], a[4 * o + 1], a[4 * o + 2], a[4 * o + 3], 0, 0, 0, 0};
          h8 xl = {b[4 * o], b[4 * o + 1], b[4 * o + 2], b[4 * o + 3], 0, 0, 0, 0};
          f4 x4 = mfma_f16(xh, al, (f4){0.f, 0.f, 0.f, 0.f});
          x4 = mfma_f16(xl, ah, x4);
          f4 t = mfma_f16(xh, ah, sT[2 * q + o]);
#pragma unroll
          for (int r = 0; r < 4; ++r) t[r] = __builtin_fmaf(x4[r], 1.0f / 2048.0f, t[r]);
          sT[2 * q + o] = t;
        }
      }
    } else {
      const float aw8[8] = {aw[0], aw[1], aw[2], aw[3], 0.f, 0.f, 0.f, 0.f};
      b8 ah, am, al;
      split_bf16(aw8, ah, am, al);
      const __bf16 zero = (__bf16)0.0f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        b8 a, b, c;
        split_bf16(z + 8 * q, a, b, c);
#pragma unroll
        for (int o = 0; o < 2; ++o) {
          b8 xh = {a[4 * o], a[4 * o + 1], a[4 * o + 2], a[4 * o + 3], zero, zero, zero, zero};
          b8 xm = {b[4 * o], b[4 * o + 1], b[4 * o + 2], b[4 * o + 3], zero, zero, zero, zero};
          b8 xl = {c[4 * o], c[4 * o + 1], c[4 * o + 2], c[4 * o + 3], zero, zero, zero, zero};
          f4 t = sT[2 * q + o];
          t = mfma_bf16(xh, ah, t);
          t = mfma_bf16(xh, am, t); t = mfma_bf16(xm, ah, t);
          t = mfma_bf16(xh, al, t); t = mfma_bf16(xl, ah, t); t = mfma_bf16(xm, am, t);
          sT[2 * q + o] = t;
        }
      }
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float acc = 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) acc += lg[r];
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc += sT[t][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (lane == 0) {
    const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    clk[2 * w] = c1 - c0;
    clk[2 * w + 1] = r1 - r0;
  }
}

// ---- numerics: one wave, one tile product  R[row, head] = sum_c Z[row, c] U[c, head]  (16 x 128 x 16) in the three forms ----
template <int MODE>
__global__ __launch_bounds__(64) void check_kernel(const float* Z /*[16][128]*/, const float* Um /*[128][16]*/, float* R /*[16][16]*/) {
  const int lane = threadIdx.x, g = lane >> 4, m = lane & 15;
  // the product kernel's layout: A lane (g, m): row m, channels 16 tq + 4 g + r; B lane (g, m): head m, the same channels
  float z[32], u[32];
#pragma unroll
  for (int tq = 0; tq < 8; ++tq)
#pragma unroll
    for (int r = 0; r < 4; ++r) { z[4 * tq + r] = Z[m * 128 + 16 * tq + 4 * g + r]; u[4 * tq + r] = Um[(16 * tq + 4 * g + r) * 16 + m]; }
  f4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
  if constexpr (MODE == 0) {
#pragma unroll
    for (int i = 0; i < 32; ++i) acc = mfma_f32(z[i], u[i], acc);
  } else if constexpr (MODE == 1 || MODE == 3) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      h8 zh, zl, uh, ul;
      split_f16<MODE == 3>(z + 8 * t, zh, zl);
      split_f16<MODE == 3>(u + 8 * t, uh, ul);
      if constexpr (MODE == 3) acc = mfma_f16(zl, ul, acc);
      acc = mfma_f16(zh, ul, acc);
      acc = mfma_f16(zl, uh, acc);
      acc = mfma_f16(zh, uh, acc);
    }
  } else if constexpr (MODE == 4) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      h8 zh, zl, uh, ul;
      split_f16s(z + 8 * t, zh, zl);
      split_f16s(u + 8 * t, uh, ul);
      acc2 = mfma_f16(zh, ul, acc2);
      acc2 = mfma_f16(zl, uh, acc2);
      acc = mfma_f16(zh, uh, acc);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { acc[r] = __builtin_fmaf(acc2[r], 1.0f / 2048.0f, acc[r]); acc2[r] = 0.f; }
  } else {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      b8 zh, zm, zl, uh, um, ul;
      split_bf16(z + 8 * t, zh, zm, zl);
      split_bf16(u + 8 * t, uh, um, ul);
      acc = mfma_bf16(zh, uh, acc);
      acc2 = mfma_bf16(zh, um, acc2); acc2 = mfma_bf16(zm, uh, acc2);
      acc2 = mfma_bf16(zh, ul, acc2); acc2 = mfma_bf16(zl, uh, acc2); acc2 = mfma_bf16(zm, um, acc2);
    }
  }
  // D: lane (g, m) reg r = R[row 4 g + r][col m]   (A's row index is the MFMA row, B's column index the MFMA column)
#pragma unroll
  for (int r = 0; r < 4; ++r) R[(4 * g + r) * 16 + m] = acc[r] + acc2[r];
}

static const char* kName[5] = {"fp32   64 x v_mfma_f32_16x16x4_f32", "f16x2  36 x v_mfma_f32_16x16x32_f16 + split", "bf16x3 72 x v_mfma_f32_16x16x32_bf16 + split",
                                "f16x2+ 48 x ..x32_f16: lo rounded + lo.lo term", "f16x2s 36 x ..x32_f16: residual scaled 2^11, folded"};

template <int MODE> void bench(float* out, unsigned long long* clk, int tiles) {
  const int wgs = 256, waves = wgs * 12;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 40; ++w) hipLaunchKernelGGL(tile_kernel<MODE>, dim3(wgs), dim3(768), 0, 0, out, clk, tiles);    // ~1 s of load: the clock settles
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(tile_kernel<MODE>, dim3(wgs), dim3(768), 0, 0, out, clk, tiles);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(2 * waves);
  hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> cyc(waves), ghz(waves);
  for (int w = 0; w < waves; ++w) { cyc[w] = (double)h[2 * w] / tiles; ghz[w] = (double)h[2 * w] / ((double)h[2 * w + 1] * 10.0); }   // s_memrealtime: 100 MHz
  std::sort(cyc.begin(), cyc.end()); std::sort(ghz.begin(), ghz.end());
  // wave cycles per tile are per WAVE with 3 waves sharing a SIMD: SIMD cycles per tile = wave cycles / 3
  // (three waves share a SIMD and run their tiles side by side: a SIMD finishes 3 tiles per wave-tile time)
  printf("%-46s wall %7.3f ms -> %6.1f ns per tile per SIMD = %6.0f SIMD cycles at the clock held, %.2f GHz  (s_memtime ticks per wave-tile %5.0f)\n",
         kName[MODE], ms, ms * 1e6 / tiles / 3.0, ms * 1e6 / tiles / 3.0 * ghz[waves / 2], ghz[waves / 2], cyc[waves / 2]);
}

template <int MODE> double check(const std::vector<float>& Z, const std::vector<float>& U, const std::vector<double>& ref, float* dZ, float* dU, float* dR) {
  hipLaunchKernelGGL(check_kernel<MODE>, dim3(1), dim3(64), 0, 0, dZ, dU, dR);
  std::vector<float> R(256);
  hipMemcpy(R.data(), dR, 1024, hipMemcpyDeviceToHost);
  double mx = 0, err = 0;
  for (int i = 0; i < 256; ++i) { mx = fmax(mx, fabs(ref[i])); err = fmax(err, fabs((double)R[i] - ref[i])); }
  return err / mx;
}

int main(int argc, char** argv) {
  const int tiles = argc > 1 ? atoi(argv[1]) : 6000;
  float* out; unsigned long long* clk;
  hipMalloc(&out, 256 * 768 * 4); hipMalloc(&clk, 256 * 12 * 16);
  for (int rep = 0; rep < 2; ++rep) { bench<0>(out, clk, tiles); bench<1>(out, clk, tiles); bench<3>(out, clk, tiles); bench<4>(out, clk, tiles); bench<2>(out, clk, tiles); }
  // ---- numerics on LayerNorm-ed rows: z = ReLU(LN(x) * gamma + beta), U = query-folded weights ~ N(0, 0.3) ----
  float *dZ, *dU, *dR; hipMalloc(&dZ, 16 * 128 * 4); hipMalloc(&dU, 128 * 16 * 4); hipMalloc(&dR, 1024);
  double worst[2][5] = {{0, 0, 0, 0, 0}, {0, 0, 0, 0, 0}};
  srand(12345);
  auto gauss = []() { double a = 0; for (int i = 0; i < 12; ++i) a += rand() / (double)RAND_MAX; return a - 6.0; };
  for (int trial = 0; trial < 200; ++trial) {
    std::vector<float> Z(16 * 128), U(128 * 16);
    for (int r = 0; r < 16; ++r) {
      double x[128], mu = 0, var = 0;
      for (int c = 0; c < 128; ++c) { x[c] = gauss() * (1.0 + trial % 7); mu += x[c]; }
      mu /= 128;
      for (int c = 0; c < 128; ++c) var += (x[c] - mu) * (x[c] - mu);
      const double rs = 1.0 / sqrt(var / 128 + 1e-5);
      for (int c = 0; c < 128; ++c) Z[r * 128 + c] = (float)fmax((x[c] - mu) * rs * (0.5 + (c % 5) * 0.3) + 0.1 * ((c % 3) - 1), 0.0);
    }
    const int small = trial & 1;          // odd trials: U ~ N(0, 0.02) -- the f16 residual of such values is subnormal (< 2^-14)
    for (auto& u : U) u = (float)(gauss() * (small ? 0.02 : 0.3));
    std::vector<double> ref(256, 0.0);
    for (int r = 0; r < 16; ++r) for (int h = 0; h < 16; ++h) { double a = 0; for (int c = 0; c < 128; ++c) a += (double)Z[r * 128 + c] * (double)U[c * 16 + h]; ref[r * 16 + h] = a; }
    hipMemcpy(dZ, Z.data(), Z.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dU, U.data(), U.size() * 4, hipMemcpyHostToDevice);
    worst[small][0] = fmax(worst[small][0], check<0>(Z, U, ref, dZ, dU, dR));
    worst[small][1] = fmax(worst[small][1], check<1>(Z, U, ref, dZ, dU, dR));
    worst[small][2] = fmax(worst[small][2], check<2>(Z, U, ref, dZ, dU, dR));
    worst[small][3] = fmax(worst[small][3], check<3>(Z, U, ref, dZ, dU, dR));
    worst[small][4] = fmax(worst[small][4], check<4>(Z, U, ref, dZ, dU, dR));
  }
  for (int mo = 0; mo < 5; ++mo)
    printf("%-52s max |error| / max |result| vs float64, 100 tiles of LayerNorm-ed rows each: U ~ N(0, 0.3): %.2e   U ~ N(0, 0.02): %.2e\n",
           kName[mo], worst[0][mo], worst[1][mo]);
  return 0;
}
