// Bond-triplet attention (BondUpdateLayer, models/uni_denoiser.py:101-165), sampling form, as TWO kernels over the same queue of
// source-atom groups (the staged form of triplet2.hip, split at the softmax):
//
//   K pass  logits of every triplet row (key MLP: factored first layer, folded LayerNorm, second layer folded into the query),
//           exact softmax over the rows of a segment, the normalised attention weights alpha[row, head] to a scratch tensor;
//   V pass  value MLP of every row, S^T[c, h] = sum_rows alpha z_v, value unfold (second value layer applied to the aggregate),
//           residual add, output row.
//
// Why split.  The one-kernel form needs U (32 registers), the transposed hidden tile (32), the value accumulator (32) and the
// per-tile state of both passes at once: 12 waves per workgroup at the 168-register cap spilled 43 registers, 8 waves had no
// spills but too little latency hiding, and both weight tables (2 x 64 KB) plus the staged rows never fitted the LDS, so the
// value unfold streamed its 64 KB through L2 for every segment.  Each pass alone
//   * fits 128 registers: 16 waves per workgroup (4 per SIMD), no scratch;
//   * needs ONE 64 KB second-layer table in LDS (W2k for the fold, W2v for the unfold): nothing streams through L2;
//   * stages only its half of the P rows (512 B per row): two stages of 80 rows fit, so the next group's rows arrive by LDS-DMA
//     (buffer_load_dwordx4 ... lds, no registers, no vector-ALU work) while the current group is being worked off: ONE barrier
//     per group instead of three, no staging phase, the queue's atomic round trip hidden behind a whole group.
// Price: alpha makes a round trip through memory (256 B per (tile, register) and segment, written and read coalesced, non-temporal)
// and the V pass recomputes the 11 angular features of a row.
//
// Lane l = (g = l>>4, m = l&15); 16x16x4 maps as in seg_attn.hip.  Per-segment arithmetic is that of triplet2.hip / triplet.hip.
#include "common.h"
#include "../../include/phoregen_hip.h"

namespace pg {

typedef float t3_f2 __attribute__((ext_vector_type(2)));
typedef int t3_i4 __attribute__((ext_vector_type(4)));

constexpr int T3_ROW = 132;          // floats per staged row: 128 + 4 (bank spread of both operand layouts)
constexpr int T3_ROWS = 80;          // staged rows per stage
constexpr int T3_XS = 320;           // floats per coordinate stage (5 DMA pieces of 64 dwords >= 3 * 96)
constexpr float T3_NEG = -1.0e30f;

template <int CTRL>
__device__ __forceinline__ float t3_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float t3_row16_sum(float v) {
  v += t3_dpp<0xB1>(v);    // quad_perm [1,0,3,2]
  v += t3_dpp<0x4E>(v);    // quad_perm [2,3,0,1]
  v += t3_dpp<0x141>(v);   // row_half_mirror
  v += t3_dpp<0x140>(v);   // row_mirror
  return v;
}
__device__ __forceinline__ float t3_from_lane(float v, int src_lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src_lane << 2, __builtin_bit_cast(int, v)));
}

// sin and cos of one argument (0 <= arg <= ~4) from ONE range reduction (same constants / polynomials as triplet.hip)
__device__ __forceinline__ void t3_sincos_pair(float arg, float& sn, float& cs) {
  const float kf = rintf(arg * 0.63661977236758134308f);
  float r = fmaf(-kf, 1.57079637050628662109375f, arg);
  r = fmaf(-kf, -4.37113900018624283e-8f, r);
  const int q = (int)kf;
  const float s = r * r;
  float ps = fmaf(s, 2.7557314297e-6f, -1.9841270114e-4f);
  ps = fmaf(ps, s, 8.3333337680e-3f);
  ps = fmaf(ps, s, -1.6666667163e-1f);
  ps = fmaf(ps * s, r, r);
  float pc = fmaf(s, 2.4801587642e-5f, -1.3888889225e-3f);
  pc = fmaf(pc, s, 4.1666667908e-2f);
  pc = fmaf(pc, s, -0.5f);
  pc = fmaf(pc, s, 1.0f);
  const float a = (q & 1) ? pc : ps, b = (q & 1) ? ps : pc;       // sin(arg) = +-a, cos(arg) = +-b
  sn = (q & 2) ? -a : a;
  cs = ((q + 1) & 2) ? -b : b;
}

// raw buffer descriptor (stride 0, byte offsets, out-of-range reads return 0)
__device__ __forceinline__ t3_i4 t3_desc(const void* base, unsigned bytes) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(base);
  t3_i4 d;
  d[0] = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu));
  d[1] = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu));
  d[2] = __builtin_amdgcn_readfirstlane((int)bytes);
  d[3] = 0x00020000;
  return d;
}
// LDS-DMA: active lanes copy 16 (4) bytes each from base + voff + soff to LDS byte address lds_dst + 16 (4) * lane
__device__ __forceinline__ void t3_dma16(unsigned lds_dst, unsigned voff, t3_i4 desc, unsigned soff) {
  asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :: "s"(lds_dst), "v"(voff), "s"(desc), "s"(soff) : "memory");
}
__device__ __forceinline__ void t3_dma4(unsigned lds_dst, unsigned voff, t3_i4 desc, unsigned soff) {
  asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds"
               :: "s"(lds_dst), "v"(voff), "s"(desc), "s"(soff) : "memory");
}

// LDS map (floats): second-layer table [64][64][4] | feature weights [3][2][64][4] | LayerNorm shift [128] | value bias [128]
//                   | coordinates 2 x T3_XS | queue words [8] | rows 2 x T3_ROWS x T3_ROW
constexpr int T3_OFF_WF = 16384, T3_OFF_B = T3_OFF_WF + 1536, T3_OFF_B2 = T3_OFF_B + 128, T3_OFF_XS = T3_OFF_B2 + 128;
constexpr int T3_OFF_CTRL = T3_OFF_XS + 2 * T3_XS, T3_OFF_ROWS = T3_OFF_CTRL + 8;
constexpr size_t t3_lds_floats() { return (size_t)T3_OFF_ROWS + 2 * (size_t)T3_ROWS * T3_ROW; }
static_assert(t3_lds_floats() * 4 <= 163840, "LDS budget");
static_assert((T3_OFF_ROWS * 4) % 16 == 0 && (T3_ROW * 4) % 16 == 0, "16-byte aligned staged rows");

template <int THREADS, int MAXT, bool VPASS>
__global__ __launch_bounds__(THREADS) void triplet3_kernel(PgTopo t, PgSegAttn p) {
  constexpr int WAVES = THREADS / 64;
  extern __shared__ __attribute__((aligned(16))) float lds[];      // the ONLY LDS object: LDS byte address = offset in `lds`
  float* const w2_l = lds;                        // K pass: lane-fixed W2k (query fold); V pass: lane-fixed W2v (value unfold)
  float* const wf = lds + T3_OFF_WF;              // feature weights of this pass, laid out for 16-byte reads
  float* const bsh = lds + T3_OFF_B;              // b' = beta/|gamma| of this pass's LayerNorm (V pass: [m][8] layout)
  float* const b2v = lds + T3_OFF_B2;             // V pass: value bias
  float* const xs0 = lds + T3_OFF_XS;
  int* const ctrl = reinterpret_cast<int*>(lds + T3_OFF_CTRL);
  float* const rows0 = lds + T3_OFF_ROWS;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, m = lane & 15;

  // queue: two entries drawn ahead (the group in work and the one whose rows are in flight)
  if (tid == 0) { ctrl[0] = atomicAdd(p.tri_counter, 1); ctrl[1] = atomicAdd(p.tri_counter, 1); }
  {
    const float* const bsrc = VPASS ? p.ln_bv : p.ln_bk;
    const float* const wsrc = VPASS ? p.Wf_v : p.Wf_k;
    for (int i = tid; i < 128; i += THREADS) {
      if (VPASS) { bsh[(i & 15) * 8 + (i >> 4)] = bsrc[i]; b2v[i] = p.b2v[i]; }
      else bsh[i] = bsrc[i];
    }
    for (int i = tid; i < 1536; i += THREADS) {             // source index i = (st * 8 + tq) * 64 + lane
      const int ln_ = i & 63, tq_ = (i >> 6) & 7, st_ = i >> 9;
      wf[((st_ * 2 + (tq_ >> 2)) * 64 + ln_) * 4 + (tq_ & 3)] = wsrc[i];
    }
    const f4* const w2src = reinterpret_cast<const f4*>(VPASS ? p.W2v_l : p.W2k_l);
    for (int i = tid; i < 4096; i += THREADS) reinterpret_cast<f4*>(w2_l)[i] = w2src[i];
  }
  const int4* const iters = reinterpret_cast<const int4*>(p.tri_iters);
  const t3_i4 descP = t3_desc(p.Csrc_k, (unsigned)t.n_bond * 1024u);
  const t3_i4 descX = t3_desc(p.x, (unsigned)t.n_ctx * 12u);

  // rows (and coordinates) of queue entry `it` -> stage `sg`, asynchronously.  One staged row = 512 B = the lower half of a wave
  auto stage_group = [&](int it, int sg) {
    const int4 d = iters[it];
    const int lig0 = __builtin_amdgcn_readfirstlane(d.x);
    const int n = __builtin_amdgcn_readfirstlane(d.y & 0xff), j0 = __builtin_amdgcn_readfirstlane((d.y >> 8) & 0xff);
    const int A = __builtin_amdgcn_readfirstlane(d.y >> 16), bond_off = __builtin_amdgcn_readfirstlane(d.z);
    const int R = A * (n - 1);
    const unsigned row0 = (unsigned)(bond_off + j0 * (n - 1));
    const unsigned dst0 = (unsigned)(T3_OFF_ROWS + sg * T3_ROWS * T3_ROW) * 4u;
    if (lane < 32) {
      for (int r = wave; r < R; r += WAVES)
        t3_dma16(dst0 + (unsigned)r * (T3_ROW * 4u), (unsigned)lane * 16u, descP, (row0 + (unsigned)r) * 1024u + (VPASS ? 512u : 0u));
    }
    if (wave == WAVES - 1) {
      const unsigned xdst = (unsigned)(T3_OFF_XS + sg * T3_XS) * 4u;
      for (int c = 0; c * 64 < n * 3; ++c)
        t3_dma4(xdst + (unsigned)c * 256u, (unsigned)lane * 4u, descX, (unsigned)lig0 * 12u + (unsigned)c * 256u);
    }
  };

  __syncthreads();
  int it_cur = __builtin_amdgcn_readfirstlane(ctrl[0]);
  if (it_cur < p.n_tri_iters) stage_group(it_cur, 0);
  int stage = 0;
  for (int round = 0;; ++round) {
    if (it_cur >= p.n_tri_iters) break;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's share of the current stage has landed
    __syncthreads();                                        // ... everybody's has; the other stage is no longer read
    const int it_nxt = __builtin_amdgcn_readfirstlane(ctrl[(round + 1) & 1]);
    if (it_nxt < p.n_tri_iters) stage_group(it_nxt, stage ^ 1);
    int it_nn = 0;
    if (tid == 0) it_nn = atomicAdd(p.tri_counter, 1);      // the entry after that: its round trip hides behind this group

    const int4 d = iters[it_cur];
    const int n = __builtin_amdgcn_readfirstlane(d.y & 0xff), j0 = __builtin_amdgcn_readfirstlane((d.y >> 8) & 0xff);
    const int A = __builtin_amdgcn_readfirstlane(d.y >> 16), bond_off = __builtin_amdgcn_readfirstlane(d.z);
    const int nm1 = n - 1;
    const int s_begin = __builtin_amdgcn_readfirstlane(d.w & 0xffff);
    const int n_seg = __builtin_amdgcn_readfirstlane((d.w >> 16) ? (d.w >> 16) : A * nm1);
    const int n_tiles = (nm1 + 15) >> 4;
    const float* const xs = xs0 + stage * T3_XS;
    const float* const rows = rows0 + stage * T3_ROWS * T3_ROW;

    for (int s = s_begin + wave; s < n_seg; s += WAVES) {
      const int a = s / nm1, ip = s - a * nm1;            // source atom of the group, target index among the other atoms
      const int j = j0 + a, i = ip + (ip >= j ? 1 : 0);
      const int seg = bond_off + i * nm1 + (j < i ? j : j - 1);          // internal id of edge j->i
      const float* const prow = rows + a * nm1 * T3_ROW;
      // opaque copies of the lane coordinates: weight reads addressed through them are not loop-invariant, so the compiler cannot
      // hoist a few hundred registers' worth of LDS weight loads out of the segment loop (and spill them)
      int lz = lane, gz = g, mz = m;
      asm volatile("" : "+v"(lz), "+v"(gz), "+v"(mz));
      float* const arow = p.tri_ws + (size_t)seg * ((size_t)p.tri_ws_tiles * 256) + lane;

      // per-segment constant Q = Wg2 . smear(d_ji) (+ the target half of the first layer) of this pass, a row of Cdst: the g == 3
      // lanes feed it to the MFMA as the weight of the constant feature 11, the other lanes carry the angular weights of step 2
      const float* const qrow = (VPASS ? p.Cdst_v : p.Cdst_k) + (size_t)seg * p.ld_cdst + m;
      float w2f[8];
#pragma unroll
      for (int tq = 0; tq < 8; ++tq) w2f[tq] = __builtin_nontemporal_load(qrow + 16 * tq);
      f4 qa, qb;
      if constexpr (!VPASS) {
        const float* qp_ = p.q + (size_t)seg * 128 + 8 * m;
        qa = __builtin_nontemporal_load(reinterpret_cast<const f4*>(qp_));
        qb = __builtin_nontemporal_load(reinterpret_cast<const f4*>(qp_ + 4));
      }
      f4 lg[VPASS ? 1 : MAXT];                  // K pass: logits of every tile; V pass: alpha of the tile in work
      f4 an = {0.f, 0.f, 0.f, 0.f};             // V pass: alpha of the next tile, fetched one tile ahead
      if constexpr (VPASS) {
#pragma unroll
        for (int r = 0; r < 4; ++r) an[r] = __builtin_nontemporal_load(arow + r * 64);      // (every segment has a first tile)
      }
#pragma unroll
      for (int tq = 0; tq < 8; ++tq)
        w2f[tq] = g == 3 ? w2f[tq] : wf[((4 + (tq >> 2)) * 64 + lz) * 4 + (tq & 3)];     // feature step 2: f = 8 + g

      const float xi0 = xs[i * 3], xi1 = xs[i * 3 + 1], xi2 = xs[i * 3 + 2];
      const float u0 = xs[j * 3] - xi0, u1 = xs[j * 3 + 1] - xi1, u2 = xs[j * 3 + 2] - xi2;
      // angle at i between j and k (uni_denoiser.py:131-135) for row m of tile g (+4 for a fifth tile): one atan2 per lane and
      // segment, the tiles then fetch theta of (tile, m) from lane 16 tile + m.  Rows past the ligand take row 0's angle
      // (finite; such rows are masked at the logits), the row k = i gives atan2(0, 0) = 0.
      float th_own[(MAXT + 3) / 4];
#pragma unroll
      for (int rep = 0; rep < (MAXT + 3) / 4; ++rep) {
        const int kp = (4 * rep + g) * 16 + m;
        const int kc = kp < nm1 ? kp : 0;
        const int k = kc + (kc >= j ? 1 : 0);
        const float v0 = xs[k * 3] - xi0, v1 = xs[k * 3 + 1] - xi1, v2 = xs[k * 3 + 2] - xi2;
        const float dt = u0 * v0 + u1 * v1 + u2 * v2;
        const float c0 = u1 * v2 - u2 * v1, c1 = u2 * v0 - u0 * v2, c2 = u0 * v1 - u1 * v0;
        th_own[rep] = atan2f(sqrtf(c0 * c0 + c1 * c1 + c2 * c2), dt);
      }
      // angular features of row (tile, m) for f = 4 step + g (common.py:85), f = 11 = the constant that carries Q.  The four
      // lanes of a row share the work: lane g evaluates sin / cos of theta, theta/2, theta/3 (one range reduction each; g = 3
      // idles), three ds_bpermutes hand round what the others need, the multiples come from sin 2t = 2 s c,
      // sin 3t = s (3 - 4 s^2), cos 2t = 1 - 2 s^2, cos 3t = c (4 c^2 - 3)
      // f = 4 st + g: [theta, sin t, sin 2t, sin 3t | sin t/2, sin t/3, cos t, cos 2t | cos 3t, cos t/2, cos t/3, 1 (Q)]
#define T3_FEATURES(tile, f0, f1, f2)                                                                                   \
      {                                                                                                                 \
        const float theta = t3_from_lane(th_own[(tile) >> 2], 16 * ((tile) & 3) + m);                                   \
        float sg_, cg_;                                                                                                 \
        t3_sincos_pair(theta * (g == 0 ? 1.0f : (g == 1 ? 0.5f : (float)(1.0 / 3.0))), sg_, cg_);                      \
        const float s1 = t3_from_lane(sg_, m), c1 = t3_from_lane(cg_, m);                                               \
        const float sx = t3_from_lane(sg_, m + (g == 0 ? 16 : 32));                                                     \
        f0 = g == 0 ? theta : (g == 1 ? s1 : (g == 2 ? 2.0f * s1 * c1 : s1 * fmaf(-4.0f * s1, s1, 3.0f)));             \
        f1 = g < 2 ? sx : (g == 2 ? c1 : fmaf(-2.0f * s1, s1, 1.0f));                                                   \
        f2 = g == 0 ? c1 * fmaf(4.0f * c1, c1, -3.0f) : (g == 3 ? 1.0f : cg_);                                          \
      }

      if constexpr (!VPASS) {
        // =============================== K pass: logits of every row ===============================
        f4 U[8];
#pragma unroll
        for (int tq = 0; tq < 8; ++tq)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int w = (tq * 4 + r) * 2;
            const f4 wa = *reinterpret_cast<const f4*>(w2_l + ((size_t)w * 64 + lz) * 4);
            const f4 wb = *reinterpret_cast<const f4*>(w2_l + ((size_t)(w + 1) * 64 + lz) * 4);
            float u = qa[0] * wa[0];
            u = fmaf(qa[1], wa[1], u); u = fmaf(qa[2], wa[2], u); u = fmaf(qa[3], wa[3], u);
            u = fmaf(qb[0], wb[0], u); u = fmaf(qb[1], wb[1], u); u = fmaf(qb[2], wb[2], u); u = fmaf(qb[3], wb[3], u);
            U[tq][r] = u;
          }
#pragma unroll
        for (int tile = 0; tile < MAXT; ++tile) {
          lg[tile] = (f4){T3_NEG, T3_NEG, T3_NEG, T3_NEG};
          if (tile < n_tiles) {
            const int kp = tile * 16 + m;                                  // row = k-th OTHER atom of j
            const int kc = kp < nm1 ? kp : 0;                              // rows past the ligand read row 0 (finite, masked below)
            float f0, f1, f2;
            T3_FEATURES(tile, f0, f1, f2)
            // hidden^T[c, row] = P_k[row][c] + Q_k[c] + Wf_k . feat
            f4 hid[8];
            const float* pk = prow + kc * T3_ROW + 4 * g;
#pragma unroll
            for (int tq = 0; tq < 8; ++tq) hid[tq] = *reinterpret_cast<const f4*>(pk + 16 * tq);
            {
              const f4 wa = *reinterpret_cast<const f4*>(wf + (0 * 64 + lz) * 4);
              const f4 wb = *reinterpret_cast<const f4*>(wf + (1 * 64 + lz) * 4);
#pragma unroll
              for (int tq = 0; tq < 4; ++tq) hid[tq] = mfma16(wa[tq], f0, hid[tq]);
#pragma unroll
              for (int tq = 0; tq < 4; ++tq) hid[4 + tq] = mfma16(wb[tq], f0, hid[4 + tq]);
            }
            {
              const f4 wa = *reinterpret_cast<const f4*>(wf + (2 * 64 + lz) * 4);
              const f4 wb = *reinterpret_cast<const f4*>(wf + (3 * 64 + lz) * 4);
#pragma unroll
              for (int tq = 0; tq < 4; ++tq) hid[tq] = mfma16(wa[tq], f1, hid[tq]);
#pragma unroll
              for (int tq = 0; tq < 4; ++tq) hid[4 + tq] = mfma16(wb[tq], f1, hid[4 + tq]);
            }
#pragma unroll
            for (int tq = 0; tq < 8; ++tq) hid[tq] = mfma16(w2f[tq], f2, hid[tq]);
            // folded LayerNorm + ReLU (packing._kv_mlp): z = ReLU(hidden + b' * sigma); 1/sigma multiplies the 16 logits
            float q2 = 0.f;
#pragma unroll
            for (int tq = 0; tq < 8; ++tq)
#pragma unroll
              for (int r = 0; r < 4; ++r) q2 = fmaf(hid[tq][r], hid[tq][r], q2);
            q2 += __shfl_xor(q2, 16);
            q2 += __shfl_xor(q2, 32);
            const float var = q2 * (1.f / 128.f) + 1e-5f;
            const float rs = __builtin_amdgcn_rsqf(var);
            const float sigma = var * rs;
            f4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int tq = 0; tq < 8; ++tq) {
              const f4 bt = *reinterpret_cast<const f4*>(bsh + 16 * tq + 4 * gz);
              acc0 = mfma16(fmaxf(fmaf(bt[0], sigma, hid[tq][0]), 0.f), U[tq][0], acc0);
              acc1 = mfma16(fmaxf(fmaf(bt[1], sigma, hid[tq][1]), 0.f), U[tq][1], acc1);
              acc0 = mfma16(fmaxf(fmaf(bt[2], sigma, hid[tq][2]), 0.f), U[tq][2], acc0);
              acc1 = mfma16(fmaxf(fmaf(bt[3], sigma, hid[tq][3]), 0.f), U[tq][3], acc1);
            }
            const f4 acc = acc0 + acc1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int kr = tile * 16 + 4 * g + r;
              const float sc_ = acc[r] * __shfl(rs, 4 * g + r);            // rstd of row 4g+r lives in lane m = 4g+r
              lg[tile][r] = (kr < nm1 && kr != ip) ? sc_ : T3_NEG;
            }
          }
        }
        // ---- exact softmax over all rows, per head m; the normalised weights leave for the V pass ----
        float mx = T3_NEG;
#pragma unroll
        for (int tile = 0; tile < MAXT; ++tile)
#pragma unroll
          for (int r = 0; r < 4; ++r) mx = fmaxf(mx, lg[tile][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float l = 0.f;
#pragma unroll
        for (int tile = 0; tile < MAXT; ++tile)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float e = lg[tile][r] > 0.5f * T3_NEG ? __builtin_amdgcn_exp2f(lg[tile][r] - mx) : 0.f;
            lg[tile][r] = e;
            l += e;
          }
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
        const float inv = l > 0.f ? 1.0f / l : 0.f;
#pragma unroll
        for (int tile = 0; tile < MAXT; ++tile)
          if (tile < n_tiles) {
#pragma unroll
            for (int r = 0; r < 4; ++r) __builtin_nontemporal_store(lg[tile][r] * inv, arow + (tile * 4 + r) * 64);
          }
      } else {
        // =============================== V pass: S^T[c, h] = sum_rows z_v[row, c] * alpha[row, h] ===============================
        f4 sT[8];
#pragma unroll
        for (int tq = 0; tq < 8; ++tq) sT[tq] = (f4){0.f, 0.f, 0.f, 0.f};
        float asum = 0.f;
#pragma unroll
        for (int tile = 0; tile < MAXT; ++tile) {
          if (tile < n_tiles) {
            lg[0] = an;
            if (tile + 1 < MAXT && tile + 1 < n_tiles) {
#pragma unroll
              for (int r = 0; r < 4; ++r) an[r] = __builtin_nontemporal_load(arow + ((tile + 1) * 4 + r) * 64);
            }
            float f0, f1, f2;
            T3_FEATURES(tile, f0, f1, f2)
            f4 hv[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int kr = tile * 16 + 4 * g + r;
              const float* pv = prow + (kr < nm1 ? kr : 0) * T3_ROW + m;     // masked rows carry alpha = 0
#pragma unroll
              for (int tq = 0; tq < 8; ++tq) hv[tq][r] = pv[16 * tq];
            }
            {
              const f4 wa = *reinterpret_cast<const f4*>(wf + (0 * 64 + lz) * 4);
              const f4 wb = *reinterpret_cast<const f4*>(wf + (1 * 64 + lz) * 4);
#pragma unroll
              for (int tq = 0; tq < 4; ++tq) hv[tq] = mfma16(f0, wa[tq], hv[tq]);
#pragma unroll
              for (int tq = 0; tq < 4; ++tq) hv[4 + tq] = mfma16(f0, wb[tq], hv[4 + tq]);
            }
            {
              const f4 wa = *reinterpret_cast<const f4*>(wf + (2 * 64 + lz) * 4);
              const f4 wb = *reinterpret_cast<const f4*>(wf + (3 * 64 + lz) * 4);
#pragma unroll
              for (int tq = 0; tq < 4; ++tq) hv[tq] = mfma16(f1, wa[tq], hv[tq]);
#pragma unroll
              for (int tq = 0; tq < 4; ++tq) hv[4 + tq] = mfma16(f1, wb[tq], hv[4 + tq]);
            }
#pragma unroll
            for (int tq = 0; tq < 8; ++tq) hv[tq] = mfma16(f2, w2f[tq], hv[tq]);
            // folded LayerNorm + ReLU per row r over c = (tau in-lane, m across the DPP row)
            f4 q2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int tq = 0; tq < 8; ++tq) q2 += hv[tq] * hv[tq];
            f4 sg, aw;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float var = t3_row16_sum(q2[r]) * (1.f / 128.f) + 1e-5f;
              const float rsq = __builtin_amdgcn_rsqf(var);
              sg[r] = var * rsq;
              aw[r] = lg[0][r] * rsq;                                         // alpha * rstd of the row
              asum += lg[0][r];
            }
#pragma unroll
            for (int tq = 0; tq < 8; ++tq) {
              const float bt = bsh[mz * 8 + tq];
#pragma unroll
              for (int r = 0; r < 4; ++r) hv[tq][r] = fmaxf(fmaf(bt, sg[r], hv[tq][r]), 0.f);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)            // r outer: 8 independent accumulator chains
#pragma unroll
              for (int tq = 0; tq < 8; ++tq) sT[tq] = mfma16(hv[tq][r], aw[r], sT[tq]);
          }
        }
        // ---- out = resid + W2v_h . S[:,h] + b2v (a segment without rows -- a 2-atom ligand -- gets the residual only) ----
        const t3_f2 rsd = __builtin_nontemporal_load(reinterpret_cast<const t3_f2*>(p.resid + (size_t)seg * 128 + 8 * m + 2 * g));
        asum += __shfl_xor(asum, 16);
        asum += __shfl_xor(asum, 32);
        float part[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < 64; ++w) {
          const int tq = w >> 3, r = (w >> 1) & 3, hf = (w & 1) * 4;
          const f4 wv_ = *reinterpret_cast<const f4*>(w2_l + ((size_t)w * 64 + lz) * 4);
          const float sv = sT[tq][r];
          part[hf + 0] = fmaf(wv_[0], sv, part[hf + 0]); part[hf + 1] = fmaf(wv_[1], sv, part[hf + 1]);
          part[hf + 2] = fmaf(wv_[2], sv, part[hf + 2]); part[hf + 3] = fmaf(wv_[3], sv, part[hf + 3]);
        }
#pragma unroll
        for (int dd = 0; dd < 8; ++dd) {
          part[dd] += __shfl_xor(part[dd], 16);
          part[dd] += __shfl_xor(part[dd], 32);
        }
        const float has = asum > 0.f ? 1.f : 0.f;
        const int o0 = 8 * m + 2 * g;
        const float p0 = g == 0 ? part[0] : (g == 1 ? part[2] : (g == 2 ? part[4] : part[6]));
        const float p1 = g == 0 ? part[1] : (g == 1 ? part[3] : (g == 2 ? part[5] : part[7]));
        const t3_f2 o = {rsd[0] + p0 + b2v[o0] * has, rsd[1] + p1 + b2v[o0 + 1] * has};
        __builtin_nontemporal_store(o, reinterpret_cast<t3_f2*>(p.out + (size_t)seg * 128 + o0));
      }
#undef T3_FEATURES
    }
    if (tid == 0) ctrl[round & 1] = it_nn;                  // read by everybody after the next barrier
    it_cur = it_nxt;
    stage ^= 1;
  }
  // the queue leaves itself ready for the next launch: the last workgroup out (all others have made their final draws before
  // they count themselves out) zeroes the head and the exit count -- no memset between the launches of a step
  if (tid == 0 && atomicAdd(p.tri_counter + 1, 1) == (int)gridDim.x - 1) {
    p.tri_counter[0] = 0;
    p.tri_counter[1] = 0;
    __threadfence();
  }
}

template <int THREADS, int MAXT>
static int launch_t3(const PgTopo* t, const PgSegAttn* p, hipStream_t st) {
  const size_t lds = t3_lds_floats() * sizeof(float);
  if (int rc = reserve_lds(reinterpret_cast<const void*>(triplet3_kernel<THREADS, MAXT, false>), lds, "pg_seg_attn(triplet, K pass)")) return rc;
  if (int rc = reserve_lds(reinterpret_cast<const void*>(triplet3_kernel<THREADS, MAXT, true>), lds, "pg_seg_attn(triplet, V pass)")) return rc;
  hipLaunchKernelGGL((triplet3_kernel<THREADS, MAXT, false>), dim3(kNumCU), dim3(THREADS), lds, st, *t, *p);
  if (int rc = check_launch("pg_seg_attn(triplet, K pass)")) return rc;
  hipLaunchKernelGGL((triplet3_kernel<THREADS, MAXT, true>), dim3(kNumCU), dim3(THREADS), lds, st, *t, *p);
  return check_launch("pg_seg_attn(triplet, V pass)");
}

// usable when the caller provides the source-atom groups (PgSegAttn.tri_iters), the alpha scratch (tri_ws), asks for the sampling
// form (out = resid + update, no S / alpha side outputs) and P is one [n_bond, 256] = [P_k | P_v] tensor; returns -1 otherwise
int launch_triplet_split(const PgTopo* t, const PgSegAttn* p, hipStream_t st) {
  if (!p->tri_iters || !p->tri_counter || p->n_tri_iters <= 0 || p->S || p->alpha || !p->out || !p->resid) return -1;
  if (!p->tri_ws || p->Csrc_v != p->Csrc_k + 128 || p->ld_csrc != 256 || ((size_t)p->Csrc_k & 15) || !p->Cdst_k || !p->Cdst_v) return -1;
  if (t->max_nlig - 1 > T3_ROWS || t->max_nlig > 96 || (unsigned long long)t->n_bond * 1024ull >= (1ull << 32)) return -1;
  const int tiles = (t->max_nlig - 1 + 15) / 16;
  if (p->tri_ws_tiles < tiles) return -1;
  if (tiles <= 3) return launch_t3<1024, 3>(t, p, st);
  if (tiles == 4) return launch_t3<1024, 4>(t, p, st);
  return launch_t3<1024, 5>(t, p, st);
}

}  // namespace pg
