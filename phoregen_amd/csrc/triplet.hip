// Bond-triplet attention (BondUpdateLayer, models/uni_denoiser.py:101-165) — the dominant kernel of a step.
//
// Occupancy-first restructuring of seg_attn.hip's TRIPLET mode:
//   * 768-thread persistent workgroups (12 waves, 3 per SIMD, <= 168 VGPRs) share one LDS copy of the lane-fixed
//     second-layer weights (2 x 64 KB), so MFMA phases of one wave overlap VALU / memory phases of its SIMD partners;
//   * two passes over the row tiles of a segment (K path -> logits for all rows -> exact softmax -> V path),
//     so the folded-key operand U (32 regs) and the value accumulator S^T (32 regs) are never live together;
//   * the per-segment constant Q = Wg2 . smear(d_ji) lives in a per-wave LDS scratch and is read at use;
//   * sin/cos of the angular code by a 2-constant Cody-Waite reduction + degree-9/8 polynomials (arguments are
//     bounded by 3*pi), row sums inside a 16-lane row by DPP adds instead of LDS-crossbar shuffles;
//   * cost-balanced static chunks of consecutive segments per workgroup (same source atom j -> shared P rows).
// Lane l = (g = l>>4, m = l&15); 16x16x4 maps as in seg_attn.hip.
#include <stdlib.h>

#include "common.h"
#include "../../include/phoregen_hip.h"

namespace pg {

constexpr int TRI_MAX_WAVES = 16;
constexpr float TRI_NEG = -1.0e30f;

template <int CTRL>
__device__ __forceinline__ float dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// sum over the 16 lanes of a DPP row (all lanes end up with the total)
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dpp<0x141>(v);   // row_half_mirror
  v += dpp<0x140>(v);   // row_mirror
  return v;
}

// sin(w*theta) or cos(w*theta) for 0 <= arg <= ~10: k = rint(arg * 2/pi), r = arg - k*pi/2 (two constants),
// sin/cos polynomials on [-pi/4, pi/4], quadrant select; cos(x) = sin-quadrant shifted by one (exact).
__device__ __forceinline__ float sincos_sel(float arg, bool want_cos) {
  const float kf = rintf(arg * 0.63661977236758134308f);
  float r = fmaf(-kf, 1.57079637050628662109375f, arg);
  r = fmaf(-kf, -4.37113900018624283e-8f, r);
  const int q = ((int)kf + (want_cos ? 1 : 0)) & 3;
  const float s = r * r;
  float ps = fmaf(s, 2.7557314297e-6f, -1.9841270114e-4f);
  ps = fmaf(ps, s, 8.3333337680e-3f);
  ps = fmaf(ps, s, -1.6666667163e-1f);
  ps = fmaf(ps * s, r, r);
  float pc = fmaf(s, 2.4801587642e-5f, -1.3888889225e-3f);
  pc = fmaf(pc, s, 4.1666667908e-2f);
  pc = fmaf(pc, s, -0.5f);
  pc = fmaf(pc, s, 1.0f);
  const float v = (q & 1) ? pc : ps;
  return (q & 2) ? -v : v;
}

__device__ __constant__ const float kTriFreq[12] = {0.f, 1.f, 2.f, 3.f, 0.5f, (float)(1.0 / 3.0), 1.f, 2.f, 3.f, 0.5f,
                                                     (float)(1.0 / 3.0), 0.f};

template <int TRI_THREADS, int TRI_MAX_TILES>
__global__ __launch_bounds__(TRI_THREADS) void triplet_kernel(PgTopo t, PgSegAttn p PG_ABL_PARAM) {
  constexpr int TRI_WAVES = TRI_THREADS / 64;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const ln = lds;                      // gk bk gv bv
  float* const wf_k = ln + 512;               // [3][8][64]
  float* const wf_v = wf_k + 1536;
  float* const w2k_l = wf_v + 1536;           // [64][64][4]
  float* const w2v_l = w2k_l + 16384;
  float* const b2v = w2v_l + 16384;           // [128]
  float* const scratch = b2v + 128;           // [16 waves][256]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, m = lane & 15;

  for (int i = tid; i < 128; i += TRI_THREADS) {
    ln[i] = p.ln_gk[i]; ln[128 + i] = p.ln_bk[i]; ln[256 + i] = p.ln_gv[i]; ln[384 + i] = p.ln_bv[i];
    b2v[i] = p.b2v[i];
  }
  for (int i = tid; i < 1536; i += TRI_THREADS) { wf_k[i] = p.Wf_k[i]; wf_v[i] = p.Wf_v[i]; }
  for (int i = tid; i < 4096; i += TRI_THREADS) {
    reinterpret_cast<f4*>(w2k_l)[i] = reinterpret_cast<const f4*>(p.W2k_l)[i];
    reinterpret_cast<f4*>(w2v_l)[i] = reinterpret_cast<const f4*>(p.W2v_l)[i];
  }
  __syncthreads();
  const float *gk = ln, *bk = ln + 128, *gv = ln + 256, *bv = ln + 384;
  float* const sc = scratch + wave * 256;

  int s_begin, s_end;
  if (p.seg_chunks) {
    s_begin = p.seg_chunks[blockIdx.x];
    s_end = p.seg_chunks[blockIdx.x + 1];
  } else {
    const int per = (p.n_seg + gridDim.x - 1) / gridDim.x;
    s_begin = blockIdx.x * per;
    s_end = min(p.n_seg, s_begin + per);
  }

  // segment = bond edge j->i, visited in source-atom order (neighbours in time share the rows P[.->j]); its 16-byte
  // descriptor is fetched one segment ahead so that no dependent index chain sits in front of a segment
  const int4* desc = reinterpret_cast<const int4*>(t.bond_desc);
  int seg_next = 0;
  int4 d_next = {0, 0, 0, 0};
  if (s_begin + wave < s_end) {
    seg_next = p.seg_ids ? p.seg_ids[s_begin + wave] : s_begin + wave;
    d_next = desc[seg_next];
  }
  for (int si = s_begin + wave; si < s_end; si += TRI_WAVES) {
    const int seg = seg_next;
    const int4 d = d_next;
    if (si + TRI_WAVES < s_end) {
      seg_next = p.seg_ids ? p.seg_ids[si + TRI_WAVES] : si + TRI_WAVES;
      d_next = desc[seg_next];
    }
    const int cj = d.x, li = d.y & 0xffff, lj = d.y >> 16, n = d.z;
    const int lig0 = cj - lj, ci = lig0 + li;
    const int* eid_g = t.eid + d.w;
    const int n_tiles = (n + 15) >> 4;

    // ---- Q = Wg2 . smear(d_ji) into the wave's scratch: [0:128] key MLP, [128:256] value MLP ----
    {
      float qk0 = 0.f, qk1 = 0.f, qv0 = 0.f, qv1 = 0.f;
      const float* Gs = p.G + (size_t)seg * 20;
#pragma unroll 5
      for (int i = 0; i < (PG_ABL(16) ? 0 : 20); ++i) {
        const float gv_ = Gs[i];
        qk0 = fmaf(p.Wg2_k[i * 128 + lane], gv_, qk0);
        qk1 = fmaf(p.Wg2_k[i * 128 + 64 + lane], gv_, qk1);
        qv0 = fmaf(p.Wg2_v[i * 128 + lane], gv_, qv0);
        qv1 = fmaf(p.Wg2_v[i * 128 + 64 + lane], gv_, qv1);
      }
      sc[lane] = qk0; sc[64 + lane] = qk1; sc[128 + lane] = qv0; sc[192 + lane] = qv1;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    const float xi0 = p.x[ci * 3], xi1 = p.x[ci * 3 + 1], xi2 = p.x[ci * 3 + 2];
    const float u0 = p.x[cj * 3] - xi0, u1 = p.x[cj * 3 + 1] - xi1, u2 = p.x[cj * 3 + 2] - xi2;

    float feat[TRI_MAX_TILES][3];
    f4 lg[TRI_MAX_TILES];

    // =============================== pass A: logits of every row ===============================
    {
      f4 U[8];
      if PG_ABL(2) {
#pragma unroll
        for (int tq = 0; tq < 8; ++tq) U[tq] = (f4){0.01f * lane, 0.02f, 0.03f, 0.04f};
      } else {
        const float* qp = p.q + (size_t)seg * 128 + 8 * m;
        const f4 qa = *reinterpret_cast<const f4*>(qp), qb = *reinterpret_cast<const f4*>(qp + 4);
#pragma unroll
        for (int tq = 0; tq < 8; ++tq)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = (tq * 4 + r) * 2;
            const f4 wa = *reinterpret_cast<const f4*>(w2k_l + ((size_t)i * 64 + lane) * 4);
            const f4 wb = *reinterpret_cast<const f4*>(w2k_l + ((size_t)(i + 1) * 64 + lane) * 4);
            U[tq][r] = (qa[0] * wa[0] + qa[1] * wa[1]) + (qa[2] * wa[2] + qa[3] * wa[3]) +
                       (qb[0] * wb[0] + qb[1] * wb[1]) + (qb[2] * wb[2] + qb[3] * wb[3]);
          }
      }
#pragma unroll
      for (int tile = 0; tile < TRI_MAX_TILES; ++tile) {
        lg[tile] = (f4){TRI_NEG, TRI_NEG, TRI_NEG, TRI_NEG};
        feat[tile][0] = feat[tile][1] = feat[tile][2] = 0.f;
        if (tile < n_tiles && !PG_ABL(64)) {
          const int k = tile * 16 + m;
          const bool valid = k < n && k != li && k != lj;
          const int e_kj = valid ? eid_g[k * n + lj] : 0;
          // angular features of row k for f = 4 step + g  (uni_denoiser.py:131-135, common.py:85)
          float theta = 0.f;
          if (valid && !PG_ABL(8)) {
            const int ck = lig0 + k;
            const float v0 = p.x[ck * 3] - xi0, v1 = p.x[ck * 3 + 1] - xi1, v2 = p.x[ck * 3 + 2] - xi2;
            const float a = u0 * v0 + u1 * v1 + u2 * v2;
            const float c0 = u1 * v2 - u2 * v1, c1 = u2 * v0 - u0 * v2, c2 = u0 * v1 - u1 * v0;
            theta = atan2f(sqrtf(c0 * c0 + c1 * c1 + c2 * c2), a);
          }
#pragma unroll
          for (int st = 0; st < 3; ++st) {
            const int f = 4 * st + g;
            float v = PG_ABL(8) ? 0.5f : sincos_sel(theta * kTriFreq[f], f >= 6);
            v = f == 0 ? theta : v;
            feat[tile][st] = f == 11 ? 1.0f : (valid ? v : 0.f);   // f = 11 carries the per-segment constant Q
          }
          // hidden^T[c, row] = P_k[e_kj][c] + Q_k[c] + Wf_k . feat
          f4 hid[8];
          const float* pk = p.Csrc_k + (size_t)e_kj * p.ld_csrc + 4 * g;
#pragma unroll
          for (int tq = 0; tq < 8; ++tq) {
            f4 c = {0.f, 0.f, 0.f, 0.f};
            if (valid && !PG_ABL(1)) c = *reinterpret_cast<const f4*>(pk + 16 * tq);
            hid[tq] = c;
          }
#pragma unroll
          for (int st = 0; st < 3; ++st)
#pragma unroll
            for (int tq = 0; tq < 8; ++tq) {
              float w = wf_k[(st * 8 + tq) * 64 + lane];
              if (st == 2) w = g == 3 ? sc[16 * tq + m] : w;          // feature row 11 = Q_k of this segment
              hid[tq] = mfma16(w, feat[tile][st], hid[tq]);
            }
          // LayerNorm + ReLU over c in the folded form (packing._kv_mlp): hidden is centred and sign-normalised,
          // z = ReLU(hidden + b' * sigma); the row's 1/sigma multiplies its 16 logits below
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
          f4 acc4[4];                                                   // four chains: dependent MFMAs wait ~10 extra cycles
#pragma unroll
          for (int r = 0; r < 4; ++r) acc4[r] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int tq = 0; tq < 8; ++tq) {
            const f4 bt = *reinterpret_cast<const f4*>(bk + 16 * tq + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc4[r] = mfma16(fmaxf(fmaf(bt[r], sigma, hid[tq][r]), 0.f), U[tq][r], acc4[r]);
          }
          f4 acc = (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[r] *= __shfl(rs, 4 * g + r);     // rstd of row 4g+r lives in lane m = 4g+r
          // rows of the logits layout: k = 16 tile + 4g + r
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int kr = tile * 16 + 4 * g + r;
            lg[tile][r] = (kr < n && kr != li && kr != lj) ? acc[r] : TRI_NEG;
          }
        }
      }
    }

    // =============================== softmax over all rows, per head m ===============================
    float mx = TRI_NEG;
#pragma unroll
    for (int tile = 0; tile < TRI_MAX_TILES; ++tile)
#pragma unroll
      for (int r = 0; r < 4; ++r) mx = fmaxf(mx, lg[tile][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float l = 0.f;
#pragma unroll
    for (int tile = 0; tile < TRI_MAX_TILES; ++tile)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = lg[tile][r] > 0.5f * TRI_NEG ? __builtin_amdgcn_exp2f(lg[tile][r] - mx) : 0.f;
        lg[tile][r] = e;
        l += e;
      }
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    if (p.alpha) {                       // training: the adjoint reads the softmax weights back instead of recomputing them
      float* ap = p.alpha + (size_t)seg * p.alpha_rows * 16 + m;
#pragma unroll
      for (int tile = 0; tile < TRI_MAX_TILES; ++tile)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int kr = tile * 16 + 4 * g + r;
          if (kr < p.alpha_rows) ap[kr * 16] = lg[tile][r] * inv;
        }
    }

    // =============================== pass B: S^T[c, h] = sum_rows z_v[row, c] * alpha[row, h] ===============================
    f4 sT[8];
#pragma unroll
    for (int tq = 0; tq < 8; ++tq) sT[tq] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tile = 0; tile < TRI_MAX_TILES; ++tile) {
      if (tile < n_tiles && !PG_ABL(32)) {
        f4 hv[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int kr = tile * 16 + 4 * g + r;
          const bool valid = kr < n && kr != li && kr != lj;
          const int e_kj = valid ? eid_g[kr * n + lj] : 0;
          const float* pv = p.Csrc_v + (size_t)e_kj * p.ld_csrc + m;
#pragma unroll
          for (int tq = 0; tq < 8; ++tq) hv[tq][r] = (valid && !PG_ABL(1)) ? pv[16 * tq] : 0.f;
        }
#pragma unroll
        for (int st = 0; st < 3; ++st)
#pragma unroll
          for (int tq = 0; tq < 8; ++tq) {
            float w = wf_v[(st * 8 + tq) * 64 + lane];
            if (st == 2) w = g == 3 ? sc[128 + 16 * tq + m] : w;      // feature row 11 = Q_v of this segment
            hv[tq] = mfma16(feat[tile][st], w, hv[tq]);
          }
        // folded LayerNorm + ReLU per row r over c = (tau in-lane, m across the DPP row)
        f4 q2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tq = 0; tq < 8; ++tq) q2 += hv[tq] * hv[tq];
        f4 sg, aw;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float var = row16_sum(q2[r]) * (1.f / 128.f) + 1e-5f;
          const float rsq = __builtin_amdgcn_rsqf(var);
          sg[r] = var * rsq;
          aw[r] = lg[tile][r] * rsq;                                      // alpha * rstd of the row
        }
#pragma unroll
        for (int tq = 0; tq < 8; ++tq) {
          const float bt = bv[16 * tq + m];
#pragma unroll
          for (int r = 0; r < 4; ++r) hv[tq][r] = fmaxf(fmaf(bt, sg[r], hv[tq][r]), 0.f);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)            // r outer: 8 independent accumulator chains
#pragma unroll
          for (int tq = 0; tq < 8; ++tq) sT[tq] = mfma16(hv[tq][r], aw[r], sT[tq]);
      }
    }

    if (p.S) {
      // training form: hand S (normalised) and the attention mass to pg_attn_unfold_value, like the node modes
      float* sp = p.S + (size_t)seg * 2048 + lane;
#pragma unroll
      for (int tq = 0; tq < 8; ++tq)
#pragma unroll
        for (int r = 0; r < 4; ++r) sp[(tq * 4 + r) * 64] = sT[tq][r] * inv;
      if (g == 0) p.swn[(size_t)seg * 16 + m] = l > 0.f ? 1.f : 0.f;
      __builtin_amdgcn_wave_barrier();
      continue;
    }
    // =============================== epilogue: out = resid + W2v_h . S[:,h] / l + b2v ===============================
    float part[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if PG_ABL(4) {
#pragma unroll
      for (int tq = 0; tq < 8; ++tq) part[tq] = sT[tq][0] + sT[tq][1] + sT[tq][2] + sT[tq][3];
    } else
#pragma unroll
    for (int tq = 0; tq < 8; ++tq)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = (tq * 4 + r) * 2;
        const f4 wa = *reinterpret_cast<const f4*>(w2v_l + ((size_t)i * 64 + lane) * 4);
        const f4 wb = *reinterpret_cast<const f4*>(w2v_l + ((size_t)(i + 1) * 64 + lane) * 4);
        const float sv = sT[tq][r];
        part[0] = fmaf(wa[0], sv, part[0]); part[1] = fmaf(wa[1], sv, part[1]);
        part[2] = fmaf(wa[2], sv, part[2]); part[3] = fmaf(wa[3], sv, part[3]);
        part[4] = fmaf(wb[0], sv, part[4]); part[5] = fmaf(wb[1], sv, part[5]);
        part[6] = fmaf(wb[2], sv, part[6]); part[7] = fmaf(wb[3], sv, part[7]);
      }
#pragma unroll
    for (int d = 0; d < 8; ++d) {
      part[d] += __shfl_xor(part[d], 16);
      part[d] += __shfl_xor(part[d], 32);
    }
    const float has = l > 0.f ? 1.f : 0.f;
    const int o0 = 8 * m + 2 * g;
    const float p0 = g == 0 ? part[0] : (g == 1 ? part[2] : (g == 2 ? part[4] : part[6]));
    const float p1 = g == 0 ? part[1] : (g == 1 ? part[3] : (g == 2 ? part[5] : part[7]));
    const size_t ro = (size_t)seg * 128 + o0;
    const float2 rsd = *reinterpret_cast<const float2*>(p.resid + ro);
    float2 o;
    o.x = rsd.x + p0 * inv + b2v[o0] * has;
    o.y = rsd.y + p1 * inv + b2v[o0 + 1] * has;
    *reinterpret_cast<float2*>(p.out + ro) = o;
    __builtin_amdgcn_wave_barrier();   // scratch is rewritten by the next segment
  }
}

template <int THREADS, int MAXT>
static int launch_tri(const PgTopo* t, const PgSegAttn* p, hipStream_t st) {
  const size_t lds = (512 + 2 * 1536 + 2 * 16384 + 128 + (THREADS / 64) * 256) * sizeof(float);
  if (int rc = reserve_lds(reinterpret_cast<const void*>(triplet_kernel<THREADS, MAXT>), lds, "triplet")) return rc;
  hipLaunchKernelGGL((triplet_kernel<THREADS, MAXT>), dim3(kNumCU), dim3(THREADS), lds, st, *t, *p PG_ABL_ARG("PG_TRI_ABLATE"));
  return check_launch("pg_seg_attn(triplet)");
}

// row tiles held in registers: 3 (ligands <= 48 atoms), 4 (<= 64), 5 (<= 80)
int launch_triplet(const PgTopo* t, const PgSegAttn* p, hipStream_t st) {
  // 768 threads = 3 waves per SIMD at <= 168 VGPRs (measured best of 512 / 768 / 1024)
  const int tiles = (t->max_nlig + 15) / 16;
  if (tiles <= 3) return launch_tri<768, 3>(t, p, st);
  if (tiles == 4) return launch_tri<768, 4>(t, p, st);
  return launch_tri<768, 5>(t, p, st);
}

}  // namespace pg
