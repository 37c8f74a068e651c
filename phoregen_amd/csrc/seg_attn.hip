// Segment attention on gfx950: the attention sub-layers of the denoiser with the first MLP layer factored
// per node / per edge and the second key/value layers folded into the query / the aggregated value.
//
// One wave owns one segment (a target node, or a target bond edge j->i for the triplet update) and walks
// its rows in tiles of 16 with v_mfma_f32_16x16x4_f32:
//   MFMA1  hidden = Cin + Wf . feat            (K path transposed [c,row], V path [row,c])
//   VALU   z = ReLU(LayerNorm_128(hidden))     (stats: in-lane + 2 or 4 cross-lane steps)
//   MFMA2  logits[row,h] = z_k[row,:] . U[:,h]          (accumulator of MFMA1 is the A operand as it stands)
//   VALU   online softmax over rows per head
//   MFMA3  S^T[c,h] += z_v[row,c] * p[row,h]            (accumulator of MFMA1 is the A operand as it stands)
// Lane l = (g = l>>4, m = l&15).  16x16x4 maps: A[row=m][k=g], B[k=g][col=m], D reg r = D[row=4g+r][col=m].
#include <stdlib.h>

#include "common.h"
#include "../../include/phoregen_hip.h"
#include "seg_common.h"

namespace pg {

int launch_triplet(const PgTopo* t, const PgSegAttn* p, hipStream_t st);     // triplet.hip
int launch_triplet_staged(const PgTopo* t, const PgSegAttn* p, hipStream_t st);   // triplet2.hip (-1: not applicable)
int launch_node_attn(const PgTopo* t, const PgSegAttn* p, hipStream_t st);   // node_attn.hip (-1: shape not covered)
bool node_attn_fused_request(const PgSegAttn* p);                            // node_attn.hip

// Folded LayerNorm + ReLU (packing._kv_mlp: hidden is centred and sign-normalised, |gamma| lives in the next Linear):
// z = ReLU(hidden + b' * sigma); returns 1/sigma, which the caller applies to the row's logits / attention weights.
// K-path tile: hid[tau][r] = hidden[c = 16 tau + 4g + r][row = m]
__device__ __forceinline__ float ln_relu_kpath(f4 (&hid)[8], const float* bp, int g) {
  float q = 0.f;
#pragma unroll
  for (int tq = 0; tq < 8; ++tq)
#pragma unroll
    for (int r = 0; r < 4; ++r) q = fmaf(hid[tq][r], hid[tq][r], q);
  q += __shfl_xor(q, 16);
  q += __shfl_xor(q, 32);
  const float var = q * (1.f / 128.f) + 1e-5f;
  const float rs = __builtin_amdgcn_rsqf(var);
  const float sigma = var * rs;
#pragma unroll
  for (int tq = 0; tq < 8; ++tq) {
    const f4 bt = *reinterpret_cast<const f4*>(bp + 16 * tq + 4 * g);
#pragma unroll
    for (int r = 0; r < 4; ++r) hid[tq][r] = fmaxf(fmaf(bt[r], sigma, hid[tq][r]), 0.f);
  }
  return rs;
}

// V-path tile: hid[tau][r] = hidden[row = 4g + r][c = 16 tau + m]; returns 1/sigma per row r
__device__ __forceinline__ f4 ln_relu_vpath(f4 (&hid)[8], const float* bp, int m) {
  f4 q = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int tq = 0; tq < 8; ++tq) q += hid[tq] * hid[tq];
#pragma unroll
  for (int o = 1; o <= 8; o <<= 1)
#pragma unroll
    for (int r = 0; r < 4; ++r) q[r] += __shfl_xor(q[r], o);
  f4 sg, rs;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float var = q[r] * (1.f / 128.f) + 1e-5f;
    rs[r] = __builtin_amdgcn_rsqf(var);
    sg[r] = var * rs[r];
  }
#pragma unroll
  for (int tq = 0; tq < 8; ++tq) {
    const float bt = bp[16 * tq + m];
#pragma unroll
    for (int r = 0; r < 4; ++r) hid[tq][r] = fmaxf(fmaf(bt, sg[r], hid[tq][r]), 0.f);
  }
  return rs;
}

struct Lds {
  float* wf_k;    // [NSTEP][8][64]
  float* wf_v;
  float* ln;      // gk, bk, gv, bv  [4][128]
  float* w2k_l;   // triplet [16384]
  float* w2v_l;   // triplet [16384]
  float* b2v;     // triplet [128]
  float* w2xv_l;  // pos [32][64]
  float* b2xv;    // pos [16]
  float* scratch; // triplet [nwaves][256]
};

template <int MODE>
__host__ __device__ inline size_t lds_floats(int nwaves) {
  using T = ModeTraits<MODE>;
  size_t n = 2 * (size_t)T::NSTEP * 512 + 512;
  if (T::TRI) n += 16384 * 2 + 128 + (size_t)nwaves * 256;
  if (T::POS) n += 32 * 64 + 16;
  return n;
}

template <int MODE>
__global__ __launch_bounds__(256, 1) void seg_attn_kernel(PgTopo t, PgSegAttn p) {
  using T = ModeTraits<MODE>;
  constexpr int NSTEP = T::NSTEP;
  extern __shared__ __attribute__((aligned(16))) float lds_raw[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
  const int g = lane >> 4, m = lane & 15;

  // ---- carve + fill LDS ----
  Lds L;
  {
    float* q = lds_raw;
    L.ln = q; q += 512;
    L.wf_k = q; q += NSTEP * 512;
    L.wf_v = q; q += NSTEP * 512;
    L.w2k_l = q; if (T::TRI) q += 16384;
    L.w2v_l = q; if (T::TRI) q += 16384;
    L.b2v = q; if (T::TRI) q += 128;
    L.w2xv_l = q; if (T::POS) q += 2048;
    L.b2xv = q; if (T::POS) q += 16;
    L.scratch = q;
  }
  for (int i = tid; i < 128; i += blockDim.x) {
    L.ln[i] = p.ln_gk[i]; L.ln[128 + i] = p.ln_bk[i]; L.ln[256 + i] = p.ln_gv[i]; L.ln[384 + i] = p.ln_bv[i];
  }
  for (int i = tid; i < NSTEP * 512; i += blockDim.x) { L.wf_k[i] = p.Wf_k[i]; L.wf_v[i] = p.Wf_v[i]; }
  if (T::TRI && p.S == nullptr) {
    for (int i = tid; i < 16384 / 4; i += blockDim.x) {
      reinterpret_cast<f4*>(L.w2k_l)[i] = reinterpret_cast<const f4*>(p.W2k_l)[i];
      reinterpret_cast<f4*>(L.w2v_l)[i] = reinterpret_cast<const f4*>(p.W2v_l)[i];
    }
    for (int i = tid; i < 128; i += blockDim.x) L.b2v[i] = p.b2v[i];
  }
  if constexpr (T::POS) {
    for (int i = tid; i < 2048; i += blockDim.x) L.w2xv_l[i] = p.W2xv_l[i];
    for (int i = tid; i < 16; i += blockDim.x) L.b2xv[i] = p.b2xv[i];
  }
  __syncthreads();
  const float *gk = L.ln, *bk = L.ln + 128, *gv = L.ln + 256, *bv = L.ln + 384;

  // ---- chunked static schedule: consecutive segments stay in one workgroup (shared Csrc rows) ----
  const int per_blk = (p.n_seg + gridDim.x - 1) / gridDim.x;
  const int s_begin = blockIdx.x * per_blk;
  const int s_end = min(p.n_seg, s_begin + per_blk);

  for (int si = s_begin + wave; si < s_end; si += nwaves) {
    const Seg<MODE> s = setup_seg<MODE>(t, p, si);
    const int dst_ctx = T::TRI ? s.ci : s.seg;

    // ---- per-segment constants: Cdst (K path: c = 16 tau + 4g + r ; V path: c = 16 tau + m) ----
    f4 cdk[8];
    float cdv[8];
    f4 cdk2[8];  // pos modes: second K-path MLP (xv)
    // triplet, training form (p.S set): Q, U come precomputed (Cdst_*, U) and S / swn are written like the node modes
    const bool tri_ext = T::TRI && p.S != nullptr;
    if (T::TRI && !tri_ext) {
      // Q[c] = sum_i Wg2[i][c] * smear(d_ji)[i]   (first-layer columns 148:168, uni_denoiser.py:146)
      float* sc = L.scratch + wave * 256;
      float qk0 = 0.f, qk1 = 0.f, qv0 = 0.f, qv1 = 0.f;
#pragma unroll 4
      for (int i = 0; i < 20; ++i) {
        const float gi_ = p.G[(size_t)s.seg * 20 + i];
        qk0 += p.Wg2_k[i * 128 + lane] * gi_;
        qk1 += p.Wg2_k[i * 128 + 64 + lane] * gi_;
        qv0 += p.Wg2_v[i * 128 + lane] * gi_;
        qv1 += p.Wg2_v[i * 128 + 64 + lane] * gi_;
      }
      sc[lane] = qk0; sc[64 + lane] = qk1; sc[128 + lane] = qv0; sc[192 + lane] = qv1;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int tq = 0; tq < 8; ++tq) {
        cdk[tq] = *reinterpret_cast<const f4*>(sc + 16 * tq + 4 * g);
        cdv[tq] = sc[128 + 16 * tq + m];
      }
      __builtin_amdgcn_wave_barrier();
    } else {
      const float* ck = p.Cdst_k + (size_t)s.seg * p.ld_cdst;
      const float* cv = p.Cdst_v + (size_t)s.seg * p.ld_cdst;
#pragma unroll
      for (int tq = 0; tq < 8; ++tq) {
        cdk[tq] = *reinterpret_cast<const f4*>(ck + 16 * tq + 4 * g);
        if constexpr (T::POS) cdk2[tq] = *reinterpret_cast<const f4*>(cv + 16 * tq + 4 * g);
        else cdv[tq] = cv[16 * tq + m];
      }
    }

    // ---- U[tau][r] = U[c = 16 tau + 4g + r][h = m] ----
    f4 U[8];
    if (T::TRI && !tri_ext) {
      const float* qp = p.q + (size_t)s.seg * 128 + 8 * m;
      const f4 qa = *reinterpret_cast<const f4*>(qp), qb = *reinterpret_cast<const f4*>(qp + 4);
#pragma unroll
      for (int tq = 0; tq < 8; ++tq)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = (tq * 4 + r) * 2;
          const f4 wa = *reinterpret_cast<const f4*>(L.w2k_l + ((size_t)i * 64 + lane) * 4);
          const f4 wb = *reinterpret_cast<const f4*>(L.w2k_l + ((size_t)(i + 1) * 64 + lane) * 4);
          U[tq][r] = (qa[0] * wa[0] + qa[1] * wa[1]) + (qa[2] * wa[2] + qa[3] * wa[3]) +
                     (qb[0] * wb[0] + qb[1] * wb[1]) + (qb[2] * wb[2] + qb[3] * wb[3]);
        }
    } else {
      const float* up = p.U + (size_t)s.seg * 2048 + lane;
#pragma unroll
      for (int tq = 0; tq < 8; ++tq)
#pragma unroll
        for (int r = 0; r < 4; ++r) U[tq][r] = up[(tq * 4 + r) * 64];
    }

    // ---- running softmax state for head h = m over this lane's rows ----
    float m_run = NEG_BIG, l_run = 0.f, sw_run = 0.f;
    f4 sT[8];
#pragma unroll
    for (int tq = 0; tq < 8; ++tq) sT[tq] = (f4){0.f, 0.f, 0.f, 0.f};
    float acc3[3] = {0.f, 0.f, 0.f};

    float xd[3] = {0.f, 0.f, 0.f}, nd[3] = {0.f, 0.f, 0.f}, xj[3] = {0.f, 0.f, 0.f};
    if constexpr (T::KNN || T::PH || T::POS || T::TRI) {
#pragma unroll
      for (int c = 0; c < 3; ++c) xd[c] = p.x[dst_ctx * 3 + c];
    }
    if constexpr (T::KNN) {
#pragma unroll
      for (int c = 0; c < 3; ++c) nd[c] = p.nrm[dst_ctx * 3 + c];
    }
    if constexpr (T::TRI) {
#pragma unroll
      for (int c = 0; c < 3; ++c) xj[c] = p.x[s.cj * 3 + c];
    }

    const int n_tiles = (s.n_rows + 15) >> 4;
    for (int tile = 0; tile < n_tiles; ++tile) {
      // ---------- rows in the two layouts ----------
      const RowInfo rk = row_info<MODE>(t, p, s, tile * 16 + m);          // K path / feature row
      RowInfo rv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) rv[r] = row_info<MODE>(t, p, s, tile * 16 + 4 * g + r);

      // ---------- features of row rk for f = 4 step + g ----------
      float feat[NSTEP > 0 ? NSTEP : 1];
      if constexpr (T::KNN) {
        float d = 0.f, dots[3] = {0.f, 0.f, 0.f};
        bool src_lig = false;
        if (rk.valid) {
          float xs[3], ns[3];
#pragma unroll
          for (int c = 0; c < 3; ++c) { xs[c] = p.x[rk.src * 3 + c]; ns[c] = p.nrm[rk.src * 3 + c]; }
          const float r0 = xd[0] - xs[0], r1 = xd[1] - xs[1], r2 = xd[2] - xs[2];
          d = sqrtf(r0 * r0 + r1 * r1 + r2 * r2);
          // common.py:316-324: vec_1 = n[src], vec_2 = n[dst], vec_3 = x[src] - x[dst]
          dots[0] = ns[0] * nd[0] + ns[1] * nd[1] + ns[2] * nd[2];
          dots[1] = -(ns[0] * r0 + ns[1] * r1 + ns[2] * r2);
          dots[2] = -(nd[0] * r0 + nd[1] * r1 + nd[2] * r2);
          src_lig = t.ctx_is_lig[rk.src] != 0;
        }
#pragma unroll
        for (int st = 0; st < 5; ++st) {
          const float sv = rk.valid ? smear(d, 4 * st + g) : 0.f;
          feat[st] = src_lig ? sv : 0.f;
          feat[5 + st] = src_lig ? 0.f : sv;
        }
        feat[10] = g == 0 ? dots[0] : (g == 1 ? dots[1] : (g == 2 ? dots[2] : ((rk.valid && src_lig) ? 1.f : 0.f)));
        feat[11] = (g == 0 && rk.valid && !src_lig) ? 1.f : 0.f;
      } else if constexpr (T::TRI) {
        float theta = 0.f;
        if (rk.valid) {
          float u[3], v[3];
#pragma unroll
          for (int c = 0; c < 3; ++c) { u[c] = xj[c] - xd[c]; v[c] = p.x[rk.src * 3 + c] - xd[c]; }
          const float a = u[0] * v[0] + u[1] * v[1] + u[2] * v[2];
          const float c0 = u[1] * v[2] - u[2] * v[1], c1 = u[2] * v[0] - u[0] * v[2], c2 = u[0] * v[1] - u[1] * v[0];
          theta = atan2f(sqrtf(c0 * c0 + c1 * c1 + c2 * c2), a);
        }
#pragma unroll
        for (int st = 0; st < 3; ++st) {
          const int f = 4 * st + g;
          float sn, cs;
          sincosf(theta * kAngFreq[f], &sn, &cs);
          float v = f >= 6 ? cs : sn;
          v = f == 0 ? theta : v;
          feat[st] = (rk.valid && f != 11) ? v : 0.f;
        }
      } else if constexpr (T::PH) {
        float d = 0.f;
        if (rk.valid) {
          if (p.efeat) {
            const int gi = t.ctx_graph[s.seg];
            d = p.efeat[(size_t)p.efeat_off[gi] + (size_t)(rk.src - s.first) * s.n_rows + (s.seg - s.first)];
          } else {
            const float r0 = xd[0] - p.x[rk.src * 3], r1 = xd[1] - p.x[rk.src * 3 + 1], r2 = xd[2] - p.x[rk.src * 3 + 2];
            d = sqrtf(r0 * r0 + r1 * r1 + r2 * r2);
          }
        }
        feat[0] = g == 0 ? d : 0.f;
      }

      // ---------- K path: hidden^T[c, row] ----------
      f4 hid[8];
      {
        const float* pk = p.Csrc_k + (size_t)rk.csrc * p.ld_csrc + 4 * g;
#pragma unroll
        for (int tq = 0; tq < 8; ++tq) {
          f4 c = {0.f, 0.f, 0.f, 0.f};
          if (rk.valid) c = *reinterpret_cast<const f4*>(pk + 16 * tq);
          hid[tq] = c + cdk[tq];
        }
      }
#pragma unroll
      for (int st = 0; st < NSTEP; ++st)
#pragma unroll
        for (int tq = 0; tq < 8; ++tq) hid[tq] = mfma16(L.wf_k[(st * 8 + tq) * 64 + lane], feat[st], hid[tq]);
      const float rs_k = ln_relu_kpath(hid, bk, g);
      f4 lg = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int tq = 0; tq < 8; ++tq)
#pragma unroll
        for (int r = 0; r < 4; ++r) lg = mfma16(hid[tq][r], U[tq][r], lg);
#pragma unroll
      for (int r = 0; r < 4; ++r) lg[r] *= __shfl(rs_k, 4 * g + r);     // rstd of row 4g+r lives in lane m = 4g+r

      // ---------- online softmax (rows 4g + r of this tile, head m) ----------
      float tmax = NEG_BIG;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        lg[r] = rv[r].valid ? lg[r] : NEG_BIG;
        tmax = fmaxf(tmax, lg[r]);
      }
      if constexpr (T::PH) {
        // training: the rows' logits are left in p.alpha and turned into softmax weights once the segment's maximum and denominator
        // are known (below); every lane re-reads exactly what it wrote (rows 4g + r, head m)
        if (p.alpha) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = tile * 16 + 4 * g + r;
            if (row < p.alpha_rows) p.alpha[((size_t)s.seg * p.alpha_rows + row) * 16 + m] = lg[r];
          }
        }
      }
      tmax = fmaxf(tmax, __shfl_xor(tmax, 16));
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
      const float m_new = fmaxf(m_run, tmax);
      const float scale = __builtin_amdgcn_exp2f(m_run - m_new);      // base-2 softmax: queries carry log2(e)/sqrt(8)
      m_run = m_new;
      f4 pw;
      float psum = 0.f, wsum = 0.f;
      f4 gate = {1.f, 1.f, 1.f, 1.f};
      if constexpr (T::KNN) {
        if (tile * 16 + 4 * g < p.knn_k) gate = *reinterpret_cast<const f4*>(p.ew + (size_t)s.seg * p.knn_k + tile * 16 + 4 * g);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pr = rv[r].valid ? __builtin_amdgcn_exp2f(lg[r] - m_new) : 0.f;
        psum += pr;
        pw[r] = pr * gate[r];
        wsum += pw[r];
      }
      l_run = l_run * scale + psum;
      sw_run = sw_run * scale + wsum;

      if constexpr (!T::POS) {
        // ---------- V path: hidden[row, c] ----------
#pragma unroll
        for (int tq = 0; tq < 8; ++tq) sT[tq] *= scale;
        f4 hv[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float* pv = p.Csrc_v + (size_t)rv[r].csrc * p.ld_csrc + m;
#pragma unroll
          for (int tq = 0; tq < 8; ++tq) hv[tq][r] = (rv[r].valid ? pv[16 * tq] : 0.f) + cdv[tq];
        }
#pragma unroll
        for (int st = 0; st < NSTEP; ++st)
#pragma unroll
          for (int tq = 0; tq < 8; ++tq) hv[tq] = mfma16(feat[st], L.wf_v[(st * 8 + tq) * 64 + lane], hv[tq]);
        const f4 rs_v = ln_relu_vpath(hv, bv, m);
        const f4 pwr = pw * rs_v;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int tq = 0; tq < 8; ++tq) sT[tq] = mfma16(hv[tq][r], pwr[r], sT[tq]);
      } else {
        // ---------- pos modes: second K-path MLP (xv), v[row,h] = z . W2xv[h,:] + b ----------
        f4 hv[8];
        {
          const float* pk = p.Csrc_v + (size_t)rk.csrc * p.ld_csrc + 4 * g;
#pragma unroll
          for (int tq = 0; tq < 8; ++tq) {
            f4 c = {0.f, 0.f, 0.f, 0.f};
            if (rk.valid) c = *reinterpret_cast<const f4*>(pk + 16 * tq);
            hv[tq] = c + cdk2[tq];
          }
        }
#pragma unroll
        for (int st = 0; st < NSTEP; ++st)
#pragma unroll
          for (int tq = 0; tq < 8; ++tq) hv[tq] = mfma16(L.wf_v[(st * 8 + tq) * 64 + lane], feat[st], hv[tq]);
        const float rs_x = ln_relu_kpath(hv, bv, g);
        f4 vv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tq = 0; tq < 8; ++tq)
#pragma unroll
          for (int r = 0; r < 4; ++r) vv = mfma16(hv[tq][r], L.w2xv_l[(tq * 4 + r) * 64 + lane], vv);
#pragma unroll
        for (int r = 0; r < 4; ++r) vv[r] *= __shfl(rs_x, 4 * g + r);
        const float bx = L.b2xv[m];
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (rv[r].valid) {
            const float w = pw[r] * (vv[r] + bx);
            a0 += w * (xd[0] - p.x[rv[r].src * 3]);
            a1 += w * (xd[1] - p.x[rv[r].src * 3 + 1]);
            a2 += w * (xd[2] - p.x[rv[r].src * 3 + 2]);
          }
        }
        acc3[0] = acc3[0] * scale + a0;
        acc3[1] = acc3[1] * scale + a1;
        acc3[2] = acc3[2] * scale + a2;
      }
    }  // tiles

    // ---------- finish: denominators over the 4 lane groups ----------
    float l_tot = l_run + __shfl_xor(l_run, 16);
    l_tot += __shfl_xor(l_tot, 32);
    float sw_tot = sw_run + __shfl_xor(sw_run, 16);
    sw_tot += __shfl_xor(sw_tot, 32);
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;

    if constexpr (T::PH) {
      if (p.alpha) {
        for (int tile = 0; tile < n_tiles; ++tile)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = tile * 16 + 4 * g + r;
            if (row < p.alpha_rows) {
              float* ap = p.alpha + ((size_t)s.seg * p.alpha_rows + row) * 16 + m;
              const float lgv = *ap;
              *ap = lgv > 0.5f * NEG_BIG ? __builtin_amdgcn_exp2f(lgv - m_run) * inv : 0.f;
            }
          }
      }
    }
    if constexpr (T::POS) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float v = acc3[c] * inv;
        v = wave_sum(v) * (1.f / 16.f);   // sum over heads (m) and lane groups (g); mean over 16 heads
        if (lane == 0) {
          if (p.accumulate_dx) p.dx[dst_ctx * 3 + c] += v;
          else p.dx[dst_ctx * 3 + c] = v;
        }
      }
    } else if (T::TRI && !tri_ext) {
      // out[8h + d] = sum_c W2v[8h+d][c] * S[c][h] + b2v[8h+d]  (alpha sums to 1; empty segment -> 0)
      float part[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int tq = 0; tq < 8; ++tq)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = (tq * 4 + r) * 2;
          const f4 wa = *reinterpret_cast<const f4*>(L.w2v_l + ((size_t)i * 64 + lane) * 4);
          const f4 wb = *reinterpret_cast<const f4*>(L.w2v_l + ((size_t)(i + 1) * 64 + lane) * 4);
          const float sv = sT[tq][r];
          part[0] += wa[0] * sv; part[1] += wa[1] * sv; part[2] += wa[2] * sv; part[3] += wa[3] * sv;
          part[4] += wb[0] * sv; part[5] += wb[1] * sv; part[6] += wb[2] * sv; part[7] += wb[3] * sv;
        }
#pragma unroll
      for (int d = 0; d < 8; ++d) {
        part[d] += __shfl_xor(part[d], 16);
        part[d] += __shfl_xor(part[d], 32);
      }
      const float has = l_tot > 0.f ? 1.f : 0.f;
      // lane group g writes d = 2g, 2g+1
      const int o0 = 8 * m + 2 * g;
      const float p0 = g == 0 ? part[0] : (g == 1 ? part[2] : (g == 2 ? part[4] : part[6]));
      const float p1 = g == 0 ? part[1] : (g == 1 ? part[3] : (g == 2 ? part[5] : part[7]));
      const size_t ro = (size_t)s.seg * 128 + o0;
      p.out[ro] = p.resid[ro] + p0 * inv + L.b2v[o0] * has;
      p.out[ro + 1] = p.resid[ro + 1] + p1 * inv + L.b2v[o0 + 1] * has;
    } else {
      float* sp = p.S + (size_t)s.seg * 2048 + lane;
#pragma unroll
      for (int tq = 0; tq < 8; ++tq)
#pragma unroll
        for (int r = 0; r < 4; ++r) sp[(tq * 4 + r) * 64] = sT[tq][r] * inv;
      if (g == 0) p.swn[(size_t)s.seg * 16 + m] = sw_tot * inv;
    }
  }  // segments
}

// ---------------- standalone query fold / value unfold for the node modes ----------------
__global__ __launch_bounds__(256) void fold_query_kernel(const float* q, int ldq, const float* W2k_l, int n, const int* ids,
                                                         float* U) {
  extern __shared__ __attribute__((aligned(16))) float w[];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) reinterpret_cast<f4*>(w)[i] = reinterpret_cast<const f4*>(W2k_l)[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, m = lane & 15;
  for (int si = blockIdx.x * 4 + (threadIdx.x >> 6); si < n; si += gridDim.x * 4) {
    const int s = ids ? ids[si] : si;
    const float* qp = q + (size_t)s * ldq + 8 * m;
    const f4 qa = *reinterpret_cast<const f4*>(qp), qb = *reinterpret_cast<const f4*>(qp + 4);
    float* up = U + (size_t)s * 2048 + lane;
#pragma unroll 4
    for (int i = 0; i < 32; ++i) {
      const f4 wa = *reinterpret_cast<const f4*>(w + ((size_t)(2 * i) * 64 + lane) * 4);
      const f4 wb = *reinterpret_cast<const f4*>(w + ((size_t)(2 * i + 1) * 64 + lane) * 4);
      up[i * 64] = (qa[0] * wa[0] + qa[1] * wa[1]) + (qa[2] * wa[2] + qa[3] * wa[3]) +
                   (qb[0] * wb[0] + qb[1] * wb[1]) + (qb[2] * wb[2] + qb[3] * wb[3]);
    }
  }
}

__global__ __launch_bounds__(256) void unfold_value_kernel(const float* S, const float* swn, const float* W2v_l,
                                                           const float* b2v, int n, const int* ids, float* out, int ldo) {
  extern __shared__ __attribute__((aligned(16))) float w[];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) reinterpret_cast<f4*>(w)[i] = reinterpret_cast<const f4*>(W2v_l)[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, g = lane >> 4, m = lane & 15;
  for (int si = blockIdx.x * 4 + (threadIdx.x >> 6); si < n; si += gridDim.x * 4) {
    const int s = ids ? ids[si] : si;
    const float* sp = S + (size_t)s * 2048 + lane;
    float part[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // the segment's 8 KB are requested at once: with the 4 loads in flight of the partially unrolled loop the kernel ran at a
    // latency-bound 2.6 TB/s (3.8 TB/s now; two segments per table pass were tried and spill)
    float svs[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) svs[i] = sp[i * 64];
    int lz = lane;                        // (opaque: the table reads must not be hoisted out of the segment loop into 256 registers)
    asm volatile("" : "+v"(lz));
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const f4 wa = *reinterpret_cast<const f4*>(w + ((size_t)(2 * i) * 64 + lz) * 4);
      const f4 wb = *reinterpret_cast<const f4*>(w + ((size_t)(2 * i + 1) * 64 + lz) * 4);
      const float sv = svs[i];
      part[0] += wa[0] * sv; part[1] += wa[1] * sv; part[2] += wa[2] * sv; part[3] += wa[3] * sv;
      part[4] += wb[0] * sv; part[5] += wb[1] * sv; part[6] += wb[2] * sv; part[7] += wb[3] * sv;
    }
#pragma unroll
    for (int d = 0; d < 8; ++d) {
      part[d] += __shfl_xor(part[d], 16);
      part[d] += __shfl_xor(part[d], 32);
    }
    const float sw = swn ? swn[(size_t)s * 16 + m] : 0.f;
    const int o0 = 8 * m + 2 * g;
    const float p0 = g == 0 ? part[0] : (g == 1 ? part[2] : (g == 2 ? part[4] : part[6]));
    const float p1 = g == 0 ? part[1] : (g == 1 ? part[3] : (g == 2 ? part[5] : part[7]));
    out[(size_t)s * ldo + o0] = p0 + (b2v ? b2v[o0] * sw : 0.f);
    out[(size_t)s * ldo + o0 + 1] = p1 + (b2v ? b2v[o0 + 1] * sw : 0.f);
  }
}

// ---------------- MFMA lane-map self test ----------------
__global__ void selftest_kernel(int* result) {
  const int lane = threadIdx.x & 63;
  int bad = 0;
  {  // 16x16x4: A[i][k] = 1 + i + 16k, B[k][j] = 2 + 3j + 5k*k
    const float a = 1.f + (lane & 15) + 16.f * (lane >> 4);
    const float b = 2.f + 3.f * (lane & 15) + 5.f * (lane >> 4) * (lane >> 4);
    f4 c = {0.f, 0.f, 0.f, 0.f};
    c = mfma16(a, b, c);
    for (int r = 0; r < 4; ++r) {
      const int i = 4 * (lane >> 4) + r, j = lane & 15;
      float ref = 0.f;
      for (int k = 0; k < 4; ++k) ref += (1.f + i + 16.f * k) * (2.f + 3.f * j + 5.f * k * k);
      if (c[r] != ref) bad |= 1;
    }
  }
  {  // 32x32x2
    const float a = 1.f + (lane & 31) + 32.f * (lane >> 5);
    const float b = 2.f + 3.f * (lane & 31) + 7.f * (lane >> 5);
    f16v c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    c = mfma32(a, b, c);
    for (int r = 0; r < 16; ++r) {
      const int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), j = lane & 31;
      float ref = 0.f;
      for (int k = 0; k < 2; ++k) ref += (1.f + i + 32.f * k) * (2.f + 3.f * j + 7.f * k);
      if (c[r] != ref) bad |= 2;
    }
  }
  if (bad) atomicOr(result, bad);
}

template <int MODE>
static int launch_seg(const PgTopo* t, const PgSegAttn* p, hipStream_t st) {
  using T = ModeTraits<MODE>;
  const int threads = 256;
  const size_t lds = lds_floats<MODE>(threads / 64) * sizeof(float);
  if (int rc = reserve_lds(reinterpret_cast<const void*>(seg_attn_kernel<MODE>), lds, "pg_seg_attn")) return rc;
  int blocks;
  if (T::TRI) blocks = kNumCU;
  else {
    const int per = threads / 64;
    blocks = (p->n_seg + per - 1) / per;
    const int cap = lds > 16 * 1024 ? kNumCU * 3 : (1 << 20);   // LDS-light modes: exactly one segment per wave
    if (blocks > cap) blocks = cap;
  }
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(seg_attn_kernel<MODE>, dim3(blocks), dim3(threads), lds, st, *t, *p);
  return check_launch("pg_seg_attn");
}

}  // namespace pg

using namespace pg;

static int g_force_generic = 0;
namespace pg { extern int g_t2_waves; }
extern "C" int pg_debug_force_generic_seg(int on) {
  const int old = g_force_generic;
  g_force_generic = on;
  pg::g_t2_waves = (on & 4) ? 8 : 12;
  return old;
}

extern "C" int pg_seg_attn(const PgTopo* t, const PgSegAttn* p, void* stream) {
  if (!t || !p) { set_error("pg_seg_attn: null argument"); return PG_ERR_ARG; }
  if (p->n_seg == 0 && p->n_seg2 <= 0) return PG_OK;
  hipStream_t st = (hipStream_t)stream;
  if (p->n_seg2 > 0) {
    // two target lists in one call (fused knn form): one launch when the two-pass kernel takes it, else list after list
    if ((p->mode != PG_SEG_KNN_NODE && p->mode != PG_SEG_KNN_POS) || !p->seg_ids || !p->seg_ids2 || !p->Wf_k2 || !p->Wf_v2) {
      set_error("pg_seg_attn: a second target list needs a knn mode, seg_ids, seg_ids2 and Wf_k2 / Wf_v2");
      return PG_ERR_ARG;
    }
    if (p->n_seg > 0 && node_attn_fused_request(p) && !(g_force_generic & 1)) {
      const int rc = launch_node_attn(t, p, st);
      if (rc >= 0) return rc;
    }
    PgSegAttn a = *p, b = *p;
    a.n_seg2 = 0; a.seg_ids2 = nullptr;
    b.n_seg2 = 0; b.seg_ids2 = nullptr; b.seg_ids = p->seg_ids2; b.n_seg = p->n_seg2; b.Wf_k = p->Wf_k2; b.Wf_v = p->Wf_v2;
    if (int rc = pg_seg_attn(t, &a, stream)) return rc;
    return pg_seg_attn(t, &b, stream);
  }
  if (p->mode <= PG_SEG_BOND_POS && !(g_force_generic & 1)) {   // two-pass kernels; pg_debug_force_generic_seg keeps the one-pass kernel testable
    const int rc = launch_node_attn(t, p, st);
    if (rc >= 0) return rc;
  }
  if (node_attn_fused_request(p)) {
    // the fused form was asked for but the one-pass kernel runs (shape outside the two-pass kernels, or forced): fold, attend
    // and unfold as separate launches through the caller's U / S / swn scratch
    const bool pos = p->mode == PG_SEG_KNN_POS || p->mode == PG_SEG_BOND_POS;
    if (!p->U || !p->seg_ids || (!pos && (!p->S || !p->swn))) { set_error("pg_seg_attn: the fused node form needs U (and S, swn) scratch and seg_ids for the one-pass kernel"); return PG_ERR_ARG; }
    if (int rc = pg_attn_fold_query(p->q, 128, p->W2k_l, p->n_seg, p->seg_ids, const_cast<float*>(p->U), stream)) return rc;
    PgSegAttn u = *p;
    u.q = nullptr; u.W2k_l = nullptr;
    if (int rc = pg_seg_attn(t, &u, stream)) return rc;
    return pos ? PG_OK : pg_attn_unfold_value(p->S, p->swn, p->W2v_l, p->b2v, p->n_seg, p->seg_ids, p->out, 128, stream);
  }
  switch (p->mode) {
    case PG_SEG_KNN_NODE: return launch_seg<PG_SEG_KNN_NODE>(t, p, st);
    case PG_SEG_KNN_POS: return launch_seg<PG_SEG_KNN_POS>(t, p, st);
    case PG_SEG_BOND_NODE: return launch_seg<PG_SEG_BOND_NODE>(t, p, st);
    case PG_SEG_BOND_POS: return launch_seg<PG_SEG_BOND_POS>(t, p, st);
    case PG_SEG_TRIPLET:
      if (!(g_force_generic & 2)) {        // source-atom rows staged in LDS (sampling form, target-major bond order)
        const int rc = launch_triplet_staged(t, p, st);
        if (rc >= 0) return rc;
      }
      // the occupancy-tuned kernel holds the logits of <= 5 row tiles in registers (ligands of <= 80 atoms)
      // S / swn output (training): the tuned kernel when the query-side inputs are given too, else the generic form
      return (t->max_nlig <= 80 && (!p->S || (p->q && p->W2k_l && p->G))) ? launch_triplet(t, p, st)
                                                                        : launch_seg<PG_SEG_TRIPLET>(t, p, st);
    case PG_SEG_PHORE: return launch_seg<PG_SEG_PHORE>(t, p, st);
  }
  set_error("pg_seg_attn: unknown mode %d", p->mode);
  return PG_ERR_ARG;
}

extern "C" int pg_attn_fold_query(const float* q, int ldq, const float* W2k_l, int n, const int* ids, float* U, void* stream) {
  if (n == 0) return PG_OK;
  if (int rc = reserve_lds(reinterpret_cast<const void*>(fold_query_kernel), 65536, "pg_attn_fold_query")) return rc;
  int blocks = (n + 3) / 4;
  if (blocks > 2 * kNumCU) blocks = 2 * kNumCU;
  hipLaunchKernelGGL(fold_query_kernel, dim3(blocks), dim3(256), 65536, (hipStream_t)stream, q, ldq, W2k_l, n, ids, U);
  return check_launch("pg_attn_fold_query");
}

extern "C" int pg_attn_unfold_value(const float* S, const float* swn, const float* W2v_l, const float* b2v, int n,
                                    const int* ids, float* out, int ldo, void* stream) {
  if (n == 0) return PG_OK;
  if (int rc = reserve_lds(reinterpret_cast<const void*>(unfold_value_kernel), 65536, "pg_attn_unfold_value")) return rc;
  int blocks = (n + 3) / 4;
  if (blocks > 2 * kNumCU) blocks = 2 * kNumCU;
  hipLaunchKernelGGL(unfold_value_kernel, dim3(blocks), dim3(256), 65536, (hipStream_t)stream, S, swn, W2v_l, b2v, n, ids, out, ldo);
  return check_launch("pg_attn_unfold_value");
}

extern "C" int pg_selftest_mfma(int* d_result, void* stream) {
  hipError_t e = hipMemsetAsync(d_result, 0, sizeof(int), (hipStream_t)stream);
  if (e != hipSuccess) { set_error("pg_selftest_mfma: %s", hipGetErrorString(e)); return PG_ERR_HIP; }
  hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, d_result);
  return check_launch("pg_selftest_mfma");
}
