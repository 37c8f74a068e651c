// Backward of the segment attention sub-layers (training path, PhoreDiff.compute_loss -> models/diffusion.py:249-352;
// the forward these gradients belong to is seg_attn.hip / node_attn.hip, reference: models/uni_denoiser.py:40-72,
// 101-165, 187-209).
//
// One wave owns one segment and recomputes the forward from the same inputs (nothing but the layer inputs is kept
// from the forward pass):
//   pass 1  per 16-row tile: both MLP paths in the transposed layout hidden^T[c,row] (seg_attn.hip "K path"),
//           logits[row,h] and tv[row,h] = rstd_v * z_v . M[:,h]  (M = dS of the segment, or W2xv in the pos modes)
//           go to a per-wave row buffer; then the exact softmax, D[h] = sum_r alpha * dalpha and dlogits per row;
//   pass 2  per tile: recompute z_k, z_v, then
//             dU[c,h]  += z_k[c,row] * dy[row,h]               dz_k[c,row] = U[c,h] * dy[row,h]
//             dz_v[c,row] = M[c,h] * cw[row,h]                 (pos: dW2xv[c,h] += z_v[c,row] * rstd * dvx[row,h])
//           folded-LayerNorm backward per row, then from dhidden: dCdst (row sum), dCsrc (scatter), dWf (x feat),
//           dfeat -> geometry (positions / direction vectors), gate gradient.
// Matrix products whose operands are not in the register layout the MFMA wants go through a per-wave LDS tile
// (stride 17), so every product is a plain 16x16x4 MFMA chain with operands read where they lie.
// Weight gradients (dWf, db', dW2xv) accumulate in registers for the whole kernel and are flushed with one global atomic
// per element per wave; nothing uses LDS atomics (ds_add_f32 measured ~700 cycles per instruction here).
#include <stdlib.h>

#include "seg_common.h"

namespace pg {

#ifdef PG_BWD_PROF
__device__ unsigned long long g_bwd_prof[16];
#define PROF_DECL() long long _pacc[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; long long _t0 = 0
#define PROF_T0() _t0 = __builtin_readcyclecounter()
#define PROF(i) do { long long _t1 = __builtin_readcyclecounter(); _pacc[i] += _t1 - _t0; _t0 = _t1; } while (0)
#define PROF_FLUSH() do { if (lane == 0) for (int _i = 0; _i < 13; ++_i) atomicAdd(&g_bwd_prof[_i], (unsigned long long)_pacc[_i]); } while (0)
#else
#define PROF_DECL() do {} while (0)
#define PROF_T0() do {} while (0)
#define PROF(i) do {} while (0)
#define PROF_FLUSH() do {} while (0)
#endif

namespace {

constexpr float LN2 = 0.69314718055994530942f;
constexpr int ROWBUF = 48;   // floats per row: alpha[16] | tv[16] | dlogit[16]

template <int MODE>
struct RowGeo {
  float rel[3];   // x_dst - x_src
  float d;
  float ns[3];    // direction vector of the source (knn)
  float v[3];     // triplet: x_k - x_i
  float theta;
  bool src_lig;
};

// sin(x) or cos(x) for 0 <= x <= ~10 (angular code arguments are bounded by 3 pi): k = rint(x * 2/pi), two-constant
// Cody-Waite reduction, degree-9 / degree-8 polynomials on [-pi/4, pi/4], quadrant select (csrc/triplet.hip uses the same)
__device__ __forceinline__ float sincos_bounded(float arg, bool want_cos) {
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

// features of row rk for f = 4 step + g (same definitions as seg_attn.hip), plus what the backward needs
template <int MODE, int NS>
__device__ __forceinline__ void row_features(const PgTopo& t, const PgSegAttn& p, const RowInfo& rk, const float (&xd)[3],
                                             const float (&nd)[3], const float (&xj)[3], int g, float (&feat)[NS],
                                             RowGeo<MODE>& geo) {
  using T = ModeTraits<MODE>;
  geo.d = 0.f; geo.theta = 0.f; geo.src_lig = false;
#pragma unroll
  for (int c = 0; c < 3; ++c) { geo.rel[c] = 0.f; geo.ns[c] = 0.f; geo.v[c] = 0.f; }
  if constexpr (T::KNN || T::PH || T::POS) {
    if (rk.valid) {
#pragma unroll
      for (int c = 0; c < 3; ++c) geo.rel[c] = xd[c] - p.x[rk.src * 3 + c];
      geo.d = sqrtf(geo.rel[0] * geo.rel[0] + geo.rel[1] * geo.rel[1] + geo.rel[2] * geo.rel[2]);
    }
  }
  if constexpr (T::KNN) {
    float dots[3] = {0.f, 0.f, 0.f};
    if (rk.valid) {
#pragma unroll
      for (int c = 0; c < 3; ++c) geo.ns[c] = p.nrm[rk.src * 3 + c];
      dots[0] = geo.ns[0] * nd[0] + geo.ns[1] * nd[1] + geo.ns[2] * nd[2];
      dots[1] = -(geo.ns[0] * geo.rel[0] + geo.ns[1] * geo.rel[1] + geo.ns[2] * geo.rel[2]);
      dots[2] = -(nd[0] * geo.rel[0] + nd[1] * geo.rel[1] + nd[2] * geo.rel[2]);
      geo.src_lig = t.ctx_is_lig[rk.src] != 0;
    }
#pragma unroll
    for (int st = 0; st < 5; ++st) {
      const float sv = rk.valid ? smear(geo.d, 4 * st + g) : 0.f;
      feat[st] = geo.src_lig ? sv : 0.f;
      feat[5 + st] = geo.src_lig ? 0.f : sv;
    }
    feat[10] = g == 0 ? dots[0] : (g == 1 ? dots[1] : (g == 2 ? dots[2] : ((rk.valid && geo.src_lig) ? 1.f : 0.f)));
    feat[11] = (g == 0 && rk.valid && !geo.src_lig) ? 1.f : 0.f;
  } else if constexpr (T::TRI) {
    if (rk.valid) {
      float u[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) { u[c] = xj[c] - xd[c]; geo.v[c] = p.x[rk.src * 3 + c] - xd[c]; }
      const float a = u[0] * geo.v[0] + u[1] * geo.v[1] + u[2] * geo.v[2];
      const float c0 = u[1] * geo.v[2] - u[2] * geo.v[1], c1 = u[2] * geo.v[0] - u[0] * geo.v[2],
                  c2 = u[0] * geo.v[1] - u[1] * geo.v[0];
      geo.theta = atan2f(sqrtf(c0 * c0 + c1 * c1 + c2 * c2), a);
    }
#pragma unroll
    for (int st = 0; st < 3; ++st) {
      const int f = 4 * st + g;
      float v = sincos_bounded(geo.theta * kAngFreq[f], f >= 6);
      v = f == 0 ? geo.theta : v;
      feat[st] = (rk.valid && f != 11) ? v : 0.f;
    }
  } else if constexpr (T::PH) {
    feat[0] = g == 0 ? geo.d : 0.f;
  }
}

// folded LayerNorm statistics of a transposed tile hid[tau][r] = hidden[c = 16 tau + 4g + r][row = m]
__device__ __forceinline__ void ln_stats(const f4 (&hid)[8], float& rs, float& sigma) {
  float q = 0.f;
#pragma unroll
  for (int tq = 0; tq < 8; ++tq)
#pragma unroll
    for (int r = 0; r < 4; ++r) q = fmaf(hid[tq][r], hid[tq][r], q);
  q += __shfl_xor(q, 16);
  q += __shfl_xor(q, 32);
  const float var = q * (1.f / 128.f) + 1e-5f;
  rs = 1.0f / sqrtf(var);
  sigma = var * rs;
}

// the row buffer is written and read by lanes of the same wave only: workgroup-scope ordering is enough (an agent-scope
// fence writes the L2 back on a multi-XCD part)
__device__ __forceinline__ void rowbuf_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

struct BwdLds {
  float *wf_k, *wf_v, *bk, *bv, *accP, *accX, *wfp_k, *wfp_v;
  float *sT, *sF, *sGF, *sR, *sC;
  int* sI;
};

}  // namespace

// SPLIT (one-pass forms of the feature modes only): the key path and the value path of a segment as two launches, so that a wave
// holds ONE path's operands, accumulators and tiles -- half the registers and, for the triplet rows, half the LDS.  That allows 8 waves
// per workgroup = two per SIMD at 256 registers each, or 4 waves that no longer spill at 512 (launch_bwd_split picks per mode and
// pass).  SPLIT = 1, the value pass, runs first: it needs nothing of the key path and leaves
// d logit [segment][row][16] (PgSegAttnGrad.dlogit) and its d feat rows (PgSegAttnGrad.gfeat_v) for SPLIT = 2, the key pass,
// which adds the two d feat parts and runs the geometry adjoint once.  SPLIT = 0: both paths in one wave (every other form).
template <int MODE, int NW, bool OP = false, int SPLIT = 0>   // OP: one pass, alpha / S of the forward given (triplet training form)
__global__ __launch_bounds__(64 * NW) void seg_attn_bwd_kernel(PgTopo t, PgSegAttn p, PgSegAttnGrad gr PG_ABL_PARAM) {
  using T = ModeTraits<MODE>;
  // An opaque kernel-uniform zero.  The d feat and d Wf products of a tile sit behind `if (live)` tests of it: always taken, but
  // the instruction scheduler cannot move their MFMA chains across the test and interleave them with the LayerNorm adjoint in
  // front and the scatter behind (which it does otherwise, at the price of longer live ranges everywhere).  Measured, same box,
  // same sources: knn-node adjoint 2.25 -> 1.91 ms per launch, its key pass 1.22 -> 1.04, the triplet key pass 5.09 -> 4.13
  // (profiles/r04_adjoint_codegen_fences.txt).
  // The triplet forms at 512 registers are the exception (one wave with both paths 7.19 -> 7.34 ms with the tests, the 4-wave key pass
  // 6.73 -> 6.84 per adjoint): they keep plain code.
  constexpr bool FENCED = T::KNN || (SPLIT != 0 && NW == 8);
  int opq_zero = 0;
  asm volatile("" : "+s"(opq_zero));
  const bool live = FENCED ? opq_zero == 0 : true;
  static_assert(SPLIT == 0 || (OP && !T::POS) || (!OP && T::POS && T::KNN),
                "the split form exists for the one-pass feature modes and for the knn position update");
  // path order inside a tile: one-pass forms value first (its projection yields d logit), the others key first
  constexpr int P_V = OP ? 0 : 1, P_K = OP ? 1 : 0;
  constexpr int P_BEGIN = SPLIT == 1 ? P_V : (SPLIT == 2 ? P_K : 0), P_END = SPLIT == 1 ? P_V + 1 : (SPLIT == 2 ? P_K + 1 : 2);
  constexpr bool DO_K = SPLIT != 1, DO_V = SPLIT != 2;
  constexpr int NPATH = SPLIT ? 1 : 2, ACCW = SPLIT ? 128 : 256;
  constexpr int NSTEP = T::NSTEP, NS = NSTEP > 0 ? NSTEP : 1, F = 4 * NSTEP, NFT = (F + 15) / 16, NF = NFT > 0 ? NFT : 1;
  constexpr int FS = 16 * NF + 1;
  constexpr int PW = 128 * 17 + 2 * 16 * FS + 64 + 32 + 256;       // per-wave floats
  extern __shared__ __attribute__((aligned(16))) float lds_raw[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, m = lane & 15;

  BwdLds L;
  {
    float* q = lds_raw;
    L.wf_k = q; if (DO_K) q += NSTEP * 512;
    L.wf_v = q; if (DO_V) q += NSTEP * 512;
    L.bk = q; q += 128;
    L.bv = q; q += 128;
    L.accP = q; if (T::TRI) q += (size_t)((t.max_nlig + 15) & ~15) * ACCW;  // triplet: d P[k -> j] of the workgroup's source atom
    L.accX = q; if (T::TRI) q += (size_t)((t.max_nlig + 15) & ~15) * 3;     // triplet: d x_k of the workgroup's graph rows
    L.wfp_k = q; if (NFT == 1 && DO_K) q += 128 * 17;          // Wf[c][f] (stride 17) for d feat = Wf^T . d hidden
    L.wfp_v = q; if (NFT == 1 && DO_V) q += 128 * 17;
    q += wave * PW;
    L.sT = q; q += 128 * 17;
    L.sF = q; q += 16 * FS;
    L.sGF = q; q += 16 * FS;
    L.sR = q; q += 64;
    L.sI = reinterpret_cast<int*>(q); q += 32;
    L.sC = q;
  }
  for (int i = tid; i < NSTEP * 512; i += blockDim.x) {
    if (DO_K) L.wf_k[i] = p.Wf_k[i];
    if (DO_V) L.wf_v[i] = p.Wf_v[i];
  }
  for (int i = tid; i < 128; i += blockDim.x) { L.bk[i] = p.ln_bk[i]; L.bv[i] = p.ln_bv[i]; }
  if constexpr (NFT == 1) {
    for (int i = tid; i < 128 * 16; i += blockDim.x) {
      const int c = i >> 4, f = i & 15;
      const int src = ((f >> 2) * 8 + (c >> 4)) * 64 + (f & 3) * 16 + (c & 15);
      if (DO_K) L.wfp_k[c * 17 + f] = f < F ? p.Wf_k[src] : 0.f;
      if (DO_V) L.wfp_v[c * 17 + f] = f < F ? p.Wf_v[src] : 0.f;
    }
  }
  for (int i = lane; i < 16 * FS; i += 64) { L.sF[i] = 0.f; L.sGF[i] = 0.f; }
  __syncthreads();

  // first-layer feature weight W[c][f] out of the lane-fixed copy ([step][tau][lane=(f&3, c&15)])
  auto wf_plain = [&](const float* wf, int c, int f) -> float {
    return f < F ? wf[((f >> 2) * 8 + (c >> 4)) * 64 + (f & 3) * 16 + (c & 15)] : 0.f;
  };

  f4 gw2_acc[8];
#pragma unroll
  for (int tq = 0; tq < 8; ++tq) gw2_acc[tq] = (f4){0.f, 0.f, 0.f, 0.f};
  float gbk0 = 0.f, gbk1 = 0.f, gbv0 = 0.f, gbv1 = 0.f;        // d b'[lane], d b'[lane + 64] of the two paths
  f4 gwf_acc[NPATH][NF][8];                                    // d Wf: [path][f tile][tau] -> (c = 16 tau + 4g + r, f = 16 ft + m)
#pragma unroll
  for (int a = 0; a < NPATH; ++a)
#pragma unroll
    for (int ft = 0; ft < NF; ++ft)
#pragma unroll
      for (int tq = 0; tq < 8; ++tq) gwf_acc[a][ft][tq] = (f4){0.f, 0.f, 0.f, 0.f};
  float gbx_acc = 0.f;
  float* const rb = gr.rowbuf + (size_t)(blockIdx.x * NW + wave) * gr.rowbuf_rows * ROWBUF;
  const float bx = T::POS ? p.b2xv[m] : 0.f;

  // triplet: merge the 4 waves' d hidden tiles (own sT each) into the workgroup's accP rows; channel-owned, no atomics.
  // Every wave of the workgroup must call this the same number of times (idle waves with a zeroed tile).
  auto tri_merge = [&](int tile, bool kp) {
    constexpr int CPW = 128 / NW, GROUPS = 64 / CPW, RPL = 16 / GROUPS;   // channels per wave, lane groups, rows per lane
    __syncthreads();
    const int cc = CPW * wave + (lane % CPW), grp = lane / CPW;
    float* ap = L.accP + (size_t)(tile * 16 + RPL * grp) * ACCW + ((SPLIT || kp) ? 0 : 128) + cc;
    const float* st0 = L.sT - wave * PW + cc * 17 + RPL * grp;      // wave 0's tile, this lane's channel / row block
#pragma unroll
    for (int rr = 0; rr < RPL; ++rr) {
      float v = 0.f;
#pragma unroll
      for (int w2 = 0; w2 < NW; ++w2) v += st0[w2 * PW + rr];
      ap[rr * ACCW] += v;
    }
    __syncthreads();
  };
  PROF_DECL();
  auto process = [&](const Seg<MODE>& s) {
    const int dst_ctx = T::TRI ? s.ci : s.seg;
    const int n_rows = s.n_rows;
    const int n_tiles = (n_rows + 15) >> 4;

    // one global round trip per segment: the Cdst rows go to LDS, U and M (dS of the segment / W2xv) to registers in
    // the B-operand layout of the projections (lane-fixed [c = 16 tau + 4g + r][h = m])
    L.sC[lane] = p.Cdst_k[(size_t)s.seg * p.ld_cdst + lane];
    L.sC[64 + lane] = p.Cdst_k[(size_t)s.seg * p.ld_cdst + 64 + lane];
    L.sC[128 + lane] = p.Cdst_v[(size_t)s.seg * p.ld_cdst + lane];
    L.sC[192 + lane] = p.Cdst_v[(size_t)s.seg * p.ld_cdst + 64 + lane];
    f4 Ur[8], Mr[8];
    {
      const float* Uk = p.U + (size_t)s.seg * 2048 + lane;
      const float* Mv = (T::POS ? p.W2xv_l : gr.gS + (size_t)s.seg * 2048) + lane;
#pragma unroll
      for (int tq = 0; tq < 8; ++tq)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          Ur[tq][r] = DO_K ? Uk[(tq * 4 + r) * 64] : 0.f;
          Mr[tq][r] = DO_V ? Mv[(tq * 4 + r) * 64] : 0.f;
        }
    }
    wave_lds_sync();
    const float* const cdk = L.sC + 4 * g;
    const float* const cdv = L.sC + 128 + 4 * g;
    float xd[3] = {0.f, 0.f, 0.f}, nd[3] = {0.f, 0.f, 0.f}, xj[3] = {0.f, 0.f, 0.f}, gdx[3] = {0.f, 0.f, 0.f};
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
    if constexpr (T::POS) {
#pragma unroll
      for (int c = 0; c < 3; ++c) gdx[c] = gr.gdx[dst_ctx * 3 + c];
    }
    const float gswn_m = T::POS ? 0.f : gr.gswn[(size_t)s.seg * 16 + m];

    // hidden^T tile of one path: gather + per-segment constant + feature product
    auto hidden_tile = [&](const float* Csrc, const float* cd, const float* wf, const RowInfo& rk, const float (&feat)[NS],
                           f4 (&hid)[8]) {
      const float* pk = Csrc + (size_t)rk.csrc * p.ld_csrc + 4 * g;
#pragma unroll
      for (int tq = 0; tq < 8; ++tq) {
        f4 c = {0.f, 0.f, 0.f, 0.f};
        if (rk.valid) c = *reinterpret_cast<const f4*>(pk + 16 * tq);
        hid[tq] = c + *reinterpret_cast<const f4*>(cd + 16 * tq);
      }
#pragma unroll
      for (int st = 0; st < NSTEP; ++st)
#pragma unroll
        for (int tq = 0; tq < 8; ++tq) hid[tq] = mfma16(wf[(st * 8 + tq) * 64 + lane], feat[st], hid[tq]);
    };
    // y[row = 4g+r][h = m] = ReLU(hidden + b' sigma) . M[:,h]   (unscaled by rstd); hidden stays as it is
    auto relu_project = [&](const f4 (&hid)[8], const float* bp, float sigma, const f4 (&M)[8]) -> f4 {
      f4 y[4];                                   // 4 independent accumulation chains (a dependent MFMA waits ~10 extra cycles)
#pragma unroll
      for (int r = 0; r < 4; ++r) y[r] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int tq = 0; tq < 8; ++tq) {
        const f4 bt = *reinterpret_cast<const f4*>(bp + 16 * tq + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) y[r] = mfma16(fmaxf(fmaf(bt[r], sigma, hid[tq][r]), 0.f), M[tq][r], y[r]);
      }
      return (y[0] + y[1]) + (y[2] + y[3]);
    };

    PROF_T0();
    PROF(0);   // segment setup
    // =============================== pass 1: logits and tv of every row ===============================
    // position modes with the forward's per-row logits / value scalars at hand (PgSegAttnGrad.alpha, 32 floats per row): copied
    // into the row buffer instead of recomputing both MLPs of every row for them
    const bool have_lt = T::POS && !OP && gr.alpha != nullptr;
    if (have_lt) {
      const float* src = gr.alpha + (size_t)s.seg * gr.alpha_rows * 32;
      for (int r = g; r < n_rows; r += 4) {
        rb[r * ROWBUF + m] = src[r * 32 + m];
        rb[r * ROWBUF + 16 + m] = src[r * 32 + 16 + m];
      }
    }
    for (int tile = 0; tile < ((OP || have_lt || PG_ABL(32)) ? 0 : n_tiles); ++tile) {
      const RowInfo rk = row_info<MODE>(t, p, s, tile * 16 + m);
      float feat[NS];
      RowGeo<MODE> geo;
      row_features<MODE, NS>(t, p, rk, xd, nd, xj, g, feat, geo);
      f4 hid[8];
      float rs, sg;
      hidden_tile(p.Csrc_k, cdk, L.wf_k, rk, feat, hid);
      ln_stats(hid, rs, sg);
      f4 y = relu_project(hid, L.bk, sg, Ur);
#pragma unroll
      for (int r = 0; r < 4; ++r) y[r] *= __shfl(rs, 4 * g + r);
      hidden_tile(p.Csrc_v, cdv, L.wf_v, rk, feat, hid);
      ln_stats(hid, rs, sg);
      f4 tv = relu_project(hid, L.bv, sg, Mr);
#pragma unroll
      for (int r = 0; r < 4; ++r) tv[r] = tv[r] * __shfl(rs, 4 * g + r) + bx;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = tile * 16 + 4 * g + r;
        if (row < n_rows) {
          const RowInfo rv = row_info<MODE>(t, p, s, row);
          rb[row * ROWBUF + m] = rv.valid ? y[r] : NEG_BIG;
          rb[row * ROWBUF + 16 + m] = tv[r];
        }
      }
    }
    rowbuf_sync();
    PROF(1);   // pass 1

    // =============================== softmax backward per head m (rows r = g, g+4, ...) ===============================
    if (!OP && !PG_ABL(16)) {
      float mx = NEG_BIG;
      for (int r = g; r < n_rows; r += 4) mx = fmaxf(mx, rb[r * ROWBUF + m]);
      mx = fmaxf(mx, __shfl_xor(mx, 16));
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      float l = 0.f;
      for (int r = g; r < n_rows; r += 4) {
        const float lg = rb[r * ROWBUF + m];
        l += lg > 0.5f * NEG_BIG ? exp2f(lg - mx) : 0.f;
      }
      l += __shfl_xor(l, 16);
      l += __shfl_xor(l, 32);
      const float inv = l > 0.f ? 1.0f / l : 0.f;
      float D = 0.f;
      for (int r = g; r < n_rows; r += 4) {
        const float lg = rb[r * ROWBUF + m];
        const float a = lg > 0.5f * NEG_BIG ? exp2f(lg - mx) * inv : 0.f;
        const float tvv = rb[r * ROWBUF + 16 + m];
        float w = 1.f;
        if constexpr (T::KNN) w = p.ew[(size_t)s.seg * p.knn_k + r];
        float ga;
        if constexpr (T::POS) {
          const RowInfo rv = row_info<MODE>(t, p, s, r);
          float e = 0.f;
          if (rv.valid) {
#pragma unroll
            for (int c = 0; c < 3; ++c) e += gdx[c] * (xd[c] - p.x[rv.src * 3 + c]);
          }
          ga = w * e * (1.f / 16.f) * tvv;
        } else {
          ga = w * (tvv + gswn_m);
        }
        D += a * ga;
        rb[r * ROWBUF + m] = a;
        rb[r * ROWBUF + 32 + m] = ga;
      }
      D += __shfl_xor(D, 16);
      D += __shfl_xor(D, 32);
      for (int r = g; r < n_rows; r += 4) {
        const float a = rb[r * ROWBUF + m];
        rb[r * ROWBUF + 32 + m] = LN2 * a * (rb[r * ROWBUF + 32 + m] - D);
      }
    }
    rowbuf_sync();
    PROF(2);   // softmax

    // one-pass form: D[h] = sum_r alpha * dalpha = <S[:,h], dS[:,h]> + swn[h] * dswn[h] (S, swn of the forward), alpha read back
    float Dm = 0.f;
    const float* arow = nullptr;
    if constexpr (OP && DO_V) {
      const float* Sp = gr.S + (size_t)s.seg * 2048 + lane;
#pragma unroll
      for (int tq = 0; tq < 8; ++tq)
#pragma unroll
        for (int r = 0; r < 4; ++r) Dm = fmaf(Sp[(tq * 4 + r) * 64], Mr[tq][r], Dm);
      Dm += __shfl_xor(Dm, 16);
      Dm += __shfl_xor(Dm, 32);
      Dm = fmaf(gr.swn[(size_t)s.seg * 16 + m], gswn_m, Dm);
    }
    if constexpr (OP) arow = gr.alpha + (size_t)s.seg * gr.alpha_rows * 16;
    // split form: d logit and the value pass's d feat rows of this segment (same row indexing as alpha)
    float* const dl_row = SPLIT ? gr.dlogit + (size_t)s.seg * gr.alpha_rows * 16 : nullptr;
    float* const gfv_row = SPLIT ? gr.gfeat_v + (size_t)s.seg * gr.alpha_rows * (16 * NF) : nullptr;
    // =============================== pass 2: gradients ===============================
    f4 gU[8];
#pragma unroll
    for (int tq = 0; tq < 8; ++tq) gU[tq] = (f4){0.f, 0.f, 0.f, 0.f};
    float gcd_k0 = 0.f, gcd_k1 = 0.f, gcd_v0 = 0.f, gcd_v1 = 0.f;
    float gxs[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // d x_j, d x_i (triplet) / d x_dst, d nrm_dst (knn, pos) summed over the segment's rows: one atomic each

    for (int tile = 0; tile < (PG_ABL(64) ? 0 : n_tiles); ++tile) {
      const int row_m = tile * 16 + m;
      const RowInfo rk = row_info<MODE>(t, p, s, row_m);
      // every global load of the tile is requested up front (the feature arithmetic runs while they are in flight)
      f4 aD_raw = {0.f, 0.f, 0.f, 0.f}, aK_raw = {0.f, 0.f, 0.f, 0.f};
      if constexpr (OP) {
        // (key pass: d logit of the value pass instead of the softmax weights, same two layouts)
        const float* const ar = SPLIT == 2 ? dl_row : arow;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = tile * 16 + 4 * g + r;
          if (row < n_rows) aD_raw[r] = ar[row * 16 + m];
        }
        if (row_m < n_rows) aK_raw = *reinterpret_cast<const f4*>(ar + row_m * 16 + 4 * g);
      }
      f4 gfv_raw[NF];
#pragma unroll
      for (int ft = 0; ft < NF; ++ft) gfv_raw[ft] = (f4){0.f, 0.f, 0.f, 0.f};
      if constexpr (SPLIT == 2 && NSTEP > 0) {
#pragma unroll
        for (int ft = 0; ft < NF; ++ft)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = tile * 16 + 4 * g + r;
            if (row < n_rows) gfv_raw[ft][r] = gfv_row[(size_t)row * (16 * NF) + 16 * ft + m];
          }
      }
      // the Csrc rows of BOTH paths are requested here: with one wave per SIMD each dependent round trip is exposed (the knn
      // forms are out of registers either way -- they spill ~430 / ~570 with or without these 64 -- and are 4 % faster with them)
      constexpr bool PRE = true;
      f4 pre[NPATH][PRE ? 8 : 1];                 // [0] key path, [1] value path; split form: [0] = the pass's own path
      if constexpr (PRE) {
        const float* pk = p.Csrc_k + (size_t)rk.csrc * p.ld_csrc + 4 * g;
        const float* pv = p.Csrc_v + (size_t)rk.csrc * p.ld_csrc + 4 * g;
#pragma unroll
        for (int tq = 0; tq < 8; ++tq) {
          if constexpr (DO_K) pre[0][tq] = rk.valid ? *reinterpret_cast<const f4*>(pk + 16 * tq) : (f4){0.f, 0.f, 0.f, 0.f};
          if constexpr (DO_V) pre[SPLIT ? 0 : 1][tq] = rk.valid ? *reinterpret_cast<const f4*>(pv + 16 * tq) : (f4){0.f, 0.f, 0.f, 0.f};
        }
      }
      float feat[NS];
      RowGeo<MODE> geo;
      row_features<MODE, NS>(t, p, rk, xd, nd, xj, g, feat, geo);
      if constexpr (NSTEP > 0) {
#pragma unroll
        for (int st = 0; st < NSTEP; ++st) L.sF[m * FS + 4 * st + g] = feat[st];
      }
      if (g == 0) L.sI[m] = rk.valid ? rk.csrc : -1;
      // per-row weight of the value path: cw = gate (non-pos) or gate * <ddx, rel_x> / 16 (pos)
      float w_m = 1.f;
      if constexpr (T::KNN) w_m = (rk.valid && row_m < p.knn_k) ? p.ew[(size_t)s.seg * p.knn_k + row_m] : 0.f;
      float e_m = 0.f;
      if constexpr (T::POS) e_m = (gdx[0] * geo.rel[0] + gdx[1] * geo.rel[1] + gdx[2] * geo.rel[2]) * (1.f / 16.f);
      const float cw_m = rk.valid ? (T::POS ? w_m * e_m : w_m) : 0.f;
      if (g == 0) L.sR[32 + m] = cw_m;
      if constexpr (OP) wave_lds_sync();      // the value path runs first in the one-pass form and reads cw right away

      f4 gfeat[NF];
#pragma unroll
      for (int ft = 0; ft < NF; ++ft) gfeat[ft] = (f4){0.f, 0.f, 0.f, 0.f};
      // row-buffer values of this tile in one batch of loads: rows 4g + r at head m, and row m at heads 4g .. 4g+3
      f4 aD = {0.f, 0.f, 0.f, 0.f}, glD = {0.f, 0.f, 0.f, 0.f}, aK = {0.f, 0.f, 0.f, 0.f}, glK = {0.f, 0.f, 0.f, 0.f};
      if constexpr (SPLIT == 2) {
#pragma unroll
        for (int ft = 0; ft < NF; ++ft) gfeat[ft] = gfv_raw[ft];
      }
      if constexpr (SPLIT == 2 && T::POS) {
        // A[row] = (1/16) gate sum_h alpha vx of the value pass (d rel_x = A * ddx, below)
        if (m < 4) {
          const int row = tile * 16 + 4 * g + m;
          L.sR[48 + 4 * g + m] = row < n_rows ? dl_row[row * 16] : 0.f;
        }
      }
      if constexpr (SPLIT == 2 && OP) {
        glD = aD_raw;
        glK = aK_raw;
      } else if constexpr (OP) {
        aD = aD_raw;
        aK = aK_raw;
        if constexpr (T::KNN) {            // the knn forward stores alpha x gate
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float wr = L.sR[32 + 4 * g + r];
            aD[r] = wr > 0.f ? aD[r] / wr : 0.f;
          }
          aK = w_m > 0.f ? aK * (1.0f / w_m) : (f4){0.f, 0.f, 0.f, 0.f};
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = tile * 16 + 4 * g + r;
          if (row < n_rows) { aD[r] = rb[row * ROWBUF + m]; glD[r] = rb[row * ROWBUF + 32 + m]; }
        }
        if (row_m < n_rows) {
          aK = *reinterpret_cast<const f4*>(rb + row_m * ROWBUF + 4 * g);
          glK = *reinterpret_cast<const f4*>(rb + row_m * ROWBUF + 32 + 4 * g);
        }
      }

#pragma unroll
      for (int path = P_BEGIN; path < (PG_ABL(8) ? P_BEGIN + 1 : P_END); ++path) {
        const bool kp = path == P_K;
        const float* bp = kp ? L.bk : L.bv;
        const float* wf = kp ? L.wf_k : L.wf_v;
        f4 hid[8];
        float rs, sg;
        if constexpr (!PRE) {
          hidden_tile(kp ? p.Csrc_k : p.Csrc_v, kp ? cdk : cdv, wf, rk, feat, hid);
        } else {
          const float* cd = kp ? cdk : cdv;
#pragma unroll
          for (int tq = 0; tq < 8; ++tq) hid[tq] = pre[SPLIT ? 0 : (kp ? 0 : 1)][tq] + *reinterpret_cast<const f4*>(cd + 16 * tq);
#pragma unroll
          for (int st = 0; st < NSTEP; ++st)
#pragma unroll
            for (int tq = 0; tq < 8; ++tq) hid[tq] = mfma16(wf[(st * 8 + tq) * 64 + lane], feat[st], hid[tq]);
        }
        ln_stats(hid, rs, sg);
        PROF(3);   // tile head: features, row buffer loads
        const f4 y = kp ? relu_project(hid, bp, sg, Ur) : relu_project(hid, bp, sg, Mr);   // unscaled, rows 4g+r, head m
        PROF(4);   // recompute
        if constexpr (OP) {
          if (!kp) {
            // dalpha[row,h] = cw * (rstd_v * y + dswn);  dlogit = ln2 * alpha * (dalpha - D); key path needs it in both layouts
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float ga = L.sR[32 + 4 * g + r] * fmaf(y[r], __shfl(rs, 4 * g + r), gswn_m);
              glD[r] = LN2 * aD[r] * (ga - Dm);
              if constexpr (SPLIT == 1) {
                const int row = tile * 16 + 4 * g + r;
                if (row < n_rows) dl_row[row * 16 + m] = glD[r];       // for the key pass
              } else {
                L.sGF[(4 * g + r) * FS + m] = glD[r];
              }
            }
            if constexpr (SPLIT != 1) {
              wave_lds_sync();
#pragma unroll
              for (int ks = 0; ks < 4; ++ks) glK[ks] = L.sGF[m * FS + 4 * g + ks];
            }
          }
        }
        // coefficient of y in the loss, rows 4g+r: k path dlogit ; v path cw * alpha
        f4 coefD;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = tile * 16 + 4 * g + r;
          coefD[r] = kp ? glD[r] : aD[r] * L.sR[32 + 4 * g + r];
        }
        wave_lds_sync();   // sR[32..] written above is read here by other lanes only after this point on later paths
        if (!kp) {
          // gate gradient, d(rel_x) weight and the bias of the pos value head
          f4 av;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = tile * 16 + 4 * g + r;
            const float a = aD[r];
            const float vfull = y[r] * __shfl(rs, 4 * g + r);          // rstd_v * z_v . M
            av[r] = row16_total(a * (T::POS ? vfull + bx : vfull + gswn_m));
          }
          if constexpr (T::KNN) {
            // d gate[row] = sum_h alpha * (rstd t + dswn)    (pos: e * sum_h alpha * vx)
            if (m < 4) {
              const int row = tile * 16 + 4 * g + m;
              const float a4 = m == 0 ? av[0] : (m == 1 ? av[1] : (m == 2 ? av[2] : av[3]));
              if (row < p.knn_k && gr.gew) {
                float e_r = 1.f;
                if constexpr (T::POS) {
                  const RowInfo rv = row_info<MODE>(t, p, s, row);
                  e_r = 0.f;
                  if (rv.valid) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) e_r += gdx[c] * (xd[c] - p.x[rv.src * 3 + c]);
                  }
                  e_r *= (1.f / 16.f);
                }
                gr.gew[(size_t)s.seg * p.knn_k + row] = row < n_rows ? a4 * e_r : 0.f;
              }
            }
          }
          if constexpr (T::POS) {
            // A[row] = (1/16) * gate * sum_h alpha vx  -> d rel_x = A * ddx
            if (m < 4) {
              const float a4 = m == 0 ? av[0] : (m == 1 ? av[1] : (m == 2 ? av[2] : av[3]));
              L.sR[48 + 4 * g + m] = a4;
              if constexpr (SPLIT == 1) {          // for the key pass, which runs the geometry adjoint
                const int row = tile * 16 + 4 * g + m;
                if (row < n_rows) dl_row[row * 16] = a4;
              }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) gbx_acc += coefD[r];
          }
        }
        // d rstd of the row = sum_h coef * y
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = row16_total(coefD[r] * y[r]);
          if (m == r) L.sR[4 * g + r] = v;
        }
        if (g == 0) L.sR[16 + m] = rs;
        // z^T tile to LDS for the product contracted over rows (d U on the key path, d W2xv on the value path of the pos modes:
        // the value path of the feature modes has no such product)
        if (kp || T::POS) {
#pragma unroll
          for (int tq = 0; tq < 8; ++tq) {
            const f4 bt = *reinterpret_cast<const f4*>(bp + 16 * tq + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) L.sT[(16 * tq + 4 * g + r) * 17 + m] = fmaxf(fmaf(bt[r], sg, hid[tq][r]), 0.f);
          }
        }
        wave_lds_sync();
        const float grs = L.sR[m];
        PROF(5);   // gate/rstd sums, z tile to LDS
        // dM[c,h] += sum_row z[c,row] * rstd[row] * coef[row,h]      (k path: dU; pos v path: dW2xv)
        if (kp || T::POS) {
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {           // contraction index (row) of this k-step: 4g + ks
            const float b = coefD[ks] * L.sR[16 + 4 * g + ks];
#pragma unroll
            for (int tq = 0; tq < 8; ++tq) {
              const float a = L.sT[(16 * tq + m) * 17 + 4 * g + ks];
              if (kp) gU[tq] = mfma16(a, b, gU[tq]);
              else gw2_acc[tq] = mfma16(a, b, gw2_acc[tq]);
            }
          }
        }
        PROF(6);   // dM product
        // dz^T[c,row] = sum_h M[c,h] * coef[row,h] * rstd[row]; M^T comes out of the registers through the LDS tile
        wave_lds_sync();   // products over sT (z) are done
#pragma unroll
        for (int tq = 0; tq < 8; ++tq)
#pragma unroll
          for (int r = 0; r < 4; ++r) L.sT[(16 * tq + 4 * g + r) * 17 + m] = kp ? Ur[tq][r] : Mr[tq][r];
        wave_lds_sync();
        f4 gz[8];
#pragma unroll
        for (int tq = 0; tq < 8; ++tq) gz[tq] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {             // contraction index (head) of this k-step: 4g + ks
          const float b = (kp ? glK[ks] : aK[ks] * cw_m) * rs;
#pragma unroll
          for (int tq = 0; tq < 8; ++tq) gz[tq] = mfma16(L.sT[(16 * tq + m) * 17 + 4 * g + ks], b, gz[tq]);
        }
        PROF(7);   // dz product
        // folded LayerNorm backward: z = ReLU(h + b' sigma), sigma = sqrt(var), rstd = 1/sigma, var = mean(h^2) + eps
        float s1 = 0.f;
#pragma unroll
        for (int tq = 0; tq < 8; ++tq) {
          const f4 bt = *reinterpret_cast<const f4*>(bp + 16 * tq + 4 * g);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            gz[tq][r] = fmaf(bt[r], sg, hid[tq][r]) > 0.f ? gz[tq][r] : 0.f;
            s1 = fmaf(gz[tq][r], bt[r], s1);
          }
        }
        s1 += __shfl_xor(s1, 16);
        s1 += __shfl_xor(s1, 32);
        const float gvar = 0.5f * rs * s1 - 0.5f * grs * rs * rs * rs;
        wave_lds_sync();   // products over sT (M^T) are done
        // d b'[c] += sum_row dpre[c,row] * sigma[row]: through the LDS tile, each lane sums its two channels
#pragma unroll
        for (int tq = 0; tq < 8; ++tq)
#pragma unroll
          for (int r = 0; r < 4; ++r) L.sT[(16 * tq + 4 * g + r) * 17 + m] = gz[tq][r] * sg;
        wave_lds_sync();
        {
          float a0 = 0.f, a1 = 0.f;
#pragma unroll
          for (int rr = 0; rr < 16; ++rr) { a0 += L.sT[lane * 17 + rr]; a1 += L.sT[(64 + lane) * 17 + rr]; }
          if (kp) { gbk0 += a0; gbk1 += a1; } else { gbv0 += a0; gbv1 += a1; }
        }
        wave_lds_sync();
#pragma unroll
        for (int tq = 0; tq < 8; ++tq)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            gz[tq][r] = fmaf(gvar * (1.f / 64.f), hid[tq][r], gz[tq][r]);       // d hidden
            L.sT[(16 * tq + 4 * g + r) * 17 + m] = gz[tq][r];
          }
        wave_lds_sync();
        PROF(8);   // LN adjoint, db', dhidden tile
        if (NSTEP > 0 && !PG_ABL(4)) {
          // d feat[row, f] += sum_c dhidden[c,row] * Wf[c,f]
          if (!T::PH && !PG_ABL(256) && live) {
            f4 gfp[NF][4];                         // independent chains per r, folded below
#pragma unroll
            for (int ft = 0; ft < NF; ++ft)
#pragma unroll
              for (int r = 0; r < 4; ++r) gfp[ft][r] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ft = 0; ft < NF; ++ft)
#pragma unroll
              for (int tq = 0; tq < 8; ++tq)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                  gfp[ft][r] = mfma16(gz[tq][r], NFT == 1 ? (kp ? L.wfp_k : L.wfp_v)[(16 * tq + 4 * g + r) * 17 + m]
                                                            : wf_plain(wf, 16 * tq + 4 * g + r, 16 * ft + m), gfp[ft][r]);
#pragma unroll
            for (int ft = 0; ft < NF; ++ft) gfeat[ft] += (gfp[ft][0] + gfp[ft][1]) + (gfp[ft][2] + gfp[ft][3]);
          }
          // d Wf[c,f] += sum_row dhidden[c,row] * feat[row,f]
          // registers for the whole kernel (LDS ds_add_f32 accumulation measured ~700 cycles per instruction)
          if (!PG_ABL(512) && live)
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {           // k-step outermost: consecutive MFMAs hit different accumulators
            float bf[NF];
#pragma unroll
            for (int ft = 0; ft < NF; ++ft) bf[ft] = L.sF[(4 * g + ks) * FS + 16 * ft + m];
#pragma unroll
            for (int tq = 0; tq < 8; ++tq) {
              const float a = L.sT[(16 * tq + m) * 17 + 4 * g + ks];
#pragma unroll
              for (int ft = 0; ft < NF; ++ft)
                gwf_acc[SPLIT ? 0 : (kp ? 0 : 1)][ft][tq] = mfma16(a, bf[ft], gwf_acc[SPLIT ? 0 : (kp ? 0 : 1)][ft][tq]);
            }
          }
        }
        PROF(9);   // dfeat, dWf
        // d Csrc (scatter) and d Cdst (row sum): lane owns channels lane and lane + 64
        if (!PG_ABL(2)) {
          float* gsrc = kp ? gr.gCsrc_k : gr.gCsrc_v;
          float a0 = 0.f, a1 = 0.f;
          if constexpr (T::TRI) {
            // per-source-atom rows in LDS without atomics (ds_add_f32 costs ~700 cycles here, profiles/r01f_*sections.md):
            // the 4 waves of the workgroup walk their segments in lockstep, every wave has its d hidden tile in its own
            // sT, and wave w adds the 4 tiles' channels [32w, 32w+32) into the rows it alone owns
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) { a0 += L.sT[lane * 17 + rr]; a1 += L.sT[(64 + lane) * 17 + rr]; }
            tri_merge(tile, kp);
          } else {
            for (int rr = 0; rr < 16; ++rr) {
              const int ci = L.sI[rr];
              if (ci >= 0) {
                const float v0 = L.sT[lane * 17 + rr], v1 = L.sT[(64 + lane) * 17 + rr];
                if constexpr (T::BOND) {          // every row is written exactly once
                  gsrc[(size_t)ci * gr.ld_gcsrc + lane] = v0;
                  gsrc[(size_t)ci * gr.ld_gcsrc + 64 + lane] = v1;
                } else {
                  atomicAdd(gsrc + (size_t)ci * gr.ld_gcsrc + lane, v0);
                  atomicAdd(gsrc + (size_t)ci * gr.ld_gcsrc + 64 + lane, v1);
                }
                a0 += v0; a1 += v1;
              }
            }
          }
          if (kp) { gcd_k0 += a0; gcd_k1 += a1; } else { gcd_v0 += a0; gcd_v1 += a1; }
        }
        wave_lds_sync();   // sT is rewritten by the next path / tile
        PROF(10);  // scatter
      }  // paths

      if constexpr (SPLIT == 1 && NSTEP > 0) {     // value pass: its d feat rows go to the key pass, which runs the geometry adjoint
#pragma unroll
        for (int ft = 0; ft < NF; ++ft)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = tile * 16 + 4 * g + r;
            if (row < n_rows) gfv_row[(size_t)row * (16 * NF) + 16 * ft + m] = gfeat[ft][r];
          }
      }
      // ---------------- geometry: d feat -> positions / direction vectors; pos modes: d rel_x ----------------
      if constexpr (SPLIT != 1) {
      if constexpr ((T::KNN || T::TRI) && NSTEP > 0) {
#pragma unroll
        for (int ft = 0; ft < NF; ++ft)
#pragma unroll
          for (int r = 0; r < 4; ++r) L.sGF[(4 * g + r) * FS + 16 * ft + m] = gfeat[ft][r];
        wave_lds_sync();
      }
      // knn: d d = sum_i d feat[i] * smear_i(d) * (off_i - d).  The smear values are the row's features and already sit in the
      // feature tile; the 20 terms are spread over the row's 4 lanes (g = 0..3) and added up across them
      float gd_row = 0.f;
      if constexpr (T::KNN) {
        const int base = geo.src_lig ? 0 : 20;
        const float* gf = L.sGF + m * FS + base;
        const float* ff = L.sF + m * FS + base;
#pragma unroll
        for (int i5 = 0; i5 < 5; ++i5) {
          const int i = 4 * i5 + g;
          gd_row = fmaf(gf[i] * ff[i], kSmearOff[i] - geo.d, gd_row);
        }
        gd_row += __shfl_xor(gd_row, 16);
        gd_row += __shfl_xor(gd_row, 32);
      }
      if (g == 0 && rk.valid && gr.gx && !PG_ABL(1)) {
        const float* gf = L.sGF + m * FS;
        float grel[3] = {0.f, 0.f, 0.f};
        if constexpr (T::KNN) {
          const float gd = gd_row;
          const float sc = geo.d > 0.f ? gd / geo.d : 0.f;
          const float g0 = gf[40], g1 = gf[41], g2 = gf[42];
          float gns[3], gnd[3];
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            grel[c] = sc * geo.rel[c] - g1 * geo.ns[c] - g2 * nd[c];
            gns[c] = g0 * nd[c] - g1 * geo.rel[c];
            gnd[c] = g0 * geo.ns[c] - g2 * geo.rel[c];
          }
          if (gr.gnrm) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              atomicAdd(gr.gnrm + rk.src * 3 + c, gns[c]);
              gxs[3 + c] += gnd[c];
            }
          }
        }
        if constexpr (T::POS) {
          const float A = L.sR[48 + m] * w_m * (1.f / 16.f);
#pragma unroll
          for (int c = 0; c < 3; ++c) grel[c] += A * gdx[c];
        }
        if constexpr (T::KNN || T::POS) {
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            gxs[c] += grel[c];
            atomicAdd(gr.gx + rk.src * 3 + c, -grel[c]);
          }
        }
        if constexpr (T::TRI) {
          // theta = atan2(|u x v|, u.v), u = x_j - x_i, v = x_k - x_i
          // d theta: the derivative of sin(w theta) is w cos(w theta) = w * feature[f + 5], of cos(w theta) it is
          // -w * feature[f - 5]: both already sit in the row's feature tile
          const float* ff = L.sF + m * FS;
          float gth = gf[0];
#pragma unroll
          for (int f = 1; f < 6; ++f) gth += kAngFreq[f] * (gf[f] * ff[f + 5] - gf[f + 5] * ff[f]);
          float u[3];
#pragma unroll
          for (int c = 0; c < 3; ++c) u[c] = xj[c] - xd[c];
          const float* v = geo.v;
          const float a = u[0] * v[0] + u[1] * v[1] + u[2] * v[2];
          const float cr[3] = {u[1] * v[2] - u[2] * v[1], u[2] * v[0] - u[0] * v[2], u[0] * v[1] - u[1] * v[0]};
          const float b = sqrtf(cr[0] * cr[0] + cr[1] * cr[1] + cr[2] * cr[2]);
          const float den = a * a + b * b;
          if (den > 0.f) {
            const float ka = -b / den * gth;                          // d theta / d a
            const float kb = b > 0.f ? a / den * gth / b : 0.f;       // d theta / d b, times 1/b of d b = c . dc / b
            // db/du = v x c, db/dv = c x u
            const float vxc[3] = {v[1] * cr[2] - v[2] * cr[1], v[2] * cr[0] - v[0] * cr[2], v[0] * cr[1] - v[1] * cr[0]};
            const float cxu[3] = {cr[1] * u[2] - cr[2] * u[1], cr[2] * u[0] - cr[0] * u[2], cr[0] * u[1] - cr[1] * u[0]};
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              const float gu = ka * v[c] + kb * vxc[c];
              const float gv = ka * u[c] + kb * cxu[c];
              gxs[c] += gu;
              gxs[3 + c] -= gu + gv;
              atomicAdd(L.accX + (tile * 16 + m) * 3 + c, gv);
            }
          }
        }
      }
      }  // SPLIT != 1
      wave_lds_sync();
      PROF(11);  // geometry
    }  // tiles

    // ---------------- per-segment outputs ----------------
    if constexpr (SPLIT == 1) {
      // (value pass: no geometry outputs)
    } else if constexpr (T::TRI) {
      if (gr.gx) {
#pragma unroll
        for (int c = 0; c < 6; ++c) {
          const float v = wave_sum(gxs[c]);
          if (lane == 0) atomicAdd(gr.gx + (c < 3 ? s.cj : dst_ctx) * 3 + (c % 3), v);
        }
      }
    } else if constexpr (T::KNN || T::POS) {
      if (gr.gx) {
#pragma unroll
        for (int c = 0; c < 6; ++c) {
          if (c >= 3 && !(T::KNN && gr.gnrm)) break;
          const float v = wave_sum(gxs[c]);
          if (lane == 0) atomicAdd((c < 3 ? gr.gx : gr.gnrm) + dst_ctx * 3 + (c % 3), v);
        }
      }
    }
    {
      float* up = gr.gU + (size_t)s.seg * 2048 + lane;
      if constexpr (DO_K) {
#pragma unroll
        for (int tq = 0; tq < 8; ++tq)
#pragma unroll
          for (int r = 0; r < 4; ++r) up[(tq * 4 + r) * 64] = gU[tq][r];
      }
      float* ck = gr.gCdst_k + (size_t)s.seg * gr.ld_gcdst;
      float* cv = gr.gCdst_v + (size_t)s.seg * gr.ld_gcdst;
      if constexpr (DO_K) { ck[lane] = gcd_k0; ck[64 + lane] = gcd_k1; }
      if constexpr (DO_V) { cv[lane] = gcd_v0; cv[64 + lane] = gcd_v1; }
    }
    __builtin_amdgcn_wave_barrier();
    PROF(12);  // segment outputs
  };  // process(segment)

  if constexpr (T::TRI) {
    // one workgroup per source atom j: its n-1 segments (j -> i) all scatter into the rows P[k -> j], which no other
    // source touches: accumulate them in LDS and store each row once
    for (int ai = blockIdx.x; ai < t.n_lig; ai += gridDim.x) {
      const int a = gr.atom_order ? gr.atom_order[ai] : ai;      // cost-sorted hand-out (PgSegAttnGrad.atom_order): levels the workgroups
      const int cj = t.lig2ctx[a];
      const int gi = t.ctx_graph[cj];
      const int n = t.g_nlig[gi], lig0 = t.g_ctx_off[gi] + t.g_nph[gi], lj = cj - lig0;
      const int* eid_g = t.eid + t.g_eid_off[gi];
      for (int i = tid; i < ((n + 15) & ~15) * ACCW; i += blockDim.x) L.accP[i] = 0.f;
      for (int i = tid; i < ((n + 15) & ~15) * 3; i += blockDim.x) L.accX[i] = 0.f;
      __syncthreads();
      const int n_tiles_j = (n + 15) >> 4;
      // the atom's n - 1 segments j -> i (i != j) are dealt out densely: n slots with the diagonal left idle cost a whole round of
      // NW segments whenever n = 1 mod NW
      for (int base = 0; base < n - 1; base += NW) {
        const int qs = base + wave;
        const int il = qs < lj ? qs : qs + 1;
        if (qs < n - 1) {
          Seg<MODE> s;
          s.seg = eid_g[lj * n + il];
          s.n_rows = s.n = n;
          s.lig0 = lig0; s.li = il; s.lj = lj; s.first = 0;
          s.ci = lig0 + il; s.cj = cj;
          s.eid_g = eid_g;
          process(s);
        } else if (!PG_ABL(2)) {       // idle wave of this round: same barrier sequence, zero contribution
          for (int tile = 0; tile < (PG_ABL(64) ? 0 : n_tiles_j); ++tile)
            for (int path = P_BEGIN; path < (PG_ABL(8) ? P_BEGIN + 1 : P_END); ++path) {
              for (int i = lane; i < 128 * 17; i += 64) L.sT[i] = 0.f;
              tri_merge(tile, path == P_K);
            }
        }
      }
      __syncthreads();
      for (int i = tid; i < n * ACCW; i += blockDim.x) {
        const int k = i / ACCW, c = i % ACCW;
        if (k == lj) continue;
        const int e = eid_g[k * n + lj];
        float* dst = ((SPLIT ? SPLIT == 2 : c < 128) ? gr.gCsrc_k : gr.gCsrc_v) + (size_t)e * gr.ld_gcsrc + (c & 127);
        *dst = L.accP[i];
      }
      if (gr.gx && DO_K)
        for (int i = tid; i < n * 3; i += blockDim.x) atomicAdd(gr.gx + (lig0 + i / 3) * 3 + (i % 3), L.accX[i]);
      __syncthreads();
    }
  } else {
    for (int si = blockIdx.x * NW + wave; si < p.n_seg; si += gridDim.x * NW) process(setup_seg<MODE>(t, p, si));
  }

  // ---------------- flush the weight-gradient accumulators ----------------
  PROF_FLUSH();
  __syncthreads();
  // d Wf: the NW waves' accumulators are added up through LDS first (every workgroup-wide value is then ONE global atomic instead of
  // NW: the flush of all workgroups goes to the same 2 x F x 128 addresses, and with one atomic per wave it was still 5 ms of a
  // 168 ms training step).  Chunk = one float4 per lane: (path a, feature tile ft, channel tile tq); 8 chunks per barrier pair.
  if constexpr (NSTEP > 0) {
    constexpr int NCH = NPATH * NF * 8, BATCH = 8;
    float* const red = lds_raw;                                   // the kernel's LDS is free now
#pragma unroll
    for (int c0 = 0; c0 < NCH; c0 += BATCH) {
#pragma unroll
      for (int j = 0; j < BATCH; ++j) {
        const int ch = c0 + j;                                    // compile-time after unrolling: a = ch / (NF * 8), ft, tq
        *reinterpret_cast<f4*>(red + ((size_t)(j * NW + wave) * 64 + lane) * 4) = gwf_acc[ch / (NF * 8)][(ch / 8) % NF][ch % 8];
      }
      __syncthreads();
      for (int j = wave; j < BATCH; j += NW) {                    // wave w sums and flushes chunks w, w + NW, ... of the batch
        const int ch = c0 + j, a = SPLIT ? (SPLIT == 2 ? 0 : 1) : ch / (NF * 8), ft = (ch / 8) % NF, tq = ch % 8;
        f4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w2 = 0; w2 < NW; ++w2) v += *reinterpret_cast<const f4*>(red + ((size_t)(j * NW + w2) * 64 + lane) * 4);
        const int f = 16 * ft + m;
        if (f < F) {
          float* dst = (a == 0 ? gr.gWf_k : gr.gWf_v) + ((f >> 2) * 8 + tq) * 64 + (f & 3) * 16 + 4 * g;   // lane-fixed layout of the forward weights
#pragma unroll
          for (int r = 0; r < 4; ++r) atomicAdd(dst + r, v[r]);
        }
      }
      __syncthreads();
    }
  }
  if constexpr (DO_K) { atomicAdd(gr.gbk + lane, gbk0); atomicAdd(gr.gbk + 64 + lane, gbk1); }
  if constexpr (DO_V) { atomicAdd(gr.gbv + lane, gbv0); atomicAdd(gr.gbv + 64 + lane, gbv1); }
  if constexpr (T::POS && DO_V) {                                   // d W2xv: the same way, one batch of 8 chunks
    float* const red = lds_raw;
#pragma unroll
    for (int tq = 0; tq < 8; ++tq) *reinterpret_cast<f4*>(red + ((size_t)(tq * NW + wave) * 64 + lane) * 4) = gw2_acc[tq];
    __syncthreads();
    for (int tq = wave; tq < 8; tq += NW) {
      f4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int w2 = 0; w2 < NW; ++w2) v += *reinterpret_cast<const f4*>(red + ((size_t)(tq * NW + w2) * 64 + lane) * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) atomicAdd(gr.gW2xv_l + (tq * 4 + r) * 64 + lane, v[r]);
    }
  }
  if constexpr (T::POS && DO_V) {
    gbx_acc += __shfl_xor(gbx_acc, 16);
    gbx_acc += __shfl_xor(gbx_acc, 32);
    if (g == 0) atomicAdd(gr.gb2xv + m, gbx_acc);
  }
}

template <int MODE, int NW, bool OP = false, int SPLIT = 0>
static int launch_bwd(const PgTopo* t, const PgSegAttn* p, const PgSegAttnGrad* gr, hipStream_t st) {
  using T = ModeTraits<MODE>;
  constexpr int NSTEP = T::NSTEP, F = 4 * NSTEP, NFT = (F + 15) / 16, NF = NFT > 0 ? NFT : 1, FS = 16 * NF + 1;
  constexpr int PW = 128 * 17 + 2 * 16 * FS + 64 + 32 + 256;
  constexpr int NPATH = SPLIT ? 1 : 2, ACCW = SPLIT ? 128 : 256;
  const size_t lds = ((size_t)NPATH * NSTEP * 512 + 256 + (T::TRI ? (size_t)((t->max_nlig + 15) & ~15) * (ACCW + 3) : 0) +
                      (NFT == 1 ? NPATH * 128 * 17 : 0) + (size_t)NW * PW) * sizeof(float);
  if (lds > 160 * 1024) { set_error("pg_seg_attn_bwd: %zu B of LDS needed (ligand of %d atoms is too large)", lds, t->max_nlig); return PG_ERR_ARG; }
  if (int rc = reserve_lds(reinterpret_cast<const void*>(seg_attn_bwd_kernel<MODE, NW, OP, SPLIT>), lds, "pg_seg_attn_bwd")) return rc;
  int blocks = T::TRI ? t->n_lig : (p->n_seg + NW - 1) / NW;
  if (blocks > gr->grid) blocks = gr->grid;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL((seg_attn_bwd_kernel<MODE, NW, OP, SPLIT>), dim3(blocks), dim3(64 * NW), lds, st, *t, *p, *gr PG_ABL_ARG("PG_BWD_ABLATE"));
  return check_launch("pg_seg_attn_bwd");
}

// The one-pass adjoint as a value pass and a key pass.  Workgroup size per pass, measured (profiles/r04_adjoint_codegen_fences.txt):
// a wave that holds ONE path fits 512 registers without spilling (336 ... 446; both paths in a wave: 22 / 431 spilled), and that is
// worth more than a second wave per SIMD at 256 registers (45 ... 307 spilled) -- except for the triplet value pass, the lightest of
// the four, which runs best as 8 waves.
template <int MODE, bool OP = true>
static int launch_bwd_split(const PgTopo* t, const PgSegAttn* p, const PgSegAttnGrad* gr, hipStream_t st) {
  constexpr int NW_V = MODE == PG_SEG_TRIPLET ? 8 : 4;
  if (int rc = launch_bwd<MODE, NW_V, OP, 1>(t, p, gr, st)) return rc;
  return launch_bwd<MODE, 4, OP, 2>(t, p, gr, st);
}

int triplet_bwd2_launch(const PgTopo* t, const PgSegAttn* p, const PgSegAttnGrad* gr, hipStream_t st);   // triplet_bwd2.hip

}  // namespace pg

using namespace pg;

#ifdef PG_BWD_PROF
extern "C" int pg_debug_bwd_prof(unsigned long long* out, int reset) {
  hipMemcpyFromSymbol(out, HIP_SYMBOL(pg::g_bwd_prof), sizeof(unsigned long long) * 16);
  if (reset) { unsigned long long z[16] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(pg::g_bwd_prof), z, sizeof(z)); }
  return 0;
}
#endif

// waves per workgroup the row buffer must be sized for (PgSegAttnGrad.rowbuf: grid x waves x rows x 48 floats): the LARGEST workgroup of any
// form this mode can be launched in -- the generic / one-wave forms use 4 waves, the triplet's value pass of the two-pass form 8 (the
// one-pass kernels do not touch the buffer today; sizing it for them anyway keeps a later change from writing past it)
extern "C" int pg_seg_attn_bwd_waves(int mode) { return mode == PG_SEG_TRIPLET ? 8 : 4; }

extern "C" int pg_seg_attn_bwd(const PgTopo* t, const PgSegAttn* p, const PgSegAttnGrad* gr, void* stream) {
  if (!t || !p || !gr) { set_error("pg_seg_attn_bwd: null argument"); return PG_ERR_ARG; }
  if (p->n_seg == 0) return PG_OK;
  if (!p->U || !p->Cdst_k || !p->Cdst_v || gr->grid < 1) {
    set_error("pg_seg_attn_bwd: U, Cdst_k/v and a grid are required");
    return PG_ERR_ARG;
  }
  // the row buffer: every form but the triplet's channel-split one (its rows live in LDS) -- a launch that needs it and has none is refused
  const bool tri_split = p->mode == PG_SEG_TRIPLET && gr->tri_form && t->max_nlig <= 64 && p->Cdst_v == p->Cdst_k + 128;
  if (!gr->rowbuf && !tri_split) {
    set_error("pg_seg_attn_bwd: mode %d in this form needs PgSegAttnGrad.rowbuf (grid x pg_seg_attn_bwd_waves x rows x 48 floats)", p->mode);
    return PG_ERR_ARG;
  }
  hipStream_t st = (hipStream_t)stream;
  switch (p->mode) {
    case PG_SEG_KNN_NODE:
      if (gr->alpha && gr->S && gr->swn && gr->dlogit && gr->gfeat_v) return launch_bwd_split<PG_SEG_KNN_NODE>(t, p, gr, st);
      return (gr->alpha && gr->S && gr->swn) ? launch_bwd<PG_SEG_KNN_NODE, 4, true>(t, p, gr, st)
                                             : launch_bwd<PG_SEG_KNN_NODE, 4>(t, p, gr, st);
    case PG_SEG_KNN_POS:
      // (two passes need the forward's logits / value scalars: each pass redoes the softmax adjoint from them)
      if (gr->alpha && gr->dlogit && gr->gfeat_v) return launch_bwd_split<PG_SEG_KNN_POS, false>(t, p, gr, st);
      return launch_bwd<PG_SEG_KNN_POS, 4>(t, p, gr, st);
    case PG_SEG_BOND_NODE:
      return (gr->alpha && gr->S && gr->swn) ? launch_bwd<PG_SEG_BOND_NODE, 4, true>(t, p, gr, st)
                                             : launch_bwd<PG_SEG_BOND_NODE, 4>(t, p, gr, st);
    case PG_SEG_BOND_POS: return launch_bwd<PG_SEG_BOND_POS, 4>(t, p, gr, st);
    case PG_SEG_TRIPLET: {
      // the per-source-atom rows (max_nlig x 259 floats) share the LDS with the per-wave tiles: 4 waves up to 64 atoms,
      // 2 waves up to the reference's maximum of 78 (and beyond, to 96)
      const bool op = gr->alpha && gr->S && gr->swn;
      if (tri_split) return triplet_bwd2_launch(t, p, gr, st);   // (Cdst k | v: one 1 KB row)
      if (op && gr->dlogit && gr->gfeat_v && t->max_nlig <= 64) return launch_bwd_split<PG_SEG_TRIPLET>(t, p, gr, st);
      if (t->max_nlig <= 64) return op ? launch_bwd<PG_SEG_TRIPLET, 4, true>(t, p, gr, st) : launch_bwd<PG_SEG_TRIPLET, 4>(t, p, gr, st);
      return op ? launch_bwd<PG_SEG_TRIPLET, 2, true>(t, p, gr, st) : launch_bwd<PG_SEG_TRIPLET, 2>(t, p, gr, st);
    }
    case PG_SEG_PHORE:
      // (the pharmacophore encoder, training: with the forward's softmax weights one pass; as a value pass + a key pass it measured
      //  slower -- 136.6 against 136.2 ms per training step -- and is not built in)
      return (gr->alpha && gr->S && gr->swn) ? launch_bwd<PG_SEG_PHORE, 4, true>(t, p, gr, st) : launch_bwd<PG_SEG_PHORE, 4>(t, p, gr, st);
  }
  set_error("pg_seg_attn_bwd: unknown mode %d", p->mode);
  return PG_ERR_ARG;
}
