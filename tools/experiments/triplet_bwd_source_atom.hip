// Adjoint of the bond-triplet attention (BondUpdateLayer, models/uni_denoiser.py:101-165; forward: triplet2.hip, training form)
// organised around the SOURCE ATOM, like the forward.
//
// All n-1 segments j->i of a source atom j read the same rows P[k->j].  seg_attn_bwd.hip gives a segment to a wave, so
// every wave gathers those rows again for every segment and the four waves of a workgroup merge their d P tiles through LDS in
// lockstep (two workgroup barriers per 16-row tile and path; 8.4 ms per launch on the config-5 batch).  Here a wave owns
// a ROW TILE (16 atoms k) of one source atom and walks the segments i = 0..n-1, once per MLP path:
//   * its P tile is copied to LDS once per pass (lane-private 16-byte slots), d P accumulates in registers and is stored once;
//   * what changes per segment -- dS or U (lane-fixed [c][h] matrices), the row Q[j->i], <S, dS> -- is read once per segment by
//     the waves of the atom together and parked in LDS in the lane-fixed order with rows padded to 65 floats: the SAME copy
//     serves the projection (B operand [c = 4g+r][h = m]) and the transposed product d z = M . coef^T (A operand
//     [c = m][h = 4g+ks]) without bank conflicts, so no per-tile transposition of U / dS is left;
//   * d b' accumulates lane-wise in registers (reduced once at the end of the kernel) instead of through an LDS transpose;
//   * per segment only d U and d Q need the other row tiles of the atom: partial tiles go through LDS, two workgroup
//     barriers per SEGMENT and pass.
// Two passes because of the register file: the VALU reaches only 256 of a wave's 512 registers, and d P + d b' + the hidden tile +
// its gradient of BOTH paths do not fit them (the one-pass form moved ~1100 values between the two halves per tile and still
// spilled ~200).  The value pass runs first and leaves d logit[row, head] of every segment in a scratch array the size of alpha;
// the key pass reads it back.  The angular features are computed in both passes (a few hundred VALU instructions per segment
// against ~170 MFMA); the geometry adjoint is linear in d feat, so each pass adds its own share to d x.
// A workgroup is 4 waves = 4 / T source atoms of one ligand with T = ceil(n / 16) row tiles each (T = 3: one atom, one
// wave idle); every group of a workgroup walks the same n-1 segments, so the barriers cost no skew.  Ligands of more than
// 64 atoms keep the segment-per-wave kernel (seg_attn_bwd.hip).
// Arithmetic, operand layouts and the one-pass softmax adjoint (alpha, S, swn handed over by the forward) are those of
// seg_attn_bwd.hip: lane l = (g = l>>4, m = l&15), hidden tiles in the transposed layout hid[tau][r] = hidden[c = 16 tau + 4g + r][row = m].
#include <type_traits>

#include "seg_common.h"

namespace pg {

// -DPG_BWD_PROF (tools/prof_bwd.sh): wave-cycle counters per section of the kernel; the product build carries none of this
#ifdef PG_BWD_PROF
__device__ unsigned long long g_tb_prof[16];
#define TB_PROF_DECL() long long _pacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; long long _t0 = __builtin_readcyclecounter()
#define TB_PROF(i) do { long long _t1 = __builtin_readcyclecounter(); _pacc[i] += _t1 - _t0; _t0 = _t1; } while (0)
#define TB_PROF_FLUSH() do { if (lane == 0) for (int _i = 0; _i < 12; ++_i) atomicAdd(&g_tb_prof[_i], (unsigned long long)_pacc[_i]); } while (0)
#else
#define TB_PROF_DECL() do {} while (0)
#define TB_PROF(i) do {} while (0)
#define TB_PROF_FLUSH() do {} while (0)
#endif

namespace {

constexpr float TB_LN2 = 0.69314718055994530942f;
constexpr int TB_ST = 19;                  // row stride of the transposed [c][row] tile (A-operand reads conflict-free per half wave)
constexpr int TB_WS = 20;                  // row stride of Wf[c][f] (B-operand reads [c = 4g+r][f = m] conflict-free)
constexpr int TB_FS = 17;
constexpr int TB_MAT = 32 * 65;            // a lane-fixed [32][64] matrix with rows padded to 65 floats
constexpr int TB_LDS_FLOATS = 2 * 1536 + 2 * 128 * TB_WS + 256 + 4 * TB_MAT + 4 * 128 + 4 * 16 + 4 * 128 + 4 * 64 + 4 * 128 * TB_ST + 4 * 2048 +
                              4 * (2 * 16 * TB_FS + 64);

}  // namespace

__global__ __launch_bounds__(256) void triplet_bwd_kernel(PgTopo t, PgSegAttn p, PgSegAttnGrad gr) {
  constexpr int ST = TB_ST, WS = TB_WS, FS = TB_FS;
  extern __shared__ __attribute__((aligned(16))) float lds_raw[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, m = lane & 15;

  float* q_ = lds_raw;
  float* const wf_k = q_; q_ += 1536;                 // first-layer feature weights, lane-fixed [step][tau][64]
  float* const wf_v = q_; q_ += 1536;
  float* const wfp_k = q_; q_ += 128 * WS;            // the same weights as Wf[c][f] for d feat = Wf^T . d hidden
  float* const wfp_v = q_; q_ += 128 * WS;
  float* const bkv = q_; q_ += 256;                   // b'_k | b'_v
  float* const slots = q_; q_ += 4 * TB_MAT;          // per group: dS (value pass) / U (key pass) of the current segment
  float* const sCall = q_; q_ += 4 * 128;             // per group: Q row of the pass's path
  float* const sDall = q_; q_ += 4 * 16;              // per wave: its share of <S, dS> per head
  float* const sQall = q_; q_ += 4 * 128;             // per wave: d Q of its rows
  int* const sSegAll = reinterpret_cast<int*>(q_); q_ += 4 * 64;   // per group: bond rows j -> i of the atom's segments
  float* const sTall = q_; q_ += 4 * 128 * ST;        // per wave: transposed tile; afterwards its d U tile for the merge
  float* const sP = q_ + wave * 2048; q_ += 4 * 2048; // per wave: its P tile, lane-private float4 slots [tau][lane]
  float* const sF = q_ + wave * (2 * 16 * FS + 64);   // per wave: features [row][f], d features [row][f], row scalars
  float* const sGF = sF + 16 * FS;
  float* const sR = sGF + 16 * FS;
  float* const sT = sTall + wave * 128 * ST;

  for (int i = tid; i < 1536; i += 256) { wf_k[i] = p.Wf_k[i]; wf_v[i] = p.Wf_v[i]; }
  bkv[tid] = tid < 128 ? p.ln_bk[tid] : p.ln_bv[tid - 128];
  for (int i = tid; i < 128 * 16; i += 256) {
    const int c = i >> 4, f = i & 15;
    const int src = ((f >> 2) * 8 + (c >> 4)) * 64 + (f & 3) * 16 + (c & 15);
    wfp_k[c * WS + f] = f < 12 ? p.Wf_k[src] : 0.f;
    wfp_v[c * WS + f] = f < 12 ? p.Wf_v[src] : 0.f;
  }
  for (int i = lane; i < 2 * 16 * FS; i += 64) sF[i] = 0.f;
  __syncthreads();

  // accumulated over the whole kernel: d Wf (c = 16 tau + 4g + r, f = m) and d b' lane-wise (c = 16 tau + 4g + r, summed over the
  // lane's row m; the 16 lanes of a DPP row are added up once, at the end)
  f4 gwf[2][8], gb[2][8];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int tq = 0; tq < 8; ++tq) { gwf[a][tq] = (f4){0.f, 0.f, 0.f, 0.f}; gb[a][tq] = (f4){0.f, 0.f, 0.f, 0.f}; }

  TB_PROF_DECL();
  for (int it = blockIdx.x; it < gr.n_tri_items; it += gridDim.x) {
    const int gi = gr.tri_items[2 * it], j0 = gr.tri_items[2 * it + 1];
    const int n = t.g_nlig[gi], lig0 = t.g_ctx_off[gi] + t.g_nph[gi];
    const int* const eid_g = t.eid + t.g_eid_off[gi];
    const int T = (n + 15) >> 4;
    auto process_item = [&](auto tt_tag) {
    constexpr int TT = decltype(tt_tag)::value, G = TT == 3 ? 1 : 4 / TT, NR = (32 + TT - 1) / TT;   // NR rows of U / dS per wave: 32, 16, 11, 8
    const int grp = wave / TT, ti = wave - grp * TT;
    const int lj = j0 + grp;
    const bool act = grp < G && lj < n;
    float* const slot = slots + grp * TB_MAT;
    float* const sCg = sCall + grp * 128;
    int* const sSeg = sSegAll + grp * 64;
    const int k_m = 16 * ti + m;                                  // the lane's row: atom k of the ligand
    const bool rowok = act && k_m < n && k_m != lj;
    const int e_m = rowok ? eid_g[k_m * n + lj] : 0;              // bond row k -> j
    float xk[3] = {0.f, 0.f, 0.f}, xj[3] = {0.f, 0.f, 0.f}, gxk[3] = {0.f, 0.f, 0.f}, gxj[3] = {0.f, 0.f, 0.f};
    if (act) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        xj[c] = p.x[(lig0 + lj) * 3 + c];
        if (rowok) xk[c] = p.x[(lig0 + k_m) * 3 + c];
      }
      if (ti == 0 && lane < n - 1) sSeg[lane] = eid_g[lj * n + (lane < lj ? lane : lane + 1)];
    }
    __syncthreads();
    TB_PROF(0);   // item setup

    auto run_pass = [&](auto kp_tag) {
      constexpr bool KP = decltype(kp_tag)::value;
      constexpr int PI = KP ? 0 : 1;
      const float* const bp = bkv + (KP ? 0 : 128);
      const float* const wf = KP ? wf_k : wf_v;
      const float* const wfp = KP ? wfp_k : wfp_v;
      f4 dP[8];
      {
        const float* pr = (KP ? p.Csrc_k : p.Csrc_v) + (size_t)e_m * p.ld_csrc + 4 * g;
#pragma unroll
        for (int tq = 0; tq < 8; ++tq) {
          dP[tq] = (f4){0.f, 0.f, 0.f, 0.f};
          *reinterpret_cast<f4*>(sP + (tq * 64 + lane) * 4) = rowok ? *reinterpret_cast<const f4*>(pr + 16 * tq) : (f4){0.f, 0.f, 0.f, 0.f};
        }
      }
      // Everything a segment needs from global memory is requested one segment AHEAD (with one wave per SIMD nothing else hides
      // an HBM round trip): the wave's rows of dS / U (+ S for <S, dS>) and its block of the Q row wait in registers until the
      // group has finished the current segment; the lane's softmax weights / d logit, d swn and x_i are carried to the next turn.
      struct Row { f4 cD, cK; float gswn, swn, xd[3]; int seg, ci, il; };
      auto fetch_row = [&](int q, Row& R) {
        R.il = q < lj ? q : q + 1;
        R.seg = sSeg[q];
        R.ci = lig0 + R.il;
#pragma unroll
        for (int c = 0; c < 3; ++c) R.xd[c] = p.x[R.ci * 3 + c];
        // value pass: softmax weights of the forward; key pass: d logit left by the value pass -- rows 4g + r at head m, and the
        // lane's row at heads 4g .. 4g + 3
        const float* const crow = (KP ? gr.rowbuf : gr.alpha) + (size_t)R.seg * gr.alpha_rows * 16;
        R.cD = (f4){0.f, 0.f, 0.f, 0.f};
        R.cK = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * ti + 4 * g + r;
          if (row < n && row != R.il && row != lj) R.cD[r] = crow[row * 16 + m];
        }
        if (rowok && k_m != R.il) R.cK = *reinterpret_cast<const f4*>(crow + k_m * 16 + 4 * g);
        R.gswn = KP ? 0.f : gr.gswn[(size_t)R.seg * 16 + m];
        R.swn = KP ? 0.f : gr.swn[(size_t)R.seg * 16 + m];
      };
      auto issue = [&](int q, float (&Lm)[NR], float (&Ls)[NR], float (&cq)[2]) {
        const int seg = sSeg[q];
        const float* Mg = (KP ? p.U : gr.gS) + (size_t)seg * 2048 + lane;
        const float* Sg = gr.S + (size_t)seg * 2048 + lane;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
          const int idx = ti * NR + i, idc = idx < 32 ? idx : 31;       // (T = 3: the last wave's 11th row repeats row 31)
          Lm[i] = Mg[idc * 64];
          if (!KP) Ls[i] = idx < 32 ? Sg[idc * 64] : 0.f;
        }
        const float* Cq = (KP ? p.Cdst_k : p.Cdst_v) + (size_t)seg * p.ld_cdst + lane;
        cq[0] = Cq[(TT == 1 ? 0 : (ti & 1)) * 64];
        cq[1] = TT == 1 ? Cq[64] : 0.f;
      };
      auto commit = [&](const float (&Lm)[NR], const float (&Ls)[NR], const float (&cq)[2]) {
        float dpart = 0.f;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
          const int idx = ti * NR + i, idc = idx < 32 ? idx : 31;
          slot[idc * 65 + lane] = Lm[i];
          if (!KP) dpart = fmaf(Ls[i], Lm[i], dpart);
        }
        if (TT == 1) {
          sCg[lane] = cq[0];
          sCg[64 + lane] = cq[1];
        } else if (ti < 2) {
          sCg[ti * 64 + lane] = cq[0];
        }
        if (!KP) {
          dpart += __shfl_xor(dpart, 16);
          dpart += __shfl_xor(dpart, 32);
          if (lane < 16) sDall[wave * 16 + lane] = dpart;
        }
      };
      Row cur;
      if (act && n > 1) {
        float Lm[NR], Ls[NR], cq[2];
        issue(0, Lm, Ls, cq);
        fetch_row(0, cur);
        commit(Lm, Ls, cq);
      }
      __syncthreads();
      TB_PROF(1);   // pass prologue: P tile, first segment

      for (int q = 0; q < n - 1; ++q) {
        float Lm[NR], Ls[NR], cq[2];
        Row nxt;
        if (act) {
          const int qn = q + 1 < n - 1 ? q + 1 : q;              // (last turn: the current segment once more, unused)
          issue(qn, Lm, Ls, cq);
          fetch_row(qn, nxt);
          TB_PROF(2);   // requests for the next segment
          // (opaque copies of the lane coordinates: the weight tables' LDS addresses must not look loop-invariant, or the compiler
          //  keeps all the table words in registers across the segment loop and spills the accumulators instead)
          int lw = lane;
          asm volatile("" : "+v"(lw));
          const int gw = lw >> 4, mw = lw & 15;
          const int il = cur.il, seg = cur.seg, ci = cur.ci;
          const f4 cD = cur.cD, cK = cur.cK;
          const bool valid = rowok && k_m != il;
          // angular features of the lane's row: f = 4 st + g
          float u[3], v[3] = {0.f, 0.f, 0.f}, theta = 0.f;
#pragma unroll
          for (int c = 0; c < 3; ++c) u[c] = xj[c] - cur.xd[c];
          if (valid) {
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] = xk[c] - cur.xd[c];
            const float a = u[0] * v[0] + u[1] * v[1] + u[2] * v[2];
            const float c0 = u[1] * v[2] - u[2] * v[1], c1 = u[2] * v[0] - u[0] * v[2], c2 = u[0] * v[1] - u[1] * v[0];
            theta = atan2f(sqrtf(c0 * c0 + c1 * c1 + c2 * c2), a);
          }
          float feat[3];
#pragma unroll
          for (int st = 0; st < 3; ++st) {
            const int f = 4 * st + g;
            float fv = sincos_bounded(theta * kAngFreq[f], f >= 6);
            fv = f == 0 ? theta : fv;
            feat[st] = (valid && f != 11) ? fv : 0.f;
            sF[m * FS + f] = feat[st];
          }
          f4 hid[8];
          {
            const float* cd = sCg + 4 * gw;
#pragma unroll
            for (int tq = 0; tq < 8; ++tq)
              hid[tq] = *reinterpret_cast<const f4*>(sP + (tq * 64 + lw) * 4) + *reinterpret_cast<const f4*>(cd + 16 * tq);
#pragma unroll
            for (int st = 0; st < 3; ++st)
#pragma unroll
              for (int tq = 0; tq < 8; ++tq) hid[tq] = mfma16(wf[(st * 8 + tq) * 64 + lw], feat[st], hid[tq]);
          }
          float rs, sg;
          ln_stats(hid, rs, sg);
          TB_PROF(3);   // features, hidden tile, statistics
          // y[row = 4g+r][h = m] = ReLU(hidden + b' sigma) . M[:, h]  (unscaled by rstd)
          f4 y;
          {
            f4 yy[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) yy[r] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int tq = 0; tq < 8; ++tq) {
              const f4 bt = *reinterpret_cast<const f4*>(bp + 16 * tq + 4 * gw);
#pragma unroll
              for (int r = 0; r < 4; ++r)
                yy[r] = mfma16(fmaxf(fmaf(bt[r], sg, hid[tq][r]), 0.f), slot[(tq * 4 + r) * 65 + lane], yy[r]);
            }
            y = (yy[0] + yy[1]) + (yy[2] + yy[3]);
          }
          if (!KP) {
            // d alpha[row, h] = rstd_v * y + d swn;  d logit = ln2 * alpha * (d alpha - D), kept for the key pass
            float Dm = 0.f;
#pragma unroll
            for (int w2 = 0; w2 < TT; ++w2) Dm += sDall[(grp * TT + w2) * 16 + m];
            Dm = fmaf(cur.swn, cur.gswn, Dm);
            float* const drow = gr.rowbuf + (size_t)seg * gr.alpha_rows * 16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int row = 16 * ti + 4 * g + r;
              const float ga = fmaf(y[r], __shfl(rs, 4 * g + r), cur.gswn);
              if (row < n) drow[row * 16 + m] = TB_LN2 * cD[r] * (ga - Dm);
            }
          }
          TB_PROF(4);   // projection (+ d logit)
          // coefficient of y in the loss, rows 4g + r: cD (value path: alpha; key path: d logit).  d rstd of the row = sum_h coef * y
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float sv = row16_total(cD[r] * y[r]);
            if (m == r) sR[4 * g + r] = sv;
          }
          if (KP) {
            if (g == 0) sR[16 + m] = rs;
            // z^T tile to LDS for d U (contracted over the rows)
#pragma unroll
            for (int tq = 0; tq < 8; ++tq) {
              const f4 bt = *reinterpret_cast<const f4*>(bp + 16 * tq + 4 * gw);
#pragma unroll
              for (int r = 0; r < 4; ++r) sT[(16 * tq + 4 * g + r) * ST + m] = fmaxf(fmaf(bt[r], sg, hid[tq][r]), 0.f);
            }
          }
          wave_lds_sync();
          const float grs = sR[m];
          f4 gU[8];
          if (KP) {
#pragma unroll
            for (int tq = 0; tq < 8; ++tq) gU[tq] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {                      // contraction index (row) of this k-step: 4g + ks
              const float b = cD[ks] * sR[16 + 4 * g + ks];
#pragma unroll
              for (int tq = 0; tq < 8; ++tq) gU[tq] = mfma16(sT[(16 * tq + m) * ST + 4 * g + ks], b, gU[tq]);
            }
          }
          TB_PROF(5);   // row sums, z tile, d U
          // d z^T[c, row] = sum_h M[c, h] * coef[row, h] * rstd[row]: M[c = 16 tau + m][h = 4g + ks] out of the lane-fixed copy
          f4 gz[8];
#pragma unroll
          for (int tq = 0; tq < 8; ++tq) gz[tq] = (f4){0.f, 0.f, 0.f, 0.f};
          {
            const float* const ma = slot + (m & 3) * 65 + (m >> 2) * 16 + 4 * g;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
              const float b = cK[ks] * rs;
#pragma unroll
              for (int tq = 0; tq < 8; ++tq) gz[tq] = mfma16(ma[tq * 4 * 65 + ks], b, gz[tq]);
            }
          }
          // folded LayerNorm backward: z = ReLU(h + b' sigma), sigma = sqrt(var), rstd = 1 / sigma, var = mean(h^2) + eps
          float s1 = 0.f;
#pragma unroll
          for (int tq = 0; tq < 8; ++tq) {
            const f4 bt = *reinterpret_cast<const f4*>(bp + 16 * tq + 4 * gw);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              gz[tq][r] = fmaf(bt[r], sg, hid[tq][r]) > 0.f ? gz[tq][r] : 0.f;
              s1 = fmaf(gz[tq][r], bt[r], s1);
              gb[PI][tq][r] = fmaf(gz[tq][r], sg, gb[PI][tq][r]);
            }
          }
          s1 += __shfl_xor(s1, 16);
          s1 += __shfl_xor(s1, 32);
          const float gvar = 0.5f * rs * s1 - 0.5f * grs * rs * rs * rs;
          wave_lds_sync();                                        // the products over sT (z) are done
#pragma unroll
          for (int tq = 0; tq < 8; ++tq) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              gz[tq][r] = fmaf(gvar * (1.f / 64.f), hid[tq][r], gz[tq][r]);       // d hidden
              sT[(16 * tq + 4 * g + r) * ST + m] = gz[tq][r];
            }
            dP[tq] += gz[tq];
          }
          wave_lds_sync();
          TB_PROF(6);   // d z, LayerNorm adjoint, d hidden tile
          // d feat[row, f] = sum_c d hidden[c, row] * Wf[c, f]
          f4 gfeat;
          {
            f4 gfp[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) gfp[r] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int tq = 0; tq < 8; ++tq)
#pragma unroll
              for (int r = 0; r < 4; ++r) gfp[r] = mfma16(gz[tq][r], wfp[(16 * tq + 4 * gw + r) * WS + mw], gfp[r]);
            gfeat = (gfp[0] + gfp[1]) + (gfp[2] + gfp[3]);
          }
          // d Wf[c, f] += sum_row d hidden[c, row] * feat[row, f]
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
            const float bf = sF[(4 * g + ks) * FS + m];
#pragma unroll
            for (int tq = 0; tq < 8; ++tq) gwf[PI][tq] = mfma16(sT[(16 * tq + m) * ST + 4 * g + ks], bf, gwf[PI][tq]);
          }
          // d Q of the wave's rows: lane owns channels lane and lane + 64
          {
            float a0 = 0.f, a1 = 0.f;
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) { a0 += sT[lane * ST + rr]; a1 += sT[(64 + lane) * ST + rr]; }
            sQall[wave * 128 + lane] = a0;
            sQall[wave * 128 + 64 + lane] = a1;
          }
          TB_PROF(7);   // d feat, d Wf, d Q
          // ---------------- geometry: this path's d feat -> positions ----------------
#pragma unroll
          for (int r = 0; r < 4; ++r) sGF[(4 * g + r) * FS + m] = gfeat[r];
          wave_lds_sync();                                        // (also: sT is free for the d U tile)
          float gxi[3] = {0.f, 0.f, 0.f};
          if (g == 0 && valid && gr.gx) {
            // theta = atan2(|u x v|, u.v), u = x_j - x_i, v = x_k - x_i;  the derivative of sin(w theta) is w * feature[f + 5], of
            // cos(w theta) it is -w * feature[f - 5]: both already sit in the row's feature tile
            const float* gf = sGF + m * FS;
            const float* ff = sF + m * FS;
            float gth = gf[0];
#pragma unroll
            for (int f = 1; f < 6; ++f) gth += kAngFreq[f] * (gf[f] * ff[f + 5] - gf[f + 5] * ff[f]);
            const float a = u[0] * v[0] + u[1] * v[1] + u[2] * v[2];
            const float cr[3] = {u[1] * v[2] - u[2] * v[1], u[2] * v[0] - u[0] * v[2], u[0] * v[1] - u[1] * v[0]};
            const float b = sqrtf(cr[0] * cr[0] + cr[1] * cr[1] + cr[2] * cr[2]);
            const float den = a * a + b * b;
            if (den > 0.f) {
              const float ka = -b / den * gth;                          // d theta / d a
              const float kb = b > 0.f ? a / den * gth / b : 0.f;       // d theta / d b, times 1/b of d b = c . dc / b
              const float vxc[3] = {v[1] * cr[2] - v[2] * cr[1], v[2] * cr[0] - v[0] * cr[2], v[0] * cr[1] - v[1] * cr[0]};
              const float cxu[3] = {cr[1] * u[2] - cr[2] * u[1], cr[2] * u[0] - cr[0] * u[2], cr[0] * u[1] - cr[1] * u[0]};
#pragma unroll
              for (int c = 0; c < 3; ++c) {
                const float gu = ka * v[c] + kb * vxc[c];
                const float gv = ka * u[c] + kb * cxu[c];
                gxj[c] += gu;
                gxi[c] -= gu + gv;
                gxk[c] += gv;
              }
            }
          }
          if (gr.gx) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              const float sv = wave_sum(gxi[c]);
              if (lane == 0) atomicAdd(gr.gx + ci * 3 + c, sv);
            }
          }
          if (KP) {                                               // the wave's d U tile for the merge (lane-fixed order)
#pragma unroll
            for (int tq = 0; tq < 8; ++tq)
#pragma unroll
              for (int r = 0; r < 4; ++r) sT[(tq * 4 + r) * 64 + lane] = gU[tq][r];
          }
          TB_PROF(8);   // geometry, d U tile
        }  // act
        __syncthreads();
        TB_PROF(9);   // barrier 1
        if (act) {
          // d U and d Q of the segment: the T row tiles added up, every wave stores its share
          if (KP) {
#pragma unroll
            for (int i = 0; i < NR; ++i) {
              const int idx = ti * NR + i;
              if (idx < 32) {
                float sv = 0.f;
#pragma unroll
                for (int w2 = 0; w2 < TT; ++w2) sv += sTall[(grp * TT + w2) * 128 * ST + idx * 64 + lane];
                gr.gU[(size_t)cur.seg * 2048 + idx * 64 + lane] = sv;
              }
            }
          }
#pragma unroll
          for (int idx = 0; idx < 2; ++idx) {
            if (TT == 1 || ti == idx) {
              float sv = 0.f;
#pragma unroll
              for (int w2 = 0; w2 < TT; ++w2) sv += sQall[(grp * TT + w2) * 128 + idx * 64 + lane];
              (KP ? gr.gCdst_k : gr.gCdst_v)[(size_t)cur.seg * gr.ld_gcdst + idx * 64 + lane] = sv;
            }
          }
          if (q + 1 < n - 1) commit(Lm, Ls, cq);
        }
        TB_PROF(10);  // merge, next segment to LDS
        __syncthreads();
        TB_PROF(11);  // barrier 2
        cur = nxt;
      }  // segments

      if (rowok) {
        float* gp = (KP ? gr.gCsrc_k : gr.gCsrc_v) + (size_t)e_m * gr.ld_gcsrc + 4 * g;
#pragma unroll
        for (int tq = 0; tq < 8; ++tq) *reinterpret_cast<f4*>(gp + 16 * tq) = dP[tq];
      }
    };  // run_pass

    run_pass(std::false_type{});                                  // value path: d logit of every segment
    __threadfence();                                              // (d logit is read back by other lanes of the wave)
    __syncthreads();
    run_pass(std::true_type{});                                   // key path

    if (act && gr.gx) {
      if (g == 0 && rowok) {
#pragma unroll
        for (int c = 0; c < 3; ++c) atomicAdd(gr.gx + (lig0 + k_m) * 3 + c, gxk[c]);
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float sv = wave_sum(gxj[c]);
        if (lane == 0) atomicAdd(gr.gx + (lig0 + lj) * 3 + c, sv);
      }
    }
    __syncthreads();                                              // (the segment ids of the next item overwrite sSeg)
    };  // process_item
    switch (T) {
      case 1: process_item(std::integral_constant<int, 1>{}); break;
      case 2: process_item(std::integral_constant<int, 2>{}); break;
      case 3: process_item(std::integral_constant<int, 3>{}); break;
      default: process_item(std::integral_constant<int, 4>{}); break;
    }
  }  // items

  TB_PROF_FLUSH();
  // ---------------- flush the weight-gradient accumulators ----------------
  if (m < 12) {
#pragma unroll
    for (int tq = 0; tq < 8; ++tq)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int idx = ((m >> 2) * 8 + tq) * 64 + (m & 3) * 16 + 4 * g + r;       // lane-fixed layout of the forward weights
        atomicAdd(gr.gWf_k + idx, gwf[0][tq][r]);
        atomicAdd(gr.gWf_v + idx, gwf[1][tq][r]);
      }
  }
#pragma unroll
  for (int tq = 0; tq < 8; ++tq)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float sk = row16_total(gb[0][tq][r]), sv = row16_total(gb[1][tq][r]);
      if (m == 0) {
        atomicAdd(gr.gbk + 16 * tq + 4 * g + r, sk);
        atomicAdd(gr.gbv + 16 * tq + 4 * g + r, sv);
      }
    }
}

// -1: not applicable (no work list, two-pass form, or a ligand of more than 64 atoms) -- the caller runs seg_attn_bwd.hip.
// gr->rowbuf is the d logit scratch here: n_bond * alpha_rows * 16 floats (the shape of alpha)
int launch_triplet_bwd(const PgTopo* t, const PgSegAttn* p, const PgSegAttnGrad* gr, hipStream_t st) {
  if (!gr->tri_items || gr->n_tri_items <= 0 || !gr->alpha || !gr->S || !gr->swn || t->max_nlig > 64 || t->max_nlig < 2) return -1;
  if ((((size_t)p->Csrc_k | (size_t)p->Csrc_v | (size_t)gr->gCsrc_k | (size_t)gr->gCsrc_v | (size_t)gr->alpha | (size_t)gr->rowbuf) & 15) != 0 ||
      ((p->ld_csrc | gr->ld_gcsrc) & 3) != 0)
    return -1;
  const size_t lds = (size_t)TB_LDS_FLOATS * sizeof(float);
  if (int rc = reserve_lds(reinterpret_cast<const void*>(triplet_bwd_kernel), lds, "pg_seg_attn_bwd(triplet)")) return rc;
  int blocks = gr->n_tri_items < kNumCU ? gr->n_tri_items : kNumCU;
  hipLaunchKernelGGL(triplet_bwd_kernel, dim3(blocks), dim3(256), lds, st, *t, *p, *gr);
  return check_launch("pg_seg_attn_bwd(triplet)");
}

}  // namespace pg

#ifdef PG_BWD_PROF
extern "C" int pg_debug_tb_prof(unsigned long long* out, int reset) {
  hipMemcpyFromSymbol(out, HIP_SYMBOL(pg::g_tb_prof), sizeof(unsigned long long) * 16);
  if (reset) { unsigned long long z[16] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(pg::g_tb_prof), z, sizeof(z)); }
  return 0;
}
#endif
