// Adjoint of the bond-triplet attention (BondUpdateLayer, models/uni_denoiser.py:101-165; training path,
// PhoreDiff.compute_loss -> models/diffusion.py:249-352) with the CHANNELS of a row tile split over the waves of a workgroup.
//
// seg_attn_bwd.hip gives a 16-row tile to ONE wave: every product that contracts over the rows (d U, d Wf, d b', d P, d Cdst) needs the
// wave's 128 x 16 tile transposed through LDS, the waves' d P tiles are merged through LDS, and a wave that holds two MLP paths needs all
// 512 registers -- one wave per SIMD, every LDS round trip exposed (0.22 of the fp32 matrix peak).
// Here a workgroup owns a source atom j like there, but ALL its waves work on the same (segment j -> i, 16-row tile) and wave w holds the
// channels [CW w, CW w + CW) of both paths (CW = 128 / waves).  Then
//   * everything that is local to a channel stays in that wave's registers for the whole source atom: the rows P[k -> j] (read ONCE per
//     atom instead of once per segment), their gradient d P (accumulated over the atom's n - 1 segments in registers, stored once, no
//     merge), d Cdst, d b', d Wf, d U;
//   * what contracts over the channels crosses the waves as partial sums through LDS: the two LayerNorm row sums, the logits / value
//     projection y = z^T . U of ALL tiles of the segment at once, the row sum of the LayerNorm adjoint and d feat = d hidden^T . Wf:
//     four workgroup barriers per segment + one per tile;
//   * nothing of the forward is read back: the softmax weights are recomputed from the logits of all tiles of the segment -- wave w takes
//     16 / waves heads and leaves rstd d logit and rstd alpha in LDS in a layout both MFMA operand forms read -- so D[h] = sum_r alpha d alpha
//     needs neither S nor alpha from HBM, and there is no value pass / key pass scratch;
//   * the LayerNorm adjoint needs no row sum of d logit x y: with z = ReLU(pre), d rstd = (1 / rstd) sum_c z d pre, so
//     d var = (rstd / 2) sum_c d pre (b' - rstd z) is one partial sum per wave;
//   * the angular features of the next segment and the geometry adjoint of the previous one run on different waves between the same two
//     barriers; the segment's operands (U, dS, the Cdst row) land in a double LDS stage by LDS-DMA one segment ahead.
// Per-wave transposes shrink to the wave's own CW x 16 block.  Lane l = (g = l >> 4, m = l & 15); 16x16x4 maps as in seg_attn.hip.
// Instances: 8 waves x 16 channels for ligands of up to 32 atoms (two waves per SIMD, 249 registers, no spills), 4 waves x 32 channels for
// up to 64 atoms (up to 512 registers).  Measurements, and what faults on the way: profiles/r05_triplet_adjoint_channel_split.txt.
#include "seg_common.h"

namespace pg {

// -DPG_TB2_PROF (tools/prof_tb2.sh): per-wave clock counters of the kernel's sections, summed into g_tb2_prof; the product build has none
#ifdef PG_TB2_PROF
__device__ unsigned long long g_tb2_prof[16];
#define TB2_STAMP(slot) do { const unsigned long long tb2_now = __builtin_readcyclecounter(); tb2_acc[slot] += tb2_now - tb2_last; tb2_last = tb2_now; } while (0)
#define TB2_WAITVM() __builtin_amdgcn_s_waitcnt(0)
#else
#define TB2_STAMP(slot) do { } while (0)
#define TB2_WAITVM() do { } while (0)
#endif

namespace {

constexpr float TB2_LN2 = 0.69314718055994530942f;

__device__ __forceinline__ float tb2_sincos(float arg, bool want_cos) {     // as seg_attn_bwd.hip sincos_bounded
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

template <int NW, int MAXT>
struct Tb2Layout {
  static constexpr int NB = 8 / NW, CW = 16 * NB, ROWS = 16 * MAXT;
  static constexpr int YT = 288;                                  // floats of one wave's partial y tile: lane (g, m) at 72 g + 4 m (bank spread
                                                                  // for the softmax lanes, which read it by (row, head))
  static constexpr int QT = 320;                                  // a [16 rows][16 heads] tile with rows 20 floats apart (b128 reads of a row)
  // floats
  static constexpr int o_op = 0;                                  // operands of a segment, landed by LDS-DMA: U [2048] | dS [2048] | Cdst k|v [256]
                                                                  // (first: the DMA's LDS base register holds 16 bits, the stage must end below 64 KB)
  static constexpr int o_b = o_op + 2 * 4352;                     // (two stages: a segment's operands are read where they are used)   b'_k[128] | b'_v[128]
  static constexpr int o_stat = o_b + 256;                        // [2 paths][MAXT][16 rows][NW]
  static constexpr int o_y = o_stat + 2 * MAXT * 16 * NW;         // [2][MAXT][NW][YT]
  static constexpr int o_rs = o_y + 2 * MAXT * NW * YT;           // rstd of every row [2][ROWS]
  static constexpr int o_q = o_rs + 2 * ROWS;                     // rstd_k d logit | rstd_v alpha  [2][MAXT][QT]
  static constexpr int o_s1 = o_q + 2 * MAXT * QT;                // [2 parity][2 paths][16][NW]
  static constexpr int o_df = o_s1 + 2 * 2 * 16 * NW;             // d feat partial sums of a segment's tiles [MAXT][NW][64][4]
  static constexpr int o_feat = o_df + MAXT * NW * 256;           // [3 segments in flight][MAXT][16][17]
  static constexpr int o_gq = o_feat + 3 * MAXT * 16 * 17;        // d feat tiles of the geometry steps [MAXT][16][17]
  static constexpr int o_gij = o_gq + MAXT * 16 * 17;             // d x_j | d x_i of a geometry step [MAXT][8]
  static constexpr int o_x = o_gij + MAXT * 8;                    // x_k [ROWS][3]
  static constexpr int o_accx = o_x + ROWS * 3;                   // d x [ROWS][3]
  static constexpr int o_wave = o_accx + ROWS * 3;                // per wave: tT [CW][17]
  static constexpr int PW = CW * 17 + 3;                          // (odd tile stride between the waves)
  static constexpr int total = o_wave + NW * PW;
};

template <int CTRL>
__device__ __forceinline__ float tb2_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float tb2_row16_max(float v) {
  v = fmaxf(v, tb2_dpp<0xB1>(v));
  v = fmaxf(v, tb2_dpp<0x4E>(v));
  v = fmaxf(v, tb2_dpp<0x141>(v));
  v = fmaxf(v, tb2_dpp<0x140>(v));
  return v;
}

typedef int tb2_i4 __attribute__((ext_vector_type(4)));
// raw buffer descriptor over `bytes` at `base` (wave-uniform)
__device__ __forceinline__ tb2_i4 tb2_desc(const void* base, unsigned bytes) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(base);
  tb2_i4 d;
  d[0] = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu));
  d[1] = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu));
  d[2] = __builtin_amdgcn_readfirstlane((int)bytes);
  d[3] = 0x00020000;
  return d;
}
// one 1 KB piece HBM -> LDS without registers (as gemm_stream.hip st_dma): lane l's 16 bytes from base + voff + soff land at lds_dst + 16 l
__device__ __forceinline__ void tb2_dma(unsigned lds_dst, unsigned voff, tb2_i4 desc, unsigned soff) {
  asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :: "s"(lds_dst), "v"(voff), "s"(desc), "s"(soff) : "memory");
}

}  // namespace

// (-DPG_TB2_MINWAVES=2 / 3: the occupancy experiments of profiles/r05_triplet_adjoint_channel_split.txt; the product build leaves it at 1)
#ifndef PG_TB2_MINWAVES
#define PG_TB2_MINWAVES 1
#endif
template <int NW, int MAXT>
__global__ __launch_bounds__(64 * NW, PG_TB2_MINWAVES) void triplet_bwd2_kernel(PgTopo t, PgSegAttn p, PgSegAttnGrad gr, int nt_lo) {
  using Ly = Tb2Layout<NW, MAXT>;
  constexpr int NB = Ly::NB, CW = Ly::CW;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, m = lane & 15;
  const int c0 = CW * wave;                       // the wave's first channel
  float* const sB = lds + Ly::o_b;
  float* const sStat = lds + Ly::o_stat;
  float* const sY = lds + Ly::o_y;
  float* const sS1 = lds + Ly::o_s1;
  float* const sDf = lds + Ly::o_df;
  float* const sFeat = lds + Ly::o_feat;
  float* const sX = lds + Ly::o_x;
  float* const sAccX = lds + Ly::o_accx;
  float* const sRs = lds + Ly::o_rs;
  float* const sQ = lds + Ly::o_q;
  float* const sGq = lds + Ly::o_gq;
  float* const sGij = lds + Ly::o_gij;
  float* const tT = lds + Ly::o_wave + wave * Ly::PW;
  float* const sOp = lds + Ly::o_op;
  // the 17 one-KB pieces of a segment's operands (8 of U, 8 of d S, the Cdst row k | v), dealt out over the waves; `lds` is the kernel's
  // only LDS object, so a float offset into it is the LDS address
  auto stage_operands = [&](int seg_next, int ob) {
    const tb2_i4 dU = tb2_desc(p.U + (size_t)seg_next * 2048, 8192), dM = tb2_desc(gr.gS + (size_t)seg_next * 2048, 8192);
    const tb2_i4 dC = tb2_desc(p.Cdst_k + (size_t)seg_next * p.ld_cdst, 1024);
    for (int pc = __builtin_amdgcn_readfirstlane(wave); pc < 17; pc += NW) {
      const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((Ly::o_op + ob * 4352 + pc * 256) * 4);
      const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane((pc & 7) * 1024);
      if (pc < 8) tb2_dma(dst, 16u * lane, dU, so);
      else if (pc < 16) tb2_dma(dst, 16u * lane, dM, so);
      else tb2_dma(dst, 16u * lane, dC, 0u);
    }
  };
  constexpr int YT = Ly::YT, QT = Ly::QT, ROWS = Ly::ROWS;

  for (int i = tid; i < 128; i += blockDim.x) { sB[i] = p.ln_bk[i]; sB[128 + i] = p.ln_bv[i]; }
  for (int i = tid; i < 3 * MAXT * 16 * 17; i += blockDim.x) sFeat[i] = 0.f;      // (columns 12..16 stay zero)
  __syncthreads();

  // kernel-long operands of the wave's channels
  //   wfA[path][st][tb]: A operand of hidden^T = Wf . feat^T   (lane-fixed forward layout [step][tau][lane])
  //   wfB[path][tb][r] : B operand of d feat = d hidden^T . Wf: Wf[c = c0 + 16 tb + 4g + r][f = m]
  //   bC[path][tb]     : b'[c0 + 16 tb + 4g + r]
  float wfA[2][3][NB];
  f4 wfB[2][NB], bC[2][NB];
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const float* wf = a == 0 ? p.Wf_k : p.Wf_v;
#pragma unroll
    for (int tb = 0; tb < NB; ++tb) {
      const int tq = NB * wave + tb;
#pragma unroll
      for (int st = 0; st < 3; ++st) wfA[a][st][tb] = wf[(st * 8 + tq) * 64 + lane];
#pragma unroll
      for (int r = 0; r < 4; ++r)      // Wf[c][f] = wf[((f >> 2) * 8 + (c >> 4)) * 64 + (f & 3) * 16 + (c & 15)], c & 15 = 4g + r
        wfB[a][tb][r] = m < 12 ? wf[((m >> 2) * 8 + tq) * 64 + (m & 3) * 16 + 4 * g + r] : 0.f;
      bC[a][tb] = *reinterpret_cast<const f4*>(sB + 128 * a + c0 + 16 * tb + 4 * g);
    }
  }
  f4 gWf[2][NB], gB[2][NB];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int tb = 0; tb < NB; ++tb) { gWf[a][tb] = (f4){0.f, 0.f, 0.f, 0.f}; gB[a][tb] = (f4){0.f, 0.f, 0.f, 0.f}; }

  // angular features of tile tt of the segment j -> i (rows k = 16 tt + m), feature f = 4 st + g  -> sFeat[par][tt][m][f]
  auto features = [&](int par, int tt, int n, int li, int lj, const float (&xi)[3], const float (&xj)[3]) {
    const int k = 16 * tt + m;
    const bool valid = k < n && k != li && k != lj;
    float theta = 0.f;
    if (valid) {
      float u[3], v[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) { u[c] = xj[c] - xi[c]; v[c] = sX[k * 3 + c] - xi[c]; }
      const float a = u[0] * v[0] + u[1] * v[1] + u[2] * v[2];
      const float cx = u[1] * v[2] - u[2] * v[1], cy = u[2] * v[0] - u[0] * v[2], cz = u[0] * v[1] - u[1] * v[0];
      theta = atan2f(sqrtf(cx * cx + cy * cy + cz * cz), a);
    }
    float* dst = sFeat + ((par * MAXT + tt) * 16 + m) * 17;
#pragma unroll
    for (int st = 0; st < 3; ++st) {
      const int f = 4 * st + g;
      float v = tb2_sincos(theta * kAngFreq[f], f >= 6);
      v = f == 0 ? theta : v;
      dst[f] = (valid && f != 11) ? v : 0.f;
    }
  };

  // geometry adjoint of tile tt of a finished segment j -> i: d feat (the waves' partial sums) -> d theta -> d x_k (sAccX rows, owned by
  // the tile), d x_j | d x_i (one slot per tile in sGij, added up by one thread behind the next barrier).  The tiles of a segment run side by
  // side on different waves, next to the waves that compute the next segment's features.
  auto geometry = [&](int tt, int fb, int n, int li, int lj, const float (&xi)[3], const float (&xj)[3]) {
    float* const qT = sGq + tt * 16 * 17;
    f4 df = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w2 = 0; w2 < NW; ++w2) df += *reinterpret_cast<const f4*>(sDf + ((tt * NW + w2) * 64 + lane) * 4);
#pragma unroll
    for (int r = 0; r < 4; ++r) qT[(4 * g + r) * 17 + m] = df[r];      // d feat[row 4g + r][f = m]
    wave_lds_sync();
    float gsum[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int k = 16 * tt + m;
    if (g == 0 && k < n && k != li && k != lj) {
      const float* gf = qT + m * 17;
      const float* ff = sFeat + ((fb * MAXT + tt) * 16 + m) * 17;
      float gth = gf[0];
#pragma unroll
      for (int f = 1; f < 6; ++f) gth += kAngFreq[f] * (gf[f] * ff[f + 5] - gf[f + 5] * ff[f]);
      float u[3], v[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) { u[c] = xj[c] - xi[c]; v[c] = sX[k * 3 + c] - xi[c]; }
      const float a = u[0] * v[0] + u[1] * v[1] + u[2] * v[2];
      const float cr[3] = {u[1] * v[2] - u[2] * v[1], u[2] * v[0] - u[0] * v[2], u[0] * v[1] - u[1] * v[0]};
      const float b = sqrtf(cr[0] * cr[0] + cr[1] * cr[1] + cr[2] * cr[2]);
      const float den = a * a + b * b;
      if (den > 0.f) {
        const float ka = -b / den * gth;
        const float kb = b > 0.f ? a / den * gth / b : 0.f;
        const float vxc[3] = {v[1] * cr[2] - v[2] * cr[1], v[2] * cr[0] - v[0] * cr[2], v[0] * cr[1] - v[1] * cr[0]};
        const float cxu[3] = {cr[1] * u[2] - cr[2] * u[1], cr[2] * u[0] - cr[0] * u[2], cr[0] * u[1] - cr[1] * u[0]};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float gu = ka * v[c] + kb * vxc[c];
          const float gv = ka * u[c] + kb * cxu[c];
          gsum[c] = gu;
          gsum[3 + c] = -(gu + gv);
          sAccX[k * 3 + c] += gv;                    // (row k belongs to this tile, and k != i, j)
        }
      }
    }
#pragma unroll
    for (int c = 0; c < 6; ++c) gsum[c] = row16_total(gsum[c]);
    if (lane == 0) {
#pragma unroll
      for (int c = 0; c < 6; ++c) sGij[tt * 8 + c] = gsum[c];
    }
  };
  // ... and the one thread that adds the tiles' d x_j | d x_i slots up (behind the barrier that follows the geometry steps)
  auto geometry_fold = [&](int nt, int li, int lj) {
    if (tid == 0) {
      for (int tt = 0; tt < nt; ++tt)
#pragma unroll
        for (int c = 0; c < 3; ++c) { sAccX[lj * 3 + c] += sGij[tt * 8 + c]; sAccX[li * 3 + c] += sGij[tt * 8 + 3 + c]; }
    }
  };

#ifdef PG_TB2_PROF
  unsigned long long tb2_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tb2_last = __builtin_readcyclecounter();
#endif
  for (int ai = blockIdx.x; ai < t.n_lig; ai += gridDim.x) {
    const int a_lig = gr.atom_order ? gr.atom_order[ai] : ai;
    const int cj = t.lig2ctx[a_lig];
    const int gi = t.ctx_graph[cj];
    const int n = t.g_nlig[gi], lig0 = t.g_ctx_off[gi] + t.g_nph[gi], lj = cj - lig0;
    const int* eid_g = t.eid + t.g_eid_off[gi];
    const int nt = (n + 15) >> 4;
    if (nt > MAXT || nt < nt_lo) continue;        // (a launch takes the ligands of nt_lo .. MAXT row tiles)

    for (int i = tid; i < n * 3; i += blockDim.x) { sX[i] = p.x[lig0 * 3 + i]; sAccX[i] = 0.f; }
    float xj[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) xj[c] = p.x[cj * 3 + c];
    // the wave's channels of the rows P[k -> j] (key | value), rows k = 16 tt + m; d P accumulators
    f4 Pk[MAXT][NB], Pv[MAXT][NB], gPk[MAXT][NB], gPv[MAXT][NB];
    int erow[MAXT];
#pragma unroll
    for (int tt = 0; tt < MAXT; ++tt) {
      const int k = 16 * tt + m;
      const bool have = tt < nt && k < n && k != lj;
      erow[tt] = eid_g[have ? k * n + lj : 0];          // (unconditional loads: the diagonal entry 0 of the table is -1)
      erow[tt] = have ? erow[tt] : -1;
      const size_t ro = (size_t)(have ? erow[tt] : 0) * p.ld_csrc + c0 + 4 * g;
#pragma unroll
      for (int tb = 0; tb < NB; ++tb) {
        const f4 z4 = {0.f, 0.f, 0.f, 0.f};
        gPk[tt][tb] = z4; gPv[tt][tb] = z4;
        const f4 lk = *reinterpret_cast<const f4*>(p.Csrc_k + ro + 16 * tb);
        const f4 lv = *reinterpret_cast<const f4*>(p.Csrc_v + ro + 16 * tb);
        Pk[tt][tb] = have ? lk : z4;
        Pv[tt][tb] = have ? lv : z4;
      }
    }
    if (n > 1) stage_operands(eid_g[lj * n + (lj == 0 ? 1 : 0)], 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();                              // sX is there
    if (n > 1) {
      const int li0 = lj == 0 ? 1 : 0;
      float xi0[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) xi0[c] = sX[li0 * 3 + c];
      for (int tt = wave; tt < nt; tt += NW) features(0, tt, n, li0, lj, xi0, xj);
    }
    __syncthreads();

    int li_prev = 0;                              // the finished segment whose geometry step is still to run
    float xi_prev[3] = {0.f, 0.f, 0.f};
    TB2_STAMP(0);
    int step = 0;                                 // tile steps of this atom so far (selects the double buffers)

    for (int sidx = 0; sidx < n - 1; ++sidx) {
      const int li = sidx < lj ? sidx : sidx + 1;
      const int seg = eid_g[lj * n + li];
      const int par = sidx % 3, ob = sidx & 1;      // feature buffer (three segments in flight), operand stage
      float xi[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) xi[c] = sX[li * 3 + c];

      // the next segment's operands start to land in the other stage (its readers, the B steps of the previous segment, are done);
      // this segment's are read out of its stage where they are used
      if (sidx + 1 < n - 1) stage_operands(eid_g[lj * n + ((sidx + 1) < lj ? sidx + 1 : sidx + 2)], ob ^ 1);
      const float* const Us = sOp + ob * 4352;
      const float* const Ms = Us + 2048;
      f4 cdk[NB], cdv[NB];
#pragma unroll
      for (int tb = 0; tb < NB; ++tb) {
        cdk[tb] = *reinterpret_cast<const f4*>(Us + 4096 + c0 + 16 * tb + 4 * g);
        cdv[tb] = *reinterpret_cast<const f4*>(Us + 4096 + 128 + c0 + 16 * tb + 4 * g);
      }
      const int hh_sm = (16 / NW) * wave + lane / (64 / (16 / NW));
      const float gsw = gr.gswn[(size_t)seg * 16 + hh_sm];       // (the softmax lanes' head; requested early)
      TB2_WAITVM();
      TB2_STAMP(1);

      // ---------------- A1: hidden^T of every tile, partial LayerNorm row sums ----------------
      f4 hK[MAXT][NB], hV[MAXT][NB];
#pragma unroll
      for (int tt = 0; tt < MAXT; ++tt) {
        if (tt < nt) {
          const float* fr = sFeat + ((par * MAXT + tt) * 16 + m) * 17 + g;
          const float f0 = fr[0], f1 = fr[4], f2 = fr[8];
          float qk = 0.f, qv = 0.f;
#pragma unroll
          for (int tb = 0; tb < NB; ++tb) {
            f4 hk = Pk[tt][tb] + cdk[tb], hv = Pv[tt][tb] + cdv[tb];
            hk = mfma16(wfA[0][0][tb], f0, hk); hv = mfma16(wfA[1][0][tb], f0, hv);
            hk = mfma16(wfA[0][1][tb], f1, hk); hv = mfma16(wfA[1][1][tb], f1, hv);
            hk = mfma16(wfA[0][2][tb], f2, hk); hv = mfma16(wfA[1][2][tb], f2, hv);
            hK[tt][tb] = hk; hV[tt][tb] = hv;
#pragma unroll
            for (int r = 0; r < 4; ++r) { qk = fmaf(hk[r], hk[r], qk); qv = fmaf(hv[r], hv[r], qv); }
          }
          qk += __shfl_xor(qk, 16); qk += __shfl_xor(qk, 32);
          qv += __shfl_xor(qv, 16); qv += __shfl_xor(qv, 32);
          if (g == 0) {
            sStat[((0 * MAXT + tt) * 16 + m) * NW + wave] = qk;
            sStat[((1 * MAXT + tt) * 16 + m) * NW + wave] = qv;
          }
        }
      }
      TB2_STAMP(2);
      __syncthreads();                            // barrier 1
      TB2_STAMP(3);

      // ---------------- A2: statistics, z, partial projections ----------------
      float rsK[MAXT], sgK[MAXT], rsV[MAXT], sgV[MAXT];
#pragma unroll
      for (int tt = 0; tt < MAXT; ++tt) {
        rsK[tt] = sgK[tt] = rsV[tt] = sgV[tt] = 0.f;
        if (tt < nt) {
          float qk = 0.f, qv = 0.f;
#pragma unroll
          for (int w4 = 0; w4 < NW; w4 += 4) {          // the waves' partial sums of a row are contiguous: 16-byte reads
            const f4 a4 = *reinterpret_cast<const f4*>(sStat + ((0 * MAXT + tt) * 16 + m) * NW + w4);
            const f4 b4 = *reinterpret_cast<const f4*>(sStat + ((1 * MAXT + tt) * 16 + m) * NW + w4);
            qk = (((qk + a4[0]) + a4[1]) + a4[2]) + a4[3];
            qv = (((qv + b4[0]) + b4[1]) + b4[2]) + b4[3];
          }
          const float vk = qk * (1.f / 128.f) + 1e-5f, vv = qv * (1.f / 128.f) + 1e-5f;
          rsK[tt] = 1.0f / sqrtf(vk); sgK[tt] = vk * rsK[tt];      // (the hardware rsq, 1 ulp, moves a bias gradient by 2e-5: kept exact)
          rsV[tt] = 1.0f / sqrtf(vv); sgV[tt] = vv * rsV[tt];
          f4 yk = {0.f, 0.f, 0.f, 0.f}, yv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int tb = 0; tb < NB; ++tb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {           // B layout of U / dS: [c = 16 tq + 4g + r][h = m]
              const int uo = ((NB * wave + tb) * 4 + r) * 64 + lane;
              yk = mfma16(fmaxf(fmaf(bC[0][tb][r], sgK[tt], hK[tt][tb][r]), 0.f), Us[uo], yk);
              yv = mfma16(fmaxf(fmaf(bC[1][tb][r], sgV[tt], hV[tt][tb][r]), 0.f), Ms[uo], yv);
            }
          *reinterpret_cast<f4*>(sY + ((0 * MAXT + tt) * NW + wave) * YT + 72 * g + 4 * m) = yk;
          *reinterpret_cast<f4*>(sY + ((1 * MAXT + tt) * NW + wave) * YT + 72 * g + 4 * m) = yv;
          if (wave == 0 && g == 0) { sRs[16 * tt + m] = rsK[tt]; sRs[ROWS + 16 * tt + m] = rsV[tt]; }
        }
      }
      TB2_STAMP(5);
      __syncthreads();                            // barrier 2
      TB2_STAMP(6);
      // features of the next segment (read from the first barrier of that segment on; the other parity's readers are done)
      if (sidx + 1 < n - 1) {
        const int li1 = (sidx + 1) < lj ? sidx + 1 : sidx + 2;
        float xi1[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) xi1[c] = sX[li1 * 3 + c];
        for (int tt = wave; tt < nt; tt += NW) features((sidx + 1) % 3, tt, n, li1, lj, xi1, xj);
      }
      TB2_STAMP(7);
      // the geometry steps of the previous segment, on the waves behind the feature waves
      if (sidx > 0)
        for (int tt = 0; tt < nt; ++tt)
          if (wave == (nt + tt) % NW) geometry(tt, (sidx + 2) % 3, n, li_prev, lj, xi_prev, xj);
      TB2_STAMP(4);

      // ---------------- softmax over the segment's rows, d logit: wave w takes the heads [HW w, HW w + HW) ----------------
      // lane = (head hs of the wave, row slot): rows slot, slot + RS, ...; the results go to LDS for every wave:
      //   sQ[0][tt][row][h] = rstd_k[row] d logit[row, h]      sQ[1][tt][row][h] = rstd_v[row] alpha[row, h]
      {
        constexpr int HW = 16 / NW, RS = 64 / HW, NQ = (16 * MAXT + RS - 1) / RS;
        const int hs = lane / RS, slot = lane % RS, hh = HW * wave + hs;
        float lg[NQ], da[NQ], rk[NQ], rv[NQ];
        float mx = NEG_BIG;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const int rho = slot + RS * q, tt = rho >> 4, r16 = rho & 15;
          lg[q] = NEG_BIG; da[q] = 0.f; rk[q] = 0.f; rv[q] = 0.f;
          if (rho < 16 * nt) {
            const bool valid = rho < n && rho != li && rho != lj;
            const float* yk = sY + (0 * MAXT + tt) * NW * YT + 72 * (r16 >> 2) + 4 * hh + (r16 & 3);
            const float* yv = sY + (1 * MAXT + tt) * NW * YT + 72 * (r16 >> 2) + 4 * hh + (r16 & 3);
            float sk = 0.f, sv = 0.f;
#pragma unroll
            for (int w2 = 0; w2 < NW; ++w2) { sk += yk[w2 * YT]; sv += yv[w2 * YT]; }
            rk[q] = sRs[rho]; rv[q] = sRs[ROWS + rho];
            lg[q] = valid ? sk * rk[q] : NEG_BIG;
            da[q] = valid ? fmaf(sv, rv[q], gsw) : 0.f;
            mx = fmaxf(mx, lg[q]);
          }
        }
        mx = tb2_row16_max(mx);
        if constexpr (RS == 32) mx = fmaxf(mx, __shfl_xor(mx, 16));
        float e[NQ], l = 0.f;
#pragma unroll
        for (int q = 0; q < NQ; ++q) { e[q] = lg[q] > 0.5f * NEG_BIG ? __builtin_amdgcn_exp2f(lg[q] - mx) : 0.f; l += e[q]; }
        l = row16_total(l);
        if constexpr (RS == 32) l += __shfl_xor(l, 16);
        const float inv = l > 0.f ? 1.0f / l : 0.f;
        float D = 0.f;
#pragma unroll
        for (int q = 0; q < NQ; ++q) { e[q] *= inv; D = fmaf(e[q], da[q], D); }
        D = row16_total(D);
        if constexpr (RS == 32) D += __shfl_xor(D, 16);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const int rho = slot + RS * q, tt = rho >> 4, r16 = rho & 15;
          if (rho < 16 * nt) {
            sQ[(0 * MAXT + tt) * QT + r16 * 20 + hh] = TB2_LN2 * e[q] * (da[q] - D) * rk[q];
            sQ[(1 * MAXT + tt) * QT + r16 * 20 + hh] = e[q] * rv[q];
          }
        }
      }
      TB2_STAMP(8);
      __syncthreads();                            // barrier 2b
      TB2_STAMP(14);
      if (sidx > 0) geometry_fold(nt, li_prev, lj);

      f4 gU[NB], gCk[NB], gCv[NB];
#pragma unroll
      for (int tb = 0; tb < NB; ++tb) { gU[tb] = (f4){0.f, 0.f, 0.f, 0.f}; gCk[tb] = gU[tb]; gCv[tb] = gU[tb]; }

      // ---------------- B: per tile ----------------
#pragma unroll
      for (int tt = 0; tt < MAXT; ++tt) {
        if (tt < nt) {
          const int buf = step & 1;
          // z_k^T of the wave's channels through LDS: A operand [c = 16 tb + m][row = 4g + ks] of d U
#pragma unroll
          for (int tb = 0; tb < NB; ++tb)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              tT[(16 * tb + 4 * g + r) * 17 + m] = fmaxf(fmaf(bC[0][tb][r], sgK[tt], hK[tt][tb][r]), 0.f);
          // rstd d logit as the B operand of d U ([row 4g + ks][h = m]) and, with the row on the lane, of d z ([row = m][h = 4g + ks])
          const float* qk = sQ + (0 * MAXT + tt) * QT;
          f4 dlB;
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) dlB[ks] = qk[(4 * g + ks) * 20 + m];
          const f4 dlT = *reinterpret_cast<const f4*>(qk + m * 20 + 4 * g);
          const f4 alT = *reinterpret_cast<const f4*>(sQ + (1 * MAXT + tt) * QT + m * 20 + 4 * g);
          wave_lds_sync();
#pragma unroll
          for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int tb = 0; tb < NB; ++tb) gU[tb] = mfma16(tT[(16 * tb + m) * 17 + 4 * g + ks], dlB[ks], gU[tb]);
          // d z^T[c, row = m] = sum_h M[c, h] (rstd coef)[row, h]; ReLU mask; the LayerNorm row sum: with z = ReLU(pre),
          // d rstd = sum_h coef y = (1 / rstd) sum_c z d pre, so  d var = (rstd / 2) sum_c d pre (b' - rstd z)  -- one partial sum per wave
          f4 dK[NB], dV[NB];
          float s1k = 0.f, s1v = 0.f;
#pragma unroll
          for (int tb = 0; tb < NB; ++tb) {
            f4 zk = {0.f, 0.f, 0.f, 0.f}, zv = {0.f, 0.f, 0.f, 0.f};
            // A layout of U / dS: [c = 16 tq + m][h = 4g + ks], ks = 0..3 contiguous in the lane-fixed rows
            const int ao = ((NB * wave + tb) * 4 + (m & 3)) * 64 + (m >> 2) * 16 + 4 * g;
            const f4 Ua = *reinterpret_cast<const f4*>(Us + ao), Ma = *reinterpret_cast<const f4*>(Ms + ao);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) { zk = mfma16(Ua[ks], dlT[ks], zk); zv = mfma16(Ma[ks], alT[ks], zv); }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float pk = fmaf(bC[0][tb][r], sgK[tt], hK[tt][tb][r]), pv = fmaf(bC[1][tb][r], sgV[tt], hV[tt][tb][r]);
              zk[r] = pk > 0.f ? zk[r] : 0.f;
              zv[r] = pv > 0.f ? zv[r] : 0.f;
              s1k = fmaf(zk[r], fmaf(-rsK[tt], pk, bC[0][tb][r]), s1k);
              s1v = fmaf(zv[r], fmaf(-rsV[tt], pv, bC[1][tb][r]), s1v);
              gB[0][tb][r] = fmaf(zk[r], sgK[tt], gB[0][tb][r]);
              gB[1][tb][r] = fmaf(zv[r], sgV[tt], gB[1][tb][r]);
            }
            dK[tb] = zk; dV[tb] = zv;
          }
          s1k += __shfl_xor(s1k, 16); s1k += __shfl_xor(s1k, 32);
          s1v += __shfl_xor(s1v, 16); s1v += __shfl_xor(s1v, 32);
          if (g == 0) {
            sS1[((buf * 2 + 0) * 16 + m) * NW + wave] = s1k;
            sS1[((buf * 2 + 1) * 16 + m) * NW + wave] = s1v;
          }
          if (tt == nt - 1) __builtin_amdgcn_s_waitcnt(0);       // the next segment's operands have landed (this wave's pieces)
          TB2_STAMP(9);
          __syncthreads();                        // barrier 3 + tt
          TB2_STAMP(10);
          {
            float tk = 0.f, tv = 0.f;
#pragma unroll
            for (int w4 = 0; w4 < NW; w4 += 4) {
              const f4 a4 = *reinterpret_cast<const f4*>(sS1 + ((buf * 2 + 0) * 16 + m) * NW + w4);
              const f4 b4 = *reinterpret_cast<const f4*>(sS1 + ((buf * 2 + 1) * 16 + m) * NW + w4);
              tk = (((tk + a4[0]) + a4[1]) + a4[2]) + a4[3];
              tv = (((tv + b4[0]) + b4[1]) + b4[2]) + b4[3];
            }
            const float gvk = 0.5f * rsK[tt] * tk * (1.f / 64.f), gvv = 0.5f * rsV[tt] * tv * (1.f / 64.f);
#pragma unroll
            for (int tb = 0; tb < NB; ++tb)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                dK[tb][r] = fmaf(gvk, hK[tt][tb][r], dK[tb][r]);       // d hidden
                dV[tb][r] = fmaf(gvv, hV[tt][tb][r], dV[tb][r]);
              }
          }
          f4 dfp = {0.f, 0.f, 0.f, 0.f};          // the wave's part of d feat[row 4g + r][f = m]
#pragma unroll
          for (int tb = 0; tb < NB; ++tb) {
            gPk[tt][tb] += dK[tb]; gPv[tt][tb] += dV[tb];
            gCk[tb] += dK[tb]; gCv[tb] += dV[tb];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              dfp = mfma16(dK[tb][r], wfB[0][tb][r], dfp);
              dfp = mfma16(dV[tb][r], wfB[1][tb][r], dfp);
            }
          }
          *reinterpret_cast<f4*>(sDf + ((tt * NW + wave) * 64 + lane) * 4) = dfp;
          // d Wf[c, f] += sum_row d hidden[c, row] feat[row, f]: d hidden^T through the LDS tile, one path after the other
          const float* fb = sFeat + ((par * MAXT + tt) * 16 + 4 * g) * 17 + m;
          f4 fB;
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) fB[ks] = fb[ks * 17];
#pragma unroll
          for (int a = 0; a < 2; ++a) {
            wave_lds_sync();
#pragma unroll
            for (int tb = 0; tb < NB; ++tb)
#pragma unroll
              for (int r = 0; r < 4; ++r) tT[(16 * tb + 4 * g + r) * 17 + m] = a == 0 ? dK[tb][r] : dV[tb][r];
            wave_lds_sync();
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
              for (int tb = 0; tb < NB; ++tb) gWf[a][tb] = mfma16(tT[(16 * tb + m) * 17 + 4 * g + ks], fB[ks], gWf[a][tb]);
          }
          wave_lds_sync();
          ++step;
          TB2_STAMP(11);
        }
      }

      // ---------------- per-segment outputs ----------------
      {
        float* up = gr.gU + (size_t)seg * 2048 + lane;
        float* ck = gr.gCdst_k + (size_t)seg * gr.ld_gcdst + c0 + 4 * g;
        float* cv = gr.gCdst_v + (size_t)seg * gr.ld_gcdst + c0 + 4 * g;
#pragma unroll
        for (int tb = 0; tb < NB; ++tb) {
          const int tq = NB * wave + tb;
          f4 sk, sv;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            up[(tq * 4 + r) * 64] = gU[tb][r];
            sk[r] = row16_total(gCk[tb][r]);
            sv[r] = row16_total(gCv[tb][r]);
          }
          if (m == 0) {
            *reinterpret_cast<f4*>(ck + 16 * tb) = sk;
            *reinterpret_cast<f4*>(cv + 16 * tb) = sv;
          }
        }
      }
      li_prev = li;
      xi_prev[0] = xi[0]; xi_prev[1] = xi[1]; xi_prev[2] = xi[2];
      TB2_STAMP(12);
    }  // segments

    __syncthreads();
    if (n > 1)
      for (int tt = 0; tt < nt; ++tt)
        if (wave == tt % NW) geometry(tt, (n - 2) % 3, n, li_prev, lj, xi_prev, xj);
    __syncthreads();
    if (n > 1) geometry_fold(nt, li_prev, lj);
    __syncthreads();
    // d P rows of the source atom: every row k -> j is written once
#pragma unroll
    for (int tt = 0; tt < MAXT; ++tt) {
      if (erow[tt] >= 0) {
#pragma unroll
        for (int tb = 0; tb < NB; ++tb) {
          *reinterpret_cast<f4*>(gr.gCsrc_k + (size_t)erow[tt] * gr.ld_gcsrc + c0 + 16 * tb + 4 * g) = gPk[tt][tb];
          *reinterpret_cast<f4*>(gr.gCsrc_v + (size_t)erow[tt] * gr.ld_gcsrc + c0 + 16 * tb + 4 * g) = gPv[tt][tb];
        }
      }
    }
    if (gr.gx)
      for (int i = tid; i < n * 3; i += blockDim.x) atomicAdd(gr.gx + lig0 * 3 + i, sAccX[i]);
    __syncthreads();
    TB2_STAMP(13);
  }  // atoms
#ifdef PG_TB2_PROF
  if (lane == 0)
    for (int i = 0; i < 16; ++i) atomicAdd(&g_tb2_prof[i], tb2_acc[i]);
#endif

  // ---------------- weight gradients: every wave owns its channels ----------------
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    float* gw = a == 0 ? gr.gWf_k : gr.gWf_v;
    float* gb = a == 0 ? gr.gbk : gr.gbv;
#pragma unroll
    for (int tb = 0; tb < NB; ++tb) {
      const int tq = NB * wave + tb;
      if (m < 12) {
        float* dst = gw + ((m >> 2) * 8 + tq) * 64 + (m & 3) * 16 + 4 * g;
#pragma unroll
        for (int r = 0; r < 4; ++r) atomicAdd(dst + r, gWf[a][tb][r]);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = row16_total(gB[a][tb][r]);
        if (m == 0) atomicAdd(gb + c0 + 16 * tb + 4 * g + r, v);
      }
    }
  }
}

template <int NW, int MAXT>
static int launch_tb2(const PgTopo* t, const PgSegAttn* p, const PgSegAttnGrad* gr, int nt_lo, hipStream_t st) {
  const size_t lds = (size_t)Tb2Layout<NW, MAXT>::total * sizeof(float);
  if (int rc = reserve_lds(reinterpret_cast<const void*>(triplet_bwd2_kernel<NW, MAXT>), lds, "pg_seg_attn_bwd (channel split)")) return rc;
  int blocks = t->n_lig < gr->grid ? t->n_lig : gr->grid;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL((triplet_bwd2_kernel<NW, MAXT>), dim3(blocks), dim3(64 * NW), lds, st, *t, *p, *gr, nt_lo);
  return check_launch("pg_seg_attn_bwd (channel split)");
}

// entry used by pg_seg_attn_bwd (seg_attn_bwd.hip) for PG_SEG_TRIPLET when PgSegAttnGrad.tri_form asks for it; ligands of up to 64 atoms.
// tri_form 1: 4 waves x 32 channels (up to 512 registers, one wave per SIMD).  tri_form 2: ligands of up to 32 atoms on 8 waves x 16 channels
// (two waves per SIMD at 256 registers, no spills: the fastest form), the larger ones of the batch in a second launch of the 4-wave form --
// every launch walks the same atom list and skips the ligands outside its range of row tiles.  (The 8-wave instances for 3 / 4 row tiles
// need more than 256 registers; they are not built into the dispatch.)
int triplet_bwd2_launch(const PgTopo* t, const PgSegAttn* p, const PgSegAttnGrad* gr, hipStream_t st) {
  const int nt = (t->max_nlig + 15) / 16;
  int lo = 1;
  if (gr->tri_form == 2) {
    if (int rc = launch_tb2<8, 2>(t, p, gr, 1, st)) return rc;
    if (nt <= 2) return PG_OK;
    lo = 3;
  }
  if (nt <= 2) return launch_tb2<4, 2>(t, p, gr, lo, st);
  if (nt == 3) return launch_tb2<4, 3>(t, p, gr, lo, st);
  return launch_tb2<4, 4>(t, p, gr, lo, st);
}

}  // namespace pg

#ifdef PG_TB2_PROF
extern "C" int pg_debug_tb2_prof(unsigned long long* out, int reset) {
  hipMemcpyFromSymbol(out, HIP_SYMBOL(pg::g_tb2_prof), sizeof(unsigned long long) * 16);
  if (reset) { unsigned long long z[16] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(pg::g_tb2_prof), z, sizeof(z)); }
  return 0;
}
#endif
