// Node-target attention sub-layers (knn edges / bond edges, node update / position update) in the two-pass,
// low-register form of triplet.hip: one wave per target node,
//   pass A  K path for every row tile -> logits (and, for the position update, the per-head value scalars);
//   exact softmax over the stored logits (base 2; queries carry log2(e)/sqrt(8));
//   pass B  V path -> S^T[c,h] (node update) or the weighted sum of relative positions (position update).
// Two forms.  Plain (training, and the standalone calls): U (query-folded keys) is read from HBM (pg_attn_fold_query), S goes
// back for pg_attn_unfold_value, so no second-layer weights sit in LDS and several workgroups share a CU.  FUSED (the sampler):
// one persistent 768-thread workgroup per CU keeps the lane-fixed W2k in LDS and folds the query in-kernel; the node-update
// modes also apply W2v to the aggregate (streamed through L2, as in triplet2.hip) and write the 128-float update itself:
// the [n][32][64] round trips of U and S (8 KB per node each way) and two launches per sub-layer disappear.  fp32 MFMA and VALU share the SIMD pipe on
// gfx950 (profiles/r01_micro_mfma_valu_coexec.md): the kernel is written for low instruction count, occupancy only
// hides the row-gather latency.
// Lane l = (g = l>>4, m = l&15); 16x16x4 maps as in seg_attn.hip.
#include "common.h"
#include "../../include/phoregen_hip.h"

namespace pg {

constexpr float NA_NEG = -1.0e30f;

template <int CTRL>
__device__ __forceinline__ float dpp_na(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row16_sum_na(float v) {
  v += dpp_na<0xB1>(v);
  v += dpp_na<0x4E>(v);
  v += dpp_na<0x141>(v);
  v += dpp_na<0x140>(v);
  return v;
}

// folded LayerNorm + ReLU on a K-path tile (hid[tau][r] = hidden[c = 16 tau + 4g + r][row = m]); returns rstd of row m
__device__ __forceinline__ float ln_fold_k(f4 (&hid)[8], const float* bp, int g) {
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

template <bool KNN, bool POS, int MAXT, int THREADS, bool FUSED>
__global__ __launch_bounds__(THREADS, FUSED ? 1 : 3) void node_attn_kernel(PgTopo t, PgSegAttn p) {
  constexpr int NSTEP = KNN ? 12 : 0;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const bpk = lds;                   // [128] b' of the key MLP
  float* const bpv = lds + 128;             // [128] b' of the value MLP
  float* const wf_k = lds + 256;            // [NSTEP][8][64]
  float* const wf_v = wf_k + NSTEP * 512;
  float* const w2xv = wf_v + NSTEP * 512;   // POS: [32][64]
  float* const b2xv = w2xv + (POS ? 2048 : 0);
  float* const w2k = b2xv + (POS ? 16 : 0); // FUSED: lane-fixed W2k [64][64][4]
  float* const b2v_s = w2k + 16384;         // FUSED node update: [128]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, m = lane & 15;
  // Two target lists in one launch (fused knn form, PgSegAttn.seg_ids2): the grid is rows of 8 workgroups (one per XCD under
  // round-robin placement); the first `rows1` rows serve list 1, the others list 2 -- in proportion to the lists' node counts, so
  // both finish after the same number of node rounds.  One list: every row serves it.
  const int n_rows_grid = FUSED ? ((int)gridDim.x + 7) / 8 : 1;
  int rows1 = n_rows_grid;
  if (FUSED && KNN && p.n_seg2 > 0) {
    rows1 = (int)(((long long)n_rows_grid * p.n_seg + (p.n_seg + p.n_seg2) / 2) / (p.n_seg + p.n_seg2));
    rows1 = rows1 < 1 ? 1 : (rows1 > n_rows_grid - 1 ? n_rows_grid - 1 : rows1);
  }
  const bool list2 = FUSED && KNN && p.n_seg2 > 0 && (int)blockIdx.x / 8 >= rows1;
  const int* const my_ids = list2 ? p.seg_ids2 : p.seg_ids;
  const int my_n = list2 ? p.n_seg2 : p.n_seg;
  const float* const my_wf_k = list2 ? p.Wf_k2 : p.Wf_k;
  const float* const my_wf_v = list2 ? p.Wf_v2 : p.Wf_v;
  for (int i = tid; i < 128; i += THREADS) { bpk[i] = p.ln_bk[i]; bpv[i] = p.ln_bv[i]; }
  for (int i = tid; i < NSTEP * 128; i += THREADS) {       // (16-byte copies: the fill is what a small-batch launch waits for)
    reinterpret_cast<f4*>(wf_k)[i] = reinterpret_cast<const f4*>(my_wf_k)[i];
    reinterpret_cast<f4*>(wf_v)[i] = reinterpret_cast<const f4*>(my_wf_v)[i];
  }
  if constexpr (POS) {
    for (int i = tid; i < 512; i += THREADS) reinterpret_cast<f4*>(w2xv)[i] = reinterpret_cast<const f4*>(p.W2xv_l)[i];
    for (int i = tid; i < 16; i += THREADS) b2xv[i] = p.b2xv[i];
  }
  if constexpr (FUSED) {
    for (int i = tid; i < 4096; i += THREADS) reinterpret_cast<f4*>(w2k)[i] = reinterpret_cast<const f4*>(p.W2k_l)[i];
    if constexpr (!POS) for (int i = tid; i < 128; i += THREADS) b2v_s[i] = p.b2v[i];
  }
  __syncthreads();

  // Node hand-out.  Fused (persistent) form: XCD-affine -- the node list (graph after graph) is cut into chunks of one node per wave,
  // the chunk range into 8 contiguous parts, and part x is served by the workgroups with blockIdx % 8 == x, which the dispatcher
  // places on one XCD (MI355X_MICROARCH.md, "Workgroup dispatch": a speed assumption only).  A graph's first-layer rows (the
  // Csrc blocks every one of its ~150 nodes gathers 32 of) are then pulled through ONE L2 instead of all eight.
  constexpr int PER = THREADS / 64;
  const int n_chunks = (my_n + PER - 1) / PER;
  // this list's workgroups: rows [row_lo, row_hi) of the grid (a row = 8 consecutive workgroups); a partial last row belongs to list 2
  const int row_lo = list2 ? rows1 : 0, row_hi = (FUSED && KNN && p.n_seg2 > 0 && !list2) ? rows1 : n_rows_grid;
  const int wg_lo = row_lo * 8, wg_hi = row_hi * 8 < (int)gridDim.x ? row_hi * 8 : (int)gridDim.x;
  const int n_wg = FUSED ? wg_hi - wg_lo : (int)gridDim.x, b_loc = FUSED ? (int)blockIdx.x - wg_lo : (int)blockIdx.x;
  const int n_x = FUSED ? (n_wg < 8 ? n_wg : 8) : 1;
  const int xcd = FUSED ? b_loc % n_x : 0, jx = FUSED ? b_loc / n_x : b_loc;
  const int n_jx = FUSED ? (n_wg - xcd + n_x - 1) / n_x : n_wg;
  const int c_lo = (int)((long long)n_chunks * xcd / n_x), c_hi = (int)((long long)n_chunks * (xcd + 1) / n_x);
  for (int ch = c_lo + jx; ch < c_hi; ch += n_jx) {
    const int si = ch * PER + wave;
    if (si >= my_n) continue;
    const int seg = my_ids ? my_ids[si] : si;              // target ctx node
    // opaque copy of the lane id: LDS weight reads addressed through it are not loop-invariant, so the compiler cannot hoist them
    // out of the node loop into registers it then has to spill (the fused knn form carried 70 spilled registers that way)
    int lw = lane;
    asm volatile("" : "+v"(lw));
    int n_rows, lig0 = 0, n = 0, li = 0;
    const int* eid_g = nullptr;
    if constexpr (KNN) {
      n_rows = p.deg[seg];
    } else {
      const int gi = t.ctx_graph[seg];
      n = t.g_nlig[gi];
      lig0 = t.g_ctx_off[gi] + t.g_nph[gi];
      li = seg - lig0;
      eid_g = t.eid + t.g_eid_off[gi];
      n_rows = n;
    }
    const int n_tiles = (n_rows + 15) >> 4;
    const float* ckp = p.Cdst_k + (size_t)seg * p.ld_cdst;
    const float* cvp = p.Cdst_v + (size_t)seg * p.ld_cdst;
    float xd[3] = {0.f, 0.f, 0.f}, nd[3] = {0.f, 0.f, 0.f};
    if constexpr (KNN || POS) {
#pragma unroll
      for (int c = 0; c < 3; ++c) xd[c] = p.x[seg * 3 + c];
    }
    if constexpr (KNN) {
#pragma unroll
      for (int c = 0; c < 3; ++c) nd[c] = p.nrm[seg * 3 + c];
    }

    float feat[MAXT][NSTEP > 0 ? NSTEP : 1];
    f4 lg[MAXT];
    f4 vv[POS ? MAXT : 1];
    // KNN: the 40 distance columns are 20 per source kind (feature steps 0-4 ligand sources, 5-9 pharmacophore sources); a tile
    // without sources of a kind has zeros there and skips those MFMA steps (wave-uniform; pg_knn_group_by_kind makes most tiles so)
    bool has_lig[MAXT], has_ph[MAXT];

    // ======================= pass A =======================
    {
      f4 U[8];
      if constexpr (FUSED) {          // U[c][h] = sum_d q[8h+d] W2k[8h+d][c]: the arithmetic of pg_attn_fold_query, from LDS
        const float* qp = p.q + (size_t)seg * 128 + 8 * m;
        const f4 qa = *reinterpret_cast<const f4*>(qp), qb = *reinterpret_cast<const f4*>(qp + 4);
#pragma unroll
        for (int tq = 0; tq < 8; ++tq)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = tq * 4 + r;
            const f4 wa = *reinterpret_cast<const f4*>(w2k + ((size_t)(2 * i) * 64 + lw) * 4);
            const f4 wb = *reinterpret_cast<const f4*>(w2k + ((size_t)(2 * i + 1) * 64 + lw) * 4);
            U[tq][r] = (qa[0] * wa[0] + qa[1] * wa[1]) + (qa[2] * wa[2] + qa[3] * wa[3]) +
                       (qb[0] * wb[0] + qb[1] * wb[1]) + (qb[2] * wb[2] + qb[3] * wb[3]);
            if (r == 3) __builtin_amdgcn_sched_barrier(0);     // at most 8 weight reads in flight: the registers are needed
          }
      } else {
        const float* up = p.U + (size_t)seg * 2048 + lane;
#pragma unroll
        for (int tq = 0; tq < 8; ++tq)
#pragma unroll
          for (int r = 0; r < 4; ++r) U[tq][r] = up[(tq * 4 + r) * 64];
      }
#pragma unroll
      for (int tile = 0; tile < MAXT; ++tile) {
        lg[tile] = (f4){NA_NEG, NA_NEG, NA_NEG, NA_NEG};
        if constexpr (POS) vv[tile] = (f4){0.f, 0.f, 0.f, 0.f};
        if (tile < n_tiles) {
          const int k = tile * 16 + m;
          bool valid = k < n_rows;
          int src = 0, crow = 0;
          if constexpr (KNN) {
            if (valid) { src = p.nbr[(size_t)seg * p.knn_k + k]; crow = src; }
          } else {
            valid = valid && k != li;
            if (valid) { src = lig0 + k; crow = eid_g[k * n + li]; }
          }
          if constexpr (KNN) {
            // 48 features of row k for f = 4 step + g (packing._knn_feat); f = 47 carries the target's constant Cdst
            float d = 0.f, dots[3] = {0.f, 0.f, 0.f};
            bool src_lig = false;
            if (valid) {
              float xs[3], ns[3];
#pragma unroll
              for (int c = 0; c < 3; ++c) { xs[c] = p.x[src * 3 + c]; ns[c] = p.nrm[src * 3 + c]; }
              const float r0 = xd[0] - xs[0], r1 = xd[1] - xs[1], r2 = xd[2] - xs[2];
              d = sqrtf(r0 * r0 + r1 * r1 + r2 * r2);
              dots[0] = ns[0] * nd[0] + ns[1] * nd[1] + ns[2] * nd[2];
              dots[1] = -(ns[0] * r0 + ns[1] * r1 + ns[2] * r2);
              dots[2] = -(nd[0] * r0 + nd[1] * r1 + nd[2] * r2);
              src_lig = t.ctx_is_lig[src] != 0;
            }
#pragma unroll
            for (int st = 0; st < 5; ++st) {
              const float sv = valid ? smear(d, 4 * st + g) : 0.f;
              feat[tile][st] = src_lig ? sv : 0.f;
              feat[tile][5 + st] = src_lig ? 0.f : sv;
            }
            has_lig[tile] = __ballot(valid && src_lig) != 0ull;
            has_ph[tile] = __ballot(valid && !src_lig) != 0ull;
            feat[tile][10] = g == 0 ? dots[0] : (g == 1 ? dots[1] : (g == 2 ? dots[2] : ((valid && src_lig) ? 1.f : 0.f)));
            feat[tile][11] = g == 0 ? ((valid && !src_lig) ? 1.f : 0.f) : (g == 3 ? 1.f : 0.f);
          }
          // ---- key MLP: hidden^T[c,row] ----
          f4 hid[8];
          {
            const float* pk = p.Csrc_k + (size_t)crow * p.ld_csrc + 4 * g;
#pragma unroll
            for (int tq = 0; tq < 8; ++tq) {
              f4 c = {0.f, 0.f, 0.f, 0.f};
              if (valid) c = *reinterpret_cast<const f4*>(pk + 16 * tq);
              if constexpr (!KNN) c += *reinterpret_cast<const f4*>(ckp + 16 * tq + 4 * g);
              hid[tq] = c;
            }
          }
#pragma unroll
          for (int blk3 = 0; blk3 < (NSTEP ? 3 : 0); ++blk3) {
            if ((blk3 == 0 && !has_lig[tile]) || (blk3 == 1 && !has_ph[tile])) continue;
#pragma unroll
            for (int st = 5 * blk3; st < (blk3 == 2 ? 12 : 5 * blk3 + 5); ++st)
#pragma unroll
              for (int tq = 0; tq < 8; ++tq) {
                float w = wf_k[(st * 8 + tq) * 64 + lw];
                if (st == 11) w = g == 3 ? ckp[16 * tq + m] : w;
                hid[tq] = mfma16(w, feat[tile][st], hid[tq]);
              }
          }
          const float rs = ln_fold_k(hid, bpk, g);
          f4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int tq = 0; tq < 8; ++tq)
#pragma unroll
            for (int r = 0; r < 4; r += 2) {
              acc = mfma16(hid[tq][r], U[tq][r], acc);
              acc2 = mfma16(hid[tq][r + 1], U[tq][r + 1], acc2);
            }
          acc += acc2;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int kr = tile * 16 + 4 * g + r;
            const bool vr = kr < n_rows && (KNN || kr != li);
            const float sc_ = acc[r] * __shfl(rs, 4 * g + r);      // shuffle outside the select: every source lane must be live
            lg[tile][r] = vr ? sc_ : NA_NEG;
          }
          if constexpr (POS) {
            // ---- value MLP of the position update (K-path form): v[row,h] = z . W2xv[h,:] + b ----
            f4 hx[8];
            const float* pk = p.Csrc_v + (size_t)crow * p.ld_csrc + 4 * g;
#pragma unroll
            for (int tq = 0; tq < 8; ++tq) {
              f4 c = {0.f, 0.f, 0.f, 0.f};
              if (valid) c = *reinterpret_cast<const f4*>(pk + 16 * tq);
              if constexpr (!KNN) c += *reinterpret_cast<const f4*>(cvp + 16 * tq + 4 * g);
              hx[tq] = c;
            }
#pragma unroll
            for (int blk3 = 0; blk3 < (NSTEP ? 3 : 0); ++blk3) {
              if ((blk3 == 0 && !has_lig[tile]) || (blk3 == 1 && !has_ph[tile])) continue;
#pragma unroll
              for (int st = 5 * blk3; st < (blk3 == 2 ? 12 : 5 * blk3 + 5); ++st)
#pragma unroll
                for (int tq = 0; tq < 8; ++tq) {
                  float w = wf_v[(st * 8 + tq) * 64 + lw];
                  if (st == 11) w = g == 3 ? cvp[16 * tq + m] : w;
                  hx[tq] = mfma16(w, feat[tile][st], hx[tq]);
                }
            }
            const float rsx = ln_fold_k(hx, bpv, g);
            f4 a1 = {0.f, 0.f, 0.f, 0.f}, a2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int tq = 0; tq < 8; ++tq)
#pragma unroll
              for (int r = 0; r < 4; r += 2) {
                a1 = mfma16(hx[tq][r], w2xv[(tq * 4 + r) * 64 + lw], a1);
                a2 = mfma16(hx[tq][r + 1], w2xv[(tq * 4 + r + 1) * 64 + lw], a2);
              }
            a1 += a2;
            const float bx = b2xv[m];
#pragma unroll
            for (int r = 0; r < 4; ++r) vv[tile][r] = a1[r] * __shfl(rsx, 4 * g + r) + bx;
          }
        }
      }
    }

    if constexpr (POS) {
      if (p.alpha) {                     // training: logits and value scalars of every row (32 floats per row), read back by the
        float* ap = p.alpha + (size_t)seg * p.alpha_rows * 32 + m;      // adjoint instead of recomputing both MLPs for them
#pragma unroll
        for (int tile = 0; tile < MAXT; ++tile)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int kr = tile * 16 + 4 * g + r;
            if (kr < p.alpha_rows) { ap[kr * 32] = lg[tile][r]; ap[kr * 32 + 16] = vv[tile][r]; }
          }
      }
    }

    // ======================= softmax over all rows, head m =======================
    float mx = NA_NEG;
#pragma unroll
    for (int tile = 0; tile < MAXT; ++tile)
#pragma unroll
      for (int r = 0; r < 4; ++r) mx = fmaxf(mx, lg[tile][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float l = 0.f, sw = 0.f;
#pragma unroll
    for (int tile = 0; tile < MAXT; ++tile) {
      f4 gate = {1.f, 1.f, 1.f, 1.f};
      if constexpr (KNN) {
        if (tile < n_tiles) gate = *reinterpret_cast<const f4*>(p.ew + (size_t)seg * p.knn_k + tile * 16 + 4 * g);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = lg[tile][r] > 0.5f * NA_NEG ? __builtin_amdgcn_exp2f(lg[tile][r] - mx) : 0.f;
        l += e;
        lg[tile][r] = e * gate[r];           // attention weight x edge gate (v = MLP(...) * e_w, uni_denoiser.py:52-54)
        sw += lg[tile][r];
      }
    }
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    if constexpr (!POS) {
      if (p.alpha) {                     // training: softmax weight x gate of every row, read back by the one-pass adjoint
        float* ap = p.alpha + (size_t)seg * p.alpha_rows * 16 + m;
#pragma unroll
        for (int tile = 0; tile < MAXT; ++tile)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int kr = tile * 16 + 4 * g + r;
            if (kr < p.alpha_rows) ap[kr * 16] = lg[tile][r] * inv;
          }
      }
    }

    if constexpr (POS) {
      // dx = mean_h sum_rows alpha * gate * v * (x_dst - x_src)   (uni_denoiser.py:200-209)
      float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
      for (int tile = 0; tile < MAXT; ++tile)
        if (tile < n_tiles) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int kr = tile * 16 + 4 * g + r;
            const bool vr = kr < n_rows && (KNN || kr != li);
            if (vr) {
              const int src = KNN ? p.nbr[(size_t)seg * p.knn_k + kr] : lig0 + kr;
              const float w = lg[tile][r] * vv[tile][r];
              a0 = fmaf(w, xd[0] - p.x[src * 3], a0);
              a1 = fmaf(w, xd[1] - p.x[src * 3 + 1], a1);
              a2 = fmaf(w, xd[2] - p.x[src * 3 + 2], a2);
            }
          }
        }
      a0 = wave_sum(a0 * inv) * (1.f / 16.f);
      a1 = wave_sum(a1 * inv) * (1.f / 16.f);
      a2 = wave_sum(a2 * inv) * (1.f / 16.f);
      if (lane == 0) {
        if (p.accumulate_dx) { p.dx[seg * 3] += a0; p.dx[seg * 3 + 1] += a1; p.dx[seg * 3 + 2] += a2; }
        else { p.dx[seg * 3] = a0; p.dx[seg * 3 + 1] = a1; p.dx[seg * 3 + 2] = a2; }
      }
    } else {
      // ======================= pass B: S^T[c,h] = sum_rows z_v[row,c] * alpha[row,h] =======================
      sw += __shfl_xor(sw, 16);
      sw += __shfl_xor(sw, 32);
      f4 sT[8];
#pragma unroll
      for (int tq = 0; tq < 8; ++tq) sT[tq] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int tile = 0; tile < MAXT; ++tile) {
        if (tile < n_tiles) {
          f4 hv[8];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int kr = tile * 16 + 4 * g + r;
            const bool vr = kr < n_rows && (KNN || kr != li);
            int crow = 0;
            if (vr) crow = KNN ? p.nbr[(size_t)seg * p.knn_k + kr] : eid_g[kr * n + li];
            const float* pv = p.Csrc_v + (size_t)crow * p.ld_csrc + m;
#pragma unroll
            for (int tq = 0; tq < 8; ++tq) hv[tq][r] = vr ? pv[16 * tq] : 0.f;
          }
          if constexpr (!KNN) {
#pragma unroll
            for (int tq = 0; tq < 8; ++tq) hv[tq] += cvp[16 * tq + m];
          }
#pragma unroll
          for (int blk3 = 0; blk3 < (NSTEP ? 3 : 0); ++blk3) {
            if ((blk3 == 0 && !has_lig[tile]) || (blk3 == 1 && !has_ph[tile])) continue;
#pragma unroll
            for (int st = 5 * blk3; st < (blk3 == 2 ? 12 : 5 * blk3 + 5); ++st)
#pragma unroll
              for (int tq = 0; tq < 8; ++tq) {
                float w = wf_v[(st * 8 + tq) * 64 + lw];
                if (st == 11) w = g == 3 ? cvp[16 * tq + m] : w;
                hv[tq] = mfma16(feat[tile][st], w, hv[tq]);
              }
          }
          f4 q2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int tq = 0; tq < 8; ++tq) q2 += hv[tq] * hv[tq];
          f4 sg, aw;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float var = row16_sum_na(q2[r]) * (1.f / 128.f) + 1e-5f;
            const float rsq = __builtin_amdgcn_rsqf(var);
            sg[r] = var * rsq;
            aw[r] = lg[tile][r] * rsq;
          }
#pragma unroll
          for (int tq = 0; tq < 8; ++tq) {
            const float bt = bpv[16 * tq + m];
#pragma unroll
            for (int r = 0; r < 4; ++r) hv[tq][r] = fmaxf(fmaf(bt, sg[r], hv[tq][r]), 0.f);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int tq = 0; tq < 8; ++tq) sT[tq] = mfma16(hv[tq][r], aw[r], sT[tq]);
        }
      }
      if constexpr (FUSED) {
        // out[8h+d] = W2v[8h+d,:] . S[:,h] + b2v[8h+d] * sum(alpha * gate): the arithmetic of pg_attn_unfold_value, W2v streamed
        // through L2 (64 float4 per lane; the opaque lane copy keeps the 64 addresses from being hoisted out of the node loop)
        int lz = lane;
        asm volatile("" : "+v"(lz));
        const f4* const w2v_g = reinterpret_cast<const f4*>(p.W2v_l);
        float part[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 32; ++i) {
          const f4 wa = w2v_g[(2 * i) * 64 + lz];
          const f4 wb = w2v_g[(2 * i + 1) * 64 + lz];
          const float sv = sT[i >> 2][i & 3] * inv;
          part[0] += wa[0] * sv; part[1] += wa[1] * sv; part[2] += wa[2] * sv; part[3] += wa[3] * sv;
          part[4] += wb[0] * sv; part[5] += wb[1] * sv; part[6] += wb[2] * sv; part[7] += wb[3] * sv;
          if ((i & 7) == 7) __builtin_amdgcn_sched_barrier(0);       // 16 weight loads in flight at most
        }
#pragma unroll
        for (int d = 0; d < 8; ++d) {
          part[d] += __shfl_xor(part[d], 16);
          part[d] += __shfl_xor(part[d], 32);
        }
        const float swn_ = sw * inv;
        const int o0 = 8 * m + 2 * g;
        const float p0 = g == 0 ? part[0] : (g == 1 ? part[2] : (g == 2 ? part[4] : part[6]));
        const float p1 = g == 0 ? part[1] : (g == 1 ? part[3] : (g == 2 ? part[5] : part[7]));
        float2 o;
        o.x = p0 + b2v_s[o0] * swn_;
        o.y = p1 + b2v_s[o0 + 1] * swn_;
        *reinterpret_cast<float2*>(p.out + (size_t)seg * 128 + o0) = o;
      } else {
        float* sp = p.S + (size_t)seg * 2048 + lane;
#pragma unroll
        for (int tq = 0; tq < 8; ++tq)
#pragma unroll
          for (int r = 0; r < 4; ++r) sp[(tq * 4 + r) * 64] = sT[tq][r] * inv;
        if (g == 0) p.swn[(size_t)seg * 16 + m] = sw * inv;
      }
    }
  }
}

template <bool KNN, bool POS, int MAXT, int THREADS, bool FUSED>
static int launch_na(const PgTopo* t, const PgSegAttn* p, hipStream_t st) {
  constexpr int NSTEP = KNN ? 12 : 0;
  const size_t lds = (256 + 2 * NSTEP * 512 + (POS ? 2048 + 16 : 0) + (FUSED ? 16384 + 128 : 0)) * sizeof(float);
  if (int rc = reserve_lds(reinterpret_cast<const void*>(node_attn_kernel<KNN, POS, MAXT, THREADS, FUSED>), lds, "node_attn")) return rc;
  const int per = THREADS / 64;
  int blocks = (p->n_seg + per - 1) / per;
  if (FUSED && KNN && p->n_seg2 > 0) {
    blocks += (p->n_seg2 + per - 1) / per;
    if (blocks < 16) blocks = 16;                            // (two lists: at least one row of 8 workgroups each)
  }
  if (FUSED) { if (blocks > kNumCU) blocks = kNumCU; }      // one persistent workgroup per CU (64 KB of W2k each)
  else if (KNN && blocks > 3 * kNumCU) blocks = 3 * kNumCU; // LDS-heavy: persistent-ish, the weights are loaded per block
  hipLaunchKernelGGL((node_attn_kernel<KNN, POS, MAXT, THREADS, FUSED>), dim3(blocks), dim3(THREADS), lds, st, *t, *p);
  return check_launch("pg_seg_attn(node)");
}

template <bool FUSED>
static int launch_node_attn_t(const PgTopo* t, const PgSegAttn* p, hipStream_t st) {
  constexpr int TH = FUSED ? 768 : 256;
  const bool knn = p->mode == PG_SEG_KNN_NODE || p->mode == PG_SEG_KNN_POS;
  const bool pos = p->mode == PG_SEG_KNN_POS || p->mode == PG_SEG_BOND_POS;
  if (knn) {
    if (p->knn_k > 32) return -1;
    return pos ? launch_na<true, true, 2, TH, FUSED>(t, p, st) : launch_na<true, false, 2, TH, FUSED>(t, p, st);
  }
  const int tiles = (t->max_nlig + 15) / 16;
  if (tiles > 5) return -1;
  if (pos) {
    if (tiles <= 3) return launch_na<false, true, 3, TH, FUSED>(t, p, st);
    if (tiles == 4) return launch_na<false, true, 4, TH, FUSED>(t, p, st);
    return launch_na<false, true, 5, TH, FUSED>(t, p, st);
  }
  if (tiles <= 3) return launch_na<false, false, 3, TH, FUSED>(t, p, st);
  if (tiles == 4) return launch_na<false, false, 4, TH, FUSED>(t, p, st);
  return launch_na<false, false, 5, TH, FUSED>(t, p, st);
}

// ------------------------------------------------------------------------------------------------------------------------------
// Position-update modes of a SMALL batch: the row tiles of a node over T waves.
// A launch of a few hundred target nodes is bound by the dependent chain inside the one wave that owns a node (fold -> T row tiles,
// each two MLP paths -> softmax -> weighted sum); neither more CUs nor fewer waves per CU change that (tools/experiments).  Here wave
// (slot, tile) of a persistent 12-wave workgroup computes ONE 16-row tile of node `slot` (the query fold redundantly per wave), the
// tiles' per-head maxima and then their softmax numerators e[row,h] and products w[row,h] = e * gate * v meet in LDS, and one wave
// per node runs the two short sequential chains (l += e ; a = fma(w, x_dst - x_src, a)) over them in the order of the one-wave
// kernel: the result is that kernel's, bit for bit.  T = 2 (knn, k <= 32) / 3 / 4 (ligands up to 48 / 64 atoms).
// ------------------------------------------------------------------------------------------------------------------------------
template <bool KNN, int T>
__global__ __launch_bounds__(768, 1) void node_attn_pos_tiled_kernel(PgTopo t, PgSegAttn p) {
  constexpr int NSTEP = KNN ? 12 : 0, THREADS = 768, NPW = 12 / T;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const bpk = lds;
  float* const bpv = lds + 128;
  float* const wf_k = lds + 256;            // [NSTEP][8][64]
  float* const wf_v = wf_k + NSTEP * 512;
  float* const w2xv = wf_v + NSTEP * 512;   // [32][64]
  float* const b2xv = w2xv + 2048;
  float* const w2k = b2xv + 16;             // lane-fixed W2k [64][64][4]
  float* const xmax = w2k + 16384;          // [12 waves][64]      lane-local maxima of a wave's tile
  float* const xew = xmax + 12 * 64;        // [12 waves][64][8]   e[4] | w[4] of a wave's tile
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int slot = wave / T, tile0 = wave - slot * T;
  for (int i = tid; i < 128; i += THREADS) { bpk[i] = p.ln_bk[i]; bpv[i] = p.ln_bv[i]; }
  for (int i = tid; i < NSTEP * 128; i += THREADS) {
    reinterpret_cast<f4*>(wf_k)[i] = reinterpret_cast<const f4*>(p.Wf_k)[i];
    reinterpret_cast<f4*>(wf_v)[i] = reinterpret_cast<const f4*>(p.Wf_v)[i];
  }
  for (int i = tid; i < 512; i += THREADS) reinterpret_cast<f4*>(w2xv)[i] = reinterpret_cast<const f4*>(p.W2xv_l)[i];
  for (int i = tid; i < 16; i += THREADS) b2xv[i] = p.b2xv[i];
  for (int i = tid; i < 4096; i += THREADS) reinterpret_cast<f4*>(w2k)[i] = reinterpret_cast<const f4*>(p.W2k_l)[i];
  __syncthreads();

  // XCD-affine hand-out of chunks of NPW nodes (as in node_attn_kernel)
  const int n_chunks = (p.n_seg + NPW - 1) / NPW;
  const int n_wg = (int)gridDim.x, n_x = n_wg < 8 ? n_wg : 8;
  const int xcd = (int)blockIdx.x % n_x, jx = (int)blockIdx.x / n_x, n_jx = (n_wg - xcd + n_x - 1) / n_x;
  const int c_lo = (int)((long long)n_chunks * xcd / n_x), c_hi = (int)((long long)n_chunks * (xcd + 1) / n_x);
  const int n_it = (c_hi - c_lo + n_jx - 1) / n_jx;          // (uniform per workgroup: every wave runs every barrier)
  for (int it = 0; it < n_it; ++it) {
    const int ch = c_lo + jx + it * n_jx;
    const int si = ch * NPW + slot;
    const bool active = ch < c_hi && si < p.n_seg;
    // opaque copies of the lane id and of the wave's tile index: everything addressed through them (the LDS weight tables, the row
    // gathers, the per-channel offsets) is then not loop-invariant, and the compiler cannot hoist ~100 address registers out of
    // the node loop and spill them (the tile index is a compile-time constant in the one-wave kernel, a per-wave constant here)
    int lw = lane, tile = __builtin_amdgcn_readfirstlane(tile0);
    asm volatile("" : "+v"(lw));
    asm volatile("" : "+s"(tile));
    const int g = lw >> 4, m = lw & 15;
    int seg = 0, n_rows = 0, lig0 = 0, n = 0, li = 0, n_tiles = 0;
    const int* eid_g = nullptr;
    float xd[3] = {0.f, 0.f, 0.f}, nd[3] = {0.f, 0.f, 0.f};
    f4 lg = {NA_NEG, NA_NEG, NA_NEG, NA_NEG}, vv = {0.f, 0.f, 0.f, 0.f};
    if (active) {
      seg = p.seg_ids ? p.seg_ids[si] : si;
      if constexpr (KNN) {
        n_rows = p.deg[seg];
      } else {
        const int gi = t.ctx_graph[seg];
        n = t.g_nlig[gi];
        lig0 = t.g_ctx_off[gi] + t.g_nph[gi];
        li = seg - lig0;
        eid_g = t.eid + t.g_eid_off[gi];
        n_rows = n;
      }
      n_tiles = (n_rows + 15) >> 4;
#pragma unroll
      for (int c = 0; c < 3; ++c) xd[c] = p.x[seg * 3 + c];
      if constexpr (KNN) {
#pragma unroll
        for (int c = 0; c < 3; ++c) nd[c] = p.nrm[seg * 3 + c];
      }
    }
    {   // (unconditional: a wave without a node or beyond the node's last tile computes masked rows of node `seg` = 0 / of the
        //  node itself; nothing of it is used -- one straight-line body keeps the register allocation that of the one-wave kernel)
      const float* ckp = p.Cdst_k + (size_t)seg * p.ld_cdst;
      const float* cvp = p.Cdst_v + (size_t)seg * p.ld_cdst;
      f4 U[8];
      {
        const float* qp = p.q + (size_t)seg * 128 + 8 * m;
        const f4 qa = *reinterpret_cast<const f4*>(qp), qb = *reinterpret_cast<const f4*>(qp + 4);
        // the table reads are addressed through a lane id that "depends" on the query: the scheduler otherwise issues all 64 of them
        // while the query is still in flight and spills every one (ds_read -> scratch_store pairs, 770 B of scratch per lane)
        int lq = lw;
        asm volatile("" : "+v"(lq) : "v"(qa[0]), "v"(qb[0]));
#pragma unroll
        for (int tq = 0; tq < 8; ++tq)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = tq * 4 + r;
            const f4 wa = *reinterpret_cast<const f4*>(w2k + ((size_t)(2 * i) * 64 + lq) * 4);
            const f4 wb = *reinterpret_cast<const f4*>(w2k + ((size_t)(2 * i + 1) * 64 + lq) * 4);
            U[tq][r] = (qa[0] * wa[0] + qa[1] * wa[1]) + (qa[2] * wa[2] + qa[3] * wa[3]) +
                       (qb[0] * wb[0] + qb[1] * wb[1]) + (qb[2] * wb[2] + qb[3] * wb[3]);
            if (r == 3) {
              // the four values are pinned here (an empty asm that "rewrites" them): otherwise the vectoriser gathers all 64 table
              // reads of the fold in front of its arithmetic and every one of them is spilled (770 B of scratch per lane)
              asm volatile("" : "+v"(U[tq][0]), "+v"(U[tq][1]), "+v"(U[tq][2]), "+v"(U[tq][3]) : : "memory");
              __builtin_amdgcn_sched_barrier(0);
            }
          }
      }
      const int k = tile * 16 + m;
      bool valid = k < n_rows;
      int src = 0, crow = 0;
      if constexpr (KNN) {
        if (valid) { src = p.nbr[(size_t)seg * p.knn_k + k]; crow = src; }
      } else {
        valid = valid && k != li;
        if (valid) { src = lig0 + k; crow = eid_g[k * n + li]; }
      }
      float feat[NSTEP > 0 ? NSTEP : 1];
      bool has_lig = false, has_ph = false;
      if constexpr (KNN) {
        float d = 0.f, dots[3] = {0.f, 0.f, 0.f};
        bool src_lig = false;
        if (valid) {
          float xs[3], ns[3];
#pragma unroll
          for (int c = 0; c < 3; ++c) { xs[c] = p.x[src * 3 + c]; ns[c] = p.nrm[src * 3 + c]; }
          const float r0 = xd[0] - xs[0], r1 = xd[1] - xs[1], r2 = xd[2] - xs[2];
          d = sqrtf(r0 * r0 + r1 * r1 + r2 * r2);
          dots[0] = ns[0] * nd[0] + ns[1] * nd[1] + ns[2] * nd[2];
          dots[1] = -(ns[0] * r0 + ns[1] * r1 + ns[2] * r2);
          dots[2] = -(nd[0] * r0 + nd[1] * r1 + nd[2] * r2);
          src_lig = t.ctx_is_lig[src] != 0;
        }
#pragma unroll
        for (int st = 0; st < 5; ++st) {
          const float sv = valid ? smear(d, 4 * st + g) : 0.f;
          feat[st] = src_lig ? sv : 0.f;
          feat[5 + st] = src_lig ? 0.f : sv;
        }
        has_lig = __ballot(valid && src_lig) != 0ull;
        has_ph = __ballot(valid && !src_lig) != 0ull;
        feat[10] = g == 0 ? dots[0] : (g == 1 ? dots[1] : (g == 2 ? dots[2] : ((valid && src_lig) ? 1.f : 0.f)));
        feat[11] = g == 0 ? ((valid && !src_lig) ? 1.f : 0.f) : (g == 3 ? 1.f : 0.f);
      }
      // ---- key MLP -> logits of the tile's rows ----
      f4 hid[8];
      {
        const float* pk = p.Csrc_k + (size_t)crow * p.ld_csrc + 4 * g;
#pragma unroll
        for (int tq = 0; tq < 8; ++tq) {
          f4 c = {0.f, 0.f, 0.f, 0.f};
          if (valid) c = *reinterpret_cast<const f4*>(pk + 16 * tq);
          if constexpr (!KNN) c += *reinterpret_cast<const f4*>(ckp + 16 * tq + 4 * g);
          hid[tq] = c;
        }
      }
#pragma unroll
      for (int blk3 = 0; blk3 < (NSTEP ? 3 : 0); ++blk3) {
        if ((blk3 == 0 && !has_lig) || (blk3 == 1 && !has_ph)) continue;
#pragma unroll
        for (int st = 5 * blk3; st < (blk3 == 2 ? 12 : 5 * blk3 + 5); ++st)
#pragma unroll
          for (int tq = 0; tq < 8; ++tq) {
            float w = wf_k[(st * 8 + tq) * 64 + lw];
            if (st == 11) w = g == 3 ? ckp[16 * tq + m] : w;
            hid[tq] = mfma16(w, feat[st], hid[tq]);
          }
      }
      const float rs = ln_fold_k(hid, bpk, g);
      f4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int tq = 0; tq < 8; ++tq)
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
          acc = mfma16(hid[tq][r], U[tq][r], acc);
          acc2 = mfma16(hid[tq][r + 1], U[tq][r + 1], acc2);
        }
      acc += acc2;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kr = tile * 16 + 4 * g + r;
        const bool vr = kr < n_rows && (KNN || kr != li);
        const float sc_ = acc[r] * __shfl(rs, 4 * g + r);
        lg[r] = vr ? sc_ : NA_NEG;
      }
      // ---- value MLP of the position update: v[row,h] ----
      f4 hx[8];
      {
        const float* pk = p.Csrc_v + (size_t)crow * p.ld_csrc + 4 * g;
#pragma unroll
        for (int tq = 0; tq < 8; ++tq) {
          f4 c = {0.f, 0.f, 0.f, 0.f};
          if (valid) c = *reinterpret_cast<const f4*>(pk + 16 * tq);
          if constexpr (!KNN) c += *reinterpret_cast<const f4*>(cvp + 16 * tq + 4 * g);
          hx[tq] = c;
        }
      }
#pragma unroll
      for (int blk3 = 0; blk3 < (NSTEP ? 3 : 0); ++blk3) {
        if ((blk3 == 0 && !has_lig) || (blk3 == 1 && !has_ph)) continue;
#pragma unroll
        for (int st = 5 * blk3; st < (blk3 == 2 ? 12 : 5 * blk3 + 5); ++st)
#pragma unroll
          for (int tq = 0; tq < 8; ++tq) {
            float w = wf_v[(st * 8 + tq) * 64 + lw];
            if (st == 11) w = g == 3 ? cvp[16 * tq + m] : w;
            hx[tq] = mfma16(w, feat[st], hx[tq]);
          }
      }
      const float rsx = ln_fold_k(hx, bpv, g);
      f4 a1 = {0.f, 0.f, 0.f, 0.f}, a2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int tq = 0; tq < 8; ++tq)
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
          a1 = mfma16(hx[tq][r], w2xv[(tq * 4 + r) * 64 + lw], a1);
          a2 = mfma16(hx[tq][r + 1], w2xv[(tq * 4 + r + 1) * 64 + lw], a2);
        }
      a1 += a2;
      const float bx = b2xv[m];
#pragma unroll
      for (int r = 0; r < 4; ++r) vv[r] = a1[r] * __shfl(rsx, 4 * g + r) + bx;
    }

    // ---- the tiles of a node meet: per-head maximum (exact in any order) ----
    xmax[wave * 64 + lane] = fmaxf(fmaxf(lg[0], lg[1]), fmaxf(lg[2], lg[3]));
    __syncthreads();
    float mx = NA_NEG;
#pragma unroll
    for (int tt = 0; tt < T; ++tt) mx = fmaxf(mx, xmax[(slot * T + tt) * 64 + lane]);
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    {
      f4 gate = {1.f, 1.f, 1.f, 1.f};
      if constexpr (KNN) {
        if (active && tile < n_tiles) gate = *reinterpret_cast<const f4*>(p.ew + (size_t)seg * p.knn_k + tile * 16 + 4 * g);
      }
      f4 e4, w4;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = lg[r] > 0.5f * NA_NEG ? __builtin_amdgcn_exp2f(lg[r] - mx) : 0.f;
        e4[r] = e;
        w4[r] = (e * gate[r]) * vv[r];
      }
      *reinterpret_cast<f4*>(xew + ((size_t)wave * 64 + lane) * 8) = e4;
      *reinterpret_cast<f4*>(xew + ((size_t)wave * 64 + lane) * 8 + 4) = w4;
    }
    __syncthreads();
    // ---- one wave per node: the sequential chains in the one-wave kernel's order (tile-major, row inner) ----
    if (active && tile == 0) {
      float l = 0.f;
#pragma unroll
      for (int tt = 0; tt < T; ++tt) {
        const f4 e4 = *reinterpret_cast<const f4*>(xew + ((size_t)(slot * T + tt) * 64 + lane) * 8);
#pragma unroll
        for (int r = 0; r < 4; ++r) l += e4[r];
      }
      l += __shfl_xor(l, 16);
      l += __shfl_xor(l, 32);
      const float inv = l > 0.f ? 1.0f / l : 0.f;
      float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
      for (int tt = 0; tt < T; ++tt)
        if (tt < n_tiles) {
          const f4 w4 = *reinterpret_cast<const f4*>(xew + ((size_t)(slot * T + tt) * 64 + lane) * 8 + 4);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int kr = tt * 16 + 4 * g + r;
            const bool vr = kr < n_rows && (KNN || kr != li);
            if (vr) {
              const int src = KNN ? p.nbr[(size_t)seg * p.knn_k + kr] : lig0 + kr;
              const float w = w4[r];
              a0 = fmaf(w, xd[0] - p.x[src * 3], a0);
              a1 = fmaf(w, xd[1] - p.x[src * 3 + 1], a1);
              a2 = fmaf(w, xd[2] - p.x[src * 3 + 2], a2);
            }
          }
        }
      a0 = wave_sum(a0 * inv) * (1.f / 16.f);
      a1 = wave_sum(a1 * inv) * (1.f / 16.f);
      a2 = wave_sum(a2 * inv) * (1.f / 16.f);
      if (lane == 0) {
        if (p.accumulate_dx) { p.dx[seg * 3] += a0; p.dx[seg * 3 + 1] += a1; p.dx[seg * 3 + 2] += a2; }
        else { p.dx[seg * 3] = a0; p.dx[seg * 3 + 1] = a1; p.dx[seg * 3 + 2] = a2; }
      }
    }
  }
}

template <bool KNN, int T>
static int launch_pos_tiled(const PgTopo* t, const PgSegAttn* p, hipStream_t st) {
  constexpr int NSTEP = KNN ? 12 : 0, NPW = 12 / T;
  const size_t lds = (256 + 2 * NSTEP * 512 + 2048 + 16 + 16384 + 12 * 64 + 12 * 64 * 8) * sizeof(float);
  if (int rc = reserve_lds(reinterpret_cast<const void*>(node_attn_pos_tiled_kernel<KNN, T>), lds, "node_attn(pos, tiled)")) return rc;
  int blocks = (p->n_seg + NPW - 1) / NPW;
  if (blocks > kNumCU) blocks = kNumCU;
  hipLaunchKernelGGL((node_attn_pos_tiled_kernel<KNN, T>), dim3(blocks), dim3(768), lds, st, *t, *p);
  return check_launch("pg_seg_attn(node pos, tiled)");
}

// the tiled form serves the fused sampler calls of the position modes (no second target list, no training outputs); -1 otherwise
static int launch_pos_tiled_any(const PgTopo* t, const PgSegAttn* p, hipStream_t st) {
  if (p->alpha || p->n_seg2 > 0) return -1;
  if (p->mode == PG_SEG_KNN_POS) return p->knn_k <= 32 ? launch_pos_tiled<true, 2>(t, p, st) : -1;
  const int tiles = (t->max_nlig + 15) / 16;
  if (tiles <= 2) return launch_pos_tiled<false, 2>(t, p, st);
  if (tiles == 3) return launch_pos_tiled<false, 3>(t, p, st);
  if (tiles == 4) return launch_pos_tiled<false, 4>(t, p, st);
  return -1;
}

// returns -1 when the shape is outside what the two-pass kernels hold in registers (caller falls back to seg_attn.hip).
// Fused form: q and W2k_l given (and, for the node-update modes, W2v_l, b2v, out) -- see node_attn_fused_request()
bool node_attn_fused_request(const PgSegAttn* p) {
  const bool pos = p->mode == PG_SEG_KNN_POS || p->mode == PG_SEG_BOND_POS;
  return p->mode <= PG_SEG_BOND_POS && p->q && p->W2k_l && (pos || (p->W2v_l && p->b2v && p->out));
}

int launch_node_attn(const PgTopo* t, const PgSegAttn* p, hipStream_t st) {
  // the weight tables go to LDS in 16-byte pieces
  if ((((size_t)p->Wf_k | (size_t)p->Wf_v | (size_t)p->Wf_k2 | (size_t)p->Wf_v2 | (size_t)p->W2xv_l) & 15) != 0) {
    set_error("pg_seg_attn: Wf_k / Wf_v / W2xv_l must be 16-byte aligned");
    return PG_ERR_ARG;
  }
  if (!node_attn_fused_request(p)) return launch_node_attn_t<false>(t, p, st);
  if (p->pos_tiled && (p->mode == PG_SEG_KNN_POS || p->mode == PG_SEG_BOND_POS)) {
    const int rc = launch_pos_tiled_any(t, p, st);
    if (rc != -1) return rc;
  }
  return launch_node_attn_t<true>(t, p, st);
}

}  // namespace pg
