// Bond-triplet attention (BondUpdateLayer, models/uni_denoiser.py:101-165), sampling form, with the rows of a SOURCE ATOM
// staged in LDS.
//
// The triplets of a bond edge j->i read the rows P[k->j] of every other atom k (uni_denoiser.py:123-135): all n-1 segments
// j->i of one source atom j read the SAME n-1 rows.  With the host mirror's target-major bond order (phoregen_amd/plan.py)
// those rows are one contiguous (n-1) x 1 KB block of P, so a workgroup
//   * takes a group of consecutive source atoms of one ligand (as many as fit 80 staged rows),
//   * streams their P blocks into LDS once (coalesced float4 copies, no edge-id lookups, no gathers),
//   * lets its waves work off the A*(n-1) segments, each row tile's MFMA C-operand coming from LDS (ds_read_b128 in the
//     key layout, ds_read_b32 in the value layout) instead of an L2 gather that was re-fetched ~39x,
//   * and pulls the next group from a global counter (longest groups first).
// LDS: lane-fixed W2k (64 KB, query fold), feature weights (12 KB), biases, the ligand's coordinates, 80 staged rows (81 KB).
// The value unfold streams W2v through L2 (64 KB per segment, software-prefetched in batches of 16 float4 per lane): the two
// 64 KB weight tables and the staged rows do not fit the 160 KB together, and the unfold has no data it must wait for.
// Per-segment arithmetic is that of triplet.hip (two passes, folded LayerNorm, base-2 softmax); the per-segment constant
// Q = Wg2 . smear(d_ji) comes in as a row of Cdst (a [n_bond,20]x[20,256] GEMM per layer) and reaches the MFMA through the
// spare feature column.
// Lane l = (g = l>>4, m = l&15); 16x16x4 maps as in seg_attn.hip.
#include "common.h"
#include "../../include/phoregen_hip.h"

namespace pg {

// -DPG_T2_PROF (tools/prof_triplet2.sh): per-wave cycle counters of the kernel's sections, added into p.alpha[0..7] as
// integers (the sampling form leaves that pointer unused); the product build contains none of this
#ifdef PG_T2_PROF
#define T2_STAMP(slot) do { const unsigned long long t2_now = __builtin_amdgcn_s_memtime(); t2_acc[slot] += t2_now - t2_last; t2_last = t2_now; } while (0)
#else
#define T2_STAMP(slot) do { } while (0)
#endif

typedef float t2_f2 __attribute__((ext_vector_type(2)));

// -DPG_DEGRADE_BITS=n (tools/degraded_build_check.sh, never the product build): the activations that enter the second-layer
// products lose their n lowest mantissa bits -- a deliberately less precise kernel, to show that the aggregate parity guard
// (tests/helpers.py AGG_MEDIAN_BOUND) fails on a uniform precision regression that the per-step bounds let through
#ifdef PG_DEGRADE_BITS
#define T2_DEGRADE(v) __builtin_bit_cast(float, __builtin_bit_cast(unsigned, (v)) & (0xffffffffu << PG_DEGRADE_BITS))
#else
#define T2_DEGRADE(v) (v)
#endif

constexpr int T2_ROW = 260;          // floats per staged row: P_k[128] | P_v[128] | 4 (bank spread of the b128 key-layout reads)
constexpr int T2_ROWS = 80;          // staged rows per workgroup
constexpr int T2_XS = 96;            // ligand atoms whose coordinates are staged
constexpr float T2_NEG = -1.0e30f;

template <int CTRL>
__device__ __forceinline__ float t2_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float t2_row16_sum(float v) {
  v += t2_dpp<0xB1>(v);    // quad_perm [1,0,3,2]
  v += t2_dpp<0x4E>(v);    // quad_perm [2,3,0,1]
  v += t2_dpp<0x141>(v);   // row_half_mirror
  v += t2_dpp<0x140>(v);   // row_mirror
  return v;
}
__device__ __forceinline__ float t2_from_lane(float v, int src_lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src_lane << 2, __builtin_bit_cast(int, v)));
}

// sin(w*theta) or cos(w*theta) for 0 <= arg <= ~10 (same reduction + polynomials as triplet.hip)
__device__ __forceinline__ float t2_sincos(float arg, bool want_cos) {
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

// sin and cos of one argument (0 <= arg <= ~4) from ONE range reduction: k = rint(arg * 2/pi), r = arg - k pi/2 (two constants),
// both polynomials on [-pi/4, pi/4], quadrant rotation
__device__ __forceinline__ void t2_sincos_pair(float arg, float& sn, float& cs) {
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

__device__ __constant__ const float kT2Freq[12] = {0.f, 1.f, 2.f, 3.f, 0.5f, (float)(1.0 / 3.0), 1.f, 2.f, 3.f, 0.5f,
                                                    (float)(1.0 / 3.0), 0.f};

constexpr size_t t2_lds_floats() { return 16384 + 2 * 1536 + 3 * 128 + 3 * T2_XS + 4 + (size_t)T2_ROWS * T2_ROW; }

// TRAIN: the training forward (PhoreDiff.compute_loss): the normalised aggregate S and the attention mass go to
// pg_attn_unfold_value like in the node modes, and the softmax weights alpha[seg][atom k][head] are left for the one-pass adjoint
// (pg_seg_attn_bwd); rows are indexed by the atom k there, so the row of the source atom j itself (which this kernel never
// visits) is written as zero
template <int THREADS, int MAXT, bool TRAIN = false>
__global__ __launch_bounds__(THREADS) void triplet2_kernel(PgTopo t, PgSegAttn p) {
  constexpr bool PRE = THREADS <= 512;          // next-segment prefetch of the per-segment global inputs
  constexpr int WAVES = THREADS / 64;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const w2k_l = lds;                     // [64][64][4]
  float* const wf_k = w2k_l + 16384;            // [3][8][64]
  float* const wf_v = wf_k + 1536;
  float* const bk = wf_v + 1536;                // b' = beta/|gamma| of the key / value LayerNorm, value bias
  float* const bv = bk + 128;
  float* const b2v = bv + 128;
  float* const xs = b2v + 128;                  // [T2_XS][3] coordinates of the current ligand
  int* const ctrl = reinterpret_cast<int*>(xs + 3 * T2_XS);
  float* const pbuf = xs + 3 * T2_XS + 4;       // [T2_ROWS][T2_ROW]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, m = lane & 15;

  // LDS copies are laid out for 16-byte reads: feature weights [step][half][lane][4] (a lane's 8 tq values = two ds_read_b128),
  // the value LayerNorm shift [m][8] (channel 16 tq + m at m*8 + tq)
  for (int i = tid; i < 128; i += THREADS) { bk[i] = p.ln_bk[i]; bv[(i & 15) * 8 + (i >> 4)] = p.ln_bv[i]; b2v[i] = TRAIN ? 0.f : p.b2v[i]; }
  for (int i = tid; i < 1536; i += THREADS) {             // source index i = (st * 8 + tq) * 64 + lane
    const int ln_ = i & 63, tq_ = (i >> 6) & 7, st_ = i >> 9;
    const int d_ = ((st_ * 2 + (tq_ >> 2)) * 64 + ln_) * 4 + (tq_ & 3);
    wf_k[d_] = p.Wf_k[i]; wf_v[d_] = p.Wf_v[i];
  }
  for (int i = tid; i < 4096; i += THREADS) reinterpret_cast<f4*>(w2k_l)[i] = reinterpret_cast<const f4*>(p.W2k_l)[i];
  const f4* const w2v_g = reinterpret_cast<const f4*>(p.W2v_l);
  const int4* const iters = reinterpret_cast<const int4*>(p.tri_iters);
#ifdef PG_T2_PROF
  unsigned long long t2_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long t2_last = __builtin_amdgcn_s_memtime();
#endif

  for (;;) {
    __syncthreads();                            // the previous group's rows / ctrl word are no longer read
    if (tid == 0) ctrl[0] = atomicAdd(p.tri_counter, 1);
    __syncthreads();
    const int it = __builtin_amdgcn_readfirstlane(ctrl[0]);
    T2_STAMP(0);                                  // waiting for the group hand-out (barriers, queue)
    if (it >= p.n_tri_iters) {
      // the queue leaves itself ready for the next launch: the last workgroup out (all others have made their final draw before
      // they count themselves out) zeroes the head and the exit count -- no memset between the six launches of a step
      if (tid == 0 && atomicAdd(p.tri_counter + 1, 1) == (int)gridDim.x - 1) {
        p.tri_counter[0] = 0;
        p.tri_counter[1] = 0;
        __threadfence();
      }
      break;
    }
    const int4 d = iters[it];
    const int lig0 = __builtin_amdgcn_readfirstlane(d.x);
    const int n = __builtin_amdgcn_readfirstlane(d.y & 0xff), j0 = __builtin_amdgcn_readfirstlane((d.y >> 8) & 0xff);
    const int A = __builtin_amdgcn_readfirstlane(d.y >> 16), bond_off = __builtin_amdgcn_readfirstlane(d.z);
    const int nm1 = n - 1;
    {   // stage the P blocks of source atoms j0 .. j0+A-1 (contiguous rows of the target-major bond order) and the coordinates
      const f4* src = reinterpret_cast<const f4*>(p.Csrc_k + (size_t)(bond_off + j0 * nm1) * 256);
      const int n4 = A * nm1 * 64;
      for (int idx = tid; idx < n4; idx += THREADS)
        *reinterpret_cast<f4*>(pbuf + (idx >> 6) * T2_ROW + (idx & 63) * 4) = __builtin_nontemporal_load(src + idx);   // read once
      for (int i = tid; i < n * 3; i += THREADS) xs[i] = p.x[(size_t)lig0 * 3 + i];
    }
    __syncthreads();
    T2_STAMP(1);                                  // staging the P blocks
    // the group's segments, or the part of them this queue entry hands out (small batches: plan.py splits groups into whole
    // 12-wave rounds so that the queue can level the workgroups)
    const int s_begin = __builtin_amdgcn_readfirstlane(d.w & 0xffff);
    const int n_seg = __builtin_amdgcn_readfirstlane((d.w >> 16) ? (d.w >> 16) : A * nm1);
    // a segment j->i visits the nm1 - 1 rows k != j, k != i of its source atom's block (round 6: the target's own row, which used to be computed
    // and masked, is skipped in the row index instead -- ligands of 34 / 50 atoms need a row tile less, 2.7 % of the headline batch's tiles)
    const int nr = nm1 - 1;
    const int n_tiles = (nr + 15) >> 4;

    // per-segment global inputs (the Q row of Cdst, the query, the residual row) are fetched ONE SEGMENT AHEAD: their HBM
    // round trip overlaps the previous segment's arithmetic instead of opening every segment with a wait
    float nQk[8], nQv[8];
    f4 nqa, nqb;
    float2 nrs = {0.f, 0.f};
#define T2_FETCH(S)                                                                                                    \
    {                                                                                                                  \
      const int a_ = (S) / nm1, ip_ = (S) - a_ * nm1, j_ = j0 + a_, i_ = ip_ + (ip_ >= j_ ? 1 : 0);                    \
      const size_t seg_ = (size_t)(bond_off + i_ * nm1 + (j_ < i_ ? j_ : j_ - 1));                                     \
      /* the vector-memory counter is in-order: the query (needed first, by the fold) goes out first, the Q rows after it */ \
      const float* qp_ = p.q + seg_ * 128 + 8 * m;                                                                     \
      nqa = __builtin_nontemporal_load(reinterpret_cast<const f4*>(qp_));                                              \
      nqb = __builtin_nontemporal_load(reinterpret_cast<const f4*>(qp_ + 4));                                          \
      const float* qk_ = p.Cdst_k + seg_ * p.ld_cdst + m;                                                              \
      const float* qv_ = p.Cdst_v + seg_ * p.ld_cdst + m;                                                              \
      _Pragma("unroll") for (int tq = 0; tq < 8; ++tq) { nQk[tq] = __builtin_nontemporal_load(qk_ + 16 * tq); nQv[tq] = __builtin_nontemporal_load(qv_ + 16 * tq); } \
      if constexpr (!TRAIN) { const t2_f2 r_ = __builtin_nontemporal_load(reinterpret_cast<const t2_f2*>(p.resid + seg_ * 128 + 8 * m + 2 * g)); nrs.x = r_[0]; nrs.y = r_[1]; } \
    }
    if (PRE && s_begin + wave < n_seg) T2_FETCH(s_begin + wave)

    for (int s = s_begin + wave; s < n_seg; s += WAVES) {
      const int a = s / nm1, ip = s - a * nm1;            // source atom of the group, target index among the other atoms
      const int j = j0 + a, i = ip + (ip >= j ? 1 : 0);
      const int seg = bond_off + i * nm1 + (j < i ? j : j - 1);          // internal id of edge j->i
      const float* const prow = pbuf + a * nm1 * T2_ROW;
      if (!PRE) T2_FETCH(s)                               // (12-wave variant: no register room for the look-ahead)
      float cQk[8], cQv[8];
#pragma unroll
      for (int tq = 0; tq < 8; ++tq) { cQk[tq] = nQk[tq]; cQv[tq] = nQv[tq]; }
      const f4 qa = nqa, qb = nqb;
      const float2 rsd = nrs;
      if (PRE && s + WAVES < n_seg) T2_FETCH(s + WAVES)
      // opaque copy of the lane id: global addresses built from it (20 rows of Wg2, 64 slices of W2v) are then not
      // loop-invariant, so the compiler cannot hoist ~150 address registers out of the segment loop and spill them
      int lz = lane;
      asm volatile("" : "+v"(lz));

      // ---- Q = Wg2 . smear(d_ji) of this segment, precomputed as a row of Cdst (one small GEMM per layer): the g == 3 lanes feed
      //      it to the MFMA as the weight of the constant feature 11, the other lanes carry the angular weights of step 2 ----
      float wk2[8], wv2[8];
#pragma unroll
      for (int tq = 0; tq < 8; ++tq) {
        wk2[tq] = g == 3 ? cQk[tq] : wf_k[((4 + (tq >> 2)) * 64 + lane) * 4 + (tq & 3)];     // feature step 2: f = 8 + g, f = 11 carries Q
        wv2[tq] = g == 3 ? cQv[tq] : wf_v[((4 + (tq >> 2)) * 64 + lane) * 4 + (tq & 3)];
      }
      T2_STAMP(2);                                // Q
      const float xi0 = xs[i * 3], xi1 = xs[i * 3 + 1], xi2 = xs[i * 3 + 2];
      const float u0 = xs[j * 3] - xi0, u1 = xs[j * 3 + 1] - xi1, u2 = xs[j * 3 + 2] - xi2;

      float feat[MAXT][3];
      f4 lg[MAXT];

      // angle at i between j and k (uni_denoiser.py:131-135) for row m of tile g (+4 for a fifth tile): one atan2 per lane and
      // segment instead of one per tile, the tiles then fetch theta of (tile, m) from lane 16 tile + m.  Rows past the ligand
      // take the first row's angle (finite; such rows are masked by selects on the logits).
      float th_own[(MAXT + 3) / 4];
#pragma unroll
      for (int rep = 0; rep < (MAXT + 3) / 4; ++rep) {
        const int kp = (4 * rep + g) * 16 + m;
        const int kq = kp < nr ? kp : 0;
        const int kc = kq + (kq >= ip ? 1 : 0);                            // staged row: the block holds every k != j, row ip is the target's own
        const int k = kc + (kc >= j ? 1 : 0);
        const float v0 = xs[k * 3] - xi0, v1 = xs[k * 3 + 1] - xi1, v2 = xs[k * 3 + 2] - xi2;
        const float dt = u0 * v0 + u1 * v1 + u2 * v2;
        const float c0 = u1 * v2 - u2 * v1, c1 = u2 * v0 - u0 * v2, c2 = u0 * v1 - u1 * v0;
        th_own[rep] = atan2f(sqrtf(c0 * c0 + c1 * c1 + c2 * c2), dt);
      }

      // =============================== pass A: logits of every row ===============================
      {
        f4 U[8];
        {
#pragma unroll
          for (int tq = 0; tq < 8; ++tq)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int w = (tq * 4 + r) * 2;
              const f4 wa = *reinterpret_cast<const f4*>(w2k_l + ((size_t)w * 64 + lane) * 4);
              const f4 wb = *reinterpret_cast<const f4*>(w2k_l + ((size_t)(w + 1) * 64 + lane) * 4);
              float u = qa[0] * wa[0];                     // one FMA chain per value: 8 VALU instead of 11 for the pairwise tree
              u = fmaf(qa[1], wa[1], u); u = fmaf(qa[2], wa[2], u); u = fmaf(qa[3], wa[3], u);
              u = fmaf(qb[0], wb[0], u); u = fmaf(qb[1], wb[1], u); u = fmaf(qb[2], wb[2], u); u = fmaf(qb[3], wb[3], u);
              U[tq][r] = u;
            }
        }
#pragma unroll
        for (int tile = 0; tile < MAXT; ++tile) {
          lg[tile] = (f4){T2_NEG, T2_NEG, T2_NEG, T2_NEG};
          feat[tile][0] = feat[tile][1] = feat[tile][2] = 0.f;
          if (tile < n_tiles) {
            const int kp = tile * 16 + m;                                  // row = k-th atom that is neither j nor i
            const int kq = kp < nr ? kp : 0;                               // rows past the ligand read the first row (finite, masked below)
            const int kc = kq + (kq >= ip ? 1 : 0);
            const float theta = t2_from_lane(th_own[tile >> 2], 16 * (tile & 3) + m);
            // angular features of row kp for f = 4 step + g  (common.py:85); f = 11 carries the per-segment constant Q
            // the four lanes of a row share the work: lane g evaluates sin / cos of theta, theta/2, theta/3 (one range reduction
            // each; g = 3 idles), three ds_bpermutes hand round what the others need, the multiples come from
            // sin 2t = 2 s c, sin 3t = s (3 - 4 s^2), cos 2t = 1 - 2 s^2, cos 3t = c (4 c^2 - 3)
            float sg, cg;
            t2_sincos_pair(theta * (g == 0 ? 1.0f : (g == 1 ? 0.5f : (float)(1.0 / 3.0))), sg, cg);
            const float s1 = t2_from_lane(sg, m), c1 = t2_from_lane(cg, m);
            const float sx = t2_from_lane(sg, m + (g == 0 ? 16 : 32));      // g = 0: sin(theta/2), g = 1: sin(theta/3)
            // f = 4 st + g: [theta, sin t, sin 2t, sin 3t | sin t/2, sin t/3, cos t, cos 2t | cos 3t, cos t/2, cos t/3, 1 (Q)]
            feat[tile][0] = g == 0 ? theta : (g == 1 ? s1 : (g == 2 ? 2.0f * s1 * c1 : s1 * fmaf(-4.0f * s1, s1, 3.0f)));
            feat[tile][1] = g < 2 ? sx : (g == 2 ? c1 : fmaf(-2.0f * s1, s1, 1.0f));
            feat[tile][2] = g == 0 ? c1 * fmaf(4.0f * c1, c1, -3.0f) : (g == 3 ? 1.0f : cg);
            // hidden^T[c, row] = P_k[row][c] + Q_k[c] + Wf_k . feat   (past-the-end rows are computed like any other: their logits are
            // replaced below, so no per-element selects are needed)
            f4 hid[8];
            const float* pk = prow + kc * T2_ROW + 4 * g;
#pragma unroll
            for (int tq = 0; tq < 8; ++tq) hid[tq] = *reinterpret_cast<const f4*>(pk + 16 * tq);
#pragma unroll
            for (int st = 0; st < 2; ++st)
              {
                const f4 wa = *reinterpret_cast<const f4*>(wf_k + ((st * 2) * 64 + lane) * 4);
                const f4 wb = *reinterpret_cast<const f4*>(wf_k + ((st * 2 + 1) * 64 + lane) * 4);
#pragma unroll
                for (int tq = 0; tq < 4; ++tq) hid[tq] = mfma16(wa[tq], feat[tile][st], hid[tq]);
#pragma unroll
                for (int tq = 0; tq < 4; ++tq) hid[4 + tq] = mfma16(wb[tq], feat[tile][st], hid[4 + tq]);
              }
#pragma unroll
            for (int tq = 0; tq < 8; ++tq) hid[tq] = mfma16(wk2[tq], feat[tile][2], hid[tq]);
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
            f4 acc4[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) acc4[r] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int tq = 0; tq < 8; ++tq) {
              const f4 bt = *reinterpret_cast<const f4*>(bk + 16 * tq + 4 * g);
#pragma unroll
              for (int r = 0; r < 4; ++r) acc4[r] = mfma16(T2_DEGRADE(fmaxf(fmaf(bt[r], sigma, hid[tq][r]), 0.f)), U[tq][r], acc4[r]);
            }
            f4 acc = (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] *= __shfl(rs, 4 * g + r);     // rstd of row 4g+r lives in lane m = 4g+r
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int kr = tile * 16 + 4 * g + r;
              lg[tile][r] = kr < nr ? acc[r] : T2_NEG;
            }
          }
        }
      }

      T2_STAMP(3);                                // theta, fold, pass A
      // =============================== softmax over all rows, per head m ===============================
      float mx = T2_NEG;
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
          const float e = lg[tile][r] > 0.5f * T2_NEG ? __builtin_amdgcn_exp2f(lg[tile][r] - mx) : 0.f;
          lg[tile][r] = e;
          l += e;
        }
      l += __shfl_xor(l, 16);
      l += __shfl_xor(l, 32);
      const float inv = l > 0.f ? 1.0f / l : 0.f;

      if constexpr (TRAIN) {
        if (p.alpha) {
          float* const ap = p.alpha + (size_t)seg * p.alpha_rows * 16 + m;
#pragma unroll
          for (int tile = 0; tile < MAXT; ++tile)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int kr = tile * 16 + 4 * g + r;                 // row among the atoms that are neither j nor i -> its atom
              const int kc = kr + (kr >= ip ? 1 : 0);
              if (kr < nr) ap[(kc + (kc >= j ? 1 : 0)) * 16] = lg[tile][r] * inv;
            }
          if (g == 0) { ap[j * 16] = 0.f; ap[i * 16] = 0.f; }      // the rows this kernel never visits: the source atom and the target
        }
      }
      T2_STAMP(4);                                // softmax
      // =============================== pass B: S^T[c, h] = sum_rows z_v[row, c] * alpha[row, h] ===============================
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
            const int kq = kr < nr ? kr : 0;
            const float* pv = prow + (kq + (kq >= ip ? 1 : 0)) * T2_ROW + 128 + m;     // past-the-end rows carry alpha = 0 below
#pragma unroll
            for (int tq = 0; tq < 8; ++tq) hv[tq][r] = pv[16 * tq];
          }
#pragma unroll
          for (int st = 0; st < 2; ++st)
            {
              const f4 wa = *reinterpret_cast<const f4*>(wf_v + ((st * 2) * 64 + lane) * 4);
              const f4 wb = *reinterpret_cast<const f4*>(wf_v + ((st * 2 + 1) * 64 + lane) * 4);
#pragma unroll
              for (int tq = 0; tq < 4; ++tq) hv[tq] = mfma16(feat[tile][st], wa[tq], hv[tq]);
#pragma unroll
              for (int tq = 0; tq < 4; ++tq) hv[4 + tq] = mfma16(feat[tile][st], wb[tq], hv[4 + tq]);
            }
#pragma unroll
          for (int tq = 0; tq < 8; ++tq) hv[tq] = mfma16(feat[tile][2], wv2[tq], hv[tq]);
          // folded LayerNorm + ReLU per row r over c = (tau in-lane, m across the DPP row)
          f4 q2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int tq = 0; tq < 8; ++tq) q2 += hv[tq] * hv[tq];
          f4 sg, aw;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float var = t2_row16_sum(q2[r]) * (1.f / 128.f) + 1e-5f;
            const float rsq = __builtin_amdgcn_rsqf(var);
            sg[r] = var * rsq;
            aw[r] = lg[tile][r] * rsq;                                      // alpha * rstd of the row
          }
#pragma unroll
          for (int tq = 0; tq < 8; ++tq) {
            const float bt = bv[m * 8 + tq];
#pragma unroll
            for (int r = 0; r < 4; ++r) hv[tq][r] = T2_DEGRADE(fmaxf(fmaf(bt, sg[r], hv[tq][r]), 0.f));
          }
#pragma unroll
          for (int r = 0; r < 4; ++r)            // r outer: 8 independent accumulator chains
#pragma unroll
            for (int tq = 0; tq < 8; ++tq) sT[tq] = mfma16(hv[tq][r], aw[r], sT[tq]);
        }
      }

      T2_STAMP(5);                                // pass B
      if constexpr (TRAIN) {
        float* const sp = p.S + (size_t)seg * 2048 + lane;
#pragma unroll
        for (int tq = 0; tq < 8; ++tq)
#pragma unroll
          for (int r = 0; r < 4; ++r) sp[(tq * 4 + r) * 64] = sT[tq][r] * inv;
        if (g == 0) p.swn[(size_t)seg * 16 + m] = l > 0.f ? 1.f : 0.f;
        continue;
      }
      // =============================== epilogue: out = resid + W2v_h . S[:,h] / l + b2v ===============================
      // W2v streams through L2: 64 float4 per lane, addresses independent of everything computed in the segment
      const size_t ro = (size_t)seg * 128 + 8 * m + 2 * g;
      float part[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int w = 0; w < 64; ++w) {           // (fully unrolled: the compiler keeps as many of the 64 loads in flight as registers allow)
        const int tq = w >> 3, r = (w >> 1) & 3, hf = (w & 1) * 4;
        const f4 wv_ = w2v_g[w * 64 + lz];
        const float sv = sT[tq][r];
        part[hf + 0] = fmaf(wv_[0], sv, part[hf + 0]); part[hf + 1] = fmaf(wv_[1], sv, part[hf + 1]);
        part[hf + 2] = fmaf(wv_[2], sv, part[hf + 2]); part[hf + 3] = fmaf(wv_[3], sv, part[hf + 3]);
      }
#pragma unroll
      for (int dd = 0; dd < 8; ++dd) {
        part[dd] += __shfl_xor(part[dd], 16);
        part[dd] += __shfl_xor(part[dd], 32);
      }
      const float has = l > 0.f ? 1.f : 0.f;
      const int o0 = 8 * m + 2 * g;
      const float p0 = g == 0 ? part[0] : (g == 1 ? part[2] : (g == 2 ? part[4] : part[6]));
      const float p1 = g == 0 ? part[1] : (g == 1 ? part[3] : (g == 2 ? part[5] : part[7]));
      float2 o;
      o.x = rsd.x + p0 * inv + b2v[o0] * has;
      o.y = rsd.y + p1 * inv + b2v[o0 + 1] * has;
      __builtin_nontemporal_store((t2_f2){o.x, o.y}, reinterpret_cast<t2_f2*>(p.out + ro));
      T2_STAMP(6);                                // unfold + store
    }
#undef T2_FETCH
    T2_STAMP(7);                                  // (loop exit)
  }
#ifdef PG_T2_PROF
  if (lane == 0 && p.alpha) {
    unsigned long long* prof = reinterpret_cast<unsigned long long*>(p.alpha);
    for (int k = 0; k < 8; ++k) atomicAdd(prof + k, t2_acc[k]);
  }
#endif
}

template <int THREADS, int MAXT, bool TRAIN = false>
static int launch_t2(const PgTopo* t, const PgSegAttn* p, hipStream_t st) {
  const size_t lds = t2_lds_floats() * sizeof(float);
  if (int rc = reserve_lds(reinterpret_cast<const void*>(triplet2_kernel<THREADS, MAXT, TRAIN>), lds, "pg_seg_attn(triplet, staged)")) return rc;
  const int grid = (p->tri_grid > 0 && p->tri_grid < kNumCU) ? p->tri_grid : kNumCU;      // persistent workgroups pulling from the queue
  hipLaunchKernelGGL((triplet2_kernel<THREADS, MAXT, TRAIN>), dim3(grid), dim3(THREADS), lds, st, *t, *p);
  return check_launch("pg_seg_attn(triplet, staged)");
}

int g_t2_waves = 12;  // waves per workgroup: 12 (3 per SIMD, no look-ahead prefetch; measured 1.98 ms) or 8 (with it, 2.07 ms);
                      // pg_debug_force_generic_seg bit 2 selects 8

// usable when the caller provides the source-atom groups (PgSegAttn.tri_iters) and P is one [n_bond, 256] = [P_k | P_v] tensor,
// for the sampling form (out = resid + update) and for the training form (S, swn and optionally alpha out; q and W2k_l given);
// returns -1 otherwise
int launch_triplet_staged(const PgTopo* t, const PgSegAttn* p, hipStream_t st) {
  if (!p->tri_iters || !p->tri_counter || p->n_tri_iters <= 0) return -1;
  const bool train = p->S != nullptr;
#ifdef PG_T2_PROF
  if (train || !p->out || !p->resid) return -1;
#else
  if (train ? (!p->swn || !p->q || !p->W2k_l || (p->alpha && p->alpha_rows < t->max_nlig)) : (p->alpha || !p->out || !p->resid)) return -1;
#endif
  if (p->Csrc_v != p->Csrc_k + 128 || p->ld_csrc != 256 || ((size_t)p->Csrc_k & 15) || !p->Cdst_k || !p->Cdst_v) return -1;
  if (t->max_nlig - 1 > T2_ROWS || t->max_nlig > T2_XS) return -1;
  const int maxn = (p->tri_max_nlig > 0 && p->tri_max_nlig < t->max_nlig) ? p->tri_max_nlig : t->max_nlig;     // largest ligand among this queue's entries
  const int tiles = (maxn - 2 + 15) / 16;          // rows a segment visits: every atom but its source and its target
  if (train) {
    if (tiles <= 3) return launch_t2<768, 3, true>(t, p, st);
    if (tiles == 4) return launch_t2<768, 4, true>(t, p, st);
    return launch_t2<768, 5, true>(t, p, st);
  }
  if (g_t2_waves == 12) {
    if (tiles <= 3) return launch_t2<768, 3>(t, p, st);
    if (tiles == 4) return launch_t2<768, 4>(t, p, st);
    return launch_t2<768, 5>(t, p, st);
  }
  if (tiles <= 3) return launch_t2<512, 3>(t, p, st);
  if (tiles == 4) return launch_t2<512, 4>(t, p, st);
  return launch_t2<512, 5>(t, p, st);
}

}  // namespace pg
