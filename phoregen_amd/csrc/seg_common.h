// Definitions shared by the generic segment-attention kernels (forward: seg_attn.hip, backward: seg_attn_bwd.hip).
#pragma once
#include "common.h"
#include "../../include/phoregen_hip.h"

namespace pg {

constexpr float NEG_BIG = -1.0e30f;

template <int MODE> struct ModeTraits;
template <> struct ModeTraits<PG_SEG_KNN_NODE>  { static constexpr int NSTEP = 12; static constexpr bool POS = false, KNN = true,  BOND = false, TRI = false, PH = false; };
template <> struct ModeTraits<PG_SEG_KNN_POS>   { static constexpr int NSTEP = 12; static constexpr bool POS = true,  KNN = true,  BOND = false, TRI = false, PH = false; };
template <> struct ModeTraits<PG_SEG_BOND_NODE> { static constexpr int NSTEP = 0;  static constexpr bool POS = false, KNN = false, BOND = true,  TRI = false, PH = false; };
template <> struct ModeTraits<PG_SEG_BOND_POS>  { static constexpr int NSTEP = 0;  static constexpr bool POS = true,  KNN = false, BOND = true,  TRI = false, PH = false; };
template <> struct ModeTraits<PG_SEG_TRIPLET>   { static constexpr int NSTEP = 3;  static constexpr bool POS = false, KNN = false, BOND = false, TRI = true,  PH = false; };
template <> struct ModeTraits<PG_SEG_PHORE>     { static constexpr int NSTEP = 1;  static constexpr bool POS = false, KNN = false, BOND = false, TRI = false, PH = true;  };

// angular features of the triplet update (models/common.py:67-87 with duplicated sin/cos(theta) columns merged)
__device__ __constant__ const float kAngFreq[12] = {0.f, 1.f, 2.f, 3.f, 0.5f, (float)(1.0 / 3.0), 1.f, 2.f, 3.f, 0.5f,
                                                    (float)(1.0 / 3.0), 0.f};

struct RowInfo {
  bool valid;
  int csrc;  // row of Csrc_{k,v}
  int src;   // ctx node the row comes from (geometry)
};

template <int MODE>
struct Seg {
  int seg;          // ctx node (node modes) / bond edge j->i (triplet)
  int n_rows;
  int lig0, n, li, lj;  // bond / triplet
  int first;            // phore
  int ci, cj;           // triplet: ctx ids of i, j
  const int* eid_g;
};

template <int MODE>
__device__ __forceinline__ RowInfo row_info(const PgTopo& t, const PgSegAttn& p, const Seg<MODE>& s, int k) {
  using T = ModeTraits<MODE>;
  RowInfo r;
  r.valid = k < s.n_rows;
  r.csrc = 0;
  r.src = 0;
  if (!r.valid) return r;
  if constexpr (T::KNN) {
    r.src = p.nbr[(size_t)s.seg * p.knn_k + k];
    r.csrc = r.src;
  } else if constexpr (T::BOND) {
    r.src = s.lig0 + k;
    r.valid = k != s.li;
    r.csrc = r.valid ? s.eid_g[k * s.n + s.li] : 0;
  } else if constexpr (T::TRI) {
    r.src = s.lig0 + k;
    r.valid = (k != s.li) && (k != s.lj);
    r.csrc = r.valid ? s.eid_g[k * s.n + s.lj] : 0;
  } else {
    r.src = s.first + k;
    r.csrc = r.src;
  }
  return r;
}


// segment descriptor of the si-th entry of the launch's segment list
template <int MODE>
__device__ __forceinline__ Seg<MODE> setup_seg(const PgTopo& t, const PgSegAttn& p, int si) {
  using T = ModeTraits<MODE>;
  Seg<MODE> s;
  s.seg = p.seg_ids ? p.seg_ids[si] : si;
  s.lig0 = s.n = s.li = s.lj = s.first = s.ci = s.cj = 0;
  s.eid_g = nullptr;
  if constexpr (T::KNN) {
    s.n_rows = p.deg[s.seg];
  } else if constexpr (T::BOND) {
    const int gi = t.ctx_graph[s.seg];
    s.n = t.g_nlig[gi];
    s.lig0 = t.g_ctx_off[gi] + t.g_nph[gi];
    s.li = s.seg - s.lig0;
    s.eid_g = t.eid + t.g_eid_off[gi];
    s.n_rows = s.n;
  } else if constexpr (T::TRI) {
    s.cj = t.bond_src[s.seg];
    s.ci = t.bond_dst[s.seg];
    const int gi = t.ctx_graph[s.cj];
    s.n = t.g_nlig[gi];
    s.lig0 = t.g_ctx_off[gi] + t.g_nph[gi];
    s.li = s.ci - s.lig0;
    s.lj = s.cj - s.lig0;
    s.eid_g = t.eid + t.g_eid_off[gi];
    s.n_rows = s.n;
  } else {
    const int gi = t.ctx_graph[s.seg];
    s.first = t.g_ctx_off[gi];
    s.n_rows = t.g_nph[gi];
  }
  return s;
}

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// sum over the 16 lanes of a DPP row (lanes with equal lane>>4); every lane ends up with the total
__device__ __forceinline__ float row16_total(float v) {
  v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);   // row_half_mirror
  v += dpp_mov<0x140>(v);   // row_mirror
  return v;
}

// LDS written by some lanes of a wave and read by others: order the accesses without a workgroup barrier
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

}  // namespace pg
