// Fused dense layers over the bond rows: every first-layer product that reads the same h_bond tile in ONE persistent launch.
//
// A denoiser layer multiplies the old bond features h_bond [E,128] (E ~ 2e5 rows) by four different weight blocks before its
// attention sub-layers can run (models/common.py:99-119 first Linear of the bond-node k/v MLPs, of the triplet k/v MLPs and of
// the triplet query MLP, call sites models/uni_denoiser.py:43-59,141-155), plus the triplet's per-segment constant from the
// bond-length smearing.  As separate GEMM launches each one re-reads h_bond, the query hidden layer round-trips HBM before its
// LayerNorm, and -- the larger loss -- every launch is only ~1.5 waves of workgroups whose load / MFMA / store phases run in
// lockstep and add up instead of overlapping (62 TF/s).  Here:
//   * a workgroup (4 waves) owns a 64-row tile: h_bond | G rows are staged ONCE in LDS as A[64][148(+1)]; two workgroups share
//     a CU (78 KB of LDS each) and run out of phase, so one's loads / stores meet the other's MFMAs;
//   * the jobs' weight matrices stream through a 64-column slab in LDS (the next slab's global loads are issued before the
//     current slab's MFMAs and wait in registers, the gathered node-row adds of the epilogue likewise);
//   * each wave holds one 32x32 accumulator per slab (v_mfma_f32_32x32x2_f32, exact fp32) and stores it straight from the
//     registers (two 128-byte row segments per store instruction) after adding bias / gathered rows;
//   * the query job keeps both slabs of its hidden layer in registers, writes ReLU(LayerNorm(.)) over the (by then unneeded)
//     h_bond columns of A and runs its second Linear from there: no HBM round trip.
// Persistent grid (two workgroups per CU), tiles dealt round-robin: after the first tile the workgroups drift apart, so the
// chip sees loads, MFMAs and stores of different tiles at the same time.
#include "common.h"
#include "../../include/phoregen_hip.h"

namespace pg {

constexpr int BR_BM = 64, BR_BN = 64, BR_THREADS = 256;      // 78 KB of LDS per workgroup: two independent workgroups per CU
constexpr int BR_NP = 10;                                    // float4 pieces of a weight slab per thread (64 x 37 / 256)
constexpr int BR_LDA = 149;                    // 148 columns (h_bond 128 | G 20) + 1: odd stride, conflict-free ds_read_b32
constexpr int BR_LDW = 149;
constexpr size_t BR_JOB_FLOATS = (sizeof(PgBondJob) * PG_BOND_MAX_JOBS + 3) / 4;
constexpr size_t BR_LDS_FLOATS = (size_t)BR_BM * BR_LDA + BR_BN * BR_LDW + 2 * BR_BM /*idx*/ + BR_JOB_FLOATS;

// slab = rows col0..col0+63 of W ([N, K] rows, K in {128, 148, 20}: 32 / 37 / 5 float4 per row), fetched as float4 pieces
// (<= 5 per thread) into registers; handed to LDS (odd row stride: scalar writes) after the current slab's MFMAs
__device__ __forceinline__ void br_piece(int e4, int K4, int& c, int& k4) {
  if (K4 == 32) { c = e4 >> 5; k4 = e4 & 31; }
  else if (K4 == 37) { c = e4 / 37; k4 = e4 - c * 37; }
  else if (K4 == 5) { c = e4 / 5; k4 = e4 - c * 5; }
  else { c = e4 / K4; k4 = e4 - c * K4; }
}

__device__ __forceinline__ void br_fetch_w(const float* W, int ldw, int K, int col0, int N, int tid, f4 (&stage)[BR_NP]) {
  const int K4 = K >> 2, total = BR_BN * K4;
#pragma unroll
  for (int i = 0; i < BR_NP; ++i) {
    int e4 = tid + i * BR_THREADS;
    e4 = e4 < total ? e4 : total - 1;                       // surplus pieces re-read the last one (not stored)
    int c, k4;
    br_piece(e4, K4, c, k4);
    c = col0 + c < N ? col0 + c : N - 1;
    stage[i] = *reinterpret_cast<const f4*>(W + (size_t)c * ldw + 4 * k4);
  }
}

__device__ __forceinline__ void br_store_w(float* wbuf, int K, int tid, const f4 (&stage)[BR_NP]) {
  const int K4 = K >> 2, total = BR_BN * K4;
#pragma unroll
  for (int i = 0; i < BR_NP; ++i) {
    const int e4 = tid + i * BR_THREADS;
    if (e4 < total) {
      int c, k4;
      br_piece(e4, K4, c, k4);
      float* d = wbuf + c * BR_LDW + 4 * k4;
      d[0] = stage[i][0]; d[1] = stage[i][1]; d[2] = stage[i][2]; d[3] = stage[i][3];
    }
  }
}

__global__ __launch_bounds__(BR_THREADS, 2) void bond_rows_kernel(PgBondRows p PG_ABL_PARAM) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const As = lds;                                   // [128][149]
  float* const Wb = As + BR_BM * BR_LDA;                   // [64][149]: one weight slab (the next one waits in registers)
  int* const idxs = reinterpret_cast<int*>(Wb + BR_BN * BR_LDW);   // [2][128] gathered-row indices of the tile (src, dst)
  // the job list lives in LDS: indexing the by-value kernel argument with a run-time job number would copy it to scratch
  PgBondJob* const jobs = reinterpret_cast<PgBondJob*>(idxs + 2 * BR_BM);
  for (int i = threadIdx.x; i < (int)(sizeof(PgBondJob) * PG_BOND_MAX_JOBS / 4); i += BR_THREADS)
    reinterpret_cast<int*>(jobs)[i] = reinterpret_cast<const int*>(p.jobs)[i];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;                 // wave tile: rows 32 wr.. (2 row blocks), slab columns 32 wc..
  const int l31 = lane & 31, lh = lane >> 5;
  const int n_tiles = (p.E + BR_BM - 1) / BR_BM;

  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int row0 = tile * BR_BM;
    __syncthreads();                                       // previous tile's A / W buffers are no longer read
    // ---- stage A = [h_bond | G] rows and the tile's gather indices: every load is issued before the first LDS write ----
    {
      f4 av[8];
      float gv[5];
#pragma unroll
      for (int i = 0; i < 8; ++i) {                         // 64 rows x 32 float4 of h_bond over 256 threads
        const int e = tid + i * BR_THREADS, r = e >> 5, c4 = (e & 31) * 4;
        const int rc = row0 + r < p.E ? row0 + r : p.E - 1;            // rows past the end re-read the last row (never stored)
        av[i] = *reinterpret_cast<const f4*>(p.hb + (size_t)rc * p.ld_hb + c4);
      }
      if (p.G) {
#pragma unroll
        for (int i = 0; i < 5; ++i) {                       // 64 x 20 floats
          const int e = tid + i * BR_THREADS, r = e / 20, k = e - r * 20;
          const int rc = row0 + r < p.E ? row0 + r : p.E - 1;
          gv[i] = p.G[(size_t)rc * 20 + k];
        }
      }
      int iv = 0;
      if (tid < 2 * BR_BM) {
        const int* src = (tid >= BR_BM) ? p.idx_b : p.idx_a;
        const int r = tid & (BR_BM - 1);
        if (src) iv = src[row0 + r < p.E ? row0 + r : p.E - 1];
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int e = tid + i * BR_THREADS, r = e >> 5, c4 = (e & 31) * 4;
        float* d = As + r * BR_LDA + c4;
        d[0] = av[i][0]; d[1] = av[i][1]; d[2] = av[i][2]; d[3] = av[i][3];
      }
      if (p.G) {
#pragma unroll
        for (int i = 0; i < 5; ++i) {
          const int e = tid + i * BR_THREADS, r = e / 20, k = e - r * 20;
          As[r * BR_LDA + 128 + k] = gv[i];
        }
      }
      if (tid < 2 * BR_BM) idxs[tid] = iv;
    }

    f4 stage[BR_NP];
    // ---- slab sequence of the tile: all first-layer slabs of all jobs, then the second-layer slabs of a query job ----
    // (the job list is tiny and uniform: walk it with scalar state)
    int job = 0, col0 = 0, phase = 0;
    float* const wcur = Wb;
    {
      const PgBondJob& jb = jobs[0];
      br_fetch_w(jb.W, jb.ldw, jb.K, 0, jb.N, tid, stage);
      br_store_w(wcur, jb.K, tid, stage);
    }
    __syncthreads();
    f16v held[2];                                          // query job: the two first-layer slabs of the wave's rows

    while (job < p.n_jobs) {
      const PgBondJob& jb = jobs[job];
      const int K = phase ? 128 : jb.K, ka = phase ? 0 : jb.k0;      // contraction range inside A's columns
      const int Nout = phase ? jb.N2 : jb.N;
      // ---- next slab (job', col0', phase') ----
      int njob = job, ncol = col0 + BR_BN, nphase = phase;
      if (ncol >= Nout) {
        ncol = 0;
        if (!phase && jb.W2) nphase = 1; else { njob = job + 1; nphase = 0; }
      }
      const bool has_next = njob < p.n_jobs;
      if (has_next && !PG_ABL(8)) {
        const PgBondJob& nj = jobs[njob];
        if (nphase) br_fetch_w(nj.W2, 128, 128, ncol, nj.N2, tid, stage);
        else br_fetch_w(nj.W, nj.ldw, nj.K, ncol, nj.N, tid, stage);
      }
      // ---- epilogue operands of THIS slab, fetched before the MFMAs: bias / gathered node rows for the wave's 32x32 block ----
      // The loaded values are NOT touched before the MFMA loop: waves issue in order, so a single add on a gathered value in
      // front of the loop would park the wave until the gather has landed; left alone, the loads fly during the MFMAs.
      const int gcol = col0 + 32 * wc + l31;
      f16v ld1, ld2;
      float bsv = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) { ld1[r] = 0.f; ld2[r] = 0.f; }
      if (!phase && !PG_ABL(1) && gcol < Nout) {
        if (jb.bias) bsv = jb.bias[gcol];
        if (jb.add1) {
          const float* a1 = jb.add1 + gcol;
          const int* ix = idxs + (jb.idx1_is_b ? BR_BM : 0) + 32 * wr + 4 * lh;
          const int ld = jb.ld_add1;
#pragma unroll
          for (int r = 0; r < 16; ++r) ld1[r] = a1[(size_t)ix[(r & 3) + 8 * (r >> 2)] * ld];
        }
        if (jb.add2) {
          const float* a2 = jb.add2 + gcol;
          const int* ix = idxs + (jb.idx2_is_b ? BR_BM : 0) + 32 * wr + 4 * lh;
          const int ld = jb.ld_add2;
#pragma unroll
          for (int r = 0; r < 16; ++r) ld2[r] = a2[(size_t)ix[(r & 3) + 8 * (r >> 2)] * ld];
        }
      }
      // ---- MFMAs: acc[32 rows x 32 cols] += A[rows][ka + k] * W[cols][k] ----
      f16v acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      {
        const float* ap = As + (32 * wr + l31) * BR_LDA + ka + lh;
        const float* bp = wcur + (32 * wc + l31) * BR_LDW + lh;
        const int ksteps = PG_ABL(4) ? 1 : (K >> 1);
#pragma unroll 4
        for (int ks = 0; ks < ksteps; ++ks) acc = mfma32(ap[2 * ks], bp[2 * ks], acc);
      }
      // ---- epilogue ----
      if (!phase && jb.W2) {
        // query job, first layer: keep (hidden + bias + gathered rows) in registers until both slabs are there
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = ((acc[r] + bsv) + ld1[r]) + ld2[r];
        if (col0 == 0) held[0] = acc; else held[1] = acc;
        if (col0 + BR_BN >= Nout) {
          // both slabs done: every wave has finished its MFMAs on A's h_bond columns only after the barrier below
          __syncthreads();
#pragma unroll
          for (int hs = 0; hs < 2; ++hs)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int rr = 32 * wr + (r & 3) + 8 * (r >> 2) + 4 * lh;
              As[rr * BR_LDA + 64 * hs + 32 * wc + l31] = held[hs][r];
            }
          __syncthreads();
          // LayerNorm statistics per row (4 threads per row), then normalise + ReLU in place (models/common.py:99-119)
          {
            const int r = tid >> 2, q4 = tid & 3;
            float s = 0.f;
            for (int c = q4; c < 128; c += 4) s += As[r * BR_LDA + c];
            s += __shfl_xor(s, 1); s += __shfl_xor(s, 2);
            const float mu = s * (1.f / 128.f);
            float q = 0.f;
            for (int c = q4; c < 128; c += 4) { const float dlt = As[r * BR_LDA + c] - mu; q = fmaf(dlt, dlt, q); }
            q += __shfl_xor(q, 1); q += __shfl_xor(q, 2);
            const float rs = 1.0f / sqrtf(q * (1.f / 128.f) + 1e-5f);
            for (int c = q4; c < 128; c += 4) {
              float* a = As + r * BR_LDA + c;
              *a = fmaxf((*a - mu) * rs * jb.ln_g[c] + jb.ln_b[c], 0.f);
            }
          }
        }
      } else {
        float* Y = phase ? jb.Y2 : jb.Y;
        const int ldy = phase ? jb.ldy2 : jb.ldy;
        const float b2 = (phase && jb.b2 && gcol < Nout) ? jb.b2[gcol] : 0.f;
        const float sc = phase ? jb.scale2 : 1.0f;
        if (gcol < Nout && !PG_ABL(2)) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int rr = 32 * wr + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (row0 + rr < p.E) Y[(size_t)(row0 + rr) * ldy + gcol] = phase ? (acc[r] + b2) * sc : ((acc[r] + bsv) + ld1[r]) + ld2[r];
          }
        }
      }
      __syncthreads();                                     // every wave is done reading the slab (and A, if it was rewritten, is settled)
      if (has_next && !PG_ABL(8)) br_store_w(wcur, nphase ? 128 : jobs[njob].K, tid, stage);
      __syncthreads();                                     // next slab is in LDS
      job = njob; col0 = ncol; phase = nphase;
    }
  }
}

}  // namespace pg

extern "C" int pg_bond_rows(const PgBondRows* p, void* stream) {
  using namespace pg;
  if (!p || !p->hb || p->E < 0 || p->n_jobs < 1 || p->n_jobs > PG_BOND_MAX_JOBS) { set_error("pg_bond_rows: bad arguments"); return PG_ERR_ARG; }
  if (p->E == 0) return PG_OK;
  if ((p->ld_hb & 3) || ((size_t)p->hb & 15)) { set_error("pg_bond_rows: h_bond rows must be 16-byte aligned"); return PG_ERR_ARG; }
  for (int j = 0; j < p->n_jobs; ++j) {
    const PgBondJob& jb = p->jobs[j];
    if (!jb.W || jb.N <= 0 || jb.K <= 0 || jb.k0 < 0 || jb.k0 + jb.K > 148 || (jb.K & 3) || (jb.k0 + jb.K > 128 && !p->G) ||
        (jb.ldw & 3) || ((size_t)jb.W & 15)) {
      set_error("pg_bond_rows: job %d: contraction range [%d, %d) must be a multiple of 4 wide, inside [0,148) (G needed beyond 128), "
                "weight rows 16-byte aligned", j, jb.k0, jb.k0 + jb.K);
      return PG_ERR_ARG;
    }
    if (jb.W2) {
      if (jb.N != 128 || jb.k0 != 0 || jb.N2 <= 0 || !jb.ln_g || !jb.ln_b || !jb.Y2 || j != p->n_jobs - 1) {
        set_error("pg_bond_rows: a two-layer (query) job must be the last job, with a 128-wide hidden layer read from h_bond");
        return PG_ERR_ARG;
      }
    } else if (!jb.Y) { set_error("pg_bond_rows: job %d has no output", j); return PG_ERR_ARG; }
    if ((jb.add1 && !(jb.idx1_is_b ? p->idx_b : p->idx_a)) || (jb.add2 && !(jb.idx2_is_b ? p->idx_b : p->idx_a))) {
      set_error("pg_bond_rows: job %d gathers rows but the index array is missing", j);
      return PG_ERR_ARG;
    }
  }
  const size_t lds = BR_LDS_FLOATS * sizeof(float);
  if (int rc = reserve_lds(reinterpret_cast<const void*>(bond_rows_kernel), lds, "pg_bond_rows")) return rc;
  const int n_tiles = (p->E + BR_BM - 1) / BR_BM;
  const int grid = n_tiles < 2 * kNumCU ? n_tiles : 2 * kNumCU;
  hipLaunchKernelGGL(bond_rows_kernel, dim3(grid), dim3(BR_THREADS), lds, (hipStream_t)stream, *p PG_ABL_ARG("PG_BR_ABLATE"));
  return check_launch("pg_bond_rows");
}
