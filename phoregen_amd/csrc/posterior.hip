// Reverse-diffusion transition step: categorical posterior + Gumbel-argmax, Gaussian posterior, guidance.
#include "common.h"
#include "../../include/phoregen_hip.h"

namespace pg {

// ---- Philox4x32-10 counter-based generator (Salmon et al., SC'11); own implementation ----
struct Philox {
  uint32_t c[4], k[2];
  __device__ Philox(uint64_t seed, uint64_t idx, uint32_t step, uint32_t stream_id) {
    k[0] = (uint32_t)seed; k[1] = (uint32_t)(seed >> 32);
    c[0] = (uint32_t)idx; c[1] = (uint32_t)(idx >> 32); c[2] = step; c[3] = stream_id;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
      const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k[0], n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k[1];
      c[1] = (uint32_t)p1; c[3] = (uint32_t)p0; c[0] = n0; c[2] = n2;
      k[0] += 0x9E3779B9u; k[1] += 0xBB67AE85u;
    }
  }
  __device__ float uniform(int i) const { return (float)(c[i] >> 8) * (1.0f / 16777216.0f); }  // [0,1)
};

// raw generator words for known-answer tests: in [n][6] = counter c0..c3, key k0 k1 -> out [n][4]
__global__ void philox_selftest_kernel(const uint32_t* in, int n, uint32_t* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t* v = in + (size_t)i * 6;
  Philox ph((uint64_t)v[4] | ((uint64_t)v[5] << 32), (uint64_t)v[0] | ((uint64_t)v[1] << 32), v[2], v[3]);
#pragma unroll
  for (int k = 0; k < 4; ++k) out[(size_t)i * 4 + k] = ph.c[k];
}

template <int K>
__global__ void posterior_cat_kernel(const float* logits, const float* log_vt_in, const int* row_graph,
                                     const int64_t* time_step, const float* q_mats, const float* q_onestep_T, int n_rows,
                                     const float* uniform, uint64_t seed, uint32_t stream_id, uint32_t step,
                                     const int* graph_row0, const int* graph_key,
                                     float* log_vt_out, float* onehot_out, float* traj_out) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n_rows) return;
  const int gr = row_graph[row];
  const int tb = (int)time_step[gr];
  // counter of element (row, k): flat index of the batch, or -- graph-keyed form -- (graph key, index inside the graph), which
  // makes a graph's noise independent of the batch / shard it is sampled in
  const uint64_t e_hi = graph_row0 ? (uint64_t)(uint32_t)(graph_key ? graph_key[gr] : gr) << 32 : 0;
  const size_t e0 = graph_row0 ? (size_t)(row - graph_row0[gr]) * K : (size_t)row * K;
  const int tm1 = tb > 0 ? tb - 1 : 0;
  // log_softmax (diffusion.py:453,462)
  float x[K], mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < K; ++k) { x[k] = logits[(size_t)row * K + k]; mx = fmaxf(mx, x[k]); }
  float se = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) se += expf(x[k] - mx);
  const float lse = logf(se);
  float lv0[K], pv0[K], pvt[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    lv0[k] = (x[k] - mx) - lse;
    pv0[k] = expf(lv0[k]);
    pvt[k] = expf(log_vt_in[(size_t)row * K + k]);
  }
  // q_v_posterior, v0_prob=True (transition.py:285-315)
  const float* QT = q_onestep_T + (size_t)tb * K * K;
  const float* QB = q_mats + (size_t)tm1 * K * K;
  float out[K], omax = -INFINITY;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    float f1 = 0.f, f2 = 0.f;
#pragma unroll
    for (int j = 0; j < K; ++j) { f1 += pvt[j] * QT[j * K + k]; f2 += pv0[j] * QB[j * K + k]; }
    out[k] = fmaxf(logf(f1 + 1e-30f), -32.f) + fmaxf(logf(f2 + 1e-30f), -32.f);
    omax = fmaxf(omax, out[k]);
  }
  float so = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) so += expf(out[k] - omax);
  const float lso = omax + logf(so);
  // Gumbel-argmax (common.py:425-431); first maximum wins
  int best = 0;
  float bestv = -INFINITY;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const float o = tb == 0 ? lv0[k] : out[k] - lso;
    log_vt_out[(size_t)row * K + k] = o;
    float u;
    if (uniform) u = uniform[(size_t)row * K + k];
    else {
      const size_t e = e0 + k;
      Philox ph(seed, e_hi | (e >> 2), step, stream_id);
      u = ph.uniform((int)(e & 3));
    }
    const float gn = -logf(-logf(u + 1e-30f) + 1e-30f);
    const float v = gn + o;
    if (v > bestv) { bestv = v; best = k; }
  }
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const float oh = k == best ? 1.f : 0.f;
    onehot_out[(size_t)row * K + k] = oh;
    if (traj_out) traj_out[(size_t)row * K + k] = oh;
  }
}

__global__ void posterior_pos_kernel(const float* x_t, const float* x0, const int* row_graph, const int64_t* time_step,
                                     const float* coef_x0, const float* coef_xt, const float* std_, const float* grad,
                                     const float* eps, uint64_t seed, uint32_t stream_id, uint32_t step, int n_rows,
                                     const int* graph_row0, const int* graph_key,
                                     const float* center, float* x_prev, float* traj_out,
                                     const int* lig2ctx, float* x_ctx_next, float* x0_out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_rows * 3) return;
  const int row = idx / 3;
  const int gr = row_graph[row];
  const int tb = (int)time_step[gr];
  // lig2ctx (pg_posterior_position_ctx): x0 is the denoiser's ctx-ordered coordinate buffer, read through the row map -- and the new
  // position goes back into the ctx-ordered buffer of the NEXT step as well (x_ctx_next may be the buffer x0 is read from: every thread
  // reads its own element before it writes it)
  const int cidx = lig2ctx ? lig2ctx[row] * 3 + (idx - row * 3) : idx;
  const float x0v = x0[cidx];
  if (x0_out) x0_out[idx] = x0v;
  // transition.py:57-62
  float mu = coef_x0[tb] * x0v + coef_xt[tb] * x_t[idx];
  if (grad) mu -= grad[idx];
  float e;
  if (eps) e = eps[idx];
  else {
    const uint64_t ctr = graph_row0 ? ((uint64_t)(uint32_t)(graph_key ? graph_key[gr] : gr) << 32) | (uint32_t)(idx - 3 * graph_row0[gr])
                                    : (uint64_t)idx;
    Philox ph(seed, ctr, step, stream_id);
    const float u1 = 1.0f - ph.uniform(0), u2 = ph.uniform(1);     // u1 in (0,1]
    e = sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
  }
  const float v = tb == 0 ? mu : mu + std_[tb] * e;
  x_prev[idx] = v;
  if (x_ctx_next) x_ctx_next[cidx] = v;
  if (traj_out) traj_out[idx] = v + (center ? center[gr * 3 + idx % 3] : 0.f);
}

// ---- guidance: closed-form gradient of the two energies (sample_utils.py:135-165) ----
__global__ void guidance_stats_kernel(PgTopo t, const float* x_lig, const float* h_edge_prev, const int* g_lig_off,
                                      float* cnt, float* mean) {
  const int g = blockIdx.x, lane = threadIdx.x;
  const int n = t.g_nlig[g], a0 = g_lig_off[g];
  const int* eid = t.eid + t.g_eid_off[g];
  float c = 0.f, m0 = 0.f, m1 = 0.f, m2 = 0.f;
  for (int i = lane; i < n * n; i += 64) {
    const int e = eid[i];
    if (e >= 0) {
      const float* h = h_edge_prev + (size_t)(t.edge_ref ? t.edge_ref[e] : e) * 6;
      bool none = true;                      // argmax > 0  <=>  some class k>0 strictly beats class 0 (first max wins)
      for (int k = 1; k < 6; ++k) none = none && !(h[k] > h[0]);
      c += none ? 0.f : 1.f;
    }
  }
  for (int a = lane; a < n; a += 64) { m0 += x_lig[(a0 + a) * 3]; m1 += x_lig[(a0 + a) * 3 + 1]; m2 += x_lig[(a0 + a) * 3 + 2]; }
  c = wave_sum(c); m0 = wave_sum(m0); m1 = wave_sum(m1); m2 = wave_sum(m2);
  if (lane == 0) {
    cnt[g] = c;
    mean[g * 3] = m0 / (float)n; mean[g * 3 + 1] = m1 / (float)n; mean[g * 3 + 2] = m2 / (float)n;
  }
}

__global__ void guidance_grad_kernel(PgTopo t, const float* x_lig, const float* h_edge_prev, const int* lig_graph,
                                     const int* g_lig_off, int use_atom, float min_d, float max_d, int use_center,
                                     const float* pc, int mean_over, const float* cnt, const float* mean, float* grad) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= t.n_lig) return;
  const int g = lig_graph[a], n = t.g_nlig[g], a0 = g_lig_off[g], la = a - a0;
  const float B = (float)mean_over;     // the energies are means over the graphs of the (logical) batch
  float gx = 0.f, gy = 0.f, gz = 0.f;
  const float xa = x_lig[a * 3], ya = x_lig[a * 3 + 1], za = x_lig[a * 3 + 2];
  if (use_atom && cnt[g] > 0.f) {
    const int* eid = t.eid + t.g_eid_off[g];
    const float w = 1.f / (cnt[g] * B);
    for (int b = 0; b < n; ++b) {
      if (b == la) continue;
      float mult = 0.f;
      const int e2[2] = {eid[la * n + b], eid[b * n + la]};
      for (int q = 0; q < 2; ++q) {
        const float* h = h_edge_prev + (size_t)(t.edge_ref ? t.edge_ref[e2[q]] : e2[q]) * 6;
        bool none = true;
        for (int k = 1; k < 6; ++k) none = none && !(h[k] > h[0]);
        mult += none ? 0.f : 1.f;
      }
      if (mult == 0.f) continue;
      const float dx = xa - x_lig[(a0 + b) * 3], dy = ya - x_lig[(a0 + b) * 3 + 1], dz = za - x_lig[(a0 + b) * 3 + 2];
      const float ln = sqrtf(dx * dx + dy * dy + dz * dz);
      const float s = (ln > max_d ? 1.f : 0.f) - (ln < min_d ? 1.f : 0.f);
      const float c = mult * s * w / ln;
      gx += c * dx; gy += c * dy; gz += c * dz;
    }
  }
  if (use_center) {
    const float dx = mean[g * 3] - pc[g * 3], dy = mean[g * 3 + 1] - pc[g * 3 + 1], dz = mean[g * 3 + 2] - pc[g * 3 + 2];
    const float nr = sqrtf(dx * dx + dy * dy + dz * dz), w = 1.f / ((float)n * B * nr);
    gx += dx * w; gy += dy * w; gz += dz * w;
  }
  grad[a * 3] = gx; grad[a * 3 + 1] = gy; grad[a * 3 + 2] = gz;
}

}  // namespace pg

using namespace pg;

extern "C" int pg_selftest_philox(const uint32_t* ctr_key, int n, uint32_t* out, void* stream) {
  if (n <= 0) return PG_OK;
  hipLaunchKernelGGL(pg::philox_selftest_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, ctr_key, n, out);
  return pg::check_launch("pg_selftest_philox");
}

extern "C" int pg_posterior_categorical(const float* logits, const float* log_vt_in, const int* row_graph,
                                        const int64_t* time_step, const float* q_mats, const float* q_onestep_T,
                                        int n_rows, int K, const float* uniform, uint64_t seed, uint32_t stream_id,
                                        uint32_t step, const int* graph_row0, const int* graph_key, float* log_vt_out,
                                        float* onehot_out, float* traj_out, void* stream) {
  if (n_rows == 0) return PG_OK;
  dim3 grid((n_rows + 255) / 256), block(256);
  if (K == 12)
    hipLaunchKernelGGL(posterior_cat_kernel<12>, grid, block, 0, (hipStream_t)stream, logits, log_vt_in, row_graph,
                       time_step, q_mats, q_onestep_T, n_rows, uniform, seed, stream_id, step, graph_row0, graph_key, log_vt_out, onehot_out,
                       traj_out);
  else if (K == 6)
    hipLaunchKernelGGL(posterior_cat_kernel<6>, grid, block, 0, (hipStream_t)stream, logits, log_vt_in, row_graph,
                       time_step, q_mats, q_onestep_T, n_rows, uniform, seed, stream_id, step, graph_row0, graph_key, log_vt_out, onehot_out,
                       traj_out);
  else { set_error("pg_posterior_categorical: K must be 12 or 6 (got %d)", K); return PG_ERR_ARG; }
  return check_launch("pg_posterior_categorical");
}

extern "C" int pg_posterior_position(const float* x_t, const float* x0, const int* row_graph, const int64_t* time_step,
                                     const float* coef_x0, const float* coef_xt, const float* std_, const float* energy_grad,
                                     const float* eps, uint64_t seed, uint32_t stream_id, uint32_t step, int n_rows,
                                     const int* graph_row0, const int* graph_key, const float* center, float* x_prev,
                                     float* traj_out, void* stream) {
  if (n_rows == 0) return PG_OK;
  hipLaunchKernelGGL(posterior_pos_kernel, dim3((n_rows * 3 + 255) / 256), dim3(256), 0, (hipStream_t)stream, x_t, x0,
                     row_graph, time_step, coef_x0, coef_xt, std_, energy_grad, eps, seed, stream_id, step, n_rows, graph_row0,
                     graph_key, center, x_prev, traj_out, (const int*)nullptr, (float*)nullptr, (float*)nullptr);
  return check_launch("pg_posterior_position");
}

extern "C" int pg_posterior_position_ctx(const float* x_t, const float* x0_ctx, const int* lig2ctx, const int* row_graph,
                                         const int64_t* time_step, const float* coef_x0, const float* coef_xt, const float* std_,
                                         const float* energy_grad, const float* eps, uint64_t seed, uint32_t stream_id, uint32_t step,
                                         int n_rows, const int* graph_row0, const int* graph_key, const float* center, float* x_prev,
                                         float* traj_out, float* x_ctx_next, float* x0_out, void* stream) {
  if (n_rows == 0) return PG_OK;
  if (!lig2ctx || !x0_ctx) { set_error("pg_posterior_position_ctx: x0_ctx and lig2ctx are required"); return PG_ERR_ARG; }
  hipLaunchKernelGGL(posterior_pos_kernel, dim3((n_rows * 3 + 255) / 256), dim3(256), 0, (hipStream_t)stream, x_t, x0_ctx,
                     row_graph, time_step, coef_x0, coef_xt, std_, energy_grad, eps, seed, stream_id, step, n_rows, graph_row0,
                     graph_key, center, x_prev, traj_out, lig2ctx, x_ctx_next, x0_out);
  return check_launch("pg_posterior_position_ctx");
}

extern "C" int pg_guidance_grad(const PgTopo* t, const float* x_lig, const float* h_edge_prev, const int* lig_graph,
                                const int* g_lig_off, int use_atom_prox, float min_d, float max_d, int use_center_prox,
                                const float* phore_center, int mean_over_graphs, float* cnt_ws, float* mean_ws, float* grad,
                                void* stream) {
  if (t->n_lig == 0) return PG_OK;
  if (mean_over_graphs <= 0) mean_over_graphs = t->n_graphs;
  hipLaunchKernelGGL(guidance_stats_kernel, dim3(t->n_graphs), dim3(64), 0, (hipStream_t)stream, *t, x_lig, h_edge_prev,
                     g_lig_off, cnt_ws, mean_ws);
  hipLaunchKernelGGL(guidance_grad_kernel, dim3((t->n_lig + 255) / 256), dim3(256), 0, (hipStream_t)stream, *t, x_lig,
                     h_edge_prev, lig_graph, g_lig_off, use_atom_prox, min_d, max_d, use_center_prox, phore_center,
                     mean_over_graphs, cnt_ws, mean_ws, grad);
  return check_launch("pg_guidance_grad");
}
