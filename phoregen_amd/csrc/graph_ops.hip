// Graph-side kernels of the denoising step: embeddings, exact in-graph kNN, direction vectors,
// global edge gate, bond-length smearing, coordinate update, atom-count pooling.
#include <stdarg.h>
#include <string.h>

#include <mutex>

#include "common.h"
#include <stdlib.h>
#include "../../include/phoregen_hip.h"

namespace pg {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int reserve_lds(const void* kernel, size_t bytes, const char* what) {
  struct Ent { const void* fn; int dev; size_t bytes; };
  static Ent tab[256];
  static int n_ent = 0;
  static std::mutex mu;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { set_error("%s: hipGetDevice failed", what); return PG_ERR_HIP; }
  std::lock_guard<std::mutex> lock(mu);
  Ent* e = nullptr;
  for (int i = 0; i < n_ent; ++i)
    if (tab[i].fn == kernel && tab[i].dev == dev) { e = &tab[i]; break; }
  if (e && e->bytes >= bytes) return PG_OK;
  hipError_t rc = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (rc != hipSuccess) {
    set_error("%s: cannot reserve %zu B of LDS on device %d: %s", what, bytes, dev, hipGetErrorString(rc));
    return PG_ERR_HIP;
  }
  if (!e) {
    if (n_ent == 256) return PG_OK;          // table full: the attribute is set, it is just not remembered
    e = &tab[n_ent++];
    e->fn = kernel; e->dev = dev;
  }
  e->bytes = bytes;
  return PG_OK;
}

// ------------------------------------------------------------------------------------------------
// embeddings  (models/diffusion.py:180-183,205 ; TimeGaussianSmearing models/common.py:51-55)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float time_smear(float t, const float* off, const float* coeff, int i) {
  t = fminf(fmaxf(t, 0.f), 1000.f);
  const float d = t - off[i];
  return expf(coeff[i] * (d * d));
}

__global__ void embed_ctx_kernel(PgTopo t, const float* h_node, const float* pos, const int64_t* time_step,
                                 const float* W_node, const float* t_off, const float* t_coeff,
                                 const float* h_phore_emb, const float* pos_phore, const int* phore2ctx,
                                 float* h_ctx, float* x_ctx) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int node = idx >> 7, c = idx & 127;
  if (node >= t.n_lig + t.n_phore) return;
  if (node < t.n_lig) {
    const int ctx = t.lig2ctx[node];
    if (h_ctx) {
      float v;
      if (c < 118) {
        v = 0.f;
#pragma unroll
        for (int k = 0; k < 12; ++k) v += h_node[node * 12 + k] * W_node[c * 12 + k];
      } else {
        v = time_smear((float)time_step[t.ctx_graph[ctx]], t_off, t_coeff, c - 118);
      }
      h_ctx[(size_t)ctx * 128 + c] = v;
    }
    if (x_ctx && c < 3) x_ctx[ctx * 3 + c] = pos[node * 3 + c];
  } else {
    const int p = node - t.n_lig, ctx = phore2ctx[p];
    if (h_ctx) h_ctx[(size_t)ctx * 128 + c] = h_phore_emb[(size_t)p * 128 + c];
    if (x_ctx && c < 3) x_ctx[ctx * 3 + c] = pos_phore[p * 3 + c];
  }
}

__global__ void embed_bond_kernel(int n_bond, const int* edge_ref, const float* h_edge, const int* bond_graph,
                                  const int64_t* time_step, const float* W_edge, const float* t_off, const float* t_coeff,
                                  float* h_bond) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t e = idx >> 7;
  const int c = idx & 127;
  if (e >= (size_t)n_bond) return;
  float v;
  if (c < 118) {
    v = 0.f;
    const size_t er = edge_ref ? (size_t)edge_ref[e] : e;        // the caller's row of internal edge e
#pragma unroll
    for (int k = 0; k < 6; ++k) v += h_edge[er * 6 + k] * W_edge[c * 6 + k];
  } else {
    v = time_smear((float)time_step[bond_graph[e]], t_off, t_coeff, c - 118);
  }
  h_bond[e * 128 + c] = v;
}

// ------------------------------------------------------------------------------------------------
// exact kNN inside a graph, one wave per centre (torch_cluster.knn_graph, loop=False)
//   key = (bits(d2) << 32) | candidate  -> ascending distance, index breaks ties
// ------------------------------------------------------------------------------------------------
constexpr int KNN_MAXC = 8;  // candidates per lane -> graphs of up to 512 nodes

__device__ __forceinline__ float dist2_rn(const float* a, const float* b) {
  // (dx^2 + dy^2) + dz^2 with every product / sum rounded separately (no fma contraction), the order
  // a torch (x[:,None]-x[None]).pow(2).sum(-1) uses
  const float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
  return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    unsigned long long w = __shfl_xor(v, o);
    v = w < v ? w : v;
  }
  return v;
}

// first..first+count = candidate ctx range; `self` excluded; writes k slots (-1 beyond deg)
__device__ void wave_knn(const float* x, int first, int count, int self, int k, int* out_slots, int* out_deg) {
  const int lane = threadIdx.x & 63;
  unsigned long long key[KNN_MAXC];
  const float xs[3] = {x[self * 3], x[self * 3 + 1], x[self * 3 + 2]};
#pragma unroll
  for (int i = 0; i < KNN_MAXC; ++i) {
    const int c = lane + 64 * i;
    key[i] = ~0ull;
    if (c < count && first + c != self) {
      const float d2 = dist2_rn(x + (size_t)(first + c) * 3, xs);
      key[i] = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)c;
    }
  }
  const int deg = min(k, count - 1);
  for (int s = 0; s < k; ++s) {
    unsigned long long best = key[0];
#pragma unroll
    for (int i = 1; i < KNN_MAXC; ++i) best = key[i] < best ? key[i] : best;
    best = wave_min_u64(best);
    const int c = (int)(best & 0xffffffffu);
    if (s < deg) {
#pragma unroll
      for (int i = 0; i < KNN_MAXC; ++i)
        if (key[i] == best) key[i] = ~0ull;
    }
    if (lane == 0) out_slots[s] = s < deg ? first + c : -1;
  }
  if (lane == 0 && out_deg) *out_deg = deg;
}

__global__ __launch_bounds__(256) void knn_ctx_kernel(PgTopo t, const float* x, int k, int* nbr, int* deg) {
  const int node = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (node >= t.n_ctx) return;
  const int g = t.ctx_graph[node];
  const int first = t.g_ctx_off[g], count = t.g_ctx_off[g + 1] - first;
  wave_knn(x, first, count, node, k, nbr + (size_t)node * k, deg + node);
}

// stable partition of every node's neighbour list by the source's kind (ligand atoms first), gate values moved along.  The knn
// attention's 40 distance columns are 20 per kind (common.py outer product with the edge type): a row tile whose sources are all of
// one kind skips the other kind's 5 MFMA k-steps (node_attn.hip), and after the partition at most one tile per node is mixed.
// Attention sums over a node's rows, so the order is free (fp32 summation order only).
__global__ __launch_bounds__(256) void knn_group_kernel(PgTopo t, int k, int* nbr, const int* deg, float* ew) {
  const int node = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (node >= t.n_ctx) return;
  const bool valid = lane < deg[node] && lane < k;
  int v = 0;
  float e = 0.f;
  if (valid) { v = nbr[(size_t)node * k + lane]; e = ew[(size_t)node * k + lane]; }
  const bool lig = valid && t.ctx_is_lig[v] != 0;
  const unsigned long long ml = __ballot(lig), mv = __ballot(valid);
  const unsigned long long below = (1ull << lane) - 1ull;
  const int pos = lig ? __popcll(ml & below) : __popcll(ml) + __popcll(mv & ~ml & below);
  if (valid) { nbr[(size_t)node * k + pos] = v; ew[(size_t)node * k + pos] = e; }
}

// direction vectors (models/common.py:300-314)
__global__ __launch_bounds__(256) void lig_normals_kernel(PgTopo t, const float* x, const float* phore_norm,
                                                          const int* phore2ctx, float* nrm) {
  __shared__ int slots[4][4];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int item = blockIdx.x * 4 + w;
  if (item >= t.n_lig + t.n_phore) return;
  if (item >= t.n_lig) {
    const int p = item - t.n_lig;
    if (lane < 3) nrm[phore2ctx[p] * 3 + lane] = phore_norm[p * 3 + lane];
    return;
  }
  const int ctx = t.lig2ctx[item], g = t.ctx_graph[ctx];
  const int first = t.g_ctx_off[g] + t.g_nph[g], count = t.g_nlig[g];
  int d;
  wave_knn(x, first, count, ctx, 3, slots[w], &d);
  d = min(3, count - 1);
  if (lane < 3) {
    float s = 0.f;
    for (int i = 0; i < d; ++i) s += x[slots[w][i] * 3 + lane];       // scatter(mean): sum in edge order / count
    nrm[ctx * 3 + lane] = s / (float)max(d, 1) - x[ctx * 3 + lane];
  }
}

// indices of the (up to) 3 nearest ligand atoms of every ligand atom, in the order lig_normals_kernel sums them
// (training path: the mean itself is composed on the host so that autograd sees it); nn3[a][i] = -1 beyond the count
__global__ __launch_bounds__(256) void lig_nn3_kernel(PgTopo t, const float* x, int* nn3) {
  __shared__ int slots[4][4];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int item = blockIdx.x * 4 + w;
  if (item >= t.n_lig) return;
  const int ctx = t.lig2ctx[item], g = t.ctx_graph[ctx];
  const int first = t.g_ctx_off[g] + t.g_nph[g], count = t.g_nlig[g];
  int d;
  wave_knn(x, first, count, ctx, 3, slots[w], &d);
  d = min(3, count - 1);
  if (lane < 3) nn3[item * 3 + lane] = lane < d ? slots[w][lane] : -1;
}

// ------------------------------------------------------------------------------------------------
// global edge gate  e_w = sigmoid(W3 . ReLU(LN(W0 . smear(d) + b0)) + b3)   (uni_denoiser.py:410-415)
// one wave per node, its k neighbour slots in tiles of 16 rows; hidden^T[c,row] by 5 K-steps of 16x16x4 MFMA
// (A = lane-fixed first-layer weights, B = the 20 Gaussians of the row), LayerNorm folded as in packing._kv_mlp.
// Arguments are in kernel layout (packing.pack_gate): Wl [5][8][64], b1c [128] centred+signed bias, bp = beta/|gamma|,
// w3g = W3 * |gamma|.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void edge_gate_kernel(int n_ctx, const float* x, const int* nbr, const int* deg,
                                                        int k, const float* Wl, const float* b1c, const float* bp,
                                                        const float* w3g, float b3, float* ew) {
  __shared__ __attribute__((aligned(16))) float wl[5 * 512];
  __shared__ __attribute__((aligned(16))) float cst[3 * 128];
  for (int i = threadIdx.x; i < 5 * 512; i += 256) wl[i] = Wl[i];
  for (int i = threadIdx.x; i < 128; i += 256) { cst[i] = b1c[i]; cst[128 + i] = bp[i]; cst[256 + i] = w3g[i]; }
  __syncthreads();
  const int lane = threadIdx.x & 63, g = lane >> 4, m = lane & 15;
  const int node = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (node >= n_ctx) return;
  const int dg = deg[node];
  const float x0 = x[node * 3], x1 = x[node * 3 + 1], x2 = x[node * 3 + 2];
  for (int tile = 0; tile * 16 < k; ++tile) {
    const int row = tile * 16 + m;
    float d = 0.f;
    const bool valid = row < dg;
    if (valid) {
      const int src = nbr[(size_t)node * k + row];
      const float r0 = x0 - x[src * 3], r1 = x1 - x[src * 3 + 1], r2 = x2 - x[src * 3 + 2];
      d = sqrtf(r0 * r0 + r1 * r1 + r2 * r2);
    }
    f4 hid[8];
#pragma unroll
    for (int tq = 0; tq < 8; ++tq) hid[tq] = *reinterpret_cast<const f4*>(cst + 16 * tq + 4 * g);
#pragma unroll
    for (int st = 0; st < 5; ++st) {
      const float f = smear(d, 4 * st + g);
#pragma unroll
      for (int tq = 0; tq < 8; ++tq) hid[tq] = mfma16(wl[(st * 8 + tq) * 64 + lane], f, hid[tq]);
    }
    float q = 0.f;
#pragma unroll
    for (int tq = 0; tq < 8; ++tq)
#pragma unroll
      for (int r = 0; r < 4; ++r) q = fmaf(hid[tq][r], hid[tq][r], q);
    q += __shfl_xor(q, 16);
    q += __shfl_xor(q, 32);
    const float var = q * (1.f / 128.f) + 1e-5f;
    const float rs = 1.0f / sqrtf(var);
    const float sigma = var * rs;
    float dot = 0.f;
#pragma unroll
    for (int tq = 0; tq < 8; ++tq) {
      const f4 bt = *reinterpret_cast<const f4*>(cst + 128 + 16 * tq + 4 * g);
      const f4 w3 = *reinterpret_cast<const f4*>(cst + 256 + 16 * tq + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) dot = fmaf(fmaxf(fmaf(bt[r], sigma, hid[tq][r]), 0.f), w3[r], dot);
    }
    dot += __shfl_xor(dot, 16);
    dot += __shfl_xor(dot, 32);
    const float o = dot * rs + b3;
    if (g == 0 && row < k) ew[(size_t)node * k + row] = valid ? 1.f / (1.f + expf(-o)) : 0.f;
  }
}

__global__ void bond_smear_kernel(int n_bond, const int* bsrc, const int* bdst, const float* x, float* G) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int e = idx / 20, i = idx % 20;
  if (e >= n_bond) return;
  const int s = bsrc[e], d = bdst[e];
  const float dx = x[d * 3] - x[s * 3], dy = x[d * 3 + 1] - x[s * 3 + 1], dz = x[d * 3 + 2] - x[s * 3 + 2];
  G[idx] = smear(sqrtf(dx * dx + dy * dy + dz * dz), i);
}

__global__ void apply_dx_kernel(int n_ctx, const uint8_t* is_lig, const float* x, const float* dx1, const float* dx2,
                                float* x_new) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_ctx * 3) return;
  const float m = is_lig[idx / 3] ? 1.f : 0.f;
  x_new[idx] = x[idx] + (dx1[idx] + dx2[idx]) * m;
}

// ------------------------------------------------------------------------------------------------
// everything of a layer that depends on the coordinates only, in ONE launch: x' = x + mask (dx1 + dx2)  (uni_denoiser.py:295-296, the
// arithmetic of apply_dx_kernel), then from x': the bond-length smearing of the next layer (bond_smear_kernel) and the direction
// vectors (lig_normals_kernel).  Three 5 us launches on three lanes cost two cross-lane event hops (20-30 us each) in front of every
// triplet kernel of a small batch; as one launch they sit on the bond chain's own stream.
// Grid (graph, part): every workgroup of a graph forms the graph's new ligand coordinates in LDS (<= 512 atoms), part 0 writes x'; the
// graph's bond rows and atoms are dealt out over the parts.  dx1 == nullptr: no update (layer 0: x' = x, nothing written).
// G == nullptr / nrm == nullptr: that product is not wanted (last layer: the update alone).
// ------------------------------------------------------------------------------------------------
constexpr int GEOM_THREADS = 512;

__device__ void wave_knn3_lds(const float* xl, int count, int self, int* out_slots) {
  // wave_knn(k = 3) on coordinates staged in LDS (candidate c = ligand atom c of the graph): same keys, same order
  const int lane = threadIdx.x & 63;
  unsigned long long key[KNN_MAXC];
  const float xs[3] = {xl[self * 3], xl[self * 3 + 1], xl[self * 3 + 2]};
#pragma unroll
  for (int i = 0; i < KNN_MAXC; ++i) {
    const int c = lane + 64 * i;
    key[i] = ~0ull;
    if (c < count && c != self) {
      const float d2 = dist2_rn(xl + c * 3, xs);
      key[i] = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)c;
    }
  }
  const int deg = min(3, count - 1);
  for (int s = 0; s < 3; ++s) {
    unsigned long long best = key[0];
#pragma unroll
    for (int i = 1; i < KNN_MAXC; ++i) best = key[i] < best ? key[i] : best;
    best = wave_min_u64(best);
    const int c = (int)(best & 0xffffffffu);
    if (s < deg) {
#pragma unroll
      for (int i = 0; i < KNN_MAXC; ++i)
        if (key[i] == best) key[i] = ~0ull;
    }
    if (lane == 0) out_slots[s] = s < deg ? c : -1;
  }
}

__global__ __launch_bounds__(GEOM_THREADS) void layer_geom_kernel(PgTopo t, const float* x, const float* dx1, const float* dx2,
                                                                  const float* nrm_phore_ctx, float* x_new, float* nrm, float* G) {
  __shared__ float xl[64 * KNN_MAXC * 3];
  __shared__ int slots[GEOM_THREADS / 64][4];
  const int g = blockIdx.x, part = blockIdx.y, n_part = gridDim.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int first = t.g_ctx_off[g], nph = t.g_nph[g], n = t.g_nlig[g];
  const int lig0 = first + nph;
  for (int i = tid; i < (nph + n) * 3; i += GEOM_THREADS) {
    const int idx = first * 3 + i;
    float v = x[idx];
    if (dx1) {
      const float m = t.ctx_is_lig[idx / 3] ? 1.f : 0.f;
      v = v + (dx1[idx] + dx2[idx]) * m;
      if (part == 0 && x_new) x_new[idx] = v;
    }
    if (i >= nph * 3) xl[i - nph * 3] = v;
    else if (nrm && part == 0) nrm[idx] = nrm_phore_ctx[idx];
  }
  __syncthreads();
  if (G) {
    const int e0 = t.g_bond_off[g], n_el = n * (n - 1) * 20;
    for (int idx = part * GEOM_THREADS + tid; idx < n_el; idx += n_part * GEOM_THREADS) {
      const int e = e0 + idx / 20, i = idx % 20;
      const int s = t.bond_src[e] - lig0, d = t.bond_dst[e] - lig0;
      const float ddx = xl[d * 3] - xl[s * 3], ddy = xl[d * 3 + 1] - xl[s * 3 + 1], ddz = xl[d * 3 + 2] - xl[s * 3 + 2];
      G[(size_t)e0 * 20 + idx] = smear(sqrtf(ddx * ddx + ddy * ddy + ddz * ddz), i);
    }
  }
  if (nrm) {
    constexpr int WAVES = GEOM_THREADS / 64;
    for (int a = part * WAVES + wave; a < n; a += n_part * WAVES) {
      wave_knn3_lds(xl, n, a, slots[wave]);
      const int d = min(3, n - 1);
      if (lane < 3) {
        float s = 0.f;
        for (int i = 0; i < d; ++i) s += xl[slots[wave][i] * 3 + lane];       // scatter(mean): sum in edge order / count
        nrm[(lig0 + a) * 3 + lane] = s / (float)max(d, 1) - xl[a * 3 + lane];
      }
    }
  }
}

// per-graph means of the two count heads (models/diffusion.py:148-159); one block per graph
__global__ void atom_count_kernel(const float* s_all, const float* s_l, const uint8_t* is_ex, const int* phore_graph,
                                  int n_phore, float* count_l, float* count_u) {
  const int g = blockIdx.x;
  __shared__ float red[4][64];
  float a = 0.f, na = 0.f, l = 0.f, nl = 0.f;
  for (int p = threadIdx.x; p < n_phore; p += 64)
    if (phore_graph[p] == g) {
      a += 1.f / (1.f + expf(-s_all[p]));
      na += 1.f;
      if (!is_ex[p]) { l += 1.f / (1.f + expf(-s_l[p])); nl += 1.f; }
    }
  a = wave_sum(a); na = wave_sum(na); l = wave_sum(l); nl = wave_sum(nl);
  if (threadIdx.x == 0) {
    const float ca = a / fmaxf(na, 1.f), cl = l / fmaxf(nl, 1.f);
    count_l[g] = cl;
    count_u[g] = cl + fmaxf(ca - cl, 0.f);
  }
  (void)red;
}

int num_cu() {
  static int cached[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (cached[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cached[dev] = n;
  }
  return cached[dev];
}

}  // namespace pg

using namespace pg;

extern "C" const char* pg_last_error(void) { return pg::g_err; }
extern "C" int pg_abi_version(void) { return 11; }
// sizeof of the argument structs as THIS build sees them (a binding's mirror is checked against it when the library is loaded)
extern "C" int pg_abi_struct_sizes(int* out, int n) {
  const int v[5] = {(int)sizeof(PgGemm), (int)sizeof(PgTopo), (int)sizeof(PgSegAttn), (int)sizeof(PgSegAttnGrad), (int)sizeof(PgLaunch)};
  for (int i = 0; i < n && i < 5; ++i) out[i] = v[i];
  return 5;
}

// ---- order points between the streams of one step (include/phoregen_hip.h) ----
// An event here only orders kernels of this device against each other: it needs neither a timestamp nor the system-scope fence (L2
// write-back for the host) a default event carries.  The record is a device-scope release (hipEventReleaseToDevice, the flag HIP
// documents for exactly this).  Round 4 shipped hipEventDisableSystemFence instead -- documented for timing-only events, its record
// carries no release of its own and visibility then rests on every kernel packet's own release/acquire, a runtime detail; that
// form stays reachable for measurement only (pg_debug_order_point_fence_free; tools/micro/stream_packets.py measured a record at
// 5.1 -> 3.7 us and a cross-stream hop at 29 -> 25 us against a default event).
static int g_order_point_fence_free = 0;
// (a process-global switch that weakens every order point created after it: honoured only when the process was started with
//  PHOREGEN_DEBUG=1 in its environment -- the product path cannot reach the weaker form by accident)
extern "C" int pg_debug_order_point_fence_free(int on) {
  const char* dbg = getenv("PHOREGEN_DEBUG");
  if (on && !(dbg && dbg[0] == '1')) {
    set_error("pg_debug_order_point_fence_free: measurement switch, needs PHOREGEN_DEBUG=1 in the environment");
    return PG_ERR_ARG;
  }
  g_order_point_fence_free = on;
  return 0;
}
extern "C" int pg_order_point_create(void** ev) {
  hipEvent_t e = nullptr;
  const unsigned flags = hipEventDisableTiming | (g_order_point_fence_free ? hipEventDisableSystemFence : hipEventReleaseToDevice);
  const hipError_t rc = hipEventCreateWithFlags(&e, flags);
  if (rc != hipSuccess) { set_error("pg_order_point_create: %s", hipGetErrorString(rc)); return 1; }
  *ev = e;
  return 0;
}
extern "C" int pg_order_point_destroy(void* ev) {
  const hipError_t rc = hipEventDestroy(static_cast<hipEvent_t>(ev));
  if (rc != hipSuccess) { set_error("pg_order_point_destroy: %s", hipGetErrorString(rc)); return 1; }
  return 0;
}
extern "C" int pg_order_point_record(void* ev, void* stream) {
  const hipError_t rc = hipEventRecord(static_cast<hipEvent_t>(ev), static_cast<hipStream_t>(stream));
  if (rc != hipSuccess) { set_error("pg_order_point_record: %s", hipGetErrorString(rc)); return 1; }
  return 0;
}
extern "C" int pg_order_point_wait(void* ev, void* stream) {
  const hipError_t rc = hipStreamWaitEvent(static_cast<hipStream_t>(stream), static_cast<hipEvent_t>(ev), 0);
  if (rc != hipSuccess) { set_error("pg_order_point_wait: %s", hipGetErrorString(rc)); return 1; }
  return 0;
}

extern "C" int pg_embed_ctx(const PgTopo* t, const float* h_node_pert, const float* pos_pert, const int64_t* time_step,
                            const float* W_node, const float* t_off, const float* t_coeff, const float* h_phore_emb,
                            const float* pos_phore, const int* phore2ctx, float* h_ctx, float* x_ctx, void* stream) {
  const long n = (long)(t->n_lig + t->n_phore) * 128;
  if (n == 0) return PG_OK;
  hipLaunchKernelGGL(embed_ctx_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, *t, h_node_pert,
                     pos_pert, time_step, W_node, t_off, t_coeff, h_phore_emb, pos_phore, phore2ctx, h_ctx, x_ctx);
  return check_launch("pg_embed_ctx");
}

extern "C" int pg_embed_bond(const PgTopo* t, const float* h_edge_pert, const int* bond_graph, const int64_t* time_step,
                             const float* W_edge, const float* t_off, const float* t_coeff, float* h_bond, void* stream) {
  const long n = (long)t->n_bond * 128;
  if (n == 0) return PG_OK;
  hipLaunchKernelGGL(embed_bond_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, t->n_bond, t->edge_ref,
                     h_edge_pert, bond_graph, time_step, W_edge, t_off, t_coeff, h_bond);
  return check_launch("pg_embed_bond");
}

// wave_knn holds KNN_MAXC candidates per lane: a graph with more context nodes would silently lose its far candidates
static int check_graph_size(const PgTopo* t, const char* what) {
  if (t->max_gctx > 64 * KNN_MAXC) {
    set_error("%s: a graph of this batch has %d context nodes (ligand + pharmacophore); the in-graph kNN holds at most %d",
              what, t->max_gctx, 64 * KNN_MAXC);
    return PG_ERR_ARG;
  }
  return PG_OK;
}

extern "C" int pg_knn_ctx(const PgTopo* t, const float* x_ctx, int k, int* nbr, int* deg, void* stream) {
  if (k < 1 || k > 64) { set_error("pg_knn_ctx: k = %d out of range [1, 64]", k); return PG_ERR_ARG; }
  if (int rc = check_graph_size(t, "pg_knn_ctx")) return rc;
  hipLaunchKernelGGL(knn_ctx_kernel, dim3((t->n_ctx + 3) / 4), dim3(256), 0, (hipStream_t)stream, *t, x_ctx, k, nbr, deg);
  return check_launch("pg_knn_ctx");
}

extern "C" int pg_knn_group_by_kind(const PgTopo* t, int k, int* nbr, const int* deg, float* ew, void* stream) {
  if (k < 1 || k > 64) { set_error("pg_knn_group_by_kind: k = %d out of range [1, 64]", k); return PG_ERR_ARG; }
  if (t->n_ctx == 0) return PG_OK;
  hipLaunchKernelGGL(knn_group_kernel, dim3((t->n_ctx + 3) / 4), dim3(256), 0, (hipStream_t)stream, *t, k, nbr, deg, ew);
  return check_launch("pg_knn_group_by_kind");
}

extern "C" int pg_lig_normals(const PgTopo* t, const float* x_ctx, const float* phore_norm, const int* phore2ctx,
                              float* nrm, void* stream) {
  if (t->max_nlig > 64 * KNN_MAXC) { set_error("pg_lig_normals: ligand of %d atoms (limit %d)", t->max_nlig, 64 * KNN_MAXC); return PG_ERR_ARG; }
  const int n = t->n_lig + t->n_phore;
  hipLaunchKernelGGL(lig_normals_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, *t, x_ctx, phore_norm,
                     phore2ctx, nrm);
  return check_launch("pg_lig_normals");
}

extern "C" int pg_lig_nn3(const PgTopo* t, const float* x_ctx, int* nn3, void* stream) {
  if (t->n_lig == 0) return PG_OK;
  if (t->max_nlig > 64 * KNN_MAXC) { set_error("pg_lig_nn3: ligand of %d atoms (limit %d)", t->max_nlig, 64 * KNN_MAXC); return PG_ERR_ARG; }
  hipLaunchKernelGGL(lig_nn3_kernel, dim3((t->n_lig + 3) / 4), dim3(256), 0, (hipStream_t)stream, *t, x_ctx, nn3);
  return check_launch("pg_lig_nn3");
}

extern "C" int pg_edge_gate(const PgTopo* t, const float* x_ctx, const int* nbr, const int* deg, int k, const float* W0,
                            const float* b0, const float* gamma, const float* beta, const float* W3, float b3,
                            float* ew, void* stream) {
  (void)gamma;
  hipLaunchKernelGGL(edge_gate_kernel, dim3((t->n_ctx + 3) / 4), dim3(256), 0, (hipStream_t)stream, t->n_ctx, x_ctx, nbr,
                     deg, k, W0, b0, beta, W3, b3, ew);
  return check_launch("pg_edge_gate");
}

extern "C" int pg_bond_smear(const PgTopo* t, const float* x_ctx, float* G, void* stream) {
  const long n = (long)t->n_bond * 20;
  if (n == 0) return PG_OK;
  hipLaunchKernelGGL(bond_smear_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, t->n_bond,
                     t->bond_src, t->bond_dst, x_ctx, G);
  return check_launch("pg_bond_smear");
}

extern "C" int pg_apply_dx(const PgTopo* t, const float* x, const float* dx1, const float* dx2, float* x_new,
                           void* stream) {
  const int n = t->n_ctx * 3;
  hipLaunchKernelGGL(apply_dx_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, t->n_ctx,
                     t->ctx_is_lig, x, dx1, dx2, x_new);
  return check_launch("pg_apply_dx");
}

extern "C" int pg_layer_geom(const PgTopo* t, const float* x, const float* dx1, const float* dx2, const float* nrm_phore_ctx,
                             float* x_new, float* nrm, float* G, void* stream) {
  if (t->n_graphs == 0) return PG_OK;
  if (t->max_nlig > 64 * KNN_MAXC) { set_error("pg_layer_geom: ligand of %d atoms (limit %d)", t->max_nlig, 64 * KNN_MAXC); return PG_ERR_ARG; }
  if ((dx1 == nullptr) != (dx2 == nullptr) || (dx1 && !x_new) || (nrm && !nrm_phore_ctx)) {
    set_error("pg_layer_geom: dx1 / dx2 / x_new come together; nrm needs nrm_phore_ctx");
    return PG_ERR_ARG;
  }
  // enough workgroups for the chip at any batch size: the parts of a graph share its bond rows and atoms
  int parts = (2 * kNumCU + t->n_graphs - 1) / t->n_graphs;
  parts = parts < 1 ? 1 : (parts > 8 ? 8 : parts);
  if (!G && !nrm) parts = 1;
  hipLaunchKernelGGL(layer_geom_kernel, dim3(t->n_graphs, parts), dim3(GEOM_THREADS), 0, (hipStream_t)stream, *t, x, dx1, dx2,
                     nrm_phore_ctx, x_new, nrm, G);
  return check_launch("pg_layer_geom");
}

extern "C" int pg_atom_count(const float* s_all, const float* s_l, const uint8_t* is_ex, const int* phore_graph,
                             int n_phore, int n_graphs, float* count_l, float* count_u, void* stream) {
  hipLaunchKernelGGL(atom_count_kernel, dim3(n_graphs), dim3(64), 0, (hipStream_t)stream, s_all, s_l, is_ex,
                     phore_graph, n_phore, count_l, count_u);
  return check_launch("pg_atom_count");
}
