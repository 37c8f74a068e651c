/*
 * phoregen_hip.h — C ABI of libphoregen_hip.so (gfx950 / MI355X).
 *
 * Drop-in boundary for PhoreGen's diffusion-denoising hot path.  The reference has no native code
 * (SURVEY.md 2.1): the arithmetic these entry points replace lives in PyTorch op chains and in the
 * un-vendored wheels torch-scatter / torch-sparse / torch-cluster.  Each entry point cites the
 * reference interface it stands in for (file:line under /root/reference).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer borrowed for the duration of the call; the library allocates
 *     nothing and keeps nothing (workspaces are passed in by the caller);
 *   - fp32 row-major tensors, int32 indices;
 *   - work is enqueued on `stream` (a hipStream_t passed as void*), no host synchronisation;
 *   - return value: 0 = ok, otherwise pg_last_error() describes the failure.
 *
 * Context ("ctx") node order is the reference's compose_context order (models/common.py:180-208):
 * per graph, pharmacophore nodes first, then ligand atoms.
 */
#ifndef PHOREGEN_HIP_H
#define PHOREGEN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

const char* pg_last_error(void);
int pg_abi_version(void);
/* sizeof(PgGemm), sizeof(PgTopo), sizeof(PgSegAttn), sizeof(PgSegAttnGrad), sizeof(PgLaunch) of the library's build -> out[0..n); returns 5.
 * A binding that mirrors the structs (phoregen_amd/hip.py) compares them with its own when it loads the library. */
int pg_abi_struct_sizes(int* out, int n);

/* ---- order points between the HIP streams one denoiser step is spread over (phoregen_amd/engine.py; the reference has no
 * counterpart: it runs on one torch stream).  An event without timestamp whose record is a DEVICE-scope release
 * (hipEventReleaseToDevice) instead of the system-scope fence of a default HIP event: it orders kernels of ONE device against each
 * other, it does not make results visible to the host -- callers synchronise the stream itself for that. */
int pg_order_point_create(void** ev);
int pg_order_point_destroy(void* ev);
int pg_order_point_record(void* ev, void* stream);
int pg_order_point_wait(void* ev, void* stream);          /* `stream` continues after the last record of `ev` */
/* 0 (default): hipEventDisableTiming | hipEventReleaseToDevice, HIP's documented device-scope release at the record.
 * 1 (measurement only, tools/): hipEventDisableTiming | hipEventDisableSystemFence, the round-4 form whose record carries no release
 * of its own.  Applies to order points created afterwards.  Refused (PG_ERR_ARG) unless the process runs with PHOREGEN_DEBUG=1. */
int pg_debug_order_point_fence_free(int on);

/* ---- one denoiser forward as ONE call: a pre-built launch list walked on the host side of the library ---------------------------
 * (reference: the ~3 000 torch ops one `self.forward` of the loop at models/diffusion.py:432-447 issues from Python.)
 * A PgLaunch stands for one entry point of this header (`op`), its arguments `a[0 .. n_arg)` in declaration order WITHOUT the
 * trailing stream -- pointers and integers as 64-bit values, a float as its bit pattern in the low 32 bits -- and the lane it is
 * enqueued on: streams[lane] of pg_program_run.  PG_OP_RECORD / PG_OP_WAIT are the order points between the lanes: `ev` indexes
 * events the program owns (pg_order_point_create flags).  Structs an argument points to (PgGemm, PgSegAttn, PgTopo) are NOT copied:
 * they, and every device buffer, must outlive the program. */
enum {
  PG_OP_RECORD = 0, PG_OP_WAIT = 1, PG_OP_GEMM = 2, PG_OP_SEG_ATTN = 3, PG_OP_EMBED_CTX = 4, PG_OP_EMBED_BOND = 5, PG_OP_KNN_CTX = 6,
  PG_OP_LIG_NORMALS = 7, PG_OP_EDGE_GATE = 8, PG_OP_KNN_GROUP_BY_KIND = 9, PG_OP_BOND_SMEAR = 10, PG_OP_ATTN_FOLD_QUERY = 11,
  PG_OP_ATTN_UNFOLD_VALUE = 12, PG_OP_APPLY_DX = 13, PG_OP_LAYER_GEOM = 14, PG_OP_ROWS_LINEAR = 15, PG_OP_ATOM_COUNT = 16
};
#define PG_PROGRAM_LANES 4
#define PG_LAUNCH_MAX_ARGS 12
typedef struct {
  int32_t op;                          /* PG_OP_* */
  int32_t lane;                        /* 0 .. PG_PROGRAM_LANES-1 */
  int32_t ev;                          /* PG_OP_RECORD / PG_OP_WAIT: order point index */
  int32_t n_arg;
  uint64_t a[PG_LAUNCH_MAX_ARGS];
} PgLaunch;
int pg_program_create(const PgLaunch* list, int n, int n_events, void** prog);   /* validates and copies the list, creates the events */
int pg_program_run(void* prog, void* const* streams /*[PG_PROGRAM_LANES] hipStream_t*/);
/* (a failing entry leaves the lanes half-enqueued: the call then drains the device, marks the program POISONED and returns the entry's
 *  error; every later run of a poisoned program is refused -- the owner destroys it and rebuilds its state) */
int pg_program_length(void* prog);
int pg_program_destroy(void* prog);

/* ---- measurement aid: `workgroups` x 4 waves issue `iters` x 16 v_mfma_f32_16x16x4_f32 each (8 independent chains) and nothing else;
 * *flops (host, optional) = the FLOPs of the launch.  bench.py times it with HIP events: the fp32 matrix rate this GPU sustains,
 * quoted beside the nominal peak.  1024 workgroups = 4 waves per SIMD of 256 CUs. */
int pg_micro_mfma_f32(int workgroups, int iters, float* sink, double* flops, void* stream);

/* ---- MFMA lane-map self test (device writes 0 on success) -------------------------------- */
int pg_selftest_mfma(int* d_result, void* stream);
/* ---- raw words of the device generator (Philox4x32-10, Salmon et al. SC'11) for known-answer tests:
 * ctr_key [n][6] = counter c0..c3, key k0 k1  ->  out [n][4].  The transition kernels call it with
 * key = seed (lo, hi), counter = (element index lo, hi, step, stream_id). */
int pg_selftest_philox(const uint32_t* ctr_key, int n, uint32_t* out, void* stream);
/* test hook (returns the old setting): bit 0 routes the node-target modes of pg_seg_attn through the generic one-pass kernel,
 * bit 1 keeps PG_SEG_TRIPLET on the gather kernel (csrc/triplet.hip) even when the staged one (csrc/triplet2.hip) applies,
 * bit 2 runs the staged kernel with 8 instead of 12 waves per workgroup (tuning) */
int pg_debug_force_generic_seg(int mask);

/* ---- dense linear layers -------------------------------------------------------------------
 * Y[R, n] = out_scale * act( sum_k Xcat[R,k] * W[n,k] + bias[n] + add1[i1(r), n] + add2[i2(r), n] ),  R = rows ? rows[r] : r
 * Xcat = [X | X2] along k; if ln_gamma != NULL, X rows first go through LayerNorm(K1, eps 1e-5)+ReLU.
 * Replaces nn.Linear / MLP second halves (models/common.py:99-119), the per-node / per-edge halves
 * of the first MLP layer that the reference computes on concatenated gathers
 * (models/uni_denoiser.py:43-59,141-155,190-201), lin_node (:288) and the heads
 * (models/diffusion.py:55-59,71-75,223,241). */
typedef struct {
  const float* X;   int ldx;  int K1;
  const float* X2;  int ldx2; int K2;
  const float* W;   int ldw;
  const float* bias;
  const float* ln_gamma; const float* ln_beta;
  const float* add1; int ld_add1; const int* idx1;
  const float* add2; int ld_add2; const int* idx2;
  float out_scale;
  int   act;                 /* 0 none, 1 shifted softplus (models/common.py:58-64), 2 ReLU */
  float* Y; int ldy; int M; int N;
  const int* rows;           /* optional [M]: logical row r reads X/X2 row rows[r] and writes Y row rows[r] (row subset) */
  int add_rows;              /* rows of add1 / add2 (upper bound of idx1 / idx2 values + 1); 0 = unknown.  Lets pg_gemm pick the
                              * streaming kernel, which addresses the gathered rows with 32-bit byte offsets */
} PgGemm;
int pg_gemm(const PgGemm* p, void* stream);
/* test hook (returns the old setting): 0 keeps every product on the tiled kernel (csrc/gemm.hip) instead of the streaming one
 * (csrc/gemm_stream.hip), so that the tests can hold the two against each other */
int pg_debug_gemm_streaming(int on);

/* ---- graph topology of one batch ------------------------------------------------------------
 * Built once per batch by the host mirror (phoregen_amd/plan.py); constant over the 1000 steps. */
typedef struct {
  int n_graphs, n_ctx, n_lig, n_phore, n_bond;
  int max_nlig;             /* largest ligand of the batch (selects the triplet kernel variant)           */
  int max_gctx;             /* most context nodes (ligand + pharmacophore) of any graph; pg_knn_ctx holds <= 512 */
  const int* g_ctx_off;     /* [B+1] first ctx node of graph g                                   */
  const int* g_nph;         /* [B]   pharmacophore nodes of graph g (ctx rows g_ctx_off[g]..+nph) */
  const int* g_nlig;        /* [B]   ligand atoms of graph g (follow the phore nodes)             */
  const int* g_eid_off;     /* [B+1] offset of graph g's n_lig x n_lig edge-id table in `eid`     */
  const int* eid;           /* eid[off_g + a_src*n + a_dst] = bond edge id, -1 on the diagonal    */
  const int* ctx_graph;     /* [n_ctx] graph of a ctx node                                        */
  const uint8_t* ctx_is_lig;/* [n_ctx]                                                            */
  const int* lig2ctx;       /* [n_lig] ctx index of ligand atom a (= l_index_in_ctx, common.py:166-177) */
  const int* bond_src;      /* [n_bond] ctx index of edge source (edge_index[0], diffusion.py:201) */
  const int* bond_dst;      /* [n_bond] ctx index of edge target                                  */
  const int* bond_desc;     /* [n_bond][4] {ctx j, local_i | local_j << 16, n_lig of the graph, eid offset of the graph} */
  const int* g_bond_off;    /* [B+1] first bond row of graph g                                    */
  const int* edge_ref;      /* [n_bond] or NULL: bond rows inside the library are in the host mirror's INTERNAL order (per graph
                               target-major: the edges k -> i of one target i are contiguous, k ascending; phoregen_amd/plan.py);
                               edge_ref[e] = row of internal edge e in the caller's edge arrays (h_edge_pert, h_edge_prev).
                               NULL: the caller's order already is the internal one */
} PgTopo;

/* ---- embeddings (models/diffusion.py:180-183,205; models/common.py:34-55) -------------------- */
int pg_embed_ctx(const PgTopo* t, const float* h_node_pert /*[n_lig,12]*/, const float* pos_pert /*[n_lig,3]*/,
                 const int64_t* time_step /*[B]*/, const float* W_node /*[118,12]*/,
                 const float* t_offset /*[10]*/, const float* t_coeff /*[10]*/,
                 const float* h_phore_emb /*[n_phore,128]*/, const float* pos_phore /*[n_phore,3]*/,
                 const int* phore2ctx /*[n_phore]*/, float* h_ctx /*[n_ctx,128]*/, float* x_ctx /*[n_ctx,3]*/,
                 void* stream);      /* h_ctx == NULL / x_ctx == NULL: that half is skipped (the features of the NEXT reverse step are embedded
                                        as soon as its types are drawn, the coordinates when its positions are: phoregen_amd/engine.py) */
int pg_embed_bond(const PgTopo* t, const float* h_edge_pert /*[n_bond,6], caller's order (t->edge_ref)*/, const int* bond_graph,
                  const int64_t* time_step, const float* W_edge /*[118,6]*/, const float* t_offset,
                  const float* t_coeff, float* h_bond /*[n_bond,128]*/, void* stream);

/* ---- neighbour search (torch_cluster.knn_graph via uni_denoiser.py:355 and common.py:301) ----
 * pg_knn_ctx: k nearest other ctx nodes of the same graph -> nbr[n_ctx,k] (ascending distance, index
 *   breaks ties), deg[n_ctx] = min(k, nodes_in_graph-1).
 * pg_lig_normals: nrm[ctx] = mean of the 3 nearest ligand atoms' positions - x  for ligand atoms
 *   (common.py:300-304), phore_norm rows for pharmacophore nodes (common.py:312-314). */
int pg_knn_ctx(const PgTopo* t, const float* x_ctx, int k, int* nbr, int* deg, void* stream);
int pg_lig_normals(const PgTopo* t, const float* x_ctx, const float* phore_norm, const int* phore2ctx,
                   float* nrm, void* stream);

/* nn3[n_lig,3]: ctx ids of the neighbours pg_lig_normals averages (-1 = fewer than 3 other atoms); training path */
int pg_lig_nn3(const PgTopo* t, const float* x_ctx, int* nn3, void* stream);

/* global edge gate e_w = sigmoid(MLP(smear(dist)))  (uni_denoiser.py:410-415); weights in kernel layout
 * (phoregen_amd/packing.py pack_gate): W0 = lane-fixed centred/sign-normalised first layer [5][8][64], b0 = its bias
 * [128], gamma unused, beta = beta/|gamma| [128], W3 = last-layer row times |gamma| [128] */
int pg_edge_gate(const PgTopo* t, const float* x_ctx, const int* nbr, const int* deg, int k,
                 const float* W0, const float* b0, const float* gamma, const float* beta,
                 const float* W3, float b3, float* ew /*[n_ctx,k]*/, void* stream);

/* optional, after pg_edge_gate: stable partition of every node's neighbour slots (and their gate values) by the kind of the source
 * node, ligand atoms first.  The attention kernels sum over a node's rows, so the order is free; with it at most one 16-row tile
 * per node mixes the two kinds and the uniform tiles skip the other kind's 20 distance columns (csrc/node_attn.hip). */
int pg_knn_group_by_kind(const PgTopo* t, int k, int* nbr /*[n_ctx,k] in place*/, const int* deg, float* ew /*[n_ctx,k] in place*/,
                         void* stream);

/* per-bond Gaussian smearing of the bond length: G[e, 0:20] (uni_denoiser.py:128,137) */
int pg_bond_smear(const PgTopo* t, const float* x_ctx, float* G /*[n_bond,20]*/, void* stream);

/* ---- segment attention (the hot path) --------------------------------------------------------
 * One call = one attention sub-layer of AttentionLayerO2TwoUpdateNodeGeneral (uni_denoiser.py:260-298)
 * with the MLP first layers factored (SURVEY.md 7 "hard parts"):
 *   hidden_m[row] = Csrc_m[src_row(row)] + Cdst_m[segment] + Wf_m * feat(row)          m in {k, v}
 *   z_m = ReLU(LayerNorm(hidden_m));  logits[row,h] = z_k . U[segment][:,h]
 *   alpha = softmax over the rows of a segment;  S[segment][:,h] = sum_row alpha*gate * z_v
 * U / S are the per-segment query-folded key weights and value pre-images.
 * Modes: */
enum {
  PG_SEG_KNN_NODE = 0,   /* NodeUpdateLayer on knn edges      (uni_denoiser.py:40-72 via :281)  */
  PG_SEG_KNN_POS  = 1,   /* PosUpdateLayer  on knn edges      (uni_denoiser.py:187-209 via :291) */
  PG_SEG_BOND_NODE = 2,  /* NodeUpdateLayer on bond edges     (:284)                             */
  PG_SEG_BOND_POS  = 3,  /* PosUpdateLayer  on bond edges     (:294)                             */
  PG_SEG_TRIPLET   = 4,  /* BondUpdateLayer                   (uni_denoiser.py:101-165 via :285) */
  PG_SEG_PHORE     = 5   /* phore encoder NodeUpdateLayer     (diffusion.py:186-191)             */
};

typedef struct {
  int mode;
  int n_seg;
  const int* seg_ids;        /* [n_seg] ctx node ids (node modes); triplet: bond edge ids in visiting order, NULL = 0..n_seg-1 */
  const int* seg_chunks;     /* triplet: [257] cost-balanced segment ranges, one per workgroup; NULL = equal split */
  /* geometry */
  const float* x;            /* [n_ctx,3] positions the features are computed from               */
  const float* nrm;          /* [n_ctx,3] direction vectors (knn modes)                          */
  const int* nbr; const int* deg; const float* ew; int knn_k;   /* knn modes                      */
  /* factored first layer */
  const float* Csrc_k; const float* Csrc_v; int ld_csrc;
  const float* Cdst_k; const float* Cdst_v; int ld_cdst;
  const float* Wf_k; const float* Wf_v;  /* lane-fixed [F/4][8][64] feature weights; node modes: 16-byte aligned (as are Wf_k2,
                                            Wf_v2 and W2xv_l: the tables go to LDS in 16-byte pieces; PG_ERR_ARG otherwise) */
  const float* Wg2_k; const float* Wg2_v;/* triplet: [20][128] weights of smear(d_ji)             */
  const float* G;                        /* triplet: [n_bond,20] from pg_bond_smear               */
  const float* ln_gk; const float* ln_bk; const float* ln_gv; const float* ln_bv;
  /* attention */
  const float* U;            /* node modes: [n_ctx][32][64] lane-fixed U (pg_attn_fold_query)      */
  const float* q;            /* triplet: [n_bond,128] queries, pre-scaled by 1/sqrt(head_dim)      */
  const float* W2k_l;        /* triplet: lane-fixed second-layer key weights [64][64][4]           */
  const float* W2v_l; const float* b2v;      /* triplet: lane-fixed value weights + bias          */
  /* Fused form of the four node modes (the sampler): q [n_ctx,128] and W2k_l given -> the query is folded in-kernel and U is not
   * read; the node-update modes (KNN_NODE, BOND_NODE) given W2v_l, b2v and out [n_ctx,128] (rows 128 floats apart) also apply
   * the value second layer and write out[seg,:] = W2v . S + b2v * sum(alpha * gate) instead of S / swn, i.e. what
   * pg_attn_fold_query -> pg_seg_attn -> pg_attn_unfold_value compute as three launches.  U (and S, swn) must still point at
   * scratch of the plain form's size: shapes the two-pass kernels do not hold (knn_k > 32, ligands above 80 atoms) run the
   * three launches internally. */
  const float* W2xv_l; const float* b2xv;    /* pos modes: lane-fixed [32][64] + [16]             */
  /* outputs */
  float* S; float* swn;      /* node modes: [n_ctx][32][64], [n_ctx][16]                          */
  const float* resid; float* out;            /* triplet: out[e,:] = resid[e,:] + update           */
  float* dx;                 /* pos modes: [n_ctx,3]                                              */
  int accumulate_dx;         /* pos modes: dx += instead of =                                     */
  float* alpha; int alpha_rows; /* optional (training): triplet S-form and the node-update modes of csrc/node_attn.hip write the
                                   softmax weights (x gate) [segments][alpha_rows][16]; its position modes write the logits and the
                                   value scalars of every row [segments][alpha_rows][32] -- what PgSegAttnGrad.alpha takes         */
  /* PG_SEG_PHORE, optional: the scalar edge feature given explicitly instead of computed as |x_dst - x_src| (the standalone
   * NodeUpdateLayer.forward(h, edge_feat, edge_index) of models/uni_denoiser.py:40-72 receives it from its caller):
   * efeat[efeat_off[g] + src_local * p_g + dst_local] for graph g with p_g nodes = the order of
   * fully_connect_two_graphs (models/common.py:329-356).  NULL: distances from `x`. */
  const float* efeat; const int* efeat_off;
  /* PG_SEG_TRIPLET, optional: source-atom groups for the LDS-staged kernel (csrc/triplet2.hip).  tri_iters [n_tri_iters][4] =
   * {ctx index of the ligand's first atom, n | j0 << 8 | A << 16, first internal bond row of the graph, s0 | s1 << 16}: the A
   * consecutive source atoms j0.. of an n-atom ligand, A*(n-1) <= 80, longest first; the entry covers the segments [s0, s1) of the
   * group's A*(n-1) (0: all of them); tri_counter: TWO ints of scratch (queue head, exit count), zero before the first launch --
   * every launch leaves them zero again.
   * Requires the target-major bond order of the host mirror, Csrc_k/Csrc_v = the two halves of one [n_bond,256] tensor and
   * Cdst_k/Cdst_v [n_bond] rows = smear(d_ji) . Wg2 of the segment's own edge (as in the adjoint's contract). */
  const int* tri_iters; int n_tri_iters; int* tri_counter;
  /* Fused knn modes, optional: a SECOND target list served by the same launch (the pharmacophore targets beside the ligand targets
   * of one sub-layer; they differ in the feature weights only): seg_ids2 [n_seg2], Wf_k2 / Wf_v2.  The persistent workgroups are
   * split between the two lists in proportion to their sizes, so that the two lists finish together instead of each rounding
   * its own number of node rounds up (one launch instead of two).  n_seg2 = 0: one list. */
  const int* seg_ids2; int n_seg2; const float* Wf_k2; const float* Wf_v2;
  /* PG_SEG_TRIPLET with tri_iters, optional: persistent workgroups of the staged kernel (0 = one per CU, 256).  A small batch
   * leaves some CUs to the launches that run beside the triplet kernel on other streams: 16 graphs of the headline shape 3.76 ->
   * 3.57 ms per step with 200 workgroups (the results do not depend on it: the queue hands out the same segments). */
  int tri_grid;
  /* Fused position modes (KNN_POS, BOND_POS), optional: 1 = the row tiles of a node over several waves of a workgroup (a small
   * batch's launch of a few hundred nodes is bound by the dependent chain inside the one wave that owns a node); the sequential
   * chains are run in the one-wave kernel's order, so the result is bit-identical.  Ignored for shapes it does not hold (ligands
   * above 64 atoms, k > 32, training outputs): those take the one-wave kernel. */
  int pos_tiled;
  /* PG_SEG_TRIPLET with tri_iters, optional: the largest ligand (atoms) among the queue's entries when that is smaller than
   * PgTopo.max_nlig.  The staged kernel is instantiated for the row tiles of the largest ligand it may meet, and the 4-tile
   * instance costs EVERY segment ~4 % (44 instead of 31 spilled registers): a batch with a few 51+-atom ligands (a segment visits n - 2 rows) runs them as a
   * second launch with its own queue and the rest on the 3-tile instance.  0 = PgTopo.max_nlig. */
  int tri_max_nlig;
} PgSegAttn;
int pg_seg_attn(const PgTopo* t, const PgSegAttn* p, void* stream);

/* U[s][c][h] = sum_d q[s,8h+d] * W2k[8h+d,c]  in lane-fixed layout (key second layer folded into the query) */
int pg_attn_fold_query(const float* q /*[n,128] pre-scaled*/, int ldq, const float* W2k_l, int n,
                       const int* ids, float* U, void* stream);
/* out[s, 8h+d] = sum_c W2v[8h+d,c] * S[s][c][h] + b2v[8h+d] * swn[s][h]   (swn / b2v NULL: no bias term) */
int pg_attn_unfold_value(const float* S, const float* swn, const float* W2v_l, const float* b2v, int n,
                         const int* ids, float* out, int ldo, void* stream);

/* x_new[i] = x[i] + (dx1[i] + dx2[i]) * is_lig[i]   (uni_denoiser.py:295-296) */
int pg_apply_dx(const PgTopo* t, const float* x, const float* dx1, const float* dx2, float* x_new, void* stream);

/* Everything of a layer that depends on the coordinates only, as ONE launch (the arithmetic of pg_apply_dx, pg_bond_smear and
 * pg_lig_normals, bit for bit): x_new = x + (dx1 + dx2) * is_lig  (uni_denoiser.py:295-296), then FROM x_new the bond-length
 * smearing G[n_bond,20] (uni_denoiser.py:128,137) and the direction vectors nrm[n_ctx,3] (common.py:300-314) the next layer reads.
 * dx1 == dx2 == NULL: no update (x_new unused; G / nrm from x).  G == NULL / nrm == NULL: that product is skipped.
 * nrm_phore_ctx [n_ctx,3]: the pharmacophore normals in ctx row order (rows of ligand atoms unused). */
int pg_layer_geom(const PgTopo* t, const float* x, const float* dx1, const float* dx2, const float* nrm_phore_ctx,
                  float* x_new, float* nrm, float* G, void* stream);

/* small per-row linear: Y[r, 0:n_out] = W[n_out,K] . X[rows ? rows[r] : r, 0:K] + b   (n_out <= 16, K <= 256) */
int pg_rows_linear(const float* X, int ldx, int K, const float* W, const float* b, int n_out, int M,
                   const int* rows, float* Y, int ldy, void* stream);

/* atom-count heads: per graph mean of sigmoid(logit) over all / non-EX phore nodes, u = l + relu(c - l)
 * (diffusion.py:148-163); s_all / s_l are the pre-sigmoid outputs of atom_mlp / atom_mlp_1 */
int pg_atom_count(const float* s_all /*[n_phore]*/, const float* s_l /*[n_phore]*/, const uint8_t* is_ex,
                  const int* phore_graph, int n_phore, int n_graphs, float* count_l, float* count_u, void* stream);

/* ---- reverse-diffusion step (models/transition.py:44-63,285-315, models/common.py:425-431) ----
 * categorical: log_softmax(logits) -> q_v_posterior(v0_prob=True) -> Gumbel-argmax -> one-hot.
 * `uniform` / `eps` != NULL replay given draws (parity mode); NULL -> counter-based Philox4x32-10 with key = seed and
 * counter = (element, step, stream_id).  element = flat index of the batch when graph_row0 == NULL; otherwise the
 * graph-keyed form (graph_key[g] << 32 | index of the element inside graph g), with graph_row0[g] = first row of graph g
 * and graph_key[g] = a caller-chosen id (NULL: g): a graph then draws the same noise in whatever batch or shard it is
 * sampled (per-graph sharding over GPUs reproduces the unsharded run). */
int pg_posterior_categorical(const float* logits, const float* log_vt_in, const int* row_graph,
                             const int64_t* time_step, const float* q_mats, const float* q_onestep_T,
                             int n_rows, int K, const float* uniform, uint64_t seed, uint32_t stream_id,
                             uint32_t step, const int* graph_row0, const int* graph_key, float* log_vt_out,
                             float* onehot_out, float* traj_out, void* stream);
int pg_posterior_position(const float* x_t, const float* x0, const int* row_graph, const int64_t* time_step,
                          const float* coef_x0, const float* coef_xt, const float* std_, const float* energy_grad,
                          const float* eps, uint64_t seed, uint32_t stream_id, uint32_t step, int n_rows,
                          const int* graph_row0, const int* graph_key, const float* center /*[B,3] per graph, added to traj_out only; or NULL*/,
                          float* x_prev, float* traj_out, void* stream);

/* The same step for a sampler loop that keeps the coordinates in the denoiser's ctx-ordered buffers (phoregen_amd/models/diffusion.py,
 * pipelined loop): x0 is read as x0_ctx[lig2ctx[row]] (the denoiser's final coordinate buffer: no gather launch in front), the new
 * position is also written to x_ctx_next[lig2ctx[row]] (the coordinates layer 0 of the NEXT step reads: no embedding launch behind;
 * may be the buffer x0_ctx points to) and x0 itself to x0_out [n_rows,3] (optional).  Same arithmetic, same noise. */
int pg_posterior_position_ctx(const float* x_t, const float* x0_ctx, const int* lig2ctx, const int* row_graph,
                              const int64_t* time_step, const float* coef_x0, const float* coef_xt, const float* std_,
                              const float* energy_grad, const float* eps, uint64_t seed, uint32_t stream_id, uint32_t step,
                              int n_rows, const int* graph_row0, const int* graph_key, const float* center, float* x_prev,
                              float* traj_out, float* x_ctx_next, float* x0_out, void* stream);

/* closed-form guidance gradient (models/diffusion.py:476-502, utils/sample_utils.py:135-165).  phore_center [B,3]: per
 * graph the mean position of its non-EX pharmacophore nodes.  Both energies are means over the graphs of the batch;
 * mean_over_graphs = that divisor (<= 0: this batch's n_graphs; a shard of a larger logical batch passes the full count). */
int pg_guidance_grad(const PgTopo* t, const float* x_lig /*[n_lig,3]*/, const float* h_edge_prev /*[n_bond,6]*/,
                     const int* lig_graph, const int* g_lig_off, int use_atom_prox, float min_d, float max_d,
                     int use_center_prox, const float* phore_center /*[B,3]*/, int mean_over_graphs, float* cnt_ws /*[B]*/,
                     float* mean_ws /*[B,3]*/, float* grad /*[n_lig,3]*/, void* stream);

/* ---- training path: backward kernels (PhoreDiff.compute_loss, models/diffusion.py:249-352) -------------------
 * The forward of a training step runs the same kernels as sampling; these entry points are their adjoints.
 * Gradient buffers marked (+=) are accumulated with atomics into caller-zeroed memory, (=) are overwritten. */

/* gW[n,k] (+=) sum_r dY[r,n] * X[r,k];  gb[n] (+=) sum_r dY[r,n] (gb may be NULL).  Adjoint of pg_gemm w.r.t. W / bias
 * (nn.Linear weight gradients). */
int pg_gemm_wgrad(const float* dY, int ldy, const float* X, int ldx, int M, int N, int K, float* gW, int ldgw,
                  float* gb, void* stream);

/* out[ctx(a)][0:ncol] (=) sum of Y[e][0:ncol] over the bond rows e whose source (by_src != 0) or target (by_src == 0) is ligand
 * atom a; rows of `out` are context nodes (pharmacophore rows are not written).  Adjoint of pg_gemm's gathered operand
 * add1[idx1[r]] with idx1 = PgTopo.bond_src / bond_dst (reference: the h_k / h_j columns of the concatenated first layers,
 * models/uni_denoiser.py:43-59,141-155,190-201).  ncol, ldy, ldo multiples of 4; rows 16-byte aligned. */
int pg_bond_rows_sum(const PgTopo* t, const float* Y, int ldy, int ncol, int by_src, float* out, int ldo, void* stream);

/* Y = ReLU(LayerNorm_128(X) * gamma + beta) and its adjoint (the LayerNorm+ReLU between the two Linear layers of
 * models/common.py:99-119 MLPs, used where the forward keeps it fused into pg_gemm's operand load).
 * gX (=), ggamma / gbeta (+=). */
int pg_ln_relu(const float* X, int ldx, const float* gamma, const float* beta, int M, float* Y, int ldy, void* stream);
int pg_ln_relu_bwd(const float* X, int ldx, const float* gamma, const float* beta, const float* gY, int ldgy, int M,
                   float* gX, int ldgx, float* ggamma, float* gbeta, void* stream);

/* adjoint of pg_seg_attn (same PgSegAttn inputs as the forward call, with U and Cdst_k/v given explicitly in every
 * mode; triplet: Cdst = smear(d_ji) . Wg2 computed by the caller, S/swn form) */
typedef struct {
  const float* gS; const float* gswn;     /* non-pos modes: gradient of S [..][32][64] and swn [..][16]            */
  const float* gdx;                        /* pos modes: gradient of dx [n_ctx,3]                                    */
  float* gU;                               /* (=) [..][32][64] lane-fixed, rows = segments                           */
  float* gCdst_k; float* gCdst_v; int ld_gcdst;   /* (=) rows = segments                                            */
  float* gCsrc_k; float* gCsrc_v; int ld_gcsrc;   /* (+=) rows as Csrc                                              */
  float* gWf_k; float* gWf_v;              /* (+=) lane-fixed [F/4][8][64]                                           */
  float* gbk; float* gbv;                  /* (+=) [128] gradient of b' = beta/|gamma| (PgSegAttn.ln_bk / ln_bv)     */
  float* gW2xv_l; float* gb2xv;            /* (+=) pos modes                                                         */
  float* gx; float* gnrm;                  /* (+=) [n_ctx,3]; NULL = not needed (pharmacophore encoder)              */
  float* gew;                              /* (=) knn modes [n_ctx,k]                                                */
  const float* alpha; int alpha_rows;      /* optional, from the forward (PgSegAttn.alpha): triplet / node-update modes: softmax weights
                                              [segments][alpha_rows][16]; position modes: logits | value scalars [..][alpha_rows][32] */
  const float* S; const float* swn;        /* ... with the forward's S / swn: one pass instead of two                */
  float* rowbuf; int rowbuf_rows; int grid;/* scratch: grid * pg_seg_attn_bwd_waves(mode) * rowbuf_rows * 48 floats,
                                              rowbuf_rows >= rows of the largest segment; grid = workgroups to launch */
  const int* atom_order;                   /* PG_SEG_TRIPLET, optional: [n_lig] ligand atoms (0 .. n_lig-1) in the order the persistent
                                              workgroups take them (workgroup b: entries b, b + grid, ...).  The kernel works off one SOURCE
                                              ATOM per workgroup round and an atom costs ~ (n-1) x ceil(n/16): in index order the slowest of
                                              256 workgroups carries 16 % more than the average on the config-5 batch; sorted by cost and
                                              dealt out in a snake it is 1 %.  NULL = index order.  Results do not depend on it. */
  float* dlogit; float* gfeat_v;           /* optional scratch, PG_SEG_TRIPLET (ligands up to 64 atoms), PG_SEG_KNN_NODE, PG_SEG_KNN_POS with the forward's
                                              per-row record (alpha) given: [segments][alpha_rows][16] and [segments][alpha_rows][16 * ceil(F / 16)]
                                              floats (F = 12 / 48 / 48).  With both, the adjoint runs as a value pass and a key pass (a wave holds ONE
                                              MLP path and fits 512 registers without spilling); the value pass leaves d logit (position update: one
                                              scalar per row) and its d feat rows there for the key pass.  Same gradients up to the order of the
                                              weight-gradient atomics. */
  int tri_form;                            /* PG_SEG_TRIPLET, ligands of up to 64 atoms, Cdst_v == Cdst_k + 128: 0 = one wave per 16-row tile
                                              (csrc/seg_attn_bwd.hip; the forms above); 1 / 2 = the channels of a tile split over the waves of a
                                              workgroup (csrc/triplet_bwd2.hip: nothing of the forward is read back -- alpha, S, swn, dlogit, gfeat_v,
                                              rowbuf unused; same gradients up to summation order): 1 = 4 waves x 32 channels, 2 = ligands of up to
                                              32 atoms on 8 waves x 16 channels and the larger ones in a second launch of the 4-wave form */
} PgSegAttnGrad;
int pg_seg_attn_bwd_waves(int mode);
int pg_seg_attn_bwd(const PgTopo* t, const PgSegAttn* p, const PgSegAttnGrad* g, void* stream);

/* gW2_l (+=) in the lane-fixed layout of W2k_l / W2v_l:  gW2[8h+d, c] += sum_s X[s, 8h+d] * T[s][c][h]
 * (adjoint of pg_attn_fold_query w.r.t. the weights with X = q, T = dU; of pg_attn_unfold_value with X = dout, T = S) */
int pg_attn_fold_wgrad(const float* X, int ldx, const float* T, int n, const int* ids, float* gW2_l, void* stream);

/* the bias side of pg_attn_unfold_value's adjoint over the rows `ids` (NULL: all n):
 * gswn[s, h] (=) sum_d gout[s, 8h+d] * b2v[8h+d];  gb2v[c] (+=) sum_s gout[s, c] * swn[s, c >> 3]  (rows outside `ids`: untouched) */
int pg_attn_unfold_bias_grad(const float* gout, int ldg, const float* swn, const float* b2v, int n, const int* ids,
                             float* gswn /*[.,16]*/, float* gb2v /*[128]*/, void* stream);

#ifdef __cplusplus
}
#endif
#endif
