// A denoiser forward as ONE call into the library: a pre-built launch list (PgLaunch records, include/phoregen_hip.h) walked on the
// host.  The list is what phoregen_amd/engine.py builds once per (weights, batch plan): the entry points of this library in launch
// order, each with the lane (HIP stream) it goes to, and the order points (record / wait pairs) between the lanes.  Walking it here
// instead of from Python makes a step one foreign call instead of ~260 (8 ranks of a node share the host's cores).
//
// Host code only: no kernel lives in this file.
#include "common.h"
#include "../../include/phoregen_hip.h"

#include <string.h>
#include <vector>

namespace {

struct Program {
  std::vector<PgLaunch> list;
  std::vector<hipEvent_t> events;
  bool poisoned = false;        // a run failed half-way: the lanes were drained, the list must not be walked again
};

// a failing entry: later records / waits of the list were skipped, so the side lanes are no longer ordered against lane 0 -- drain the
// device (nothing half-ordered stays in flight) and refuse the program from now on
int fail_run(Program* p, int rc) {
  p->poisoned = true;
  (void)hipDeviceSynchronize();
  return rc;
}

inline float as_float(uint64_t bits) {
  const uint32_t b = (uint32_t)bits;
  float f;
  memcpy(&f, &b, 4);
  return f;
}

#define PTR(i, T) (reinterpret_cast<T>((uintptr_t)L.a[i]))
#define INT(i) ((int)(int64_t)L.a[i])
#define FLT(i) (as_float(L.a[i]))
#define FP(i) PTR(i, float*)
#define CFP(i) PTR(i, const float*)
#define IP(i) PTR(i, int*)
#define CIP(i) PTR(i, const int*)
#define TOPO(i) PTR(i, const PgTopo*)

int n_args_of(int op) {
  switch (op) {
    case PG_OP_RECORD: case PG_OP_WAIT: return 0;
    case PG_OP_GEMM: return 1;
    case PG_OP_SEG_ATTN: return 2;
    case PG_OP_EMBED_CTX: return 12;
    case PG_OP_EMBED_BOND: return 8;
    case PG_OP_KNN_CTX: return 5;
    case PG_OP_LIG_NORMALS: return 5;
    case PG_OP_EDGE_GATE: return 12;
    case PG_OP_KNN_GROUP_BY_KIND: return 5;
    case PG_OP_BOND_SMEAR: return 3;
    case PG_OP_ATTN_FOLD_QUERY: return 6;
    case PG_OP_ATTN_UNFOLD_VALUE: return 8;
    case PG_OP_APPLY_DX: return 5;
    case PG_OP_LAYER_GEOM: return 8;
    case PG_OP_ROWS_LINEAR: return 10;
    case PG_OP_ATOM_COUNT: return 8;
    default: return -1;
  }
}

int run_one(const PgLaunch& L, void* s) {
  switch (L.op) {
    case PG_OP_GEMM: return pg_gemm(PTR(0, const PgGemm*), s);
    case PG_OP_SEG_ATTN: return pg_seg_attn(TOPO(0), PTR(1, const PgSegAttn*), s);
    case PG_OP_EMBED_CTX:
      return pg_embed_ctx(TOPO(0), CFP(1), CFP(2), PTR(3, const int64_t*), CFP(4), CFP(5), CFP(6), CFP(7), CFP(8), CIP(9), FP(10),
                          FP(11), s);
    case PG_OP_EMBED_BOND: return pg_embed_bond(TOPO(0), CFP(1), CIP(2), PTR(3, const int64_t*), CFP(4), CFP(5), CFP(6), FP(7), s);
    case PG_OP_KNN_CTX: return pg_knn_ctx(TOPO(0), CFP(1), INT(2), IP(3), IP(4), s);
    case PG_OP_LIG_NORMALS: return pg_lig_normals(TOPO(0), CFP(1), CFP(2), CIP(3), FP(4), s);
    case PG_OP_EDGE_GATE:
      return pg_edge_gate(TOPO(0), CFP(1), CIP(2), CIP(3), INT(4), CFP(5), CFP(6), CFP(7), CFP(8), CFP(9), FLT(10), FP(11), s);
    case PG_OP_KNN_GROUP_BY_KIND: return pg_knn_group_by_kind(TOPO(0), INT(1), IP(2), CIP(3), FP(4), s);
    case PG_OP_BOND_SMEAR: return pg_bond_smear(TOPO(0), CFP(1), FP(2), s);
    case PG_OP_ATTN_FOLD_QUERY: return pg_attn_fold_query(CFP(0), INT(1), CFP(2), INT(3), CIP(4), FP(5), s);
    case PG_OP_ATTN_UNFOLD_VALUE: return pg_attn_unfold_value(CFP(0), CFP(1), CFP(2), CFP(3), INT(4), CIP(5), FP(6), INT(7), s);
    case PG_OP_APPLY_DX: return pg_apply_dx(TOPO(0), CFP(1), CFP(2), CFP(3), FP(4), s);
    case PG_OP_LAYER_GEOM: return pg_layer_geom(TOPO(0), CFP(1), CFP(2), CFP(3), CFP(4), FP(5), FP(6), FP(7), s);
    case PG_OP_ROWS_LINEAR: return pg_rows_linear(CFP(0), INT(1), INT(2), CFP(3), CFP(4), INT(5), INT(6), CIP(7), FP(8), INT(9), s);
    case PG_OP_ATOM_COUNT:
      return pg_atom_count(CFP(0), CFP(1), PTR(2, const uint8_t*), CIP(3), INT(4), INT(5), FP(6), FP(7), s);
    default: pg::set_error("pg_program_run: unknown op %d", L.op); return PG_ERR_ARG;
  }
}

}  // namespace

extern "C" int pg_program_create(const PgLaunch* list, int n, int n_events, void** prog) {
  if (!list || n < 0 || n_events < 0 || !prog) { pg::set_error("pg_program_create: bad arguments"); return PG_ERR_ARG; }
  for (int i = 0; i < n; ++i) {
    const PgLaunch& L = list[i];
    const int want = n_args_of(L.op);
    if (want < 0 || L.n_arg != want) {
      pg::set_error("pg_program_create: entry %d: op %d takes %d arguments, %d given", i, L.op, want, L.n_arg);
      return PG_ERR_ARG;
    }
    if (L.lane < 0 || L.lane >= PG_PROGRAM_LANES) { pg::set_error("pg_program_create: entry %d: lane %d", i, L.lane); return PG_ERR_ARG; }
    if ((L.op == PG_OP_RECORD || L.op == PG_OP_WAIT) && (L.ev < 0 || L.ev >= n_events)) {
      pg::set_error("pg_program_create: entry %d: order point %d of %d", i, L.ev, n_events);
      return PG_ERR_ARG;
    }
  }
  Program* p = new Program();
  p->list.assign(list, list + n);
  p->events.resize(n_events, nullptr);
  for (int i = 0; i < n_events; ++i) {
    void* e = nullptr;
    if (pg_order_point_create(&e)) {
      for (int j = 0; j < i; ++j) (void)hipEventDestroy(p->events[j]);
      delete p;
      return PG_ERR_HIP;
    }
    p->events[i] = static_cast<hipEvent_t>(e);
  }
  *prog = p;
  return PG_OK;
}

extern "C" int pg_program_destroy(void* prog) {
  Program* p = static_cast<Program*>(prog);
  if (!p) return PG_OK;
  int rc = PG_OK;
  for (hipEvent_t e : p->events)
    if (e && hipEventDestroy(e) != hipSuccess) rc = PG_ERR_HIP;
  delete p;
  return rc;
}

extern "C" int pg_program_length(void* prog) { return prog ? (int)static_cast<Program*>(prog)->list.size() : -1; }

extern "C" int pg_program_run(void* prog, void* const* streams) {
  Program* p = static_cast<Program*>(prog);
  if (!p || !streams) { pg::set_error("pg_program_run: bad arguments"); return PG_ERR_ARG; }
  if (p->poisoned) { pg::set_error("pg_program_run: this program failed half-way in an earlier run (poisoned): destroy and rebuild it"); return PG_ERR_ARG; }
  for (const PgLaunch& L : p->list) {
    void* s = streams[L.lane];
    if (L.op == PG_OP_RECORD) {
      const hipError_t rc = hipEventRecord(p->events[L.ev], static_cast<hipStream_t>(s));
      if (rc != hipSuccess) { pg::set_error("pg_program_run: record: %s", hipGetErrorString(rc)); return fail_run(p, PG_ERR_HIP); }
    } else if (L.op == PG_OP_WAIT) {
      const hipError_t rc = hipStreamWaitEvent(static_cast<hipStream_t>(s), p->events[L.ev], 0);
      if (rc != hipSuccess) { pg::set_error("pg_program_run: wait: %s", hipGetErrorString(rc)); return fail_run(p, PG_ERR_HIP); }
    } else {
      const int rc = run_one(L, s);
      if (rc) return fail_run(p, rc);
    }
  }
  return PG_OK;
}
