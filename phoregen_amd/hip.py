"""ctypes binding of libphoregen_hip.so (the C ABI declared in include/phoregen_hip.h).

There is NO fallback: if the library is missing or no GPU is visible the product path raises.
"""
import ctypes as C
import struct
import sys
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, '_lib', 'libphoregen_hip.so')

c_fp = C.c_void_p
c_ip = C.c_void_p


class PgGemm(C.Structure):
    _fields_ = [('X', c_fp), ('ldx', C.c_int), ('K1', C.c_int),
                ('X2', c_fp), ('ldx2', C.c_int), ('K2', C.c_int),
                ('W', c_fp), ('ldw', C.c_int),
                ('bias', c_fp),
                ('ln_gamma', c_fp), ('ln_beta', c_fp),
                ('add1', c_fp), ('ld_add1', C.c_int), ('idx1', c_ip),
                ('add2', c_fp), ('ld_add2', C.c_int), ('idx2', c_ip),
                ('out_scale', C.c_float), ('act', C.c_int),
                ('Y', c_fp), ('ldy', C.c_int), ('M', C.c_int), ('N', C.c_int), ('rows', c_ip), ('add_rows', C.c_int)]


class PgTopo(C.Structure):
    _fields_ = [('n_graphs', C.c_int), ('n_ctx', C.c_int), ('n_lig', C.c_int), ('n_phore', C.c_int),
                ('n_bond', C.c_int), ('max_nlig', C.c_int), ('max_gctx', C.c_int),
                ('g_ctx_off', c_ip), ('g_nph', c_ip), ('g_nlig', c_ip), ('g_eid_off', c_ip), ('eid', c_ip),
                ('ctx_graph', c_ip), ('ctx_is_lig', c_ip), ('lig2ctx', c_ip), ('bond_src', c_ip),
                ('bond_dst', c_ip), ('bond_desc', c_ip), ('g_bond_off', c_ip), ('edge_ref', c_ip)]


class PgSegAttn(C.Structure):
    _fields_ = [('mode', C.c_int), ('n_seg', C.c_int), ('seg_ids', c_ip), ('seg_chunks', c_ip),
                ('x', c_fp), ('nrm', c_fp), ('nbr', c_ip), ('deg', c_ip), ('ew', c_fp), ('knn_k', C.c_int),
                ('Csrc_k', c_fp), ('Csrc_v', c_fp), ('ld_csrc', C.c_int),
                ('Cdst_k', c_fp), ('Cdst_v', c_fp), ('ld_cdst', C.c_int),
                ('Wf_k', c_fp), ('Wf_v', c_fp), ('Wg2_k', c_fp), ('Wg2_v', c_fp), ('G', c_fp),
                ('ln_gk', c_fp), ('ln_bk', c_fp), ('ln_gv', c_fp), ('ln_bv', c_fp),
                ('U', c_fp), ('q', c_fp), ('W2k_l', c_fp), ('W2v_l', c_fp), ('b2v', c_fp),
                ('W2xv_l', c_fp), ('b2xv', c_fp),
                ('S', c_fp), ('swn', c_fp), ('resid', c_fp), ('out', c_fp), ('dx', c_fp),
                ('accumulate_dx', C.c_int), ('alpha', c_fp), ('alpha_rows', C.c_int), ('efeat', c_fp), ('efeat_off', c_ip),
                ('tri_iters', c_ip), ('n_tri_iters', C.c_int), ('tri_counter', c_ip),
                ('seg_ids2', c_ip), ('n_seg2', C.c_int), ('Wf_k2', c_fp), ('Wf_v2', c_fp), ('tri_grid', C.c_int), ('pos_tiled', C.c_int), ('tri_max_nlig', C.c_int)]


class PgSegAttnGrad(C.Structure):
    _fields_ = [('gS', c_fp), ('gswn', c_fp), ('gdx', c_fp), ('gU', c_fp),
                ('gCdst_k', c_fp), ('gCdst_v', c_fp), ('ld_gcdst', C.c_int),
                ('gCsrc_k', c_fp), ('gCsrc_v', c_fp), ('ld_gcsrc', C.c_int),
                ('gWf_k', c_fp), ('gWf_v', c_fp), ('gbk', c_fp), ('gbv', c_fp),
                ('gW2xv_l', c_fp), ('gb2xv', c_fp), ('gx', c_fp), ('gnrm', c_fp), ('gew', c_fp),
                ('alpha', c_fp), ('alpha_rows', C.c_int), ('S', c_fp), ('swn', c_fp),
                ('rowbuf', c_fp), ('rowbuf_rows', C.c_int), ('grid', C.c_int), ('atom_order', c_ip),
                ('dlogit', c_fp), ('gfeat_v', c_fp), ('tri_form', C.c_int)]


PG_PROGRAM_LANES, PG_LAUNCH_MAX_ARGS = 4, 12


class PgLaunch(C.Structure):
    _fields_ = [('op', C.c_int32), ('lane', C.c_int32), ('ev', C.c_int32), ('n_arg', C.c_int32), ('a', C.c_uint64 * PG_LAUNCH_MAX_ARGS)]


OP_RECORD, OP_WAIT = 0, 1
# entry point -> PG_OP_* (include/phoregen_hip.h): what a launch list handed to pg_program_create may hold
PROGRAM_OPS = {'pg_gemm': 2, 'pg_seg_attn': 3, 'pg_embed_ctx': 4, 'pg_embed_bond': 5, 'pg_knn_ctx': 6, 'pg_lig_normals': 7,
               'pg_edge_gate': 8, 'pg_knn_group_by_kind': 9, 'pg_bond_smear': 10, 'pg_attn_fold_query': 11,
               'pg_attn_unfold_value': 12, 'pg_apply_dx': 13, 'pg_layer_geom': 14, 'pg_rows_linear': 15, 'pg_atom_count': 16}

SEG_KNN_NODE, SEG_KNN_POS, SEG_BOND_NODE, SEG_BOND_POS, SEG_TRIPLET, SEG_PHORE = range(6)
ACT_NONE, ACT_SSP, ACT_RELU = 0, 1, 2

ABI_VERSION = 11
_lib = None

_PROTOS = {
    'pg_last_error': (C.c_char_p, []),
    'pg_abi_version': (C.c_int, []),
    'pg_abi_struct_sizes': (C.c_int, [C.POINTER(C.c_int), C.c_int]),
    'pg_order_point_create': (C.c_int, [C.POINTER(C.c_void_p)]),
    'pg_order_point_destroy': (C.c_int, [C.c_void_p]),
    'pg_order_point_record': (C.c_int, [C.c_void_p, C.c_void_p]),
    'pg_order_point_wait': (C.c_int, [C.c_void_p, C.c_void_p]),
    'pg_debug_order_point_fence_free': (C.c_int, [C.c_int]),
    'pg_program_create': (C.c_int, [C.POINTER(PgLaunch), C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    'pg_program_run': (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    'pg_program_length': (C.c_int, [C.c_void_p]),
    'pg_program_destroy': (C.c_int, [C.c_void_p]),
    'pg_micro_mfma_f32': (C.c_int, [C.c_int, C.c_int, c_fp, C.POINTER(C.c_double), C.c_void_p]),
    'pg_selftest_mfma': (C.c_int, [c_ip, C.c_void_p]),
    'pg_selftest_philox': (C.c_int, [c_ip, C.c_int, c_ip, C.c_void_p]),
    'pg_debug_force_generic_seg': (C.c_int, [C.c_int]),
    'pg_gemm': (C.c_int, [C.POINTER(PgGemm), C.c_void_p]),
    'pg_debug_gemm_streaming': (C.c_int, [C.c_int]),
    'pg_embed_ctx': (C.c_int, [C.POINTER(PgTopo)] + [c_fp] * 11 + [C.c_void_p]),
    'pg_embed_bond': (C.c_int, [C.POINTER(PgTopo)] + [c_fp] * 7 + [C.c_void_p]),
    'pg_knn_ctx': (C.c_int, [C.POINTER(PgTopo), c_fp, C.c_int, c_ip, c_ip, C.c_void_p]),
    'pg_lig_normals': (C.c_int, [C.POINTER(PgTopo), c_fp, c_fp, c_ip, c_fp, C.c_void_p]),
    'pg_lig_nn3': (C.c_int, [C.POINTER(PgTopo), c_fp, c_ip, C.c_void_p]),
    'pg_edge_gate': (C.c_int, [C.POINTER(PgTopo), c_fp, c_ip, c_ip, C.c_int, c_fp, c_fp, c_fp, c_fp, c_fp,
                               C.c_float, c_fp, C.c_void_p]),
    'pg_knn_group_by_kind': (C.c_int, [C.POINTER(PgTopo), C.c_int, c_ip, c_ip, c_fp, C.c_void_p]),
    'pg_bond_smear': (C.c_int, [C.POINTER(PgTopo), c_fp, c_fp, C.c_void_p]),
    'pg_seg_attn': (C.c_int, [C.POINTER(PgTopo), C.POINTER(PgSegAttn), C.c_void_p]),
    'pg_attn_fold_query': (C.c_int, [c_fp, C.c_int, c_fp, C.c_int, c_ip, c_fp, C.c_void_p]),
    'pg_attn_unfold_value': (C.c_int, [c_fp, c_fp, c_fp, c_fp, C.c_int, c_ip, c_fp, C.c_int, C.c_void_p]),
    'pg_apply_dx': (C.c_int, [C.POINTER(PgTopo), c_fp, c_fp, c_fp, c_fp, C.c_void_p]),
    'pg_layer_geom': (C.c_int, [C.POINTER(PgTopo), c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, C.c_void_p]),
    'pg_attn_unfold_bias_grad': (C.c_int, [c_fp, C.c_int, c_fp, c_fp, C.c_int, c_ip, c_fp, c_fp, C.c_void_p]),
    'pg_bond_rows_sum': (C.c_int, [C.POINTER(PgTopo), c_fp, C.c_int, C.c_int, C.c_int, c_fp, C.c_int, C.c_void_p]),
    'pg_rows_linear': (C.c_int, [c_fp, C.c_int, C.c_int, c_fp, c_fp, C.c_int, C.c_int, c_ip, c_fp, C.c_int,
                                 C.c_void_p]),
    'pg_atom_count': (C.c_int, [c_fp, c_fp, c_ip, c_ip, C.c_int, C.c_int, c_fp, c_fp, C.c_void_p]),
    'pg_posterior_categorical': (C.c_int, [c_fp, c_fp, c_ip, c_ip, c_fp, c_fp, C.c_int, C.c_int, c_fp,
                                           C.c_uint64, C.c_uint32, C.c_uint32, c_ip, c_ip, c_fp, c_fp, c_fp, C.c_void_p]),
    'pg_posterior_position': (C.c_int, [c_fp, c_fp, c_ip, c_ip, c_fp, c_fp, c_fp, c_fp, c_fp, C.c_uint64,
                                        C.c_uint32, C.c_uint32, C.c_int, c_ip, c_ip, c_fp, c_fp, c_fp, C.c_void_p]),
    'pg_posterior_position_ctx': (C.c_int, [c_fp, c_fp, c_ip, c_ip, c_ip, c_fp, c_fp, c_fp, c_fp, c_fp, C.c_uint64,
                                            C.c_uint32, C.c_uint32, C.c_int, c_ip, c_ip, c_fp, c_fp, c_fp, c_fp, c_fp, C.c_void_p]),
    'pg_gemm_wgrad': (C.c_int, [c_fp, C.c_int, c_fp, C.c_int, C.c_int, C.c_int, C.c_int, c_fp, C.c_int, c_fp, C.c_void_p]),
    'pg_ln_relu': (C.c_int, [c_fp, C.c_int, c_fp, c_fp, C.c_int, c_fp, C.c_int, C.c_void_p]),
    'pg_ln_relu_bwd': (C.c_int, [c_fp, C.c_int, c_fp, c_fp, c_fp, C.c_int, C.c_int, c_fp, C.c_int, c_fp, c_fp,
                                 C.c_void_p]),
    'pg_seg_attn_bwd_waves': (C.c_int, [C.c_int]),
    'pg_seg_attn_bwd': (C.c_int, [C.POINTER(PgTopo), C.POINTER(PgSegAttn), C.POINTER(PgSegAttnGrad), C.c_void_p]),
    'pg_attn_fold_wgrad': (C.c_int, [c_fp, C.c_int, c_fp, C.c_int, c_ip, c_fp, C.c_void_p]),
    'pg_guidance_grad': (C.c_int, [C.POINTER(PgTopo), c_fp, c_fp, c_ip, c_ip, C.c_int, C.c_float, C.c_float,
                                   C.c_int, c_fp, C.c_int, c_fp, c_fp, c_fp, C.c_void_p]),
}

EXPORTS = tuple(_PROTOS)


def load_library(path=None):
    """Load the shared library and declare every prototype (no GPU needed for this)."""
    global _lib
    if _lib is not None:
        return _lib
    if path is None and os.environ.get('PHOREGEN_DEBUG') == '1':      # an instrumented / ablation build (tools/): only on explicit request
        path = os.environ.get('PHOREGEN_HIP_LIB')
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise RuntimeError(f'phoregen_amd: HIP extension not built: {path} is missing. '
                           f'Run `python -c "import __graft_entry__ as g; g.build()"` (hipcc, gfx950). '
                           f'There is no CPU fallback.')
    lib = C.CDLL(path)
    for name, (res, args) in _PROTOS.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    # the mirrors above against the build that was loaded: a stale .so (or a header edited without its mirror) fails here, not in a kernel
    if lib.pg_abi_version() != ABI_VERSION:
        raise RuntimeError(f'phoregen_amd: {path} has ABI version {lib.pg_abi_version()}, this package binds version {ABI_VERSION}: rebuild it')
    sizes = (C.c_int * 5)()
    lib.pg_abi_struct_sizes(sizes, 5)
    mine = [C.sizeof(x) for x in (PgGemm, PgTopo, PgSegAttn, PgSegAttnGrad, PgLaunch)]
    if list(sizes) != mine:
        raise RuntimeError(f'phoregen_amd: struct sizes of {path} {list(sizes)} differ from the ctypes mirrors {mine} '
                           '(PgGemm, PgTopo, PgSegAttn, PgSegAttnGrad, PgLaunch)')
    _lib = lib
    return lib


_gpu_checked = False


def lib():
    """The library, for compute calls: additionally requires a visible GPU (checked once, then cached)."""
    global _gpu_checked
    if _gpu_checked:
        return _lib
    l = load_library()
    if not torch.cuda.is_available():
        raise RuntimeError('phoregen_amd: no MI355X / ROCm device visible; the HIP path has no CPU fallback.')
    _gpu_checked = True
    return l


class OrderPoint:
    """A point of one stream that other streams of the same device can wait for (pg_order_point_*: a HIP event without timestamp
    and without the host-visibility fence).  Same two calls as torch.cuda.Event, so the engine can take either."""
    __slots__ = ('h', '_lib')

    def __init__(self):
        self._lib = lib()
        self.h = C.c_void_p()
        check(self._lib.pg_order_point_create(C.byref(self.h)), 'pg_order_point_create')

    def record(self, stream):
        if self._lib.pg_order_point_record(self.h, stream.cuda_stream):
            check(1, 'pg_order_point_record')

    def wait(self, stream):
        if self._lib.pg_order_point_wait(self.h, stream.cuda_stream):
            check(1, 'pg_order_point_wait')

    def __del__(self):
        # (at interpreter shutdown the runtime -- and this module's globals -- may already be gone: the process's events go with it)
        try:
            if self.h and not sys.is_finalizing():
                self._lib.pg_order_point_destroy(self.h)
                self.h = None
        except Exception:
            pass


def launch_record(fn, args, lane):
    """One call of the launch list as a PgLaunch: the arguments as ctypes would pass them (without the trailing stream)."""
    L = PgLaunch()
    L.op, L.lane, L.ev, L.n_arg = PROGRAM_OPS[fn.__name__], lane, -1, len(args)
    assert len(args) == len(fn.argtypes) - 1 <= PG_LAUNCH_MAX_ARGS, fn.__name__
    for i, (a, ty) in enumerate(zip(args, fn.argtypes)):
        if a is None:
            v = 0
        elif ty is C.c_float:
            v = int.from_bytes(struct.pack('<f', a.value if isinstance(a, C.c_float) else float(a)), 'little')
        elif hasattr(a, '_obj'):                    # byref(struct): the struct's address (the engine keeps the struct alive)
            v = C.addressof(a._obj)
        elif isinstance(a, C._SimpleCData):
            v = a.value or 0
        else:
            v = int(a)
        L.a[i] = v & 0xFFFFFFFFFFFFFFFF
    return L


class Program:
    """A launch list inside the library (pg_program_*): one foreign call per run."""
    __slots__ = ('h', '_lib', 'n')

    def __init__(self, records, n_events):
        self._lib = lib()
        arr = (PgLaunch * max(len(records), 1))(*records)
        self.h = C.c_void_p()
        check(self._lib.pg_program_create(arr, len(records), n_events, C.byref(self.h)), 'pg_program_create')
        self.n = len(records)

    def run(self, stream_ptrs):
        if self._lib.pg_program_run(self.h, stream_ptrs):
            check(1, 'pg_program_run')

    def __del__(self):
        try:
            if self.h and not sys.is_finalizing():
                self._lib.pg_program_destroy(self.h)
                self.h = None
        except Exception:
            pass


def check(rc, what=''):
    if rc != 0:
        msg = load_library().pg_last_error().decode()
        raise RuntimeError(f'phoregen_hip {what} failed ({rc}): {msg}')


def ptr(t):
    """Device pointer of a tensor (or None)."""
    if t is None:
        return None
    return t.data_ptr()


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream
