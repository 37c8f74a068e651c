"""Variant switches of the engine and of the training path.

The values in `DEFAULTS` are the product.  They change in two ways only: `options.override(...)` (a context manager; the tests and
the measurement tools use it), or -- with `PHOREGEN_DEBUG=1` in the environment, for profiling a stock `bench.py` run from a shell
script -- through the `PG_*` variables listed in `_ENV`.  Without `PHOREGEN_DEBUG=1` the ambient environment cannot change which
kernels a model runs.  None of the variants routes around the HIP library.
"""
import contextlib
import os

DEFAULTS = dict(
    streams=True,         # four lanes (HIP streams) per layer; False: one stream
    tri_staged=True,      # csrc/triplet2.hip (False: the gather kernel, triplet.hip)
    node_fused=True,      # node attention folds the query / unfolds the value in-kernel
    knn_group=True,       # neighbour slots partitioned by source kind
    knn_merge='auto',     # ligand + pharmacophore targets of a knn sub-layer in one launch ('auto' = 'always' since round 4), 'never' = two launches
    layer_ahead=True,     # small batches: the next layer's x-independent products inside this layer's position phase
    ahead_v2='auto',      # next layer's Y1 on the node chain's lane behind Y2, P waits for it alone, layer 0's bond-node attention on lane 3:
                          # 'auto' = small batches (8 / 16 / 32 graphs - 3 % / - 3.5 % / - 1 %; 64 / 128 graphs + 0.4 % / + 1.7 %), 'never', 'always'
    ahead_v2_below=82000, # ... 'auto': fewer bond edges than this (~55 graphs of the headline shape: 48 graphs 7.75 -> 7.63 ms, 64 graphs 9.97 -> 10.04)
    pos_tiled='auto',     # position-update attention with a node's row tiles over several waves: 'auto' = launches of few nodes, 'never', 'always'
    pos_tiled_below=1500, # ... 'auto': up to this many target nodes (32 graphs 5.29 -> 5.21 ms; at 64 graphs = 2 560 nodes it loses)
    step_ahead=True,      # the sampler loop as a software pipeline over reverse steps: the categorical posteriors behind their heads on the side
                          # lanes, the NEXT step's feature embedding and layer 0's coordinate-free products (first-layer blocks, queries, bond-node
                          # sub-layer) behind them, beside the last layer's position phase / the Gaussian posterior (`Engine.pipelined_programs`)
    sa_head_early=True,   # ... the node head of a pipelined step right behind the last lin_node (it needs h' only), not behind the triplet kernel
    c_program=True,       # a forward = ONE call into the library (pg_program_run walks the launch list); False: the list is walked from Python
    order_points=True,    # cross-lane order points as device-scope HIP events (pg_order_point_*); False: torch.cuda.Event()
    geom_split='auto',    # ahead_v2: the layer's closing launch (pg_layer_geom) once per chain, on the chain's own lane: 'auto' = batches
                          # whose node chain is the longer one (4 / 8 graphs 2.29 -> 2.22 / 2.38 -> 2.30 ms; 12 graphs equal; 16 / 32 graphs + 1 %)
    geom_split_below=16000,   # ... 'auto': fewer bond edges than this
    tune_grid=True,       # small batches: time the neighbouring triplet grids (multiples of 32 workgroups) during the first forwards, keep the fastest
    chain_q_from=150000,  # the Q rows of the triplet MLPs behind P on lane 0 from this many bond edges up (below: beside P on lane 2)
    tri_split=True,       # the triplet kernel as two launches when a few ligands need more row tiles than the rest (BatchPlan.tri_split):
                          # True = from `tri_split_from` bond edges up, 'always', False
    tri_split_from=82000,
    tri_overlap=3,        # the side lane the larger ligands' launch runs on, BESIDE the other one (disjoint ligands; both queues drain into the
                          # same workgroup slots, one tail instead of two); 0 = behind it on lane 0; < 0: that lane, launched second
    tri_grid=-1,          # persistent workgroups of the staged triplet kernel (-1: by batch size)
    graph=False,          # hipGraph replay of the forward launch list
    fused_geom='auto',    # coordinate update + bond smearing + direction vectors as one launch on the bond chain's lane (pg_layer_geom):
                          # 'auto' = 'always' (round 4: it pays at every batch size), 'never' = three launches on three lanes
    dgrad_mm=True,        # training: input gradients through the library GEMM
    rows_sum=True,        # training: pg_bond_rows_sum instead of atomic index_add_
    tri_onepass=True,     # training: one-pass triplet / node adjoints fed by the forward's softmax weights
    wide_gemm=True,       # training: one wide first-layer GEMM per layer (ColumnBlocksFn)
    bwd_atom_sort=True,   # training: the triplet adjoint takes its source atoms cost-sorted (PgSegAttnGrad.atom_order)
    bwd_split='all',      # training: the triplet, knn-node and knn-position adjoints as a value pass + a key pass (one MLP path per wave:
                          # no spills at 512 registers; PgSegAttnGrad.dlogit): 'all', 'knn' = the knn adjoints only, 'none' = both paths in one wave
    bwd_grid=256,         # training: persistent workgroups of pg_seg_attn_bwd (one per CU)
    tri_bwd_form=2,       # training: triplet adjoint 0 = one wave per row tile (csrc/seg_attn_bwd.hip); the channels of a tile over the waves of
                          # a workgroup (csrc/triplet_bwd2.hip; ligands of up to 64 atoms): 1 = 4 waves, 2 = 8 waves for ligands of up to
                          # 32 atoms + the 4-wave form for the larger ones
    tri_bwd_grid=256,     # ... persistent workgroups of the channel-split form
    ph_onepass=True,      # training: the pharmacophore encoder's forward leaves its softmax weights for a one-pass adjoint instead of the generic
                          # form that recomputes both MLP paths twice
)

_tri = lambda v: {'0': 'never', '1': 'auto', '2': 'always'}[v]
_flag = lambda v: v != '0'
_ENV = {
    'PG_STREAMS': ('streams', _flag), 'PG_TRI_STAGED': ('tri_staged', _flag),
    'PG_NODE_FUSED': ('node_fused', _flag), 'PG_KNN_GROUP': ('knn_group', _flag), 'PG_KNN_MERGE': ('knn_merge', _tri),
    'PG_LAYER_AHEAD': ('layer_ahead', _flag), 'PG_AHEAD_V2': ('ahead_v2', _tri), 'PG_AHEAD_V2_BELOW': ('ahead_v2_below', int), 'PG_TRI_GRID': ('tri_grid', int), 'PG_POS_TILED': ('pos_tiled', _tri), 'PG_POS_TILED_BELOW': ('pos_tiled_below', int), 'PG_GRAPH': ('graph', _flag), 'PG_ORDER_POINTS': ('order_points', _flag), 'PG_C_PROGRAM': ('c_program', _flag), 'PG_STEP_AHEAD': ('step_ahead', _flag), 'PG_CHAIN_Q_FROM': ('chain_q_from', int), 'PG_TRI_SPLIT': ('tri_split', _flag), 'PG_TUNE_GRID': ('tune_grid', _flag), 'PG_GEOM_SPLIT': ('geom_split', _tri),
    'PG_FUSED_GEOM': ('fused_geom', _tri), 'PG_DGRAD_MM': ('dgrad_mm', _flag),
    'PG_ROWS_SUM': ('rows_sum', _flag), 'PG_TRI_ONEPASS': ('tri_onepass', _flag), 'PG_WIDE_GEMM': ('wide_gemm', _flag), 'PG_BWD_GRID': ('bwd_grid', int), 'PG_BWD_SPLIT': ('bwd_split', lambda v: {'0': 'none', '1': 'knn', '2': 'all'}[v]), 'PG_BWD_ATOM_SORT': ('bwd_atom_sort', _flag),
    'PG_TRI_BWD_FORM': ('tri_bwd_form', int), 'PG_TRI_BWD_GRID': ('tri_bwd_grid', int), 'PG_PH_ONEPASS': ('ph_onepass', _flag),
}
_overrides = {}


def get(name):
    if name in _overrides:
        return _overrides[name]
    if os.environ.get('PHOREGEN_DEBUG') == '1':
        for var, (key, conv) in _ENV.items():
            if key == name and var in os.environ:
                return conv(os.environ[var])
    return DEFAULTS[name]


def snapshot():
    return {k: get(k) for k in DEFAULTS}


@contextlib.contextmanager
def override(**kw):
    """Run a block with other variants (an Engine reads them when it is built, the training path per call)."""
    unknown = set(kw) - set(DEFAULTS)
    if unknown:
        raise KeyError(f'phoregen_amd.options: unknown switch(es) {sorted(unknown)}')
    old = dict(_overrides)
    _overrides.update(kw)
    try:
        yield
    finally:
        _overrides.clear()
        _overrides.update(old)
