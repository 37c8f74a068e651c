"""Launch sequence of the denoiser on one batch plan.

`Engine` owns the persistent workspace of a (weights, plan) pair and a pre-built launch list: every
C-ABI argument struct is created once, so a denoise step is ~200 ctypes calls with no allocation, no
host synchronisation and no Python tensor arithmetic.  Stream = torch's current stream at run time.

Data flow of one layer (reference: AttentionLayerO2TwoUpdateNodeGeneral.forward, uni_denoiser.py:260-298):
  Y1   = h @ W_node1^T            15 first-layer blocks (knn-node, bond-node, triplet; k/v dst+src halves, q)
  q*   = LN/ReLU/W2 of the q blocks (pre-scaled by 1/sqrt(8)),  U = fold(q, W2k)
  Cs*  = h_bond @ W_hb^T + gathered node halves                (bond-node, triplet P, triplet q)
  aggE = unfold(seg_attn<KNN_NODE>),  aggB = unfold(seg_attn<BOND_NODE>),  h_bond' = h_bond + seg_attn<TRIPLET>
  h'   = h + lin_node(aggE + aggB)
  Y2   = h' @ W_node2^T; same for the two position sub-layers;  x' = x + mask * (dx_knn + dx_bond)
"""
import ctypes as C

import torch

from . import hip, options
from .packing import HEAD_SCALE, ModelPack


_SIDE_STREAMS = {}


_TRI_GRID_CACHE = {}       # batch shape -> (persistent triplet grid chosen by Engine.calibrate_tri_grid, its timings)


def side_streams(device_index, stream_set=0):
    """The three side lanes of a device, shared by every Engine of the process: the runtime maps HIP streams onto a handful of
    hardware queues (4 by default), and streams that share a queue serialise -- an Engine that created its own streams made the
    lanes of every later Engine slower (measured: a second model's 16-graph step 3.4 -> 5.4 ms).  tools/micro/hw_queues.py
    (profiles/r04_micro_hw_queues.txt): exactly four streams of a process run side by side (the null stream + three); a fifth
    shares a queue and waits for its partner unless GPU_MAX_HW_QUEUES is raised BEFORE the runtime starts.  Four lanes = lane 0
    (the caller's stream) + these three is therefore the most this design can use without asking the host program for more."""
    key = (device_index, stream_set)
    if key not in _SIDE_STREAMS:
        with torch.cuda.device(device_index):
            _SIDE_STREAMS[key] = [torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()]
    return _SIDE_STREAMS[key]


def _f(*shape, device, zero=False):
    return (torch.zeros if zero else torch.empty)(*shape, dtype=torch.float32, device=device)


class _TorchPoint:
    """torch.cuda.Event behind the two calls of hip.OrderPoint."""
    __slots__ = ('ev',)

    def __init__(self):
        self.ev = torch.cuda.Event()

    def record(self, stream):
        self.ev.record(stream)

    def wait(self, stream):
        stream.wait_event(self.ev)


class _Tap:
    """A launch-list entry that is not a kernel launch: 'order' (records / waits of order points between the lanes), 'event' (a timing
    event when Engine.timers is set) or 'mark' (clones of tensors when Engine.debug is set)."""
    __slots__ = ('kind', '__name__', 'ops', 'start', 'lane', 'tensors')

    def __init__(self, kind, name, ops=None, start=None, lane=0, tensors=()):
        self.kind, self.__name__, self.ops, self.start, self.lane, self.tensors = kind, name, ops, start, lane, tensors

    @property
    def name(self):
        return self.__name__


class Engine:
    def __init__(self, pack: ModelPack, plan, knn_k=32, full=True, phore_only=False):
        self.lib = hip.lib()
        self.pack, self.plan, self.k = pack, plan, knn_k
        self.dev = plan.device
        self._keep = []          # ctypes structs / tensors referenced by raw pointer
        self.debug = None
        self.timers = None
        self.trace = None        # list: when set, `_run` brackets every launch with HIP events on its lane (tools/event_timeline.py)
        self.tri_calls = []      # indices of the triplet launches in the forward program (profiling)
        # independent sub-layer chains of a layer run on side streams ("lanes" 1, 2) next to the triplet chain (lane 0):
        # kernels of different chains interleave on the CUs, so one chain's load/store phases meet another's MFMA phases.
        # The variants below are read from phoregen_amd.options when an Engine is built (defaults = the product; tests and tools
        # switch them with options.override(...), the ambient environment only counts under PHOREGEN_DEBUG=1)
        o = options.snapshot()
        self.multi_stream = o['streams']
        self.staged_triplet = o['tri_staged']          # csrc/triplet2.hip (False: the gather kernel)
        self.fused_node = o['node_fused']              # node attention folds / unfolds in-kernel
        self.group_knn = o['knn_group']                # neighbour slots partitioned by source kind
        self.merge_knn_lists = o['knn_merge'] != 'never'      # ligand + pharmacophore targets of a knn sub-layer in one launch
        self.merge_knn_always = o['knn_merge'] == 'always'    # ... also below the batch size where it pays (tests)
        # the next layer's first-layer blocks, triplet queries and bond-node rows (everything in front of its triplet kernel that does
        # not depend on the new coordinates) run on lane 2 during this layer's position updates
        self.layer_ahead = o['layer_ahead']
        # cross-lane order points without the host-visibility fence of a default event (hip.OrderPoint; tools/micro/stream_packets.py)
        self.order_points = o['order_points'] and not o['graph']
        self.tri_split, self.tri_split_from = o['tri_split'], o['tri_split_from']
        self.tri_overlap = o['tri_overlap']
        self.chain_q_from = o['chain_q_from']
        self.tune_grid = o['tune_grid']
        self.tuned_tri_grid = None
        self.geom_split = o['geom_split'] == 'always' or (o['geom_split'] == 'auto' and self.plan.n_bond < o['geom_split_below'])
        # (hipGraph capture of the v2 launch list segfaults inside the runtime -- lanes 2 / 3 wait on lane 1 there while lane 1 waits on
        #  lane 3; the replay variant, off by default and measured slower, keeps the previous list)
        self.ahead_v2 = (o['ahead_v2'] == 'always' or (o['ahead_v2'] == 'auto' and self.plan.n_bond < o['ahead_v2_below'])) and not o['graph']
        self.step_ahead = o['step_ahead']
        self.head_early = o['sa_head_early']
        self.pos_tiled, self.pos_tiled_below = o['pos_tiled'], o['pos_tiled_below']
        self.tri_grid = o['tri_grid']                  # persistent triplet workgroups (-1: by batch size)
        # apply_dx + bond smearing + direction vectors as one launch on lane 0: small batches (measured, same box: 16 graphs 3.52 -> 3.40 ms
        # per step, 32 graphs 5.69 -> 5.43; 128 graphs 20.20 -> 20.28: there the three launches hide beside the first-layer GEMM)
        self.fused_geom = o['fused_geom'] in ('always', 'auto')
        # hipGraph replay of the forward launch list.  Off by default: measured on MI355X it buys nothing, a step
        # is bound by the ~225 dependent kernels themselves, not by their launches (tools/bench_graph.py: B=1 3.19 -> 2.95,
        # B=10 3.89 -> 4.07, B=30 5.33 -> 5.92 ms/step; identical results)
        self.graph_mode = '1' if o['graph'] else '0'
        self._graph = None
        self._lane = 0
        self._n_points = 0       # order points of this engine's launch lists
        self._points = {}        # ... their events in the Python runner (created at first use)
        # the launch lists run inside the library (pg_program_run: ONE foreign call per forward) unless a debug / timing / trace tap is
        # active, the order points are torch events, or the list is being captured into a hipGraph
        self.c_program = o['c_program'] and (self.order_points or not self.multi_stream)
        self._compiled = {}
        self._side = None
        self.stream_set = 0      # (experiments: a second set of side lanes)
        self._alloc()
        if full:
            self.prog_phore = self._build_phore_program()
            if not phore_only:                       # (sample_nodes: pharmacophore encoder + count heads only)
                self.prog_fwd = self._build_forward_program()
                self.prog_step = self.prog_ahead = None      # the pipelined sampler step (built at first use: `pipelined_programs`)

    # ------------------------------------------------------------------ workspace
    def _alloc(self):
        p, d = self.plan, self.dev
        n, E, k = p.n_ctx, p.n_bond, self.k
        w = self.ws = type('WS', (), {})()
        w.in_h_node, w.in_pos = _f(p.n_lig, 12, device=d), _f(p.n_lig, 3, device=d)
        w.in_h_edge = _f(E, 6, device=d)
        w.in_t = torch.zeros(p.n_graphs, dtype=torch.int64, device=d)
        w.in_t_next = torch.zeros(p.n_graphs, dtype=torch.int64, device=d)      # t of the NEXT reverse step (its features are embedded one step ahead)
        w.h_phore_ctx, w.x_phore_ctx = _f(n, 18, device=d, zero=True), _f(n, 3, device=d, zero=True)
        w.pos_phore, w.phore_norm = _f(p.n_phore, 3, device=d), _f(p.n_phore, 3, device=d)
        w.nrm_phore_ctx = _f(n, 3, device=d, zero=True)          # phore_norm in ctx row order (pg_layer_geom)
        w.is_ex = torch.zeros(p.n_phore, dtype=torch.uint8, device=d)
        w.hp_ctx, w.Yp = _f(n, 128, device=d), _f(n, 640, device=d)
        w.enc_ctx = _f(n, 128, device=d, zero=True)
        w.hp_emb = _f(p.n_phore, 128, device=d)
        w.cnt_hid = _f(n, 256, device=d)
        w.s_all, w.s_l = _f(p.n_phore, device=d), _f(p.n_phore, device=d)
        w.count_l, w.count_u = _f(p.n_graphs, device=d), _f(p.n_graphs, device=d)
        w.h = [_f(n, 128, device=d), _f(n, 128, device=d)]
        # the embedded features = layer 0's input, in a buffer of their own: the layers ping-pong h[1] / h[0] behind it, so the NEXT step's embedding
        # (`prog_ahead`, lane 2 of the pipelined loop) never overwrites what the current step's last layer still reads (round 5's advisor finding:
        # with the embedding in h[0] the position phase's first-layer products on lane 1 raced it; no order point is needed this way)
        w.h_in = _f(n, 128, device=d)
        w.x = [_f(n, 3, device=d), _f(n, 3, device=d)]
        w.hb = [_f(E, 128, device=d), _f(E, 128, device=d)]
        w.nbr = torch.zeros(n, k, dtype=torch.int32, device=d)
        w.deg = torch.zeros(n, dtype=torch.int32, device=d)
        w.ew, w.nrm, w.G = _f(n, k, device=d), _f(n, 3, device=d), _f(E, 20, device=d)
        w.Y1, w.Y2 = _f(n, 1920, device=d), _f(n, 1280, device=d)
        w.Y1b = None                      # second first-layer buffer of the `ahead_v2` schedule (allocated by the program builder)
        # per concurrent chain: 0 knn, 1 bond, 2 the bond-node sub-layer launched one layer ahead, 3 (q only) the knn-node query ahead
        w.q = [_f(n, 128, device=d) for _ in range(4)]
        w.U, w.S = [_f(n, 2048, device=d) for _ in range(3)], [_f(n, 2048, device=d) for _ in range(3)]
        w.swn = [_f(n, 16, device=d) for _ in range(3)]
        w.aggE, w.aggB = _f(n, 128, device=d, zero=True), _f(n, 128, device=d, zero=True)
        w.lin_tmp = _f(n, 128, device=d)
        w.CsB, w.P = _f(E, 256, device=d), _f(E, 256, device=d)     # (CsB: bond-pos edge rows)
        w.CsB2 = _f(E, 256, device=d)                                  # bond-node edge rows (written one layer ahead, see _denoiser_program)
        w.Qd = _f(E, 256, device=d)                                    # triplet: per-segment constant smear(d_ji) . Wg2 (k | v)
        w.qhid, w.qT = _f(E, 128, device=d), _f(E, 128, device=d)
        w.dxe, w.dxb = _f(n, 3, device=d, zero=True), _f(n, 3, device=d, zero=True)
        w.head = _f(n, 128, device=d)
        w.head_b = _f(E, 128, device=d)
        w.out_v, w.out_bond = _f(p.n_lig, 12, device=d), _f(E, 6, device=d)

    # ------------------------------------------------------------------ call builders
    def _call(self, prog, fn, *args):
        assert len(args) + 1 == len(fn.argtypes), (fn.__name__, len(args) + 1, len(fn.argtypes))
        prog.append((fn, args, self._lane if self.multi_stream else 0))

    def _point(self):
        """A new cross-lane order point: the library's device-scope event, or a torch event (option `order_points` off; hipGraph capture)."""
        if self.order_points:
            return hip.OrderPoint()
        return _TorchPoint()

    # Entries of a launch list that are not kernel launches are `_Tap` records (plain data: a launch list holds no reference to its
    # Engine, so an Engine is freed by reference counting alone).  Order points are numbered per Engine (`_n_points`); the Python
    # runner creates the event of a point at its first record and re-records it every step (a wait refers to the record that precedes
    # it, so re-use across steps is safe), the C runner (pg_program_*) owns one event per point.
    def _new_point(self):
        self._n_points += 1
        return self._n_points - 1

    def _order(self, prog, name, ops):
        if self.multi_stream and ops:
            prog.append((_Tap('order', name, ops=ops), None, -1))

    def _fork(self, prog, lanes):
        """Side lanes start after everything enqueued so far on lane 0."""
        if not self.multi_stream:
            return
        pt = self._new_point()
        self._order(prog, 'fork', [('record', pt, 0)] + [('wait', pt, l) for l in lanes])

    def _join(self, prog, lanes):
        """Lane 0 continues after the side lanes have drained."""
        self._sync(prog, 0, lanes, name='join')

    def _sync(self, prog, waiter, on, name='sync'):
        """Lane `waiter` continues after everything enqueued SO FAR on the lanes `on` (a join of a point, not of the whole lane)."""
        if not self.multi_stream:
            return
        ops = []
        for l in on:
            pt = self._new_point()
            ops += [('record', pt, l), ('wait', pt, waiter)]
        self._order(prog, name, ops)

    def _record(self, prog, lane):
        """Mark this point of `lane`; `_wait` lets another lane continue after it (a `_sync` whose wait is placed later in the list)."""
        if not self.multi_stream:
            return None
        pt = self._new_point()
        self._order(prog, 'record', [('record', pt, lane)])
        return pt

    def _wait(self, prog, waiter, pt):
        if not self.multi_stream:
            return
        self._order(prog, 'wait', [('wait', pt, waiter)])

    def _event(self, prog, name, start):
        """Timing tap: when `self.timers` is a dict, record a HIP event around a kernel -- on the lane (stream) the kernel is
        launched on, i.e. the lane that is current while the launch list is built."""
        prog.append((_Tap('event', name, start=start, lane=self._lane if self.multi_stream else 0), None, -1))

    def _mark(self, prog, name, *tensors, lane=0):
        """Debug tap: when `self.debug` is a dict, clone the named tensors at this point of the launch list (of `lane`)."""
        prog.append((_Tap('mark', name, tensors=tensors, lane=lane if self.multi_stream else 0), None, -1))

    def _tap(self, tap, streams):
        """The Python runner's side of a `_Tap`."""
        if tap.kind == 'order':
            for what, pt, lane in tap.ops:
                ev = self._points.get(pt)
                if ev is None:
                    ev = self._points[pt] = self._point()
                if what == 'record':
                    ev.record(streams[lane])
                else:
                    ev.wait(streams[lane])
        elif tap.kind == 'event':
            if self.timers is not None:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record(streams[tap.lane])
                self.timers.setdefault(tap.name, []).append((tap.start, ev))
        elif self.debug is not None:                 # 'mark'
            with torch.cuda.stream(streams[tap.lane]):
                self.debug[tap.name] = tuple(t.clone() for t in tap.tensors)

    def kernel_ms(self, name):
        """Per-launch durations (ms) collected since `self.timers = {}`; call after a device synchronize."""
        evs = self.timers.get(name, [])
        return [a.elapsed_time(b) for (sa, a), (sb, b) in zip(evs[0::2], evs[1::2]) if sa and not sb]

    def _gemm(self, prog, X, K1, W, Y, M, N, bias=None, X2=None, K2=0, ln=None, add1=None, idx1=None, add2=None,
              idx2=None, scale=1.0, act=hip.ACT_NONE, rows=None):
        g = hip.PgGemm()
        g.X, g.ldx, g.K1 = X.data_ptr(), X.stride(0), K1
        g.X2, g.ldx2, g.K2 = (X2.data_ptr(), X2.stride(0), K2) if X2 is not None else (None, 0, 0)
        g.W, g.ldw = W.data_ptr(), W.stride(0)
        g.bias = hip.ptr(bias)
        g.ln_gamma, g.ln_beta = (ln[0].data_ptr(), ln[1].data_ptr()) if ln is not None else (None, None)
        g.add1, g.ld_add1, g.idx1 = (add1.data_ptr(), add1.stride(0), hip.ptr(idx1)) if add1 is not None else (None, 0, None)
        g.add2, g.ld_add2, g.idx2 = (add2.data_ptr(), add2.stride(0), hip.ptr(idx2)) if add2 is not None else (None, 0, None)
        g.out_scale, g.act = scale, act
        g.Y, g.ldy, g.M, g.N = Y.data_ptr(), Y.stride(0), M, N
        g.rows = hip.ptr(rows)
        g.add_rows = add1.size(0) if add1 is not None else 0      # the gathered operands are whole [rows, ld] tensors
        self._keep += [g, X, W, Y, bias, X2, ln, add1, idx1, add2, idx2, rows]
        self._call(prog, self.lib.pg_gemm, C.byref(g))

    def _seg(self, prog, mode, n_seg, seg_ids, a, **kw):
        s = hip.PgSegAttn()
        s.mode, s.n_seg, s.seg_ids = mode, n_seg, hip.ptr(seg_ids)
        s.knn_k = self.k
        s.ln_gk, s.ln_bk, s.ln_gv, s.ln_bv = (t.data_ptr() for t in (a.ln_gk, a.ln_bk, a.ln_gv, a.ln_bv))
        for key, val in kw.items():
            setattr(s, key, val.data_ptr() if torch.is_tensor(val) else val)
            self._keep.append(val)
        self._keep += [s, seg_ids, a]
        self._call(prog, self.lib.pg_seg_attn, self.plan.topo_ref, C.byref(s))

    def _query_gemm(self, prog, a, Y, col0, h_dst_lists, buf):
        """q = W2q . ReLU(LN(q_hid)) + b2q, scaled by 1/sqrt(head_dim); q_hid = block 4 of the sub-layer's first-layer columns."""
        n, wq = self.plan.n_ctx, self.ws.q[buf]
        qh = Y[:, col0 + 4 * 128: col0 + 5 * 128]
        self._gemm(prog, qh, 128, a.W2q, wq, n, 128, bias=a.b2q, ln=(a.q_ln_g, a.q_ln_b), scale=HEAD_SCALE)

    def _node_attention(self, prog, mode, a, Y, col0, x, h_dst_lists, out=None, dx=None, csrc=None, buf=0, extra=None,
                        query_done=False, qbuf=None):
        """Shared tail of the four node-target sub-layers.  Y[:, col0 + 128*b] blocks: k_dst, v_dst, k_src, v_src, q_hid."""
        w, p, n = self.ws, self.plan, self.plan.n_ctx
        wq, wU, wS, wsw = w.q[buf if qbuf is None else qbuf], w.U[buf], w.S[buf], w.swn[buf]
        blk = lambda b: Y[:, col0 + 128 * b: col0 + 128 * (b + 1)]
        if not query_done:
            self._query_gemm(prog, a, Y, col0, h_dst_lists, buf)
        knn = mode in (hip.SEG_KNN_NODE, hip.SEG_KNN_POS)
        pos = mode in (hip.SEG_KNN_POS, hip.SEG_BOND_POS)
        fused = self.fused_node and mode != hip.SEG_PHORE      # in-kernel query fold / value unfold (csrc/node_attn.hip)
        # fused knn form with two target lists (ligand + pharmacophore targets, different feature weights): ONE launch, the
        # persistent workgroups split between the lists in proportion to their sizes (PgSegAttn.seg_ids2)
        # (at every size since round 4: with the node chain on 32 ... 96 CUs beside the triplet kernel, two launches of 50 + 85 workgroups
        #  take three rounds where the merged one takes two -- 8 graphs 2.29 -> 2.04 ms per step, 16 graphs 3.05 -> 3.02, 24+ as before)
        merged = fused and knn and len(h_dst_lists) == 2 and self.merge_knn_lists and all(n > 0 for _, n, _ in h_dst_lists)
        lists = [h_dst_lists[0]] if merged else h_dst_lists
        for seg_ids, n_seg, is_lig in lists:
            if not fused:
                self._call(prog, self.lib.pg_attn_fold_query, wq.data_ptr(), 128, a.W2k_l.data_ptr(), n_seg,
                           seg_ids.data_ptr(), wU.data_ptr())
            kw = dict(x=x, Cdst_k=blk(0), Cdst_v=blk(1), ld_cdst=Y.stride(0), U=wU)
            if knn:
                kw.update(nrm=w.nrm, nbr=w.nbr, deg=w.deg, ew=w.ew, Csrc_k=blk(2), Csrc_v=blk(3), ld_csrc=Y.stride(0),
                          Wf_k=a.Wf_k[is_lig], Wf_v=a.Wf_v[is_lig])
                if merged:
                    ids2, n2, lig2 = h_dst_lists[1]
                    kw.update(seg_ids2=ids2, n_seg2=n2, Wf_k2=a.Wf_k[lig2], Wf_v2=a.Wf_v[lig2])
            elif mode == hip.SEG_PHORE:
                kw.update(Csrc_k=blk(2), Csrc_v=blk(3), ld_csrc=Y.stride(0), Wf_k=a.Wf_k, Wf_v=a.Wf_v)
            else:
                kw.update(Csrc_k=csrc[:, 0:128], Csrc_v=csrc[:, 128:256], ld_csrc=csrc.stride(0))
            if pos:
                kw.update(W2xv_l=a.W2xv_l, b2xv=a.b2xv, dx=dx, accumulate_dx=0)
                # small batches: a node's row tiles over several waves (csrc/node_attn.hip, bit-identical): the launch is bound by the
                # dependent chain inside the wave that owns a node, not by the chip
                tiled = self.pos_tiled == 'always' or (self.pos_tiled == 'auto' and n_seg <= self.pos_tiled_below)
                if fused and tiled:
                    kw.update(pos_tiled=1)
            else:
                kw.update(S=wS, swn=wsw)
            if fused:      # (U / S / swn stay attached as scratch for the one-pass fallback inside pg_seg_attn)
                kw.update(q=wq, W2k_l=a.W2k_l)
                if not pos:
                    assert out.stride(0) == 128
                    kw.update(W2v_l=a.W2v_l, b2v=a.b2v, out=out)
            if extra:
                kw.update(extra)
            self._seg(prog, mode, n_seg, seg_ids, a, **kw)
            if not pos and not fused:
                self._call(prog, self.lib.pg_attn_unfold_value, wS.data_ptr(), wsw.data_ptr(), a.W2v_l.data_ptr(),
                           a.b2v.data_ptr(), n_seg, seg_ids.data_ptr(), out.data_ptr(), out.stride(0))

    # ------------------------------------------------------------------ phore encoder + count heads (per plan)
    def _build_phore_program(self):
        w, p, pk, prog = self.ws, self.plan, self.pack, []
        n = p.n_ctx
        # diffusion.py:186: phore_embedding on ctx-ordered rows (ligand rows are zero and unused)
        self._gemm(prog, w.h_phore_ctx, 18, pk.W_pe, w.hp_ctx, n, 128, bias=pk.b_pe)
        self._gemm(prog, w.hp_ctx, 128, pk.W_ph, w.Yp, n, 640, bias=pk.b_ph)
        self._node_attention(prog, hip.SEG_PHORE, pk.PH, w.Yp, 0, w.x_phore_ctx,
                             [(p.phore2ctx, p.n_phore, False)], out=w.enc_ctx)
        # diffusion.py:148-163 count heads on the encoded pharmacophore
        for (W0, b0, W2, b2), dst in zip(pk.cnt, (w.s_all, w.s_l)):
            self._gemm(prog, w.enc_ctx, 128, W0, w.cnt_hid, n, 256, bias=b0, act=hip.ACT_RELU)
            self._call(prog, self.lib.pg_rows_linear, w.cnt_hid.data_ptr(), 256, 256, W2.data_ptr(), b2.data_ptr(), 1,
                       p.n_phore, p.phore2ctx.data_ptr(), dst.data_ptr(), 1)
        self._call(prog, self.lib.pg_atom_count, w.s_all.data_ptr(), w.s_l.data_ptr(), w.is_ex.data_ptr(),
                   p.phore_graph.data_ptr(), p.n_phore, p.n_graphs, w.count_l.data_ptr(), w.count_u.data_ptr())
        return prog

    def encode_phore(self, h_phore, pos_phore, phore_norm, ex_col=12):
        """Time-independent part of PhoreDiff.forward (diffusion.py:186-191,244): run once per batch."""
        w, p = self.ws, self.plan
        w.h_phore_ctx.index_copy_(0, p.phore2ctx_long, h_phore.float())
        w.x_phore_ctx.index_copy_(0, p.phore2ctx_long, pos_phore.float())
        w.pos_phore.copy_(pos_phore)
        w.phore_norm.copy_(phore_norm)
        w.nrm_phore_ctx.index_copy_(0, p.phore2ctx_long, phore_norm.float())
        w.is_ex.copy_((h_phore[:, ex_col] == 1).to(torch.uint8))
        self._run(self.prog_phore)
        torch.index_select(w.enc_ctx, 0, p.phore2ctx_long, out=w.hp_emb)

    # ------------------------------------------------------------------ one denoiser forward
    def _embed_bond(self, prog, in_t):
        w, p, pk = self.ws, self.plan, self.pack
        self._call(prog, self.lib.pg_embed_bond, p.topo_ref, w.in_h_edge.data_ptr(), p.bond_graph.data_ptr(), in_t.data_ptr(),
                   pk.W_edge_emb.data_ptr(), pk.t_off.data_ptr(), pk.t_coeff.data_ptr(), w.hb[0].data_ptr())

    def _embed_ctx(self, prog, in_t, features=True, coords=True):
        w, p, pk = self.ws, self.plan, self.pack
        self._call(prog, self.lib.pg_embed_ctx, p.topo_ref, w.in_h_node.data_ptr(), w.in_pos.data_ptr(), in_t.data_ptr(),
                   pk.W_node_emb.data_ptr(), pk.t_off.data_ptr(), pk.t_coeff.data_ptr(), w.hp_emb.data_ptr(),
                   w.pos_phore.data_ptr(), p.phore2ctx.data_ptr(), w.h_in.data_ptr() if features else None,
                   w.x[0].data_ptr() if coords else None)

    def _build_forward_program(self, step_ahead=False):
        """One denoiser forward.  step_ahead (the pipelined sampler step, `pipelined_programs`): the features of this step were embedded,
        and layer 0's coordinate-free products launched, by `prog_ahead` at the end of the PREVIOUS step, whose Gaussian posterior wrote the
        new coordinates straight into x[0] (pg_posterior_position_ctx): nothing is embedded here, layer 0 runs like every later layer, and
        the program ends without joining lanes 2 / 3 (the categorical posteriors and the next step's `prog_ahead` follow on them)."""
        w, p, pk, prog, lib = self.ws, self.plan, self.pack, [], self.lib
        n, E, t = p.n_ctx, p.n_bond, p.topo_ref
        lig = [(p.lig2ctx, p.n_lig, True)]
        both = [(p.lig2ctx, p.n_lig, True), (p.phore2ctx, p.n_phore, False)]
        h, x, hb = w.h, w.x, w.hb
        if step_ahead:
            self._lane = 0          # (the coordinates are in x[0] already: the Gaussian posterior of the previous step wrote them there)
        else:
            # the bond embedding runs beside the node embedding, the knn search and the edge gate (joined before layer 0 forks)
            self._fork(prog, (2,))
            self._lane = 2
            self._embed_bond(prog, w.in_t)
            self._lane = 0
            self._embed_ctx(prog, w.in_t)
        # heads (diffusion.py:221-241).  Neither waits for the last layer's position update: the bond head starts on a side lane as
        # soon as the last triplet kernel has written h_bond, the node head as soon as the last lin_node has written h
        def bond_head(hb_final):
            W0, b0, W2, b2 = pk.b0
            self._gemm(prog, hb_final, 128, W0, w.head_b, E, 128, bias=b0, act=hip.ACT_SSP)
            # out_bond in the caller's edge order: row r reads the internal row of caller edge r
            self._call(prog, lib.pg_rows_linear, w.head_b.data_ptr(), 128, 128, W2.data_ptr(), b2.data_ptr(), 6, E,
                       None if p.edge_identity else p.edge_int.data_ptr(), w.out_bond.data_ptr(), 6)

        def node_head(h_final):
            W0, b0, W2, b2 = pk.v0
            self._gemm(prog, h_final, 128, W0, w.head, n, 128, bias=b0, act=hip.ACT_SSP)
            self._call(prog, lib.pg_rows_linear, w.head.data_ptr(), 128, 128, W2.data_ptr(), b2.data_ptr(), 12, p.n_lig,
                       p.lig2ctx.data_ptr(), w.out_v.data_ptr(), 12)
        self._denoiser_program(prog, lig, both, heads=(bond_head, node_head), pre_join=None if step_ahead else (2,), step_ahead=step_ahead)
        return prog

    def _build_ahead_program(self):
        """What the NEXT reverse step can do before its coordinates exist (they come last, from the Gaussian posterior): embed its atom /
        bond features from the types the categorical posteriors have just drawn (its t is in `ws.in_t_next`), layer 0's first-layer
        blocks, knn-node query, triplet queries and bond-node sub-layer -- the products every later layer gets from the layer in front
        of it (`layer_ahead`).  Lane 2 behind the node posterior, lane 3 behind the bond posterior; `prog_step` picks the results up with
        whole-lane order points (the programs own their events, so an event cannot cross from one to the other)."""
        w, p, pk, prog = self.ws, self.plan, self.pack, []
        lig = [(p.lig2ctx, p.n_lig, True)]
        both = [(p.lig2ctx, p.n_lig, True), (p.phore2ctx, p.n_phore, False)]
        L0 = pk.layers[0]
        self._sync(prog, 3, (2,))         # (the caller writes `in_t_next` on lane 2 right in front of this program)
        self._lane = 3
        self._embed_bond(prog, w.in_t_next)
        hb_done = self._record(prog, 3)
        self._lane = 2
        self._embed_ctx(prog, w.in_t_next, coords=False)
        self._first_layer_gemm(prog, L0, w.h_in, w.Y1)
        y1_done = self._record(prog, 2)
        self._query_gemm(prog, L0.NE, w.Y1, 0, both, 3)
        self._wait(prog, 2, hb_done)
        self._triplet_queries(prog, L0, w.hb[0], w.Y1)
        self._lane = 3
        self._wait(prog, 3, y1_done)
        self._bond_node_rows(prog, L0, w.hb[0], w.Y1)
        self._node_attention(prog, hip.SEG_BOND_NODE, L0.NB, w.Y1, 5 * 128, w.x[0], lig, out=w.aggB, csrc=w.CsB2, buf=2)
        self._lane = 0
        return prog

    def pipelined_programs(self):
        """(prog_step, prog_ahead) of the pipelined sampler step, or None when this engine's schedule has no layer-ahead form."""
        if not (self.step_ahead and self.multi_stream and self.layer_ahead and self.fused_geom and self.staged_triplet and self.plan.n_tri_iters
                and self.order_points):
            return None
        if self.prog_step is None:
            tri_calls, self.tri_calls = self.tri_calls, []
            self.prog_step = self._build_forward_program(step_ahead=True)
            self.step_tri_calls, self.tri_calls = self.tri_calls, tri_calls
            self.prog_ahead = self._build_ahead_program()
        return self.prog_step, self.prog_ahead

    def _first_layer_gemm(self, prog, L, h_in, Y1):
        # first-layer blocks: knn-node blocks for every ctx node, bond-node / triplet blocks only where they are read
        # (ligand atoms: targets and sources of bond edges)
        p, n = self.plan, self.plan.n_ctx
        self._gemm(prog, h_in, 128, L.W_node1, Y1, n, 1920, bias=L.b_node1)

    def _triplet_queries(self, prog, L, hb_in, Y1):
        w, p, E = self.ws, self.plan, self.plan.n_bond
        self._gemm(prog, hb_in, 128, L.TB.W_q_hb, w.qhid, E, 128, add1=Y1[:, 14 * 128:15 * 128], idx1=p.bond_dst)
        self._gemm(prog, w.qhid, 128, L.TB.W2q, w.qT, E, 128, bias=L.TB.b2q, ln=(L.TB.q_ln_g, L.TB.q_ln_b), scale=HEAD_SCALE)

    def _bond_node_rows(self, prog, L, hb_in, Y1):
        w, p, E = self.ws, self.plan, self.plan.n_bond
        self._gemm(prog, hb_in, 128, L.NB.W_hb, w.CsB2, E, 256, add1=Y1[:, 7 * 128:9 * 128], idx1=p.bond_src)

    def _denoiser_program(self, prog, lig, both, heads=None, pre_join=None, step_ahead=False):
        """uni_denoiser.py:396-430: knn graph + gate once, then the layers.  State starts in slot 0."""
        w, p, pk, lib = self.ws, self.plan, self.pack, self.lib
        n, E, t = p.n_ctx, p.n_bond, p.topo_ref
        h, x, hb = w.h, w.x, w.hb
        g = pk.gate
        # the knn graph and its gate are read by the knn attention only (lane 1): they run there, beside layer 0's first GEMMs
        self._fork(prog, (1,))
        self._lane = 1
        self._call(prog, lib.pg_knn_ctx, t, x[0].data_ptr(), self.k, w.nbr.data_ptr(), w.deg.data_ptr())
        self._call(prog, lib.pg_edge_gate, t, x[0].data_ptr(), w.nbr.data_ptr(), w.deg.data_ptr(), self.k,
                   g['W0'].data_ptr(), g['b0'].data_ptr(), g['g'].data_ptr(), g['b'].data_ptr(), g['W3'].data_ptr(),
                   C.c_float(g['b3']), w.ew.data_ptr())
        self._mark(prog, 'graph', w.nbr, w.deg, w.ew, lane=1)
        if self.group_knn:      # ligand-source rows first: uniform row tiles skip the other kind's distance columns (node_attn.hip)
            self._call(prog, lib.pg_knn_group_by_kind, t, self.k, w.nbr.data_ptr(), w.deg.data_ptr(), w.ew.data_ptr())
        self._lane = 0
        cur = 0
        staged = bool(p.n_tri_iters and self.staged_triplet)
        n_layers = len(pk.layers)
        # next layer's x-independent products inside this layer's position phase: measured (same box, alternating runs) 16 graphs
        # 3.93 -> 3.83 ms, 32 graphs 5.94 -> 5.67, 64 graphs 10.55 -> 10.36, 128 graphs 20.12 -> 20.17 (the triplet kernel then
        # shares the chip with more side work: 2.03 -> 2.19 ms per launch) -- so it is used below ~100 graphs of the headline shape
        ahead = self.layer_ahead and self.multi_stream
        v2 = self.ahead_v2 and ahead        # Y1 of the next layer on lane 1 behind Y2, bond-node of layer 0 on lane 3, finer waits (round 4)
        # the Q rows (a K = 20 product, bound by its 256-wide output) behind P on the bond chain's own lane, or beside P on lane 2: lane 0 in
        # the v2 schedule (there lane 2 would need its own release after the layer's closing launch: measured equal or slower, 8 graphs
        # 2.05 vs 2.03 ms) and for the largest batches (128 graphs + 0.3 % on the side lane), lane 2 in between (64 graphs 9.98 -> 9.94 ms)
        chain_q = self.multi_stream and (v2 or E >= self.chain_q_from)
        # (round 6, with the queue's tail in half-groups: the best grid moved up by 32 -- 8 graphs 2.01 / 1.90 / 1.92 ms with 160 / 192 / 224
        #  workgroups, 16 graphs 3.31 / 2.99 / 2.81 / 2.93 with 160 / 192 / 224 / 256, 32 graphs 5.33 / 4.88 / 5.10 with 192 / 224 / 256; the defaults
        #  below follow, the calibration in begin_sampling still tries the neighbours)
        # small batches leave some CUs to the side lanes while the persistent triplet kernel runs (measured on the headline shape:
        # 16 graphs = 25 k bond edges 3.76 -> 3.57 ms per step with 192 workgroups, 24 / 32 graphs 4.9 -> 4.7 / 5.87 -> 5.61 with
        # 224 and the merged knn launch; 48 graphs equal either way, from 64 graphs up the full grid is fastest)
        tri_grid = self.tri_grid if self.tri_grid >= 0 else ((192 if E < 15000 else 224 if E < 80000 else 0) if self.multi_stream else 0)
        # ... and which multiple of 32 is best flips from batch to batch of the SAME size (whole 12-segment rounds per workgroup against
        # the CUs left to the node chain: rank shares of the headline batch with 25.7 k bond edges each: 3.48 / 3.31 ms with 192 / 224
        # workgroups for one, 3.27 / 3.48 for the next; profiles/r04_share_tri_grid.txt), so small batches TIME the neighbours of this
        # default once, before the sampler loop, and keep the fastest (`calibrate_tri_grid`; the result does not depend on the grid: the queue
        # hands out the same segments)
        self._tune = None
        if (self.tune_grid and self.tri_grid < 0 and self.multi_stream and staged and 0 < tri_grid <= 224 and not self.graph_mode == '1'):
            self._tune = dict(cands=[g for g in (tri_grid, tri_grid + 32, tri_grid - 32) if 128 <= g <= 256])

        if v2 and w.Y1b is None:
            w.Y1b = torch.empty_like(w.Y1)
        # v2: the layer's closing launch (x' = x + dx, and the next layer's geometry from x') once per chain -- on lane 0 for the bond
        # chain (x', bond smearing), on lane 1 for the node chain (its own copy of x', direction vectors) -- so that neither chain waits
        # for a launch on the other's lane: each waits for the other's position update only.  The two copies of x' are the same
        # expression of the same operands.
        split = v2 and self.fused_geom and self.geom_split
        if split and getattr(w, 'x_node', None) is None:
            w.x_node = [torch.empty_like(x[0]), torch.empty_like(x[0])]
            w.dxe2 = torch.zeros_like(w.dxe)

        first_layer_gemm = lambda L_, h_in, Y1_: self._first_layer_gemm(prog, L_, h_in, Y1_)
        triplet_queries = lambda L_, hb_in, Y1_: self._triplet_queries(prog, L_, hb_in, Y1_)
        bond_node_rows = lambda L_, hb_in, Y1_: self._bond_node_rows(prog, L_, hb_in, Y1_)
        assert not step_ahead or (ahead and self.fused_geom and staged)

        for li, L in enumerate(pk.layers):
            nxt = 1 - cur
            hc, xc, hbc, hn, xn, hbn = (w.h_in if li == 0 else h[cur]), x[cur], hb[cur], h[nxt], x[nxt], hb[nxt]
            # v2 writes the next layer's first-layer blocks while this layer's are still being read: two buffers, alternating (the
            # other schedules overwrite in place, behind the last reader)
            Y1c, Y1n = (w.Y1, w.Y1) if not v2 else ((w.Y1, w.Y1b) if li % 2 == 0 else (w.Y1b, w.Y1))
            xc1 = w.x_node[cur] if (split and li > 0) else xc      # the node chain's coordinates
            dxe = w.dxe2 if (split and li % 2) else w.dxe           # (alternating: lane 1 writes the next while lane 0's closing launch may read this one)
            pre = ahead and (li > 0 or step_ahead)      # Y1, the triplet queries and the bond-node rows of this layer were launched by the previous one
            pre0 = step_ahead and li == 0                # ... layer 0 of a pipelined step: by `prog_ahead` at the end of the previous STEP (lanes 2 / 3)
            if self.fused_geom:
                # direction vectors and bond-length smearing of this layer came with the previous layer's coordinate update (ONE
                # launch on lane 0, pg_layer_geom); layer 0 forms them from the embedded coordinates the same way
                if li == 0:
                    self._call(prog, lib.pg_layer_geom, t, xc.data_ptr(), None, None, w.nrm_phore_ctx.data_ptr(), None,
                               w.nrm.data_ptr(), w.G.data_ptr())
            else:
                # direction vectors (read by the knn attention, lane 1) and bond-length smearing (read by the P product on lane 0 and
                # the Q rows on lane 2) depend on x only: they run beside the first-layer GEMM instead of in front of it
                self._fork(prog, (1, 3))
                self._lane = 1
                self._call(prog, lib.pg_lig_normals, t, xc.data_ptr(), w.phore_norm.data_ptr(), p.phore2ctx.data_ptr(),
                           w.nrm.data_ptr())
                self._lane = 3
                self._call(prog, lib.pg_bond_smear, t, xc.data_ptr(), w.G.data_ptr())
                self._lane = 0
            if (pre and not v2) or pre0:
                # lane 2 carried this layer's first-layer blocks and triplet queries through the previous layer's position updates
                # (pipelined step, layer 0: through the end of the previous step)
                self._sync(prog, 0, (2,))
            elif not pre:
                first_layer_gemm(L, hc, Y1c)
            if li == 0 and pre_join:
                self._join(prog, pre_join)
            # (v2 from layer 1 on: lanes 2 / 3 have nothing in the first half of the layer, only the node chain's lane is released here)
            q_side = staged and not (chain_q and self.fused_geom)       # the Q rows on lane 2: it reads this layer's smearing (lane 0)
            if pre and split and not pre0:         # (split: lane 1 went on behind its own closing launch)
                if q_side:
                    self._fork(prog, (2,))
            else:
                self._fork(prog, ((1, 2) if q_side else (1,)) if (pre and v2) else (1, 2, 3))
            # (v2: this layer's first-layer blocks came from lane 1 behind the previous layer's Y2, which lane 0 has joined since; the
            #  triplet queries on lane 2 are waited for in front of the triplet kernel, not in front of P)
            if not self.fused_geom:
                self._sync(prog, 0, (3,))              # the smearing (alone on lane 3 so far) is read by P on lane 0
                self._sync(prog, 2, (3,))              # ... and by the Q rows on lane 2; the queries on lane 3 do not wait for it
            last = heads is not None and li == n_layers - 1
            # Launch order of a layer.  Lane 0 carries the bond chain (P -> triplet -> bond position update), lane 1 the node
            # chain (knn attention -> lin_node -> second first-layer GEMM -> knn position update), lane 2 the Q rows and the
            # bond-node attention, lane 3 the triplet queries: nothing on lane 1 waits for the triplet kernel, and lane 0 picks
            # the node chain's results up at the two points where the bond chain needs them.
            # ---- bond update over triplets, its operands (:285)
            # P[k->j] = W.[h_bond | G] + (source half)[k] + (target half)[j].  Every row of the block a segment j->i reads has the
            # same target j, so in the staged form that half rides on the segment's own row Q[j->i] instead (one gathered operand
            # per product; the gather kernel builds Q in-kernel and keeps it on P)
            self._lane = 0
            self._gemm(prog, hbc, 128, L.TB.W_hbg, w.P, E, 256, X2=w.G, K2=20,
                       add1=Y1c[:, 10 * 128:12 * 128], idx1=p.bond_src,
                       **({} if staged else dict(add2=Y1c[:, 12 * 128:14 * 128], idx2=p.bond_dst)))
            # the per-segment constant of the triplet MLPs as rows: Wg2 . smear(d_ji) + (target half)[j].  Small batches: behind P on
            # lane 0 (a 25 us product costs less there than the two cross-lane hops around it on a side lane)
            q_lane = 0 if (chain_q and self.fused_geom) else 2
            self._lane = q_lane
            if staged:
                self._gemm(prog, w.G, 20, L.TB.W_g2, w.Qd, E, 256, add1=Y1c[:, 12 * 128:14 * 128], idx1=p.bond_src)
                if q_lane != 0 and not (pre and v2):   # (v2 from layer 1 on: the wait for lane 2 in front of the triplet kernel covers it)
                    self._sync(prog, 0, (2,))          # lane 0 (the triplet kernel) waits for the Q rows, not for all of lane 2
            if not pre:
                self._lane = 3                                                          # triplet queries
                triplet_queries(L, hbc, Y1c)
                q3_done = self._record(prog, 3)
            self._lane = 0
            if not pre:                            # (pre: lane 3 carries the bond-node attention, which lin_node waits for)
                if v2:
                    self._wait(prog, 0, q3_done)   # (lane 3 goes on with the bond-node attention: not joined)
                else:
                    self._join(prog, (3,))
            if pre and v2 and not pre0:            # (pre0: lane 0 has waited for lane 2 at the layer's start)
                self._sync(prog, 0, (2,))          # this layer's triplet queries (lane 2, launched one layer ahead)
            a = L.TB
            self._event(prog, 'triplet', True)
            # one launch, or two when a few ligands need more row tiles than the rest (BatchPlan.tri_split): the ligands of up to 50 atoms on
            # the 3-tile instance of the kernel, the larger ones with their own queue (the 4-tile instance costs every segment ~4 %:
            # tools/experiments/triplet_maxt_penalty.py) on lane 3 BESIDE it -- the two launches cover disjoint ligands, a workgroup slot that
            # one queue no longer fills goes to the other, and the sub-layer ends with one tail instead of two (one kernel that picks its
            # unrolling per group was 12 % slower: tools/experiments/README.md).  Per step, one launch / two in a row / two side by side:
            # 128 graphs 19.21 / 19.00 / 18.89 ms, 64 graphs 9.60 / 9.59 / 9.44, 32 and 16 graphs equal (profiles/r06_triplet_side_by_side.txt)
            queues = [(p.tri_iters, p.n_tri_iters, 0, p.tri_counter)]
            if staged and self.tri_split and p.tri_split is not None and (E >= self.tri_split_from or self.tri_split == 'always'):
                queues = [p.tri_split['small'], p.tri_split['big']]
            side = abs(self.tri_overlap) if len(queues) == 2 else 0
            big_first = self.tri_overlap > 0
            if side:                               # the larger ligands' queue on a side lane, beside (not behind) the other launch
                if big_first:
                    queues = queues[::-1]
                self._fork(prog, (side,))
            for qi, (q_iters, q_n, q_maxn, q_ctr) in enumerate(queues):
                if side:
                    self._lane = side if (qi == 0) == big_first else 0
                self.tri_calls.append(len(prog))
                self._seg(prog, hip.SEG_TRIPLET, E, p.tri_order, a, x=xc, Csrc_k=w.P[:, 0:128], Csrc_v=w.P[:, 128:256],
                          ld_csrc=w.P.stride(0), Wf_k=a.Wf_k, Wf_v=a.Wf_v, Wg2_k=a.Wg2_k, Wg2_v=a.Wg2_v, G=w.G, q=w.qT,
                          W2k_l=a.W2k_l, W2v_l=a.W2v_l, b2v=a.b2v, resid=hbc, out=hbn, seg_chunks=p.tri_chunks,
                          **(dict(tri_iters=q_iters, n_tri_iters=q_n, tri_counter=q_ctr, tri_grid=tri_grid, tri_max_nlig=q_maxn,
                                  Cdst_k=w.Qd[:, 0:128], Cdst_v=w.Qd[:, 128:256], ld_cdst=256) if staged else {}))
            if side:
                self._lane = 0
                self._join(prog, (side,))
            self._event(prog, 'triplet', False)
            if last:                               # lane 3 (the triplet queries) has been joined: the bond head takes it
                self._fork(prog, (3,))
                self._lane = 3
                heads[0](hbn)
            # ---- node update over bond edges (:284)                                   [lane 2]
            # (the sub-layer reads no coordinates: from layer 1 on a small batch has launched it during the previous layer's position
            #  updates, see below)
            self._lane = 3 if v2 else 2            # (v2: lane 1 then never waits on lane 2, so lane 2 may wait on lane 1 -- the other
            if not pre:                            #  order of the two edges breaks hipGraph capture)
                bond_node_rows(L, hbc, Y1c)
                self._node_attention(prog, hip.SEG_BOND_NODE, L.NB, Y1c, 5 * 128, xc, lig, out=w.aggB, csrc=w.CsB2, buf=1)
                if v2:
                    bn_done = self._record(prog, 3)
            # ---- node update over knn edges (:281), then h' = h + lin_node(aggE + aggB) (:288)   [lane 1]
            self._lane = 1
            if not pre:
                self._query_gemm(prog, L.NE, Y1c, 0, both, 0)
            if pre0:
                self._sync(prog, 1, (2,))              # this layer's first-layer blocks and knn-node query (lane 2, `prog_ahead`)
            elif pre and v2:
                self._wait(prog, 1, ne_done)           # this layer's knn-node query (lane 3, launched one layer ahead)
            self._event(prog, 'knn_node', True)       # (the launches of the sub-layer: ligand targets, pharmacophore targets)
            self._node_attention(prog, hip.SEG_KNN_NODE, L.NE, Y1c, 0, xc1, both, out=w.aggE, buf=0, query_done=True,
                                 qbuf=3 if pre else None)
            self._event(prog, 'knn_node', False)
            if pre0:
                self._sync(prog, 1, (3,))              # aggB: the bond-node attention of `prog_ahead` (lane 3)
            elif pre or v2:
                self._wait(prog, 1, bn_done)           # aggB: the bond-node attention on lane 3 (launched one layer ahead from layer 1 on)
            else:
                self._sync(prog, 1, (2,))              # aggB
            # two K = 128 launches of the streaming kernel instead of one K = 256 launch of the tiled one (56 -> 2 x ~17 us)
            self._gemm(prog, w.aggE, 128, L.W_lin2[:, :128], w.lin_tmp, n, 128, bias=L.b_lin, add1=hc)
            self._gemm(prog, w.aggB, 128, L.W_lin2[:, 128:], hn, n, 128, add1=w.lin_tmp)
            self._mark(prog, f'A{li}', w.aggE, w.aggB, lane=1)      # (the next layer's bond-node attention may overwrite aggB from here on)
            if step_ahead and self.head_early and li == n_layers - 1 and heads is not None:
                hn_done = self._record(prog, 1)        # h' of the last layer: the node head's only input
            # ---- position updates from h', h_bond' and the OLD geometry (:291-296)
            # knn-pos k/v source halves for every node (cols 256:512); target halves, queries and the bond-pos blocks
            # only for ligand atoms
            self._gemm(prog, hn, 128, L.W_node2, w.Y2, n, 1280, bias=L.b_node2)
            if v2:
                # the bond position update's query right behind Y2 on the node chain's lane (Y2 is out well before the triplet kernel ends):
                # the one event lane 0 waits for below then covers it, instead of a second cross-lane hop in front of the attention
                self._query_gemm(prog, L.PB, w.Y2, 5 * 128, lig, 1)
            more_ahead = ahead and li + 1 < n_layers and not last
            self._sync(prog, 0, (1,))                  # lane 0 continues after the triplet kernel AND Y2 (which implies lane 2)
            if v2 and more_ahead:
                # h' is final and Y2 is out: the next layer's first-layer blocks follow on lane 1 (beside the triplet kernel / the bond
                # position update), so that the position phase's side lanes start with the triplet queries right away and the next P
                # waits for these blocks only.  They go to the OTHER first-layer buffer: its last readers belong to the previous layer,
                # and lane 1 has waited for every one of them since (layer-start fork, ne_done, bn_done)
                self._lane = 1
                first_layer_gemm(pk.layers[li + 1], hn, Y1n)
                y1_done = self._record(prog, 1)
            # the bond position update's query MLP runs beside its edge product / the knn position update, not in front of the
            # attention on lane 0: lane 3 is free after the triplet queries (last layer: lane 2, in front of the node head --
            # forked from lane 0, which has just seen h' and Y2: a lane-2-waits-lane-1 edge after lane 1 waited on lane 2 crashes
            # hipGraph capture, PG_GRAPH=1)
            qlane = 2 if last else 3
            # (releasing lanes 2 / 3 in FRONT of lane 0's wait above -- they need nothing of lane 1 -- measured slower: 16 graphs 3.10 vs 3.05 ms)
            early = step_ahead and last and self.head_early
            if early:
                # pipelined step: the node head (and behind it the node posterior and the next step's products) needs h' only -- it does not
                # wait for the triplet kernel (this lane-2-waits-lane-1 edge is what the hipGraph capture of the plain list cannot have)
                self._wait(prog, 2, hn_done)
                self._lane = 2
                heads[1](hn)
                if not v2:
                    self._fork(prog, (2,))             # Y2: lane 0 has waited for lane 1 above
                # (v2: lane 2 goes on behind `hn_done` alone -- what follows on it, `prog_ahead`, embeds the next step's features into `h_in`,
                #  which nothing of this step reads any more, and never touches h[0])
            else:
                self._fork(prog, (2, 3) if (ahead and li + 1 < n_layers) else (qlane,))
            self._lane = qlane
            if not v2:
                self._query_gemm(prog, L.PB, w.Y2, 5 * 128, lig, 1)
                q_done = self._record(prog, qlane)     # (this point of the lane: the node head below is not waited for)
            if last and not early:
                heads[1](hn)                           # lane 2 has nothing else left in this step: the node head takes it
            elif ahead and li + 1 < n_layers:
                # ---- the NEXT layer's products that do not depend on the new coordinates: its first-layer blocks (h' is final:
                #      every reader of this layer's Y1 has run), its triplet queries and bond-node rows (h_bond' is final, the
                #      triplet kernel has read qT): they fill lane 2 during this layer's position updates instead of standing in
                #      front of the next triplet kernel.  (Forked from lane 0 like the node head, for the same reason.)
                #      The bond-node sub-layer and the knn-node query read no coordinates either: they go along (lane 3 behind the
                #      bond position update's query; lin_node of the next layer is what waits for it, not its triplet kernel).
                Ln = pk.layers[li + 1]
                self._lane = 2
                if v2:
                    self._wait(prog, 2, y1_done)
                else:
                    first_layer_gemm(Ln, hn, Y1n)
                    y1_done = self._record(prog, 2)
                triplet_queries(Ln, hbn, Y1n)
                if not v2:
                    self._query_gemm(prog, Ln.NE, Y1n, 0, both, 3)
                self._lane = 3
                self._wait(prog, 3, y1_done)
                if v2:
                    self._query_gemm(prog, Ln.NE, Y1n, 0, both, 3)
                    ne_done = self._record(prog, 3)
                bond_node_rows(Ln, hbn, Y1n)
                self._node_attention(prog, hip.SEG_BOND_NODE, Ln.NB, Y1n, 5 * 128, xc, lig, out=w.aggB, csrc=w.CsB2, buf=2)
                bn_done = self._record(prog, 3)
            self._lane = 1
            self._node_attention(prog, hip.SEG_KNN_POS, L.PE, w.Y2, 0, xc1, lig, dx=dxe, buf=0)
            self._lane = 0
            self._gemm(prog, hbn, 128, L.PB.W_hb, w.CsB, E, 256, add1=w.Y2[:, 7 * 128:9 * 128], idx1=p.bond_src)
            if step_ahead and last:
                csb_done = self._record(prog, 0)       # the last reader of this step's final h_bond outside lane 3: `prog_ahead` may embed the next
            if not v2:
                self._wait(prog, 0, q_done)
            self._node_attention(prog, hip.SEG_BOND_POS, L.PB, w.Y2, 5 * 128, xc, lig, dx=w.dxb, csrc=w.CsB, buf=1, query_done=True)
            more = li + 1 < n_layers
            if split and more:
                bp_done = self._record(prog, 0)
            self._join(prog, (1,))
            if self.fused_geom:      # x' = x + dx, and from x' the next layer's smearing + direction vectors (last layer: the update alone)
                self._call(prog, lib.pg_layer_geom, t, xc.data_ptr(), dxe.data_ptr(), w.dxb.data_ptr(), w.nrm_phore_ctx.data_ptr(),
                           xn.data_ptr(), w.nrm.data_ptr() if (more and not split) else None, w.G.data_ptr() if more else None)
                if split and more:
                    self._lane = 1
                    self._wait(prog, 1, bp_done)
                    self._call(prog, lib.pg_layer_geom, t, xc1.data_ptr(), dxe.data_ptr(), w.dxb.data_ptr(), w.nrm_phore_ctx.data_ptr(),
                               w.x_node[nxt].data_ptr(), w.nrm.data_ptr(), None)
                    self._lane = 0
            else:
                self._call(prog, lib.pg_apply_dx, t, xc.data_ptr(), dxe.data_ptr(), w.dxb.data_ptr(), xn.data_ptr())
            self._mark(prog, f'L{li}', hn, hbn, xn, dxe, w.dxb, hbc)
            cur = nxt
        if heads is not None and step_ahead:
            # no join: the categorical posteriors and `prog_ahead` follow on lanes 2 / 3 (the caller joins when it needs the results:
            # `join_lanes`); lane 3 -- where the next step's bond embedding overwrites h_bond -- waits for the bond-pos rows' read of it
            self._wait(prog, 3, csb_done)
        elif heads is not None:
            self._join(prog, (2, 3))
        self.final_idx = cur

    def _compile(self, prog):
        """The launch list as PgLaunch records inside the library (order points renumbered densely per program)."""
        recs, slot = [], {}
        for fn, args, lane in prog:
            if lane >= 0:
                recs.append(hip.launch_record(fn, args, lane))
            elif fn.kind == 'order':
                for what, pt, l in fn.ops:
                    L = hip.PgLaunch()
                    L.op, L.lane, L.ev, L.n_arg = (hip.OP_RECORD if what == 'record' else hip.OP_WAIT), l, slot.setdefault(pt, len(slot)), 0
                    recs.append(L)
        return hip.Program(recs, len(slot))

    def _run(self, prog, compiled=True):
        cur = torch.cuda.current_stream()
        if self._side is None:
            self._side = side_streams(cur.device_index, self.stream_set)
        streams = [cur] + self._side
        sp = [st.cuda_stream for st in streams]
        if self.trace is not None:
            return self._run_traced(prog, streams, sp)
        if compiled and self.c_program and self.timers is None and self.debug is None and not torch.cuda.is_current_stream_capturing():
            ent = self._compiled.get(id(prog))
            if ent is None or ent[0] is not prog:
                ent = self._compiled[id(prog)] = (prog, self._compile(prog))
            ent[1].run((C.c_void_p * hip.PG_PROGRAM_LANES)(*sp))
            return
        for fn, args, lane in prog:
            if lane < 0:
                self._tap(fn, streams)
                continue
            rc = fn(*args, sp[lane])
            if rc:
                hip.check(rc, fn.__name__)

    def _run_traced(self, prog, streams, sp):
        """`_run` with a timing event before and after every launch, on the launch's lane: the step's timeline as the device sees
        it without a profiler attached (a kernel-trace profiler adds device-side latency to every dispatch)."""
        for fn, args, lane in prog:
            if lane < 0:
                self._tap(fn, streams)
                continue
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(streams[lane])
            rc = fn(*args, sp[lane])
            b.record(streams[lane])
            what = fn.__name__
            if what in ('pg_gemm', 'pg_seg_attn'):
                s = args[-1]._obj
                what += f'[{s.M}x{s.N}x{s.K1}+{s.K2}]' if what == 'pg_gemm' else f'[mode {s.mode}, {s.n_seg}+{s.n_seg2}]'
            self.trace.append((what, lane, a, b))
            if rc:
                hip.check(rc, fn.__name__)

    def forward(self, h_node_pert, pos_pert, h_edge_pert, time_step):
        """Returns (logits_v [N_lig,12], x0 [N_lig,3], logits_bond [E,6]) as views of the workspace."""
        w = self.ws
        w.in_h_node.copy_(h_node_pert)
        w.in_pos.copy_(pos_pert)
        w.in_h_edge.copy_(h_edge_pert)
        w.in_t.copy_(time_step)
        return self.forward_inplace()

    def _graph_wanted(self):
        return self.graph_mode == '1' and self.timers is None and self.debug is None

    def _capture(self):
        """Capture the (static) forward launch list, side-stream forks / joins included, into one hipGraph."""
        self._run(self.prog_fwd)              # eager once: lazy kernel attributes, side streams
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._run(self.prog_fwd)
        self._graph = g

    TUNE_REPS = 6          # timing marks per candidate grid (five timed steps between them, behind one settling step)

    def _set_tri_grid(self, grid):
        for i in self.tri_calls:
            self.prog_fwd[i][1][-1]._obj.tri_grid = grid
        if getattr(self, 'prog_step', None) is not None:
            for i in self.step_tri_calls:
                self.prog_step[i][1][-1]._obj.tri_grid = grid

    def calibrate_tri_grid(self, run_step):
        """Small batches: which persistent grid of the triplet kernel (the default by size or its neighbours, multiples of 32 workgroups) makes a
        whole sampler step fastest -- the kernel shares the chip with the node chain's launches, and which multiple wins flips between batches of
        the SAME size (profiles/r04_share_tri_grid.txt).  Decided ONCE, in `PhoreDiff.begin_sampling`, before the caller's loop: `run_step(k)` runs
        one real sampler step (denoiser, posteriors, whatever overlaps) on scratch state; per candidate one settling step and TUNE_REPS - 1 timed
        ones, HIP events between the steps, one host wait at the end.  The loop itself then has no host synchronisation and a fixed launch
        configuration; the choice is remembered per batch shape for the life of the process (`_TRI_GRID_CACHE`).  Results do not depend on
        the grid (the queue hands out the same segments)."""
        t, self._tune = self._tune, None
        if t is None:
            return
        # remembered per batch SHAPE, coarsely (bond rows in steps of 2 048, context nodes in steps of 512, graphs, row tiles of the largest ligand): the
        # repeated `sample()` calls of a sample_all.py-style loop (same batch size, atom counts drawn anew: sample_all.py:79-94) calibrate once, not
        # 21 extra steps per call
        key = (self.plan.n_bond // 2048, self.plan.n_ctx // 512, self.plan.n_graphs, (max(self.plan.topo.max_nlig, 2) + 14) // 16)
        if key in _TRI_GRID_CACHE:
            self.tuned_tri_grid, self.tuned_tri_grid_ms = _TRI_GRID_CACHE[key]
            self._set_tri_grid(self.tuned_tri_grid)
            return
        k = 0
        for _ in range(3):                            # the first steps of an engine pay one-time costs (kernel attributes, allocator)
            run_step(k)
            k += 1
        marks = {}
        for g in t['cands']:
            self._set_tri_grid(g)
            run_step(k)                               # settles
            k += 1
            marks[g] = []
            for _ in range(self.TUNE_REPS):
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                marks[g].append(ev)
                if len(marks[g]) < self.TUNE_REPS:
                    run_step(k)
                    k += 1
        torch.cuda.current_stream().synchronize()    # the one host wait (inside begin_sampling)
        ms = {}
        for g, evs in marks.items():
            per = sorted(a.elapsed_time(b) for a, b in zip(evs[:-1], evs[1:]))
            ms[g] = 0.5 * (per[len(per) // 2 - 1] + per[len(per) // 2]) if len(per) % 2 == 0 else per[len(per) // 2]
        best = min(ms, key=ms.get)
        if ms[best] > 0.99 * ms[t['cands'][0]]:      # the default stays unless a neighbour is clearly faster
            best = t['cands'][0]
        self._set_tri_grid(best)
        self.tuned_tri_grid, self.tuned_tri_grid_ms = best, ms
        _TRI_GRID_CACHE[key] = (best, ms)

    def lane_stream(self, lane):
        """hipStream_t of a lane for launches the caller adds beside a program (lane 0 = torch's current stream)."""
        cur = torch.cuda.current_stream()
        if self._side is None:
            self._side = side_streams(cur.device_index, self.stream_set)
        return cur if lane == 0 else self._side[lane - 1]

    def join_lanes(self, lanes=(2, 3)):
        """The caller's stream continues after everything enqueued so far on the side lanes."""
        cur = torch.cuda.current_stream()
        for l in lanes:
            ev = torch.cuda.Event()
            ev.record(self.lane_stream(l))
            cur.wait_event(ev)

    def fork_lanes(self, lanes=(2, 3)):
        """The side lanes continue after everything enqueued so far on the caller's stream."""
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        for l in lanes:
            self.lane_stream(l).wait_event(ev)

    def step_forward(self):
        """The denoiser forward of a pipelined sampler step (`prog_step`): (out_v, the ctx-ordered final coordinates, out_bond) WITHOUT joining
        lanes 2 / 3 (out_v is complete on lane 2, out_bond on lane 3)."""
        w = self.ws
        self._run(self.prog_step)
        return w.out_v, w.x[self.final_idx], w.out_bond        # (x0 in ctx rows: the Gaussian posterior reads it through lig2ctx)

    def forward_inplace(self):
        w = self.ws
        if self._graph_wanted():
            if self._graph is None:
                self._capture()
            self._graph.replay()
        else:
            self._run(self.prog_fwd)
        x0 = torch.index_select(w.x[self.final_idx], 0, self.plan.lig2ctx_long)
        return w.out_v, x0, w.out_bond


def denoiser_forward_standalone(module, h, x, bond_index, h_bond, mask_ligand, batch, phore_norm, return_all, record=False):
    """Entry used by models.uni_denoiser.UniTransformerO2TwoUpdateGeneralBond.forward (ctx-ordered inputs).  `record`: gradients are
    being recorded -> the autograd-composed form of the same kernels (training.TrainForward.denoise) instead of the launch list."""
    from .plan import BatchPlan
    dev = h.device
    if dev.type != 'cuda':
        raise RuntimeError('phoregen_amd: the denoiser runs on the MI355X HIP path only (no CPU fallback)')
    mask = mask_ligand.bool()
    B = int(batch.max().item()) + 1
    lig_ctx = mask.nonzero().squeeze(-1)
    ph_ctx = (~mask).nonzero().squeeze(-1)
    ctx2lig = torch.full((h.size(0),), -1, dtype=torch.long, device=dev)
    ctx2lig[lig_ctx] = torch.arange(lig_ctx.numel(), device=dev)
    edge_index = ctx2lig[bond_index]
    plan = BatchPlan(batch[lig_ctx], batch[ph_ctx], edge_index, batch[lig_ctx][edge_index[0]], B, dev)
    if not torch.equal(plan.lig2ctx_long, lig_ctx):
        raise ValueError('phoregen_amd: context must be ordered [phore..., ligand...] per graph (compose_context)')
    if record:
        from .training import TrainForward
        params = {'denoiser.' + k: v for k, v in {**dict(module.named_buffers()), **dict(module.named_parameters())}.items()}
        tf = TrainForward(params, plan, knn_k=module.k, num_layers=module.num_layers, denoiser_only=True)
        hb_in = h_bond.float() if plan.edge_identity else h_bond.float().index_select(0, plan.edge_ref_long)    # internal bond order
        nrm_ph = torch.zeros(h.size(0), 3, dtype=torch.float32, device=dev).index_copy(0, plan.phore2ctx_long, phore_norm.float())
        h_out, x_out, hb_out = tf.denoise(h.float(), x.float(), hb_in, nrm_ph)
        out = {'x': x_out, 'h': h_out, 'h_bond': hb_out if plan.edge_identity else hb_out.index_select(0, plan.edge_int_long)}
        if return_all:
            out.update(all_x=[x, out['x']], all_h=[h, out['h']], all_h_bond=[h_bond, out['h_bond']])
        return out
    sd = {'denoiser.' + k: v for k, v in module.state_dict().items()}
    eng = Engine(_DenoiserOnlyPack(sd, module.num_layers), plan, knn_k=module.k, full=False)
    w = eng.ws
    w.h_in.copy_(h)
    w.x[0].copy_(x)
    w.hb[0].copy_(h_bond if plan.edge_identity else h_bond.index_select(0, plan.edge_ref_long))   # internal bond order
    w.phore_norm.copy_(phore_norm)
    w.nrm_phore_ctx.index_copy_(0, plan.phore2ctx_long, phore_norm.float())
    prog = []
    eng._denoiser_program(prog, [(plan.lig2ctx, plan.n_lig, True)],
                          [(plan.lig2ctx, plan.n_lig, True), (plan.phore2ctx, plan.n_phore, False)])
    eng._run(prog)
    c = eng.final_idx
    out = {'x': w.x[c].clone(), 'h': w.h[c].clone(),
           'h_bond': w.hb[c].clone() if plan.edge_identity else w.hb[c].index_select(0, plan.edge_int_long)}
    if return_all:
        out.update(all_x=[x, out['x']], all_h=[h, out['h']], all_h_bond=[h_bond, out['h_bond']])
    return out


def phore_encoder_standalone(module, h, edge_feat, edge_index, e_w=None):
    """Entry used by models.uni_denoiser.NodeUpdateLayer.forward for the pharmacophore-encoder configuration
    (models/__init__.py:29-35: edge_feat_dim = 1, no out_fc; called at models/diffusion.py:186-191 with the edges of
    fully_connect_two_graphs and the distance as the edge feature).  Runs the PG_SEG_PHORE program on the given `h` with
    the caller's edge feature (PgSegAttn.efeat)."""
    from .packing import fuse_blocks, pack_phore
    from .plan import BatchPlan
    dev = h.device
    if dev.type != 'cuda':
        raise RuntimeError('phoregen_amd: NodeUpdateLayer.forward runs on the MI355X HIP path only (no CPU fallback)')
    if torch.is_grad_enabled() and (h.requires_grad or edge_feat.requires_grad or any(p.requires_grad for p in module.parameters())):
        # this entry runs the forward kernels only; inside compute_loss the encoder is differentiated by training.TrainForward
        raise RuntimeError('phoregen_amd: NodeUpdateLayer.forward (standalone pharmacophore encoder) does not record gradients; '
                           'call it under torch.no_grad(), or train through PhoreDiff.compute_loss')
    if e_w is not None or edge_feat.dim() != 2 or edge_feat.size(1) != 1:
        raise NotImplementedError('phoregen_amd: NodeUpdateLayer.forward is callable in the pharmacophore-encoder form '
                                  '(one scalar edge feature, no edge gate); inside the denoiser it is fused into the layer')
    N = h.size(0)
    src, dst = edge_index[0].long(), edge_index[1].long()
    # fully_connect_two_graphs(batch, batch): all (i, j) of equal graph id, row-major, self pairs kept (common.py:329-356)
    first = torch.full((N,), N, dtype=torch.long, device=dev).scatter_reduce_(0, src, dst, 'amin')
    deg = torch.zeros(N, dtype=torch.long, device=dev).index_add_(0, src, torch.ones_like(src))
    starts = torch.unique_consecutive(first)
    batch = torch.searchsorted(starts, first, right=True) - 1
    sizes = torch.bincount(batch, minlength=starts.numel())
    if int((sizes * sizes).sum()) != src.numel() or not torch.equal(deg, sizes[batch]) or \
            not torch.equal(src, torch.repeat_interleave(torch.arange(N, device=dev), deg)):
        raise NotImplementedError('phoregen_amd: NodeUpdateLayer.forward expects the edges of fully_connect_two_graphs '
                                  '(every ordered pair of nodes of a graph, self pairs included, grouped by source)')
    B = int(starts.numel())
    z = torch.zeros(0, dtype=torch.long)
    plan = BatchPlan(z, batch.cpu(), torch.zeros(2, 0, dtype=torch.long), z, B, dev)
    eng = Engine(None, plan, full=False)
    sd = {'phore_encoder.' + k: v.detach() for k, v in module.state_dict().items()}
    PH, blocks = pack_phore(sd)
    W_ph, b_ph = fuse_blocks(blocks)
    w = eng.ws
    off = torch.zeros(B + 1, dtype=torch.long, device=dev)
    off[1:] = (sizes * sizes).cumsum(0)
    efeat, efeat_off = edge_feat.detach().float().contiguous().view(-1), off.to(torch.int32)
    prog = []
    eng._gemm(prog, h.detach().float().contiguous(), 128, W_ph, w.Yp, N, 640, bias=b_ph)
    eng._node_attention(prog, hip.SEG_PHORE, PH, w.Yp, 0, w.x_phore_ctx, [(plan.phore2ctx, plan.n_phore, False)],
                        out=w.enc_ctx, extra=dict(efeat=efeat, efeat_off=efeat_off))
    eng._run(prog)
    return w.enc_ctx.clone()


class _DenoiserOnlyPack(ModelPack):
    def __init__(self, sd, num_layers):
        from .packing import LayerPack, pack_gate
        sd = {k: v.detach() for k, v in sd.items()}
        self.layers = [LayerPack(sd, f'denoiser.base_block.{l}') for l in range(num_layers)]
        self.gate = pack_gate(sd)
