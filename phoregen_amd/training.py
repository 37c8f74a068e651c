"""Training path (PhoreDiff.compute_loss, reference models/diffusion.py:249-352): the denoiser forward with
autograd, every matrix product and every attention sub-layer a HIP kernel with a hand-written HIP adjoint.

torch.autograd is the tape and nothing more: each `Function` below launches the forward kernel of the C ABI and, in
`backward`, its adjoint (`pg_gemm` with the transposed weight, `pg_gemm_wgrad`, `pg_ln_relu_bwd`, `pg_seg_attn_bwd`,
`pg_attn_fold_wgrad`, fold <-> unfold as each other's adjoints).  Gradients reach the reference-schema parameters
through `packing.py` run WITHOUT detaching (centring, sign/|gamma| folding and the lane-fixed gathers are plain
differentiable tensor ops on 128x128-sized weights).  Index plumbing of the geometry that the sampler does inside
fused kernels (distances for the edge gate, bond-length smearing, mean of the 3 nearest atoms, residual adds, the loss
itself) is composed from elementwise tensor ops here; the kNN searches stay in the HIP kernels.
No CPU fallback: every entry point raises without the library / a GPU.
"""
import ctypes as C

import torch
import torch.nn.functional as F

from . import hip, options
from .packing import HEAD_SCALE, NODE1_PARTS, NODE2_PARTS, PHORE_PARTS, ModelPack

from .packing import _SMEAR_OFF


def _st():
    return hip.stream_ptr()


class _ZeroPool:
    """Small zero-filled buffers of one training step from ONE fill launch.  The adjoints accumulate weight / bias / coordinate gradients
    with atomics into caller-zeroed memory: ~800 buffers per step, most of them a few KB, each a `torch.zeros` = one 4 us fill launch on
    the critical stream (profiles/r05_train_kernel_stats.md: 817 fills = 3.4 ms per step).  A step takes them as 64-byte-aligned views
    of one block, allocated and filled at the step's first request with the size the previous step needed (a request that does not fit,
    or is larger than `max_bytes`, is an ordinary torch.zeros).  The block is NEW memory every step (never re-zeroed in place): a
    gradient that autograd hands on as a view of it -- p.grad of an identity-packed parameter -- keeps its block alive and is never
    overwritten by a later step."""

    def __init__(self, max_bytes=1 << 20):
        self.max_bytes, self.need, self.block, self.off, self.dev = max_bytes, 0, None, 0, None

    def begin_step(self):
        self.capacity = (self.need + 1023) // 1024 * 1024       # floats requested by the previous step
        self.need, self.block, self.off = 0, None, 0

    def zeros(self, *shape, device, dtype=torch.float32):
        n = 1
        for d in shape:
            n *= d
        if dtype != torch.float32 or n * 4 > self.max_bytes or n == 0:
            return torch.zeros(*shape, dtype=dtype, device=device)
        n16 = (n + 15) // 16 * 16
        self.need += n16
        if self.block is None and getattr(self, 'capacity', 0) >= n16:
            self.block, self.off, self.dev = torch.zeros(self.capacity, dtype=torch.float32, device=device), 0, device
        if self.block is None or self.dev != device or self.off + n16 > self.block.numel():
            return torch.zeros(*shape, dtype=dtype, device=device)
        v = self.block[self.off:self.off + n].view(*shape)
        self.off += n16
        return v

    def zeros_like(self, t):
        return self.zeros(*t.shape, device=t.device, dtype=t.dtype)


zero_pool = _ZeroPool()


def _rowmajor(t):
    return t if (t.dim() == 2 and t.stride(1) == 1) else t.contiguous()


def _dgrad(gY, W, gX):
    """gX = gY W: the input gradient of Y = X W^T is a plain product with the weight as stored -- the library GEMM (rocBLAS /
    hipBLASLt through torch.mm), which needs no transposed copy of W and runs the K = 256 ... 1920 contractions of the adjoint
    pass at about twice the rate of the tiled pg_gemm fallback (the streaming pg_gemm kernel covers K = 128 only; training
    step 195 -> 188 ms).  `PG_DGRAD_MM=0`: pg_gemm on W^T, as before."""
    if options.get('dgrad_mm'):
        if lib_timers is not None:      # measurement (tools/bench_train.py, bench.py): HIP events around the library call, on its stream
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
            torch.mm(gY, W, out=gX)
            ev[1].record()
            lib_timers.append(ev)
        else:
            torch.mm(gY, W, out=gX)
    else:
        _gemm_raw(gY, W.t().contiguous(), gX)


def _wgrad(gY, X, N, K, want_bias):
    """(gW [N,K], gb [N] | None) = (gY^T X, column sums of gY): pg_gemm_wgrad (row-split partial tiles + atomics).  The library
    GEMM was measured on this contraction over 10^5 rows and is slower (training step 188.5 -> 204.9 ms)."""
    buf = zero_pool.zeros(N * K + (N if want_bias else 0), device=X.device)   # one fill for both (small ones: a view of the step's zero block)
    gW = buf[:N * K].view(N, K)
    gb = buf[N * K:] if want_bias else None
    hip.check(hip.lib().pg_gemm_wgrad(gY.data_ptr(), gY.stride(0), X.data_ptr(), X.stride(0), X.shape[0], N, K,
                                      gW.data_ptr(), gW.stride(0), hip.ptr(gb), _st()), 'pg_gemm_wgrad')
    return gW, gb


def _gemm_raw(X, W, Y, bias=None):
    """Y = X @ W^T (+ bias) through pg_gemm."""
    g = hip.PgGemm()
    M, K = X.shape
    g.X, g.ldx, g.K1 = X.data_ptr(), X.stride(0), K
    g.W, g.ldw = W.data_ptr(), W.stride(0)
    g.bias = hip.ptr(bias)
    g.out_scale, g.act = 1.0, hip.ACT_NONE
    g.Y, g.ldy, g.M, g.N = Y.data_ptr(), Y.stride(0), M, W.shape[0]
    hip.check(hip.lib().pg_gemm(C.byref(g), _st()), 'pg_gemm')


class LinearFn(torch.autograd.Function):
    """nn.Linear: Y = X W^T + b."""

    @staticmethod
    def forward(ctx, X, W, b):
        X, W = _rowmajor(X), _rowmajor(W)
        Y = torch.empty(X.shape[0], W.shape[0], dtype=torch.float32, device=X.device)
        if X.shape[0]:
            _gemm_raw(X, W, Y, b.contiguous() if b is not None else None)
        ctx.save_for_backward(X, W)
        ctx.has_bias = b is not None
        return Y

    @staticmethod
    def backward(ctx, gY):
        X, W = ctx.saved_tensors
        gY = _rowmajor(gY)
        lib = hip.lib()
        gX = gW = gb = None
        M = X.shape[0]
        if ctx.needs_input_grad[0]:
            gX = torch.empty_like(X)
            if M:
                _dgrad(gY, W, gX)
        if ctx.needs_input_grad[1] or ctx.has_bias:
            gW, gb = _wgrad(gY, X, W.shape[0], W.shape[1], ctx.has_bias)
        return gX, gW, gb


def linear(X, W, b=None):
    return LinearFn.apply(X, W, b)


class ColumnBlocksFn(torch.autograd.Function):
    """Y = X W^T + b computed ONCE for the fused first-layer columns of a layer, handed out as the consumers' column blocks
    (views with the wide row stride: every kernel of the C ABI takes a leading dimension).  The backward assembles the
    blocks' gradients into one wide buffer (one concatenation) and runs ONE data-gradient and ONE weight-gradient product --
    instead of a GEMM pair per consumer plus the accumulation of their input gradients (9 consumers of h per layer)."""

    @staticmethod
    def forward(ctx, X, W, b, sizes):
        X, W = _rowmajor(X), _rowmajor(W)
        Y = torch.empty(X.shape[0], W.shape[0], dtype=torch.float32, device=X.device)
        if X.shape[0]:
            _gemm_raw(X, W, Y, b.contiguous())
        ctx.save_for_backward(X, W)
        ctx.sizes = sizes
        return Y.split(sizes, 1)

    @staticmethod
    def backward(ctx, *gYs):
        X, W = ctx.saved_tensors
        M, (N, K) = X.shape[0], W.shape
        gY = torch.cat([g if g is not None else torch.zeros(M, n, dtype=torch.float32, device=X.device)
                        for g, n in zip(gYs, ctx.sizes)], 1)
        gX = torch.empty_like(X)
        if M:
            _dgrad(gY, W, gX)
        gW, gb = _wgrad(gY, X, N, K, True)
        return gX, gW, gb, None


def column_blocks(X, W, b, sizes):
    """blk(c0, c1) -> columns [c0, c1) of Y = X W^T + b; the ranges are the consecutive ones of the given sizes."""
    out, c0 = {}, 0
    for y, n in zip(ColumnBlocksFn.apply(X, W, b, tuple(sizes)), sizes):
        out[(c0, c0 + n)] = y
        c0 += n
    return lambda c0, c1: out[(c0, c1)]


class LinearGatherAddFn(torch.autograd.Function):
    """Y[r] = X[r] W^T + A1[i1[r]] (+ A2[i2[r]]): the per-edge half of a factored first layer plus the gathered per-node
    halves, in pg_gemm's epilogue (no materialised gathers)."""

    @staticmethod
    def forward(ctx, X, W, A1, i1, A2, i2, topo=None, kinds=None):
        """`topo`, `kinds` (a 'src' / 'dst' tag per gathered operand): i1 / i2 are the plan's bond_src / bond_dst, so the adjoint of
        the gathers is a walk over the edge-id table (pg_bond_rows_sum) instead of an atomic index_add_."""
        X, W, A1 = _rowmajor(X), _rowmajor(W), _rowmajor(A1)
        A2 = _rowmajor(A2) if A2 is not None else None
        M, K = X.shape
        Y = torch.empty(M, W.shape[0], dtype=torch.float32, device=X.device)
        g = hip.PgGemm()
        g.X, g.ldx, g.K1 = X.data_ptr(), X.stride(0), K
        g.W, g.ldw = W.data_ptr(), W.stride(0)
        g.add1, g.ld_add1, g.idx1 = A1.data_ptr(), A1.stride(0), i1.data_ptr()
        g.add_rows = A1.shape[0]
        if A2 is not None:
            g.add2, g.ld_add2, g.idx2 = A2.data_ptr(), A2.stride(0), i2.data_ptr()
            g.add_rows = max(A1.shape[0], A2.shape[0])
        g.out_scale, g.act = 1.0, hip.ACT_NONE
        g.Y, g.ldy, g.M, g.N = Y.data_ptr(), Y.stride(0), M, W.shape[0]
        if M:
            hip.check(hip.lib().pg_gemm(C.byref(g), _st()), 'pg_gemm')
        ctx.save_for_backward(X, W, i1, i2 if i2 is not None else i1)
        ctx.shapes = (A1.shape, None if A2 is None else A2.shape)
        ctx.topo, ctx.kinds = topo, kinds
        return Y

    @staticmethod
    def backward(ctx, gY):
        X, W, i1, i2 = ctx.saved_tensors
        gY = _rowmajor(gY)
        N, K = W.shape
        gX = torch.empty_like(X)
        if X.shape[0]:
            _dgrad(gY, W, gX)
        gW, _ = _wgrad(gY, X, N, K, False)
        s1, s2 = ctx.shapes

        def gathered_adjoint(shape, idx, kind):
            out = torch.zeros(shape, dtype=torch.float32, device=gY.device)
            if ctx.topo is not None and kind is not None and options.get('rows_sum') and shape[1] % 4 == 0 and \
                    gY.stride(0) % 4 == 0:
                hip.check(hip.lib().pg_bond_rows_sum(ctx.topo, gY.data_ptr(), gY.stride(0), shape[1], 1 if kind == 'src' else 0,
                                                     out.data_ptr(), out.stride(0), _st()), 'pg_bond_rows_sum')
                return out
            return out.index_add_(0, idx.long(), gY)
        kinds = ctx.kinds or (None, None)
        gA1 = gathered_adjoint(s1, i1, kinds[0])
        gA2 = None if s2 is None else gathered_adjoint(s2, i2, kinds[1])
        return gX, gW, gA1, None, gA2, None, None, None


def linear_gather_add(X, W, A1, i1, A2=None, i2=None, topo=None, kinds=None):
    return LinearGatherAddFn.apply(X, W, A1, i1, A2, i2, topo, kinds)


class LnReluFn(torch.autograd.Function):
    """ReLU(LayerNorm_128(X) * gamma + beta)  (middle of models/common.py:99-119 MLP)."""

    @staticmethod
    def forward(ctx, X, gamma, beta):
        X = _rowmajor(X)
        gamma, beta = gamma.contiguous(), beta.contiguous()
        Y = torch.empty(X.shape[0], 128, dtype=torch.float32, device=X.device)
        hip.check(hip.lib().pg_ln_relu(X.data_ptr(), X.stride(0), gamma.data_ptr(), beta.data_ptr(), X.shape[0],
                                       Y.data_ptr(), Y.stride(0), _st()), 'pg_ln_relu')
        ctx.save_for_backward(X, gamma, beta)
        return Y

    @staticmethod
    def backward(ctx, gY):
        X, gamma, beta = ctx.saved_tensors
        gY = _rowmajor(gY)
        gX = torch.empty_like(X)
        gg, gb = zero_pool.zeros_like(gamma), zero_pool.zeros_like(beta)
        hip.check(hip.lib().pg_ln_relu_bwd(X.data_ptr(), X.stride(0), gamma.data_ptr(), beta.data_ptr(), gY.data_ptr(),
                                           gY.stride(0), X.shape[0], gX.data_ptr(), gX.stride(0), gg.data_ptr(),
                                           gb.data_ptr(), _st()), 'pg_ln_relu_bwd')
        return gX, gg, gb


def mlp(X, W1, b1, g, b, W2, b2):
    """models/common.py:99-119 (Linear -> LayerNorm -> ReLU -> Linear) on 128 hidden channels."""
    return linear(LnReluFn.apply(linear(X, W1, b1), g, b), W2, b2)


def _fold(q, W2_l, ids, n_ids, out):
    hip.check(hip.lib().pg_attn_fold_query(q.data_ptr(), q.stride(0), W2_l.data_ptr(), n_ids, hip.ptr(ids),
                                           out.data_ptr(), _st()), 'pg_attn_fold_query')


def _unfold(S, swn, W2_l, b2, ids, n_ids, out):
    hip.check(hip.lib().pg_attn_unfold_value(S.data_ptr(), hip.ptr(swn), W2_l.data_ptr(), hip.ptr(b2), n_ids,
                                             hip.ptr(ids), out.data_ptr(), out.stride(0), _st()), 'pg_attn_unfold_value')


def _fold_wgrad(X, T, ids, n_ids, W2_l):
    gW = zero_pool.zeros_like(W2_l)
    hip.check(hip.lib().pg_attn_fold_wgrad(X.data_ptr(), X.stride(0), T.data_ptr(), n_ids, hip.ptr(ids), gW.data_ptr(),
                                           _st()), 'pg_attn_fold_wgrad')
    return gW


class FoldFn(torch.autograd.Function):
    """U[s][c][h] = sum_d q[s,8h+d] W2k[8h+d,c]: second key Linear folded into the query (rows `ids`)."""

    @staticmethod
    def forward(ctx, q, W2_l, ids, n_ids):
        q, W2_l = _rowmajor(q), W2_l.contiguous()
        # rows outside `ids` are never read by any kernel and no tensor op touches U / S / dU / dS: no fill needed
        U = torch.empty(q.shape[0], 2048, dtype=torch.float32, device=q.device)
        _fold(q, W2_l, ids, n_ids, U)
        ctx.save_for_backward(q, W2_l)
        ctx.ids, ctx.n_ids = ids, n_ids
        return U

    @staticmethod
    def backward(ctx, gU):
        q, W2_l = ctx.saved_tensors
        gU = gU.contiguous()
        gq = torch.zeros_like(q) if ctx.ids is not None else torch.empty_like(q)
        _unfold(gU, None, W2_l, None, ctx.ids, ctx.n_ids, gq)
        return gq, _fold_wgrad(q, gU, ctx.ids, ctx.n_ids, W2_l), None, None


class UnfoldFn(torch.autograd.Function):
    """out[s,8h+d] = sum_c W2v[8h+d,c] S[s][c][h] + b2v[8h+d] swn[s][h]: second value Linear after the aggregation."""

    @staticmethod
    def forward(ctx, S, swn, W2_l, b2, ids, n_ids):
        S, swn, W2_l, b2 = S.contiguous(), swn.contiguous(), W2_l.contiguous(), b2.contiguous()
        out = (torch.zeros if ids is not None else torch.empty)(S.shape[0], 128, dtype=torch.float32, device=S.device)
        _unfold(S, swn, W2_l, b2, ids, n_ids, out)
        ctx.save_for_backward(S, swn, W2_l, b2)
        ctx.ids, ctx.n_ids = ids, n_ids
        return out

    @staticmethod
    def backward(ctx, gout):
        S, swn, W2_l, b2 = ctx.saved_tensors
        gout = _rowmajor(gout)
        gS = torch.empty_like(S)
        _fold(gout, W2_l, ctx.ids, ctx.n_ids, gS)
        # bias side: gswn[s, h] = sum_d gout[s, 8h+d] b2[8h+d], gb2[c] = sum_s gout[s, c] swn[s, c >> 3] over the rows `ids` -- one
        # launch (five elementwise / reduction passes over [n, 128] in tensor ops; an indexed assignment of a Python scalar there
        # once cost 149 ms per step: it uploads the scalar with a BLOCKING copy)
        gswn = (torch.zeros_like if ctx.ids is not None else torch.empty_like)(swn)
        gb2 = zero_pool.zeros_like(b2)
        hip.check(hip.lib().pg_attn_unfold_bias_grad(gout.data_ptr(), gout.stride(0), swn.data_ptr(), b2.data_ptr(), ctx.n_ids,
                                                     hip.ptr(ctx.ids), gswn.data_ptr(), gb2.data_ptr(), _st()), 'pg_attn_unfold_bias_grad')
        return gS, gswn, _fold_wgrad(gout, S, ctx.ids, ctx.n_ids, W2_l), gb2, None, None


class SegCoreFn(torch.autograd.Function):
    """pg_seg_attn between the folded query and the unfolded value: (S, swn) for the feature-update modes, dx for the
    coordinate-update modes.  `cfg` carries the non-differentiable launch description."""

    @staticmethod
    def forward(ctx, cfg, Ydst, Ysrc, U, x, nrm, ew, Wf_k, Wf_v, bk, bv, W2xv_l, b2xv):
        """cfg['tri_fwd'] (triplet only): dict(q, W2k_l, W2v_l, b2v, Wg2_k, Wg2_v, G, seg_ids, seg_chunks) lets the forward run
        in the tuned triplet kernel (it folds q and smear(d_ji) itself); U / Ydst still feed the adjoint."""
        lib = hip.lib()
        pos = cfg['mode'] in (hip.SEG_KNN_POS, hip.SEG_BOND_POS)
        dev = Ydst.device
        Ydst, Ysrc, U = _rowmajor(Ydst), _rowmajor(Ysrc), U.contiguous()
        x = x.contiguous()
        tensors = dict(Ydst=Ydst, Ysrc=Ysrc, U=U, x=x, nrm=None if nrm is None else nrm.contiguous(),
                       ew=None if ew is None else ew.contiguous(), Wf_k=None if Wf_k is None else Wf_k.contiguous(),
                       Wf_v=None if Wf_v is None else Wf_v.contiguous(), bk=bk.contiguous(), bv=bv.contiguous(),
                       W2xv_l=None if W2xv_l is None else W2xv_l.contiguous(),
                       b2xv=None if b2xv is None else b2xv.contiguous())
        n_rows = cfg['n_out_rows']
        full = cfg['seg_ids'] is None and cfg['n_seg'] == n_rows        # every output row is written by the kernel
        alloc = torch.empty if full else torch.zeros
        if pos:
            out = (alloc(n_rows, 3, dtype=torch.float32, device=dev),)
        else:
            out = (torch.empty(n_rows, 2048, dtype=torch.float32, device=dev), alloc(n_rows, 16, dtype=torch.float32, device=dev))
        s = SegCoreFn._struct(cfg, tensors)
        if pos:
            s.dx, s.accumulate_dx = out[0].data_ptr(), 0
        else:
            s.S, s.swn = out[0].data_ptr(), out[1].data_ptr()
        tf = cfg.get('tri_fwd')
        alpha = None
        onepass = options.get('tri_onepass')
        ph_onepass = cfg['mode'] == hip.SEG_PHORE and onepass and options.get('ph_onepass') and 0 < cfg['max_rows'] <= 256
        if ph_onepass or cfg['mode'] in (hip.SEG_KNN_NODE, hip.SEG_BOND_NODE, hip.SEG_KNN_POS, hip.SEG_BOND_POS) and onepass and \
                (cfg['k'] <= 32 if cfg['mode'] in (hip.SEG_KNN_NODE, hip.SEG_KNN_POS) else cfg['max_rows'] <= 80):
            # the two-pass node kernels run (csrc/node_attn.hip): they hand alpha x gate to a one-pass adjoint (node update), or
            # the logits and value scalars of every row to the adjoint's softmax step (position update: 32 floats per row); the
            # pharmacophore encoder's generic kernel leaves its softmax weights the same way
            arows = (cfg['max_rows'] + 15) // 16 * 16
            alpha = torch.empty(n_rows * arows * (32 if pos else 16), dtype=torch.float32, device=dev)
            s.alpha, s.alpha_rows = alpha.data_ptr(), arows
            ctx.alpha_rows = arows
        if tf is not None:
            for k in ('q', 'W2k_l', 'W2v_l', 'b2v', 'Wg2_k', 'Wg2_v', 'G', 'seg_ids', 'seg_chunks'):
                setattr(s, k, tf[k].data_ptr())
            if tf.get('n_tri_iters') and options.get('tri_staged'):
                # source-atom groups of the plan: the LDS-staged kernel (csrc/triplet2.hip, training form) takes the launch
                s.tri_iters, s.n_tri_iters, s.tri_counter = tf['tri_iters'].data_ptr(), tf['n_tri_iters'], tf['tri_counter'].data_ptr()
            # the tuned kernel runs: it can hand the softmax weights to the one-wave-per-tile adjoint (the channel-split adjoint,
            # options.tri_bwd_form, recomputes them and reads nothing of the forward back)
            if cfg['max_rows'] <= 80 and onepass and not (options.get('tri_bwd_form') and cfg['max_rows'] <= 64):
                arows = (cfg['max_rows'] + 15) // 16 * 16
                alpha = torch.empty(cfg['n_seg'] * arows * 16, dtype=torch.float32, device=dev)
                s.alpha, s.alpha_rows = alpha.data_ptr(), arows
                ctx.alpha_rows = arows
        ctx.has_alpha = alpha is not None
        if alpha is not None:
            ctx.save_for_backward(alpha, *(() if pos else out))      # (the adjoint reads S / swn of the node modes; dx of a position mode is not needed)
        hip.check(lib.pg_seg_attn(cfg['topo'], C.byref(s), _st()), 'pg_seg_attn')
        ctx.cfg, ctx.tensors, ctx.pos = cfg, tensors, pos
        return out if not pos else out[0]

    @staticmethod
    def _struct(cfg, t):
        s = hip.PgSegAttn()
        s.mode, s.n_seg, s.seg_ids, s.knn_k = cfg['mode'], cfg['n_seg'], hip.ptr(cfg['seg_ids']), cfg['k']
        s.x, s.nrm, s.ew = t['x'].data_ptr(), hip.ptr(t['nrm']), hip.ptr(t['ew'])
        s.nbr, s.deg = hip.ptr(cfg.get('nbr')), hip.ptr(cfg.get('deg'))
        Yd, Ys = t['Ydst'], t['Ysrc']
        s.Cdst_k, s.Cdst_v, s.ld_cdst = Yd.data_ptr(), Yd.data_ptr() + 128 * 4, Yd.stride(0)
        s.Csrc_k, s.Csrc_v, s.ld_csrc = Ys.data_ptr(), Ys.data_ptr() + 128 * 4, Ys.stride(0)
        s.Wf_k, s.Wf_v = hip.ptr(t['Wf_k']), hip.ptr(t['Wf_v'])
        s.ln_gk, s.ln_bk, s.ln_gv, s.ln_bv = (t['bk'].data_ptr(), t['bk'].data_ptr(), t['bv'].data_ptr(),
                                              t['bv'].data_ptr())
        s.U = t['U'].data_ptr()
        s.W2xv_l, s.b2xv = hip.ptr(t['W2xv_l']), hip.ptr(t['b2xv'])
        return s

    @staticmethod
    def backward(ctx, *gouts):
        lib, cfg, t, pos = hip.lib(), ctx.cfg, ctx.tensors, ctx.pos
        dev = t['x'].device
        z = lambda ref: None if ref is None else torch.zeros_like(ref)
        full = cfg['seg_ids'] is None and cfg['n_seg'] == cfg['n_out_rows']
        gYdst = torch.empty_like(t['Ydst']) if full else z(t['Ydst'])
        gU = torch.empty_like(t['U'])
        # bond / triplet modes store every dCsrc row exactly once; the knn / phore modes accumulate with atomics
        gYsrc = torch.empty_like(t['Ysrc']) if (full and cfg['mode'] == hip.SEG_TRIPLET) else z(t['Ysrc'])
        # the small gradients (weights, coordinates, direction vectors, gate) share ONE zero-filled buffer: one fill launch instead of nine
        small = [t[k] for k in ('Wf_k', 'Wf_v', 'bk', 'bv', 'W2xv_l', 'b2xv')] + \
                [t['x'] if cfg['need_gx'] else None, t['nrm'] if cfg['need_gx'] else None, t['ew']]
        pad4 = lambda k: (k + 3) // 4 * 4                      # (every view starts on a 16-byte boundary)
        buf = zero_pool.zeros(sum(pad4(x.numel()) for x in small if x is not None), device=dev)
        views, off = [], 0
        for x in small:
            if x is None:
                views.append(None)
            else:
                views.append(buf[off:off + x.numel()].view(x.shape))
                off += pad4(x.numel())
        gWf_k, gWf_v, gbk, gbv, gW2, gb2, gx, gnrm, gew = views
        g = hip.PgSegAttnGrad()
        keep = []
        if pos:
            gdx = gouts[0].contiguous()
            g.gdx = gdx.data_ptr()
            keep.append(gdx)
        else:
            gS = gouts[0].contiguous() if gouts[0] is not None else torch.zeros(cfg['n_out_rows'], 2048, device=dev)
            gsw = gouts[1].contiguous() if gouts[1] is not None else torch.zeros(cfg['n_out_rows'], 16, device=dev)
            g.gS, g.gswn = gS.data_ptr(), gsw.data_ptr()
            keep += [gS, gsw]
        g.gU = gU.data_ptr()
        g.gCdst_k, g.gCdst_v, g.ld_gcdst = gYdst.data_ptr(), gYdst.data_ptr() + 128 * 4, gYdst.stride(0)
        g.gCsrc_k, g.gCsrc_v, g.ld_gcsrc = gYsrc.data_ptr(), gYsrc.data_ptr() + 128 * 4, gYsrc.stride(0)
        g.gWf_k, g.gWf_v, g.gbk, g.gbv = hip.ptr(gWf_k), hip.ptr(gWf_v), gbk.data_ptr(), gbv.data_ptr()
        g.gW2xv_l, g.gb2xv = hip.ptr(gW2), hip.ptr(gb2)
        g.gx, g.gnrm, g.gew = hip.ptr(gx), hip.ptr(gnrm), hip.ptr(gew)
        waves = lib.pg_seg_attn_bwd_waves(cfg['mode'])
        # one persistent workgroup per CU (the adjoints hold a CU's LDS / registers alone): every workgroup stages its weight tables and
        # flushes its weight-gradient accumulators (hundreds of atomics per wave) ONCE, and no partial last round of workgroups is
        # left -- training step 179.5 ms with 1 024 workgroups, 171-173 with 512, 168 with 256 (320: 205, 8 192: 206)
        grid = max(1, min((cfg['n_seg'] + 3) // 4, options.get('bwd_grid')))          # (the forms' smallest workgroup has 4 waves)
        rows = (cfg['max_rows'] + 15) // 16 * 16
        if cfg['mode'] == hip.SEG_TRIPLET and options.get('tri_bwd_form') and cfg['max_rows'] <= 64:
            # the channel-split form (csrc/triplet_bwd2.hip) keeps its rows in LDS: no row buffer (the library rejects a launch that would need one)
            g.tri_form = int(options.get('tri_bwd_form'))
            grid = max(1, min(cfg['n_seg'], options.get('tri_bwd_grid')))
            g.rowbuf, g.rowbuf_rows, g.grid = None, rows, grid
        else:
            rowbuf = torch.empty(grid * waves * rows * 48, dtype=torch.float32, device=dev)
            g.rowbuf, g.rowbuf_rows, g.grid = rowbuf.data_ptr(), rows, grid
            keep.append(rowbuf)
        if cfg['mode'] == hip.SEG_TRIPLET and cfg.get('plan') is not None and options.get('bwd_atom_sort'):
            ao = cfg['plan'].bwd_atom_order(grid)          # cost-sorted source atoms, dealt out in a snake (levels the persistent workgroups)
            g.atom_order = ao.data_ptr()
            keep.append(ao)
        if getattr(ctx, 'has_alpha', False):
            a_ = ctx.saved_tensors[0]
            g.alpha, g.alpha_rows = a_.data_ptr(), ctx.alpha_rows
            keep.append(a_)
            if not pos:
                S_, sw_ = ctx.saved_tensors[1:3]
                g.S, g.swn = S_.data_ptr(), sw_.data_ptr()
                keep += [S_, sw_]
            # (the pharmacophore encoder's adjoint stays one pass: as a value pass + a key pass it measured 0.4 ms per step slower)
            split_modes = {'knn': (hip.SEG_KNN_NODE, hip.SEG_KNN_POS), 'all': (hip.SEG_TRIPLET, hip.SEG_KNN_NODE, hip.SEG_KNN_POS)}
            # (the library takes the triplet's two-pass form for ligands of up to 64 atoms only: no scratch for a launch that will not use it)
            if cfg['mode'] in split_modes.get(options.get('bwd_split'), ()) and (cfg['mode'] != hip.SEG_TRIPLET or cfg['max_rows'] <= 64):
                # scratch of the two-pass form (value pass, then key pass): d logit (position update: one value per row) and the value
                # pass's d feat rows, indexed like the forward's per-row record
                n_rows_rec = a_.numel() // (32 if pos else 16)
                dl_ = torch.empty(n_rows_rec * 16, dtype=torch.float32, device=dev)
                gf_ = torch.empty(n_rows_rec * (16 if cfg['mode'] == hip.SEG_TRIPLET else 48), dtype=torch.float32, device=dev)
                g.dlogit, g.gfeat_v = dl_.data_ptr(), gf_.data_ptr()
                keep += [dl_, gf_]
        s = SegCoreFn._struct(cfg, t)
        if bwd_timers is not None:       # measurement (tools/bench_train.py): HIP events around the adjoint launch, on its stream
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        hip.check(lib.pg_seg_attn_bwd(cfg['topo'], C.byref(s), C.byref(g), _st()), 'pg_seg_attn_bwd')
        if bwd_timers is not None:
            ev[1].record()
            bwd_timers.setdefault(cfg['mode'], []).append(ev)
        # first-layer blocks: the k|v target halves live in Ydst[:, 0:256], the source halves in Ysrc[:, 0:256]
        return (None, gYdst, gYsrc, gU, gx, gnrm, gew, gWf_k, gWf_v, gbk, gbv, gW2, gb2)


bwd_timers = None       # dict mode -> [(start, end) events] when a benchmark wants the adjoint launches timed
lib_timers = None       # list of (start, end) events around every LIBRARY GEMM of the step (the input gradients, `_dgrad`): the one place
                        # where the training path leaves the in-tree kernels -- benchmarks report it as `library_ms`


def seg_core(cfg, Ydst, Ysrc, U, x, nrm=None, ew=None, Wf_k=None, Wf_v=None, bk=None, bv=None, W2xv_l=None, b2xv=None):
    return SegCoreFn.apply(cfg, Ydst, Ysrc, U, x, nrm, ew, Wf_k, Wf_v, bk, bv, W2xv_l, b2xv)


# ------------------------------------------------------------------------------------------------------------------
# model
# ------------------------------------------------------------------------------------------------------------------
def shifted_softplus(x):
    return F.softplus(x) - 0.6931471805599453


_smear_off_cache = {}


def gaussian_smearing(d):
    off = _smear_off_cache.get(d.device)
    if off is None:
        off = _smear_off_cache[d.device] = torch.tensor(_SMEAR_OFF, dtype=torch.float32, device=d.device)
    return torch.exp(-0.5 * (d.unsqueeze(-1) - off) ** 2)


def rows(x, idx):
    """x[idx] for a (possibly multi-dimensional) long index: index_select, whose backward is an atomic index_add (advanced
    indexing would go through a sort-based index_put: dozens of small kernels per call)."""
    return x.index_select(0, idx.reshape(-1)).view(*idx.shape, *x.shape[1:])


class TrainForward:
    """Differentiable PhoreDiff.forward on one batch plan (same launch order as engine.Engine, autograd-composed)."""

    def __init__(self, params, plan, knn_k=32, num_layers=6, ex_col=12, denoiser_only=False):
        self.lib = hip.lib()
        self.sd, self.plan, self.k, self.L, self.ex_col = params, plan, knn_k, num_layers, ex_col
        if denoiser_only:       # the denoiser module on its own (models/uni_denoiser.py forward under autograd): its layers, nothing else
            from .packing import LayerPack
            self.pack = type('DenoiserPack', (), {})()
            self.pack.layers = LayerPack(params, [f'denoiser.base_block.{l}' for l in range(num_layers)]).unstack(num_layers)
        else:
            self.pack = ModelPack(params, num_layers, detach=False)
        self.dev = plan.device

    # -- one attention sub-layer. `blk(c0, c1)` yields columns [c0, c1) of its first-layer blocks
    #    (k_dst | v_dst | k_src | v_src | q_hid = 640 columns) as a tensor of its own: one GEMM per consumer instead of
    #    views of one wide GEMM, so autograd never pads a slice gradient back to the wide shape
    def _attention(self, mode, a, blk, x, dst_lists, Ysrc=None, nrm=None, ew=None, nbr=None, deg=None, max_rows=None,
                   need_gx=True):
        p = self.plan
        pos = mode in (hip.SEG_KNN_POS, hip.SEG_BOND_POS)
        knn = mode in (hip.SEG_KNN_NODE, hip.SEG_KNN_POS)
        n = p.n_ctx
        q = linear(LnReluFn.apply(blk(512, 640), a.q_ln_g, a.q_ln_b), a.W2q, a.b2q) * HEAD_SCALE
        Ydst = blk(0, 256)
        if Ysrc is None:
            Ysrc = blk(256, 512)
        total = None
        for seg_ids, n_seg, is_lig in dst_lists:
            if n_seg == 0:
                continue
            U = FoldFn.apply(q, a.W2k_l, seg_ids, n_seg)
            cfg = dict(mode=mode, n_seg=n_seg, seg_ids=seg_ids, k=self.k, topo=p.topo_ref, nbr=nbr, deg=deg,
                       n_out_rows=n, max_rows=max_rows, need_gx=need_gx)
            kw = {}
            if knn:
                kw.update(Wf_k=a.Wf_k[is_lig], Wf_v=a.Wf_v[is_lig])
            elif mode == hip.SEG_PHORE:
                kw.update(Wf_k=a.Wf_k, Wf_v=a.Wf_v)
            if pos:
                out = seg_core(cfg, Ydst, Ysrc, U, x, nrm, ew, bk=a.ln_bk, bv=a.ln_bv, W2xv_l=a.W2xv_l, b2xv=a.b2xv, **kw)
            else:
                S, swn = seg_core(cfg, Ydst, Ysrc, U, x, nrm, ew, bk=a.ln_bk, bv=a.ln_bv, **kw)
                out = UnfoldFn.apply(S, swn, a.W2v_l, a.b2v, seg_ids, n_seg)
            total = out if total is None else total + out
        return total

    def phore_encode(self, h_phore, pos_phore):
        """diffusion.py:186-191: embedding -> p x p attention (output replaces the embedding)."""
        p, pk = self.plan, self.pack
        n = p.n_ctx
        h_ctx = torch.zeros(n, h_phore.shape[1], dtype=torch.float32, device=self.dev).index_copy(0, p.phore2ctx_long, h_phore.float())
        x_ctx = torch.zeros(n, 3, dtype=torch.float32, device=self.dev).index_copy(0, p.phore2ctx_long, pos_phore.float())
        hp = linear(h_ctx, pk.W_pe, pk.b_pe)
        Yp = column_blocks(hp, pk.W_ph, pk.b_ph, PHORE_PARTS)
        max_rows = int(p.g_nph.max()) if p.n_graphs else 0
        enc = self._attention(hip.SEG_PHORE, pk.PH, Yp, x_ctx, [(p.phore2ctx, p.n_phore, False)], max_rows=max_rows,
                              need_gx=False)
        return enc.index_select(0, p.phore2ctx_long)

    def atom_count(self, hp_emb, h_phore, batch_phore):
        """diffusion.py:148-163."""
        B = self.plan.n_graphs
        sd = self.sd

        def head(name, xx):
            return torch.sigmoid(linear(torch.relu(linear(xx, sd[name + '.0.weight'], sd[name + '.0.bias'])),
                                        sd[name + '.2.weight'], sd[name + '.2.bias']))

        def seg_mean(v, idx):
            c = torch.bincount(idx, minlength=B).clamp(min=1).unsqueeze(-1).to(v.dtype)
            return torch.zeros(B, v.shape[1], dtype=v.dtype, device=v.device).index_add(0, idx, v) / c
        c_all = seg_mean(head('atom_mlp', hp_emb), batch_phore)
        m = (h_phore[:, self.ex_col] != 1).to(hp_emb.dtype).unsqueeze(-1)        # non-EX nodes (diffusion.py:152-155)
        cnt = torch.zeros(B, 1, dtype=hp_emb.dtype, device=hp_emb.device).index_add(0, batch_phore, m).clamp(min=1)
        c_l = torch.zeros(B, 1, dtype=hp_emb.dtype, device=hp_emb.device).index_add(0, batch_phore, head('atom_mlp_1', hp_emb) * m) / cnt
        return c_l, c_l + F.relu(c_all - c_l)

    def time_smearing(self, t):
        off, coeff = self.sd['time_emb.0.offset'], self.sd['time_emb.0.coeff']
        return torch.exp(coeff * (t.float().unsqueeze(-1) - off.view(1, -1)) ** 2)

    def forward(self, h_node_pert, pos_pert, h_edge_pert, time_step, h_phore, pos_phore, phore_norm, batch_phore):
        zero_pool.begin_step()
        p, pk, sd, dev, lib = self.plan, self.pack, self.sd, self.dev, self.lib
        n, E = p.n_ctx, p.n_bond
        hp_emb = self.phore_encode(h_phore, pos_phore)
        counts = self.atom_count(hp_emb, h_phore, batch_phore)
        # embeddings (diffusion.py:180-183,205) and the ctx order of compose_context (common.py:180-208)
        h_lig = torch.cat([linear(h_node_pert.float(), sd['node_embedder.weight']), self.time_smearing(time_step[p.batch_node])], -1)
        hb = torch.cat([linear(h_edge_pert.float(), sd['edge_embedder.weight']), self.time_smearing(time_step[p.batch_edge])], -1)
        if not p.edge_identity:        # bond rows live in the plan's internal (target-major) order inside the network
            hb = hb.index_select(0, p.edge_ref_long)
        h = torch.zeros(n, 128, dtype=torch.float32, device=dev).index_copy(0, p.lig2ctx_long, h_lig)
        h = h.index_copy(0, p.phore2ctx_long, hp_emb)
        x = torch.zeros(n, 3, dtype=torch.float32, device=dev).index_copy(0, p.lig2ctx_long, pos_pert.float())
        x = x.index_copy(0, p.phore2ctx_long, pos_phore.float())
        nrm_ph = torch.zeros(n, 3, dtype=torch.float32, device=dev).index_copy(0, p.phore2ctx_long, phore_norm.float())
        h, x, hb = self.denoise(h, x, hb, nrm_ph)
        v0, b0 = pk.v0, pk.b0
        v = linear(shifted_softplus(linear(h.index_select(0, p.lig2ctx_long), v0[0], v0[1])), v0[2], v0[3])
        bond = linear(shifted_softplus(linear(hb, b0[0], b0[1])), b0[2], b0[3])
        if not p.edge_identity:
            bond = bond.index_select(0, p.edge_int_long)                    # back to the caller's edge order
        return v, x.index_select(0, p.lig2ctx_long), bond, counts

    def denoise(self, h, x, hb, nrm_ph):
        """UniTransformerO2TwoUpdateGeneralBond.forward (uni_denoiser.py:396-430) on ctx-ordered h [n,128], x [n,3], bond rows hb [E,128] in
        the plan's internal order, pharmacophore normals in ctx rows: knn graph + gate once, then the layers.  -> (h, x, hb)."""
        p, pk, sd, dev, lib = self.plan, self.pack, self.sd, self.dev, self.lib
        n, E = p.n_ctx, p.n_bond
        is_lig = p.ctx_is_lig.bool().unsqueeze(-1)

        # knn graph + global edge gate, once per forward (uni_denoiser.py:396-415)
        nbr = torch.zeros(n, self.k, dtype=torch.int32, device=dev)
        deg = torch.zeros(n, dtype=torch.int32, device=dev)
        x0 = x.detach().contiguous()
        hip.check(lib.pg_knn_ctx(p.topo_ref, x0.data_ptr(), self.k, nbr.data_ptr(), deg.data_ptr(), _st()), 'pg_knn_ctx')
        slot_ok = torch.arange(self.k, device=dev).view(1, -1) < deg.view(-1, 1)
        nbr_safe = torch.where(slot_ok, nbr.long(), torch.arange(n, device=dev).view(-1, 1).expand(-1, self.k))
        dist = (x.unsqueeze(1) - rows(x, nbr_safe)).pow(2).sum(-1).clamp(min=1e-24).sqrt()
        gp = 'denoiser.edge_pred_layer.net.'
        ew = torch.sigmoid(mlp(gaussian_smearing(dist.reshape(-1)), sd[gp + '0.weight'], sd[gp + '0.bias'], sd[gp + '1.weight'],
                               sd[gp + '1.bias'], sd[gp + '3.weight'], sd[gp + '3.bias'])).view(n, self.k)
        ew = ew * slot_ok.to(ew.dtype)

        lig = [(p.lig2ctx, p.n_lig, True)]
        both = [(p.lig2ctx, p.n_lig, True), (p.phore2ctx, p.n_phore, False)]
        max_lig = int(p.num_atoms.max()) if p.n_graphs else 0
        bsrc, bdst = p.bond_src.long(), p.bond_dst.long()
        for L in pk.layers:
            # direction vectors: mean of the 3 nearest ligand atoms - x (common.py:300-304), file normals for phore nodes
            nn3 = torch.full((p.n_lig, 3), -1, dtype=torch.int32, device=dev)
            xc = x.detach().contiguous()
            hip.check(lib.pg_lig_nn3(p.topo_ref, xc.data_ptr(), nn3.data_ptr(), _st()), 'pg_lig_nn3')
            ok = (nn3 >= 0).to(x.dtype).unsqueeze(-1)
            nsum = (rows(x, nn3.clamp(min=0).long()) * ok).sum(1)
            l_norm = nsum / ok.sum(1).clamp(min=1.0) - x.index_select(0, p.lig2ctx_long)
            nrm = nrm_ph.index_copy(0, p.lig2ctx_long, l_norm)
            G = gaussian_smearing((x.index_select(0, bsrc) - x.index_select(0, bdst)).pow(2).sum(-1).clamp(min=1e-24).sqrt())   # [E,20]

            # first-layer blocks of the three feature sub-layers (15 x 128 columns of W_node1), one GEMM per consumer
            if options.get('wide_gemm'):      # one wide GEMM, one adjoint pair (ColumnBlocksFn)
                Y1 = column_blocks(h, L.W_node1, L.b_node1, NODE1_PARTS)
            else:                                               # A/B knob: one GEMM (and one adjoint pair) per consumer
                Y1 = lambda c0, c1: linear(h, *L.node1_parts[(c0, c1)])
            aggE = self._attention(hip.SEG_KNN_NODE, L.NE, Y1, x, both, nrm=nrm, ew=ew, nbr=nbr, deg=deg, max_rows=self.k)
            CsB = linear_gather_add(hb, L.NB.W_hb, Y1(7 * 128, 9 * 128), p.bond_src, topo=p.topo_ref, kinds=('src',))
            aggB = self._attention(hip.SEG_BOND_NODE, L.NB, lambda c0, c1: Y1(640 + c0, 640 + c1), x, lig, Ysrc=CsB,
                                   max_rows=max_lig)
            # bond update over triplets (uni_denoiser.py:101-165)
            a = L.TB
            P = linear_gather_add(torch.cat([hb, G], -1), a.W_hbg, Y1(10 * 128, 12 * 128), p.bond_src,
                                  Y1(12 * 128, 14 * 128), p.bond_dst, topo=p.topo_ref, kinds=('src', 'dst'))
            Q = linear(G, a.W_g2)                                                      # smear(d_ji) columns, per segment
            qhid = linear_gather_add(hb, a.W_q_hb, Y1(14 * 128, 15 * 128), p.bond_dst, topo=p.topo_ref, kinds=('dst',))
            qT = linear(LnReluFn.apply(qhid, a.q_ln_g, a.q_ln_b), a.W2q, a.b2q) * HEAD_SCALE
            U = FoldFn.apply(qT, a.W2k_l, None, E)
            cfg = dict(mode=hip.SEG_TRIPLET, n_seg=E, seg_ids=None, k=self.k, topo=p.topo_ref, plan=p, n_out_rows=E,
                       max_rows=max_lig, need_gx=True,
                       tri_fwd=dict(q=qT.detach(), W2k_l=a.W2k_l.detach(), W2v_l=a.W2v_l.detach(), b2v=a.b2v.detach(),
                                    Wg2_k=a.Wg2_k.detach(), Wg2_v=a.Wg2_v.detach(), G=G.detach().contiguous(),
                                    seg_ids=p.tri_order, seg_chunks=p.tri_chunks, tri_iters=p.tri_iters,
                                    n_tri_iters=p.n_tri_iters, tri_counter=p.tri_counter))
            S, swn = seg_core(cfg, Q, P, U, x, Wf_k=a.Wf_k, Wf_v=a.Wf_v, bk=a.ln_bk, bv=a.ln_bv)
            hb_new = hb + UnfoldFn.apply(S, swn, a.W2v_l, a.b2v, None, E)
            h_new = h + linear(aggE + aggB, L.W_lin, L.b_lin)
            # coordinate updates from h', h_bond' and the old geometry (uni_denoiser.py:291-296)
            if options.get('wide_gemm'):
                Y2 = column_blocks(h_new, L.W_node2, L.b_node2, NODE2_PARTS)
            else:
                Y2 = lambda c0, c1: linear(h_new, *L.node2_parts[(c0, c1)])
            dxe = self._attention(hip.SEG_KNN_POS, L.PE, Y2, x, lig, nrm=nrm, ew=ew, nbr=nbr, deg=deg, max_rows=self.k)
            CsB2 = linear_gather_add(hb_new, L.PB.W_hb, Y2(7 * 128, 9 * 128), p.bond_src, topo=p.topo_ref, kinds=('src',))
            dxb = self._attention(hip.SEG_BOND_POS, L.PB, lambda c0, c1: Y2(640 + c0, 640 + c1), x, lig, Ysrc=CsB2,
                                  max_rows=max_lig)
            x = x + (dxe + dxb) * is_lig.to(x.dtype)
            h, hb = h_new, hb_new
        return h, x, hb
