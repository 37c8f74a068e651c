"""Weight layouts for the HIP kernels, derived from the reference-schema state_dict.

First-layer column maps (validated numerically there):
  knn  k/v/xk/xv : [ type(x)smear 0:80 | type 80:84 | dir 84:93 | h_dst 93:221 | h_src 221:349 ]
  bond k/v/xk/xv : [ h_bond 0:128 | h_dst 128:256 | h_src 256:384 ]
  triplet k/v    : [ h_bond_kj 0:128 | smear(d_kj) 128:148 | smear(d_ji) 148:168 | ang 168:181 | h_k 181:309 | h_j 309:437 ]
  triplet q      : [ h_bond_ji 0:128 | h_i 128:256 ]
  phore k/v      : [ dist 0:1 | h_dst 1:129 | h_src 129:257 ]
"""
import math

import torch

# fixed offsets of every GaussianSmearing on the path (models/common.py:18; fix_offset=True)
_SMEAR_OFF = (0., 1., 1.25, 1.5, 1.75, 2., 2.25, 2.5, 2.75, 3., 3.5, 4., 4.5, 5., 5.5, 6., 7., 8., 9., 10.)

# queries are pre-scaled by 1/sqrt(head_dim) (np.sqrt(k.shape[-1]) = sqrt(8), uni_denoiser.py:62,158,204) and by
# log2(e): the kernels' segment softmax runs in base 2 (one v_exp_f32 per weight), which is the same softmax
HEAD_SCALE = math.log2(math.e) / math.sqrt(8.0)
# LayerNorm channels with |gamma| below this are constant ReLU(beta) channels (see _kv_mlp): beta / 1e-25 is still far
# inside the fp32 range for any |beta| < 1e12, and a channel that small contributes < 1e-25 * |x_hat| anyway
DEAD_GAMMA = 1e-25


def _lane():
    lane = torch.arange(64)
    return lane >> 4, lane & 15


_index_cache = {}


def _cached(key, device, build):
    """Constant gather indices of the lane-fixed layouts, built once per device (the training path packs every step)."""
    k = (key, str(device))
    if k not in _index_cache:
        _index_cache[k] = build().to(device)
    return _index_cache[k]


def _gather_last2(W, flat_idx, shape):
    """out[..., i] = W[..., r_i, c_i] for a constant list flat_idx = r * n_cols + c over the LAST TWO dims, any leading (layer)
    dims: index_select, whose backward is one index_add (advanced indexing would run a sort-based index_put)."""
    lead = W.shape[:-2]
    return W.reshape(*lead, W.shape[-2] * W.shape[-1]).index_select(-1, flat_idx).view(*lead, *shape)


def lane_fixed_w2(W2):
    """[..., 128 (8h+d), 128 (c)] -> [..., 64][64 lanes][4]: element n=i*4+j -> tau=n>>5, r=(n>>3)&3, d=n&7;
    lane=(g,h) holds W2[8h+d][16 tau + 4g + r]."""
    def build():
        g, h = _lane()
        n = (4 * torch.arange(64)[:, None, None] + torch.arange(4)[None, None, :]).expand(64, 64, 4)      # [i][lane][j]
        tau, r, d = n >> 5, (n >> 3) & 3, n & 7
        rows, cols = 8 * h[None, :, None] + d, 16 * tau + 4 * g[None, :, None] + r
        return (rows * 128 + cols).reshape(-1)
    return _gather_last2(W2, _cached('w2', W2.device, build), (64, 64, 4))


def lane_fixed_feat(Wfeat):
    """[..., 128 (c), F] -> [..., F/4][8][64]: value[step][tau][lane=(g,m)] = Wfeat[16 tau + m][4 step + g]."""
    F = Wfeat.shape[-1]
    assert F % 4 == 0
    steps = F // 4

    def build():
        g, m = _lane()
        c = (16 * torch.arange(8)[:, None] + m[None, :])[None].expand(steps, -1, -1)
        f = (4 * torch.arange(steps)[:, None, None] + g[None, None, :]).expand(-1, 8, -1)
        return (c * F + f).reshape(-1)
    return _gather_last2(Wfeat, _cached(('feat', steps), Wfeat.device, build), (steps, 8, 64))


def lane_fixed_xv(W2xv):
    """[..., 16 (h), 128 (c)] -> [..., 32 (tau*4+r)][64]: lane=(g,h) holds W2xv[h][16 tau + 4g + r]."""
    def build():
        g, h = _lane()
        i = torch.arange(32)
        tau, r = i >> 2, i & 3
        return (h[None, :].expand(32, -1) * 128 + 16 * tau[:, None] + 4 * g[None, :] + r[:, None]).reshape(-1)
    return _gather_last2(W2xv, _cached('xv', W2xv.device, build), (32, 64))


def _param(sd, p, suffix):
    """One parameter (`p` a prefix) or the same parameter of several layers stacked along a new leading dim (`p` a list of
    prefixes): every function below works on the last one / two dims, so the six denoiser layers are packed by ONE pass of
    tensor ops in the training path (which packs, and differentiates through the packing, every step)."""
    if isinstance(p, str):
        return sd[p + suffix]
    return torch.stack([sd[q + suffix] for q in p])


def _mlp(sd, p, name=''):
    return dict(W1=_param(sd, p, name + '.net.0.weight'), b1=_param(sd, p, name + '.net.0.bias'),
                g=_param(sd, p, name + '.net.1.weight').contiguous(), b=_param(sd, p, name + '.net.1.bias').contiguous(),
                W2=_param(sd, p, name + '.net.3.weight'), b2=_param(sd, p, name + '.net.3.bias'))


def _kv_mlp(sd, p, name=''):
    """Key/value MLP rewritten so that the kernels' LayerNorm+ReLU costs 3 VALU ops per element instead of 6
    (exact in real arithmetic):
      * first layer centred over its 128 outputs (W1 - mean_rows, b1 - mean)  =>  hidden has zero mean, the
        LayerNorm mean pass disappears (the mean of a linear map is the map with averaged weights);
      * ReLU(g*x_hat + b) = |g| * ReLU(s*x_hat + b/|g|), s = sign(g): s goes into the first layer (row signs),
        |g| into the COLUMNS of the second Linear, and with sigma = 1/rstd
        ReLU(s*x_hat + b') = rstd * ReLU(s*x + b'*sigma), the per-row rstd is applied to the 16 logits /
        attention weights of the row instead of its 128 channels.
    Dead channels (gamma == 0, or |gamma| below DEAD_GAMMA where beta/|gamma| would leave the fp32 range): the channel's
    activation is the constant ReLU(beta), so its column of the second Linear is zeroed and W2[:, c] * ReLU(beta_c)
    moves into the second bias -- the exact reference result.  (The channel still takes part in the LayerNorm
    statistics, which are computed before gamma.)  |gamma| = gamma * sign(gamma) keeps d|gamma|/dgamma = sign for training;
    a dead channel gets no gradient through gamma (the kernels never see it).
    Returns W1', b1' (centred, sign-normalised), bp = b/|g|, W2' = W2 * |g| (columns), b2' (b2 + dead-channel constants)."""
    m = _mlp(sd, p, name)
    W1c = m['W1'] - m['W1'].mean(-2, keepdim=True)
    b1c = m['b1'] - m['b1'].mean(-1, keepdim=True)
    sgn = torch.where(m['g'] < 0, -torch.ones_like(m['g']), torch.ones_like(m['g']))
    dead = (m['g'].detach().abs() < DEAD_GAMMA)
    ag = torch.where(dead, torch.ones_like(m['g']), m['g'] * sgn)
    live = (~dead).to(m['g'].dtype)
    const = torch.relu(m['b']) * (1.0 - live)
    b2 = m['b2'] + (m['W2'] @ const if const.dim() == 1 else (m['W2'] @ const.unsqueeze(-1)).squeeze(-1))
    return dict(W1=(sgn.unsqueeze(-1) * W1c).contiguous(), b1=(sgn * b1c).contiguous(), bp=(m['b'] / ag * live).contiguous(),
                W2=(m['W2'] * (ag * live).unsqueeze(-2)).contiguous(), b2=b2)


_KNN_COLS = (20, 20, 20, 20, 1, 1, 1, 1, 9, 128, 128)    # type(x)smear 4 x 20 | type 4 x 1 | dir 9 | h_dst | h_src


def _knn_feat(W1, W_dd, dst_is_lig, pieces=None):
    """48 feature rows of a knn first layer for one target kind (csrc/seg_attn.hip KNN features):
    [smear if src lig (20) | smear if src phore (20) | 3 direction dots | src-lig flag | src-phore flag | 0 0 0].
    Edge types (uni_denoiser.py:373-378): (src lig, dst lig)=0, (src lig, dst ph)=1, (src ph, dst lig)=2, (ph, ph)=3."""
    t_ls, t_ps = (0, 2) if dst_is_lig else (1, 3)
    pc = pieces if pieces is not None else W1.split(_KNN_COLS, -1)     # (one split: its backward is one concatenation)
    z = torch.zeros(*W_dd.shape[:-1], 3, device=W_dd.device, dtype=W_dd.dtype)
    return torch.cat([pc[t_ls], pc[t_ps], W_dd, pc[4 + t_ls], pc[4 + t_ps], z], -1)


_tri_mix = {}


def _tri_feat(W1, ang=None):
    """12 angular feature rows: [theta, sin t, sin 2t, sin 3t, sin t/2, sin t/3, cos t, cos 2t, cos 3t, cos t/2, cos t/3, 0];
    AngularEncoding columns (common.py:85): 168 theta | 169:175 sin(f t) | 175:181 cos(f t), f=[1,2,3,1,1/2,1/3]: the two
    frequency-1 columns of each kind add up.  Written as the 13 columns times a constant 0/1 matrix (at most two non-zero
    terms per output: the same rounding as the plain sum)."""
    A = ang if ang is not None else W1[..., 168:181]
    key = (str(A.device), A.dtype)
    if key not in _tri_mix:
        M = torch.zeros(13, 12, dtype=A.dtype)
        for dst, srcs in enumerate(((0,), (1, 4), (2,), (3,), (5,), (6,), (7, 10), (8,), (9,), (11,), (12,))):
            for s_ in srcs:
                M[s_, dst] = 1.0
        _tri_mix[key] = M.to(A.device)
    return A @ _tri_mix[key]


class AttnPack:
    """Everything one attention sub-layer needs besides its slice of the fused first-layer GEMMs."""
    pass


def pack_knn(sd, p, names, Wd, bd, pos):
    """names = (k, v, q) MLP names. Returns (AttnPack, node GEMM blocks [(W [128,128], bias or None)] x 5:
    k_dst, v_dst, k_src, v_src, q_hid)."""
    k, v, q = _kv_mlp(sd, p, '.' + names[0]), _kv_mlp(sd, p, '.' + names[1]), _mlp(sd, p, '.' + names[2])
    a = AttnPack()
    blocks = []
    feats = {}
    for tag, m in (('k', k), ('v', v)):
        pc = m['W1'].split(_KNN_COLS, -1)
        W_dd = pc[8] @ Wd                              # dire_embedding folded (uni_denoiser.py:279)
        c_dir = (pc[8] @ bd.unsqueeze(-1)).squeeze(-1) if bd.dim() > 1 else pc[8] @ bd
        feats[tag] = (pc, W_dd)
        blocks.append((pc[9], m['b1'] + c_dir))
    for tag in ('k', 'v'):
        blocks.append((feats[tag][0][10], None))
    blocks.append((q['W1'], q['b1']))
    a.Wf_k = {kind: lane_fixed_feat(_knn_feat(None, feats['k'][1], kind, feats['k'][0])) for kind in (True, False)}
    a.Wf_v = {kind: lane_fixed_feat(_knn_feat(None, feats['v'][1], kind, feats['v'][0])) for kind in (True, False)}
    _common(a, k, v, q, pos)
    return a, blocks


def pack_bond(sd, p, names, pos):
    """Node GEMM blocks: k_dst, v_dst, k_src, v_src, q_hid; bond GEMM weight [256,128] (k | v halves of h_bond)."""
    k, v, q = _kv_mlp(sd, p, '.' + names[0]), _kv_mlp(sd, p, '.' + names[1]), _mlp(sd, p, '.' + names[2])
    a = AttnPack()
    kp, vp = k['W1'].split((128, 128, 128), -1), v['W1'].split((128, 128, 128), -1)      # h_bond | h_dst | h_src
    blocks = [(kp[1], k['b1']), (vp[1], v['b1']), (kp[2], None), (vp[2], None), (q['W1'], q['b1'])]
    a.W_hb = torch.cat([kp[0], vp[0]], -2).contiguous()
    _common(a, k, v, q, pos)
    return a, blocks


def pack_triplet(sd, p, name='.bond_layer'):
    """Node GEMM blocks: k_hk, v_hk (gathered at src), k_hj(+b1), v_hj(+b1) (gathered at dst), q_hi(+b1q) (dst)."""
    k, v, q = _kv_mlp(sd, p, name + '.hk_func'), _kv_mlp(sd, p, name + '.hv_func'), _mlp(sd, p, name + '.hq_func')
    a = AttnPack()
    cols = (148, 20, 13, 128, 128)                 # h_bond_kj + smear(d_kj) | smear(d_ji) | angular | h_k | h_j
    kp, vp = k['W1'].split(cols, -1), v['W1'].split(cols, -1)
    qp = q['W1'].split((128, 128), -1)             # h_bond_ji | h_i
    blocks = [(kp[3], None), (vp[3], None), (kp[4], k['b1']), (vp[4], v['b1']), (qp[1], q['b1'])]
    a.W_hbg = torch.cat([kp[0], vp[0]], -2).contiguous()                             # [256, 128+20]: h_bond_kj | smear(d_kj)
    a.W_q_hb = qp[0].contiguous()
    a.Wg2_k = kp[1].transpose(-1, -2).contiguous()                                    # [20,128]
    a.Wg2_v = vp[1].transpose(-1, -2).contiguous()
    a.W_g2 = torch.cat([kp[1], vp[1]], -2).contiguous()                              # [256,20]: Q = smear(d_ji) . W_g2^T as one GEMM
    a.Wf_k = lane_fixed_feat(_tri_feat(None, kp[2]))
    a.Wf_v = lane_fixed_feat(_tri_feat(None, vp[2]))
    _common(a, k, v, q, pos=False)
    return a, blocks


def pack_phore(sd, p='phore_encoder'):
    k, v, q = _kv_mlp(sd, p, '.hk_func'), _kv_mlp(sd, p, '.hv_func'), _mlp(sd, p, '.hq_func')
    a = AttnPack()
    kp, vp = k['W1'].split((1, 128, 128), -1), v['W1'].split((1, 128, 128), -1)          # dist | h_dst | h_src
    blocks = [(kp[1], k['b1']), (vp[1], v['b1']), (kp[2], None), (vp[2], None), (q['W1'], q['b1'])]
    z = torch.zeros(*kp[0].shape[:-1], 3, device=kp[0].device, dtype=kp[0].dtype)
    a.Wf_k = lane_fixed_feat(torch.cat([kp[0], z], -1))
    a.Wf_v = lane_fixed_feat(torch.cat([vp[0], z], -1))
    _common(a, k, v, q, pos=False)
    return a, blocks


def _common(a, k, v, q, pos):
    a.ln_gk, a.ln_bk, a.ln_gv, a.ln_bv = k['bp'], k['bp'], v['bp'], v['bp']      # kernels read only b' = beta/|gamma|
    a.W2k_l = lane_fixed_w2(k['W2'])                     # key bias cancels inside the segment softmax
    a.q_ln_g, a.q_ln_b, a.W2q, a.b2q = q['g'], q['b'], q['W2'].contiguous(), q['b2'].contiguous()
    if pos:
        a.W2xv_l = lane_fixed_xv(v['W2'])
        a.b2xv = v['b2'].contiguous()
    else:
        a.W2v_l = lane_fixed_w2(v['W2'])
        a.b2v = v['b2'].contiguous()


def fuse_blocks(blocks):
    """[(W [..., 128, K], bias|None)] -> W [..., 128*n, K], bias [..., 128*n]."""
    W = torch.cat([w for w, _ in blocks], -2).contiguous()
    b = torch.cat([bb if bb is not None else torch.zeros(w.shape[:-1], device=w.device, dtype=w.dtype)
                   for w, bb in blocks], -1).contiguous()
    return W, b


# consumer ranges of the fused first-layer columns (one GEMM per consumer in the training path, training.py): rows of W_node1 =
# knn-node k_dst|v_dst, k_src|v_src, q ; bond-node k_dst|v_dst, k_src|v_src, q ; triplet k_hk|v_hk, k_hj|v_hj, q_hi
NODE1_PARTS = (256, 256, 128, 256, 256, 128, 256, 256, 128)
NODE2_PARTS = (256, 256, 128, 256, 256, 128)              # knn-pos, bond-pos
PHORE_PARTS = (256, 256, 128)


def _parts(W, b, sizes):
    """{(c0, c1): (W rows c0:c1, b[c0:c1])} from ONE split each (backward: one concatenation, no zero-padded slice gradients)."""
    Ws, bs = W.split(sizes, -2), b.split(sizes, -1)
    out, c0 = {}, 0
    for w_, b_, n in zip(Ws, bs, sizes):
        out[(c0, c0 + n)] = (w_, b_)
        c0 += n
    return out


class LayerPack:
    """Kernel-layout weights of one denoiser layer (`p` a prefix) or of several layers at once (`p` a list of prefixes: every
    tensor carries a leading layer dim; `unstack()` then gives the per-layer objects)."""

    def __init__(self, sd, p):
        Wd, bd = _param(sd, p, '.dire_embedding.weight'), _param(sd, p, '.dire_embedding.bias')
        sub = lambda name: p + name if isinstance(p, str) else [q + name for q in p]
        self.NE, b_ne = pack_knn(sd, sub('.node_layer_with_edge'), ('hk_func', 'hv_func', 'hq_func'), Wd, bd, pos=False)
        self.NB, b_nb = pack_bond(sd, sub('.node_layer_with_bond'), ('hk_func', 'hv_func', 'hq_func'), pos=False)
        self.TB, b_tb = pack_triplet(sd, p)
        self.PE, b_pe = pack_knn(sd, sub('.pos_layer_with_edge'), ('xk_func', 'xv_func', 'xq_func'), Wd, bd, pos=True)
        self.PB, b_pb = pack_bond(sd, sub('.pos_layer_with_bond'), ('xk_func', 'xv_func', 'xq_func'), pos=True)
        self.W_node1, self.b_node1 = fuse_blocks(b_ne + b_nb + b_tb)       # [1920,128]
        self.W_node2, self.b_node2 = fuse_blocks(b_pe + b_pb)              # [1280,128]
        self.node1_parts = _parts(self.W_node1, self.b_node1, NODE1_PARTS)
        self.node2_parts = _parts(self.W_node2, self.b_node2, NODE2_PARTS)
        Wl = _param(sd, p, '.lin_node.weight')
        self.W_lin = Wl
        self.W_lin2 = torch.cat([Wl, Wl], -1).contiguous()                 # (aggE | aggB) @ [W | W]^T
        self.b_lin = _param(sd, p, '.lin_node.bias').contiguous()

    def unstack(self, n):
        """Per-layer views of a stacked pack (`unbind`: the backward of each is one stack)."""
        def split(v):
            if torch.is_tensor(v):
                return v.unbind(0)
            if isinstance(v, dict):
                cols = {k: split(x) for k, x in v.items()}
                return [{k: c[i] for k, c in cols.items()} for i in range(n)]
            if isinstance(v, tuple):
                cols = [split(x) for x in v]
                return [tuple(c[i] for c in cols) for i in range(n)]
            if isinstance(v, AttnPack):
                cols = {k: split(x) for k, x in vars(v).items()}
                outs = [AttnPack() for _ in range(n)]
                for k, c in cols.items():
                    for i in range(n):
                        setattr(outs[i], k, c[i])
                return outs
            raise TypeError(type(v))
        outs = [LayerPack.__new__(LayerPack) for _ in range(n)]
        for k, v in vars(self).items():
            for i, x in enumerate(split(v)):
                setattr(outs[i], k, x)
        return outs


def pack_gate(sd, p='denoiser.edge_pred_layer'):
    """Global edge gate MLP (20 -> 128 -> LN -> ReLU -> 1) in the folded form of _kv_mlp, kernel layout of pg_edge_gate."""
    m = _kv_mlp(sd, p)
    return dict(W0=lane_fixed_feat(m['W1']), b0=m['b1'], g=m['bp'], b=m['bp'], W3=m['W2'].reshape(-1).contiguous(),
                b3=float(m['b2'].reshape(-1)[0]))


class ModelPack:
    """All kernel-layout weights of a PhoreDiff state_dict (tensors must already be on the GPU)."""

    def __init__(self, sd, num_layers=6, detach=True):
        """detach=False (training.py): the packed tensors stay attached to the parameters' autograd graph, and the layers are
        packed together (stacked parameters, one pass of tensor ops for all of them)."""
        if detach:
            sd = {k: v.detach() for k, v in sd.items()}
            self.layers = [LayerPack(sd, f'denoiser.base_block.{l}') for l in range(num_layers)]
        else:
            self.layers = LayerPack(sd, [f'denoiser.base_block.{l}' for l in range(num_layers)]).unstack(num_layers)
        self.PH, b_ph = pack_phore(sd)
        self.W_ph, self.b_ph = fuse_blocks(b_ph)                           # [640,128]
        self.ph_parts = _parts(self.W_ph, self.b_ph, PHORE_PARTS)
        self.gate = pack_gate(sd) if detach else None       # training composes the gate MLP from the raw parameters
        c = lambda k: sd[k].contiguous()
        self.W_node_emb, self.W_edge_emb = c('node_embedder.weight'), c('edge_embedder.weight')
        self.t_off, self.t_coeff = c('time_emb.0.offset'), c('time_emb.0.coeff')
        self.W_pe, self.b_pe = c('phore_embedding.weight'), c('phore_embedding.bias')
        self.v0 = (c('v_inference.0.weight'), c('v_inference.0.bias'), c('v_inference.2.weight'), c('v_inference.2.bias'))
        self.b0 = (c('bond_inference.0.weight'), c('bond_inference.0.bias'), c('bond_inference.2.weight'),
                   c('bond_inference.2.bias'))
        self.cnt = tuple((c(f'{n}.0.weight'), c(f'{n}.0.bias'), c(f'{n}.2.weight'), c(f'{n}.2.bias'))
                         for n in ('atom_mlp', 'atom_mlp_1'))
        self.pos_tab = tuple(c(f'pos_transition.{n}') for n in ('coef_x0', 'coef_xt', 'std'))
        self.node_tab = (c('node_transition.q_mats'), c('node_transition.transpopse_q_onestep_mats'))
        self.edge_tab = (c('edge_transition.q_mats'), c('edge_transition.transpopse_q_onestep_mats'))
