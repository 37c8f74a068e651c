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
        _index_cache[k] = tuple(t.to(device) for t in build())
    return _index_cache[k]


def lane_fixed_w2(W2):
    """[128 (8h+d), 128 (c)] -> [64][64 lanes][4]: element n=i*4+j -> tau=n>>5, r=(n>>3)&3, d=n&7;
    lane=(g,h) holds W2[8h+d][16 tau + 4g + r]."""
    def build():
        g, h = _lane()
        n = torch.arange(256)
        tau, r, d = n >> 5, (n >> 3) & 3, n & 7
        return 8 * h[None, :] + d[:, None], 16 * tau[:, None] + 4 * g[None, :] + r[:, None]
    rows, cols = _cached('w2', W2.device, build)
    out = W2[rows, cols]                                              # [256, 64]
    return out.view(64, 4, 64).permute(0, 2, 1).contiguous()


def lane_fixed_feat(Wfeat):
    """[128 (c), F] -> [F/4][8][64]: value[step][tau][lane=(g,m)] = Wfeat[16 tau + m][4 step + g]."""
    F = Wfeat.shape[1]
    assert F % 4 == 0
    steps = F // 4

    def build():
        g, m = _lane()
        c = (16 * torch.arange(8)[:, None] + m[None, :])[None].expand(steps, -1, -1)
        f = (4 * torch.arange(steps)[:, None, None] + g[None, None, :]).expand(-1, 8, -1)
        return c.contiguous(), f.contiguous()
    c, f = _cached(('feat', steps), Wfeat.device, build)
    return Wfeat[c, f].contiguous()


def lane_fixed_xv(W2xv):
    """[16 (h), 128 (c)] -> [32 (tau*4+r)][64]: lane=(g,h) holds W2xv[h][16 tau + 4g + r]."""
    def build():
        g, h = _lane()
        i = torch.arange(32)
        tau, r = i >> 2, i & 3
        return h[None, :].expand(32, -1).contiguous(), 16 * tau[:, None] + 4 * g[None, :] + r[:, None]
    rows, cols = _cached('xv', W2xv.device, build)
    return W2xv[rows, cols].contiguous()


def _mlp(sd, p):
    return dict(W1=sd[p + '.net.0.weight'], b1=sd[p + '.net.0.bias'], g=sd[p + '.net.1.weight'].contiguous(),
                b=sd[p + '.net.1.bias'].contiguous(), W2=sd[p + '.net.3.weight'], b2=sd[p + '.net.3.bias'])


def _kv_mlp(sd, p):
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
    m = _mlp(sd, p)
    W1c = m['W1'] - m['W1'].mean(0, keepdim=True)
    b1c = m['b1'] - m['b1'].mean()
    sgn = torch.where(m['g'] < 0, -torch.ones_like(m['g']), torch.ones_like(m['g']))
    dead = (m['g'].detach().abs() < DEAD_GAMMA)
    ag = torch.where(dead, torch.ones_like(m['g']), m['g'] * sgn)
    live = (~dead).to(m['g'].dtype)
    b2 = m['b2'] + m['W2'] @ (torch.relu(m['b']) * (1.0 - live))
    return dict(W1=(sgn[:, None] * W1c).contiguous(), b1=(sgn * b1c).contiguous(), bp=(m['b'] / ag * live).contiguous(),
                W2=(m['W2'] * (ag * live)[None, :]).contiguous(), b2=b2)


def _knn_feat(W1, W_dd, dst_is_lig):
    """48 feature rows of a knn first layer for one target kind (csrc/seg_attn.hip KNN features):
    [smear if src lig (20) | smear if src phore (20) | 3 direction dots | src-lig flag | src-phore flag | 0 0 0].
    Edge types (uni_denoiser.py:373-378): (src lig, dst lig)=0, (src lig, dst ph)=1, (src ph, dst lig)=2, (ph, ph)=3."""
    t_ls, t_ps = (0, 2) if dst_is_lig else (1, 3)
    z = torch.zeros(128, 3, device=W1.device, dtype=W1.dtype)
    return torch.cat([W1[:, t_ls * 20:(t_ls + 1) * 20], W1[:, t_ps * 20:(t_ps + 1) * 20], W_dd,
                      W1[:, 80 + t_ls:81 + t_ls], W1[:, 80 + t_ps:81 + t_ps], z], 1)


def _tri_feat(W1):
    """12 angular feature rows: [theta, sin t, sin 2t, sin 3t, sin t/2, sin t/3, cos t, cos 2t, cos 3t, cos t/2, cos t/3, 0];
    AngularEncoding columns (common.py:85): 168 theta | 169:175 sin(f t) | 175:181 cos(f t), f=[1,2,3,1,1/2,1/3]."""
    a = 168
    cols = [W1[:, a], W1[:, a + 1] + W1[:, a + 4], W1[:, a + 2], W1[:, a + 3], W1[:, a + 5], W1[:, a + 6],
            W1[:, a + 7] + W1[:, a + 10], W1[:, a + 8], W1[:, a + 9], W1[:, a + 11], W1[:, a + 12],
            torch.zeros_like(W1[:, a])]
    return torch.stack(cols, 1)


class AttnPack:
    """Everything one attention sub-layer needs besides its slice of the fused first-layer GEMMs."""
    pass


def pack_knn(sd, p, names, Wd, bd, pos):
    """names = (k, v, q) MLP names. Returns (AttnPack, node GEMM blocks [(W [128,128], bias or None)] x 5:
    k_dst, v_dst, k_src, v_src, q_hid)."""
    k, v, q = _kv_mlp(sd, f'{p}.{names[0]}'), _kv_mlp(sd, f'{p}.{names[1]}'), _mlp(sd, f'{p}.{names[2]}')
    a = AttnPack()
    blocks = []
    feats = {}
    for tag, m in (('k', k), ('v', v)):
        W1 = m['W1']
        W_dd = W1[:, 84:93] @ Wd                       # dire_embedding folded (uni_denoiser.py:279)
        c_dir = W1[:, 84:93] @ bd
        feats[tag] = (W1, W_dd)
        blocks.append((W1[:, 93:221], m['b1'] + c_dir))
    for tag, m in (('k', k), ('v', v)):
        blocks.append((m['W1'][:, 221:349], None))
    blocks.append((q['W1'], q['b1']))
    a.Wf_k = {kind: lane_fixed_feat(_knn_feat(*feats['k'], kind)) for kind in (True, False)}
    a.Wf_v = {kind: lane_fixed_feat(_knn_feat(*feats['v'], kind)) for kind in (True, False)}
    _common(a, k, v, q, pos)
    return a, blocks


def pack_bond(sd, p, names, pos):
    """Node GEMM blocks: k_dst, v_dst, k_src, v_src, q_hid; bond GEMM weight [256,128] (k | v halves of h_bond)."""
    k, v, q = _kv_mlp(sd, f'{p}.{names[0]}'), _kv_mlp(sd, f'{p}.{names[1]}'), _mlp(sd, f'{p}.{names[2]}')
    a = AttnPack()
    blocks = [(k['W1'][:, 128:256], k['b1']), (v['W1'][:, 128:256], v['b1']),
              (k['W1'][:, 256:384], None), (v['W1'][:, 256:384], None), (q['W1'], q['b1'])]
    a.W_hb = torch.cat([k['W1'][:, 0:128], v['W1'][:, 0:128]], 0).contiguous()
    _common(a, k, v, q, pos)
    return a, blocks


def pack_triplet(sd, p):
    """Node GEMM blocks: k_hk, v_hk (gathered at src), k_hj(+b1), v_hj(+b1) (gathered at dst), q_hi(+b1q) (dst)."""
    k, v, q = _kv_mlp(sd, p + '.hk_func'), _kv_mlp(sd, p + '.hv_func'), _mlp(sd, p + '.hq_func')
    a = AttnPack()
    blocks = [(k['W1'][:, 181:309], None), (v['W1'][:, 181:309], None),
              (k['W1'][:, 309:437], k['b1']), (v['W1'][:, 309:437], v['b1']), (q['W1'][:, 128:256], q['b1'])]
    a.W_hbg = torch.cat([k['W1'][:, 0:148], v['W1'][:, 0:148]], 0).contiguous()     # [256, 128+20]: h_bond_kj | smear(d_kj)
    a.W_q_hb = q['W1'][:, 0:128].contiguous()
    a.Wg2_k = k['W1'][:, 148:168].t().contiguous()                                    # [20,128]
    a.Wg2_v = v['W1'][:, 148:168].t().contiguous()
    a.W_g2 = torch.cat([k['W1'][:, 148:168], v['W1'][:, 148:168]], 0).contiguous()   # [256,20]: Q = smear(d_ji) . W_g2^T as one GEMM
    a.Wf_k = lane_fixed_feat(_tri_feat(k['W1']))
    a.Wf_v = lane_fixed_feat(_tri_feat(v['W1']))
    _common(a, k, v, q, pos=False)
    return a, blocks


def pack_phore(sd, p='phore_encoder'):
    k, v, q = _kv_mlp(sd, p + '.hk_func'), _kv_mlp(sd, p + '.hv_func'), _mlp(sd, p + '.hq_func')
    a = AttnPack()
    blocks = [(k['W1'][:, 1:129], k['b1']), (v['W1'][:, 1:129], v['b1']),
              (k['W1'][:, 129:257], None), (v['W1'][:, 129:257], None), (q['W1'], q['b1'])]
    z = torch.zeros(128, 3, device=k['W1'].device, dtype=k['W1'].dtype)
    a.Wf_k = lane_fixed_feat(torch.cat([k['W1'][:, 0:1], z], 1))
    a.Wf_v = lane_fixed_feat(torch.cat([v['W1'][:, 0:1], z], 1))
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
    """[(W [128,K], bias|None)] -> W [128*n, K], bias [128*n]."""
    W = torch.cat([w for w, _ in blocks], 0).contiguous()
    b = torch.cat([bb if bb is not None else torch.zeros(w.shape[0], device=w.device, dtype=w.dtype)
                   for w, bb in blocks]).contiguous()
    return W, b


class LayerPack:
    def __init__(self, sd, p):
        Wd, bd = sd[p + '.dire_embedding.weight'], sd[p + '.dire_embedding.bias']
        self.NE, b_ne = pack_knn(sd, p + '.node_layer_with_edge', ('hk_func', 'hv_func', 'hq_func'), Wd, bd, pos=False)
        self.NB, b_nb = pack_bond(sd, p + '.node_layer_with_bond', ('hk_func', 'hv_func', 'hq_func'), pos=False)
        self.TB, b_tb = pack_triplet(sd, p + '.bond_layer')
        self.PE, b_pe = pack_knn(sd, p + '.pos_layer_with_edge', ('xk_func', 'xv_func', 'xq_func'), Wd, bd, pos=True)
        self.PB, b_pb = pack_bond(sd, p + '.pos_layer_with_bond', ('xk_func', 'xv_func', 'xq_func'), pos=True)
        self.W_node1, self.b_node1 = fuse_blocks(b_ne + b_nb + b_tb)       # [1920,128]
        self.W_node2, self.b_node2 = fuse_blocks(b_pe + b_pb)              # [1280,128]
        Wl = sd[p + '.lin_node.weight']
        self.W_lin = Wl
        self.W_lin2 = torch.cat([Wl, Wl], 1).contiguous()                  # (aggE | aggB) @ [W | W]^T
        self.b_lin = sd[p + '.lin_node.bias'].contiguous()


def pack_gate(sd, p='denoiser.edge_pred_layer'):
    """Global edge gate MLP (20 -> 128 -> LN -> ReLU -> 1) in the folded form of _kv_mlp, kernel layout of pg_edge_gate."""
    m = _kv_mlp(sd, p)
    return dict(W0=lane_fixed_feat(m['W1']), b0=m['b1'], g=m['bp'], b=m['bp'], W3=m['W2'].reshape(-1).contiguous(),
                b3=float(m['b2'].reshape(-1)[0]))


class ModelPack:
    """All kernel-layout weights of a PhoreDiff state_dict (tensors must already be on the GPU)."""

    def __init__(self, sd, num_layers=6, detach=True):
        """detach=False (training.py): the packed tensors stay attached to the parameters' autograd graph."""
        if detach:
            sd = {k: v.detach() for k, v in sd.items()}
        self.layers = [LayerPack(sd, f'denoiser.base_block.{l}') for l in range(num_layers)]
        self.PH, b_ph = pack_phore(sd)
        self.W_ph, self.b_ph = fuse_blocks(b_ph)                           # [640,128]
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
