"""CPU oracle: a plain-PyTorch (fp32, CPU) restatement of PhoreGen's denoising hot path.

TEST INFRASTRUCTURE — NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and bench.py's
``cpu_baseline`` leg may import this file; nothing under phoregen_amd/ does.  It follows the
reference's *dataflow* (unfactored MLP inputs, materialised triplet tensors, per-step recomputation of
constant pieces) so that it can also stand in as the "reference CPU path" timed beside the GPU result
(the reference's Python cannot travel to the GPU box).

Parity pin: every function below is checked in tests/test_oracle_golden.py against vectors recorded
from the reference's own code by oracle/make_golden.py (tests/golden/*.npz).  The third-party ops the
reference calls (torch-scatter 2.0.9, torch-cluster 1.6.0, torch-sparse 0.6.15) are absent from
/root/reference and restated from their documented behaviour: parity is unpinned at THAT boundary only
(see oracle/standins/README.md).

All citations are file:line in /root/reference.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# fixed Gaussian offsets: models/common.py:18 (fix_offset=True ignores start/stop/num_gaussians)
SMEAR_OFFSETS = [0, 1, 1.25, 1.5, 1.75, 2, 2.25, 2.5, 2.75, 3, 3.5, 4, 4.5, 5, 5.5, 6, 7, 8, 9, 10]
SMEAR_COEFF = -0.5 / (1.0 - 0.0) ** 2  # common.py:23
LOG2 = math.log(2.0)


# ----------------------------------------------------------------------------------------------
# small ops
# ----------------------------------------------------------------------------------------------
def gaussian_smearing(dist):
    """common.py:29-31."""
    off = torch.tensor(SMEAR_OFFSETS, dtype=dist.dtype)
    d = dist.reshape(-1, 1) - off.reshape(1, -1)
    return torch.exp(SMEAR_COEFF * d.pow(2))


def time_smearing(t, num_timesteps=1000, num_gaussians=10):
    """common.py:34-55 with type_='linear' (diffusion.py:42)."""
    off = torch.linspace(0.0, float(num_timesteps), num_gaussians)
    diff = torch.diff(off)
    diff = torch.cat([diff[:1], diff])
    coeff = -0.5 / diff ** 2
    d = t.clamp(0.0, float(num_timesteps)).reshape(-1, 1) - off.reshape(1, -1)
    return torch.exp(coeff * d.pow(2))


def angular_encoding(theta):
    """common.py:67-87: [x, sin(x f), cos(x f)], f = [1,2,3,1,1/2,1/3]."""
    f = torch.tensor([1., 2., 3., 1., 1. / 2, 1. / 3], dtype=theta.dtype)
    x = theta.unsqueeze(-1)
    return torch.cat([x, torch.sin(x * f), torch.cos(x * f)], -1)


def shifted_softplus(x):
    """common.py:58-64."""
    return F.softplus(x) - LOG2


def seg_sum(src, index, n):
    out = torch.zeros((n,) + tuple(src.shape[1:]), dtype=src.dtype)
    return out.index_add_(0, index, src)


def seg_softmax(logits, index, n):
    """torch_scatter.scatter_softmax(dim=0): max-shifted, no epsilon (2.0.9)."""
    idx = index.reshape(-1, *([1] * (logits.dim() - 1))).expand_as(logits)
    mx = torch.full((n,) + tuple(logits.shape[1:]), -float('inf'), dtype=logits.dtype)
    mx = mx.scatter_reduce(0, idx, logits, 'amax', include_self=True)
    ex = torch.exp(logits - mx[index])
    return ex / seg_sum(ex, index, n)[index]


def knn_graph(x, k, batch):
    """torch_cluster.knn_graph(loop=False, flow='source_to_target'): row 0 = neighbour, row 1 = centre."""
    src, dst = [], []
    for g in torch.unique(batch).tolist():
        idx = (batch == g).nonzero().squeeze(-1)
        xg = x[idx]
        d2 = ((xg[:, None, :] - xg[None, :, :]) ** 2).sum(-1)
        kk = min(k + 1, idx.numel())
        order = torch.argsort(d2, dim=1, stable=True)[:, :kk]
        centre = idx[:, None].expand(-1, kk)
        neigh = idx[order]
        keep = neigh != centre
        dst.append(centre[keep])
        src.append(neigh[keep])
    return torch.stack([torch.cat(src), torch.cat(dst)], 0)


def make_edge_data(num_atoms):
    """utils/sample_utils.py:40-54: per graph all (a<b) pairs then all reversed pairs."""
    ei, eb, start = [], [], 0
    for g, n in enumerate(num_atoms.tolist()):
        half = torch.triu_indices(n, n, offset=1)
        full = torch.cat([half, half.flip(0)], 1)
        ei.append(full + start)
        eb.append(torch.full((full.size(1),), g, dtype=torch.long))
        start += n
    return torch.cat(ei, 1), torch.cat(eb)


def triplets(bond_index, num_nodes):
    """uni_denoiser.py:101-121 (SparseTensor row gather restated with sorting).

    For each edge e=(j->i) in order, the edges (k->j) in ascending k, dropping k == i."""
    row, col = bond_index  # j -> i
    E = row.numel()
    order = torch.argsort(col * num_nodes + row, stable=True)      # adjacency rows = dst, cols = src
    s_dst, s_src = col[order], row[order]
    counts = torch.bincount(s_dst, minlength=num_nodes)
    ptr = torch.zeros(num_nodes + 1, dtype=torch.long)
    ptr[1:] = counts.cumsum(0)
    cnt = counts[row]                                              # incoming edges of j, per edge ji
    idx_ji = torch.repeat_interleave(torch.arange(E), cnt)
    offs = torch.arange(int(cnt.sum())) - torch.repeat_interleave(cnt.cumsum(0) - cnt, cnt)
    take = torch.repeat_interleave(ptr[row], cnt) + offs
    idx_k, idx_kj = s_src[take], order[take]
    idx_i, idx_j = col[idx_ji], row[idx_ji]
    m = idx_i != idx_k
    return idx_i[m], idx_j[m], idx_k[m], idx_kj[m], idx_ji[m]


def compose_context(h_p, h_l, pos_p, pos_l, batch_p, batch_l):
    """common.py:180-208: stable sort by graph id -> per graph [phore..., ligand...]."""
    batch = torch.cat([batch_p, batch_l])
    perm = torch.sort(batch, stable=True).indices
    mask_l = torch.cat([torch.zeros(len(batch_p), dtype=torch.bool), torch.ones(len(batch_l), dtype=torch.bool)])[perm]
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(len(perm))
    l_index = inv[len(batch_p):]                 # position of ligand atom a in the context (== common.py:166-177)
    p_index = inv[:len(batch_p)]
    return (torch.cat([h_p, h_l])[perm], torch.cat([pos_p, pos_l])[perm], batch[perm], mask_l, p_index, l_index)


# ----------------------------------------------------------------------------------------------
# schedules / transition tables (float64 numpy exactly as the reference builds them)
# ----------------------------------------------------------------------------------------------
def _sigmoid(x):
    return 1 / (np.exp(-x) + 1)  # common.py:501-502


def advance_alphas_bar(T, scale_start, scale_end, width):
    """common.py:459-480."""
    a = (scale_end - scale_start) / (_sigmoid(-width) - _sigmoid(width))
    b = 0.5 * (scale_end + scale_start - a)
    return a * _sigmoid(-width * np.linspace(-1, 1, T)) + b


def betas_from_alphas_bar(ab):
    al = np.zeros_like(ab)
    al[0] = ab[0]
    al[1:] = ab[1:] / ab[:-1]
    return np.clip(1 - al, 0, 1)


def beta_schedule(T, cfg):
    """common.py:505-544 for the two schedules the shipped configs use."""
    if cfg['beta_schedule'] == 'advance':
        return betas_from_alphas_bar(advance_alphas_bar(T, cfg.get('scale_start', 0.999),
                                                        cfg.get('scale_end', 0.001), cfg.get('width', 2)))
    if cfg['beta_schedule'] == 'segment':     # common.py:483-498
        ab = []
        for seg, p in zip(cfg['time_segment'], cfg['segment_diff']):
            ab.extend(advance_alphas_bar(seg + 1, p['scale_start'], p['scale_end'], p['width'])[1:])
        assert len(ab) == T
        return betas_from_alphas_bar(np.array(ab))
    raise NotImplementedError(cfg['beta_schedule'])


def continuous_tables(betas):
    """transition.py:14-26."""
    al = 1. - betas
    ab = np.cumprod(al)
    abp = np.concatenate([[1.], ab[:-1]])
    f = lambda a: torch.from_numpy(a).float()
    return dict(betas=f(betas), alphas=f(al), alphas_bar=f(ab), alphas_bar_prev=f(abp),
                coef_x0=f(np.sqrt(abp) * betas / (1 - ab)),
                coef_xt=f(np.sqrt(al) * (1 - abp) / (1 - ab)),
                std=f(np.sqrt((1 - abp) * betas / (1 - ab))))


def categorical_init_prob(K, kind):
    """transition.py:183-196."""
    if kind == 'absorb':
        p = 0.01 * np.ones(K)
        p[0] = 1
    elif kind == 'tomask':
        p = 0.001 * np.ones(K)
        p[-1] = 1.
    else:
        p = np.ones(K)
    return p / p.sum()


def categorical_tables(betas, K, kind):
    """transition.py:200-243: one-step Q_t = beta_t * 1 p^T + (1-beta_t) I; cumulative products."""
    p = categorical_init_prob(K, kind)
    one = np.stack([b * np.repeat(p[None], K, 0) + np.eye(K) * (1. - b) for b in betas])
    cum = [one[0]]
    for t in range(1, len(betas)):
        cum.append(cum[-1] @ one[t])
    f = lambda a: torch.from_numpy(a).float()
    return dict(q_mats=f(np.stack(cum)), transpopse_q_onestep_mats=f(np.transpose(one, (0, 2, 1))), init_prob=p)


# ----------------------------------------------------------------------------------------------
# posterior / sampling steps
# ----------------------------------------------------------------------------------------------
def q_v_posterior(tab, log_v0, log_vt, t, batch):
    """transition.py:285-315 with v0_prob=True."""
    tb = t[batch]
    tm1 = torch.clamp(tb - 1, min=0)
    f1 = torch.einsum('bj,bjk->bk', log_vt.exp(), tab['transpopse_q_onestep_mats'][tb])
    f2 = torch.einsum('bj,bjk->bk', log_v0.exp(), tab['q_mats'][tm1])
    out = torch.log(f1 + 1e-30).clamp_min(-32.) + torch.log(f2 + 1e-30).clamp_min(-32.)
    out = out - torch.logsumexp(out, -1, keepdim=True)
    return torch.where((tb == 0).unsqueeze(-1), log_v0, out)


def qd_loss(y_true, y_l, y_u, a=0.05, s=160, nd=15, factor=1, epsilon=1e-12):
    """common.py:261-281 (mode='soft'): quality-driven interval loss of the atom-count heads."""
    n = y_true.shape[0]
    k_h = torch.relu(torch.sign(y_u - y_true)) * torch.relu(torch.sign(y_true - y_l))
    k_s = torch.sigmoid((y_u - y_true) * s) * torch.sigmoid((y_true - y_l) * s)
    mpiw = torch.sum((y_u - y_l) * k_h) / (torch.sum(k_h) + epsilon) * factor
    return mpiw + (torch.relu((1 - a) - torch.mean(k_s)) ** 2) * (n ** 0.5) * nd


def gumbel_argmax(logits, uniform):
    """common.py:425-431."""
    g = -torch.log(-torch.log(uniform + 1e-30) + 1e-30)
    return (g + logits).argmax(-1)


def pos_prev_from_recon(tab, x_t, x0, t, batch, eps, energy_grad=0.):
    """transition.py:44-63."""
    tb = t[batch]
    mu = tab['coef_x0'][tb].unsqueeze(-1) * x0 + tab['coef_xt'][tb].unsqueeze(-1) * x_t - energy_grad
    return torch.where((tb == 0).unsqueeze(-1), mu, mu + tab['std'][tb].unsqueeze(-1) * eps)


# ----------------------------------------------------------------------------------------------
# network
# ----------------------------------------------------------------------------------------------
class Oracle:
    """Functional PhoreDiff over a plain state_dict (SURVEY.md Appendix C key names)."""

    def __init__(self, state_dict, n_heads=16, knn=32, num_timesteps=1000, num_layers=6,
                 diff_cfg=None, data_name='zinc_300', dtype=torch.float32):
        """dtype=torch.float64 (with float64 inputs): the same dataflow in double precision, used by the tests to measure
        how much of an fp32 discrepancy is conditioning of the network rather than an implementation difference."""
        self.sd = {k: v.detach().to(dtype).cpu() if v.is_floating_point() else v.detach().cpu()
                   for k, v in state_dict.items()}
        self.H, self.k, self.T, self.L = n_heads, knn, num_timesteps, num_layers
        self.ex_col = 12 if data_name in ('zinc_300', 'pdbbind') else 10     # diffusion.py:152-155
        self.max_atom, self.min_atom = 78, 4                                  # diffusion.py:30-31
        if diff_cfg is not None:
            self.tab_pos = continuous_tables(beta_schedule(self.T, diff_cfg['diff_pos']))
            self.tab_node = categorical_tables(beta_schedule(self.T, diff_cfg['diff_atom']), 12,
                                               diff_cfg['diff_atom']['init_prob'])
            self.tab_edge = categorical_tables(beta_schedule(self.T, diff_cfg['diff_bond']), 6,
                                               diff_cfg['diff_bond']['init_prob'])

    # ---- building blocks ----
    def lin(self, p, x):
        b = self.sd.get(p + '.bias')
        return F.linear(x, self.sd[p + '.weight'], b)

    def mlp(self, p, x):
        """common.py:99-119: Linear -> LayerNorm -> ReLU -> Linear."""
        h = self.lin(p + '.net.0', x)
        h = F.layer_norm(h, (h.size(-1),), self.sd[p + '.net.1.weight'], self.sd[p + '.net.1.bias'], 1e-5)
        return self.lin(p + '.net.3', F.relu(h))

    def node_update(self, p, h, edge_feat, src, dst, e_w=None):
        """uni_denoiser.py:40-72 (out_fc=False)."""
        N, H = h.size(0), self.H
        kv = torch.cat([edge_feat, h[dst], h[src]], -1)
        k = self.mlp(p + '.hk_func', kv).view(-1, H, 128 // H)
        v = self.mlp(p + '.hv_func', kv)
        if e_w is not None:
            v = v * e_w.view(-1, 1)
        v = v.view(-1, H, 128 // H)
        q = self.mlp(p + '.hq_func', h).view(-1, H, 128 // H)
        alpha = seg_softmax((q[dst] * k / np.sqrt(k.shape[-1])).sum(-1), dst, N)
        return seg_sum(alpha.unsqueeze(-1) * v, dst, N).view(N, -1)

    def pos_update(self, p, h, rel_x, edge_feat, src, dst, e_w=None):
        """uni_denoiser.py:187-209."""
        N, H = h.size(0), self.H
        kv = torch.cat([edge_feat, h[dst], h[src]], -1)
        k = self.mlp(p + '.xk_func', kv).view(-1, H, 128 // H)
        v = self.mlp(p + '.xv_func', kv)
        if e_w is not None:
            v = v * e_w.view(-1, 1)
        v = v.unsqueeze(-1) * rel_x.unsqueeze(1)
        q = self.mlp(p + '.xq_func', h).view(-1, H, 128 // H)
        alpha = seg_softmax((q[dst] * k / np.sqrt(k.shape[-1])).sum(-1), dst, N)
        return seg_sum(alpha.unsqueeze(-1) * v, dst, N).mean(1)

    def bond_update(self, p, h, h_bond, pos, bond_index):
        """uni_denoiser.py:123-165 (include_h_node=True): materialised [E3, 437] / [E3, 256] inputs."""
        E, H = h_bond.size(0), self.H
        src, dst = bond_index
        idx_i, idx_j, idx_k, idx_kj, idx_ji = triplets(bond_index, h.size(0))
        dist = (pos[dst] - pos[src]).pow(2).sum(-1).sqrt()
        pji, pki = pos[idx_j] - pos[idx_i], pos[idx_k] - pos[idx_i]
        a = (pji * pki).sum(-1)
        b = torch.linalg.cross(pji, pki).norm(dim=-1)
        ang = angular_encoding(torch.atan2(b, a))
        r = gaussian_smearing(dist)
        kv = torch.cat([h_bond[idx_kj], r[idx_kj], r[idx_ji], ang, h[idx_k], h[idx_j]], -1)
        qin = torch.cat([h_bond[idx_ji], h[idx_i]], -1)
        k = self.mlp(p + '.hk_func', kv).view(-1, H, 128 // H)
        v = self.mlp(p + '.hv_func', kv).view(-1, H, 128 // H)
        q = self.mlp(p + '.hq_func', qin).view(-1, H, 128 // H)
        alpha = seg_softmax((q * k / np.sqrt(k.shape[-1])).sum(-1), idx_ji, E)
        return seg_sum(alpha.unsqueeze(-1) * v, idx_ji, E).view(E, -1)

    def direction_feat(self, x, phore_norm, src, dst, mask_ligand, batch):
        """common.py:300-326: ligand 'normal' = mean of 3-NN positions - x; phore normal from file."""
        lx = x[mask_ligand]
        nsrc, ndst = knn_graph(lx, 3, batch[mask_ligand])
        cnt = torch.bincount(ndst, minlength=lx.size(0)).clamp(min=1).unsqueeze(-1).to(lx.dtype)
        l_norm = seg_sum(lx[nsrc], ndst, lx.size(0)) / cnt - lx
        comb = torch.zeros_like(x)
        comb[~mask_ligand] = phore_norm
        comb[mask_ligand] = l_norm
        v1, v2, v3 = comb[src], comb[dst], x[src] - x[dst]
        return torch.stack([(v1 * v2).sum(-1), (v1 * v3).sum(-1), (v2 * v3).sum(-1)], -1)

    def attention_layer(self, p, h, x, edge_type, edge_index, h_bond, bond_index, mask_ligand, e_w, phore_norm, batch):
        """uni_denoiser.py:260-298."""
        src, dst = edge_index
        rel_x = x[dst] - x[src]
        dist = rel_x.norm(dim=-1)
        g = gaussian_smearing(dist)
        et = edge_type.to(g.dtype)
        outer = (et.unsqueeze(-1) * g.unsqueeze(1)).reshape(g.size(0), -1)     # common.py:156-163: idx = type*20+gauss
        dire = self.lin(p + '.dire_embedding', self.direction_feat(x, phore_norm, src, dst, mask_ligand, batch))
        edge_feat = torch.cat([outer, et, dire], -1)
        h_edge = self.node_update(p + '.node_layer_with_edge', h, edge_feat, src, dst, e_w)
        bsrc, bdst = bond_index
        h_bnd = self.node_update(p + '.node_layer_with_bond', h, h_bond, bsrc, bdst)
        new_h_bond = h_bond + self.bond_update(p + '.bond_layer', h, h_bond, x, bond_index)
        new_h = h + self.lin(p + '.lin_node', h_edge + h_bnd)
        dx = self.pos_update(p + '.pos_layer_with_edge', new_h, rel_x, edge_feat, src, dst, e_w)
        dx = dx + self.pos_update(p + '.pos_layer_with_bond', new_h, x[bdst] - x[bsrc], new_h_bond, bsrc, bdst)
        return new_h, new_h_bond, x + dx * mask_ligand[:, None].to(x.dtype)

    def denoiser(self, h, x, bond_index, h_bond, mask_ligand, batch, phore_norm, capture=None):
        """uni_denoiser.py:396-430 (num_blocks=1, cutoff_mode=knn, use_global_ew)."""
        edge_index = knn_graph(x, self.k, batch)
        src, dst = edge_index
        n_src, n_dst = mask_ligand[src], mask_ligand[dst]
        etype = torch.where(n_src & n_dst, 0, torch.where(n_src & ~n_dst, 1, torch.where(~n_src & n_dst, 2, 3)))
        edge_type = F.one_hot(etype, 4)                                        # uni_denoiser.py:363-379
        dist = (x[dst] - x[src]).norm(dim=-1)
        e_w = torch.sigmoid(self.mlp('denoiser.edge_pred_layer', gaussian_smearing(dist)))
        if capture is not None:
            capture.update(edge_index=edge_index, edge_type=edge_type, e_w=e_w)
        for l in range(self.L):
            h, h_bond, x = self.attention_layer(f'denoiser.base_block.{l}', h, x, edge_type, edge_index, h_bond,
                                                bond_index, mask_ligand, e_w, phore_norm, batch)
            if capture is not None:
                capture[f'L{l}_out'] = (h, h_bond, x)
        return h, h_bond, x

    def phore_encode(self, h_phore, pos_phore, batch_phore):
        """diffusion.py:186-191 + common.py:329-356 + models/__init__.py:29-35 (p x p, self loops kept)."""
        hp = self.lin('phore_embedding', h_phore)
        same = batch_phore[:, None] == batch_phore[None, :]
        i1, i2 = same.nonzero(as_tuple=True)        # fully_connect_two_graphs: index_1 tiled, index_2 inner
        src, dst = i1, i2
        dist = (pos_phore[dst] - pos_phore[src]).norm(dim=-1, keepdim=True)
        return self.node_update('phore_encoder', hp, dist, src, dst)

    def atom_count(self, hp_emb, batch_phore, h_phore, B):
        """diffusion.py:148-163."""
        def head(p, x):
            return torch.sigmoid(self.lin(p + '.2', F.relu(self.lin(p + '.0', x))))

        def seg_mean(v, idx):
            c = torch.bincount(idx, minlength=B).clamp(min=1).unsqueeze(-1).to(v.dtype)
            return seg_sum(v, idx, B) / c
        c_all = seg_mean(head('atom_mlp', hp_emb), batch_phore)
        m = h_phore[:, self.ex_col] != 1
        c_l = seg_mean(head('atom_mlp_1', hp_emb[m]), batch_phore[m])
        return c_l, c_l + F.relu(c_all - c_l)

    def forward(self, h_node_pert, pos_pert, batch_node, h_edge_pert, edge_index, batch_edge, time_step,
                h_phore, pos_phore, phore_norm, batch_phore, capture=None):
        """diffusion.py:175-246."""
        B = int(time_step.numel())
        t_node = time_smearing(time_step[batch_node].to(pos_pert.dtype), self.T)
        h_node = torch.cat([self.lin('node_embedder', h_node_pert), t_node], -1)
        t_edge = time_smearing(time_step[batch_edge].to(pos_pert.dtype), self.T)
        hp = self.phore_encode(h_phore, pos_phore, batch_phore)
        h_all, pos_all, batch_all, mask_l, p_idx, l_idx = compose_context(hp, h_node, pos_phore, pos_pert,
                                                                          batch_phore, batch_node)
        bond_index = l_idx[edge_index]
        h_bond = torch.cat([self.lin('edge_embedder', h_edge_pert), t_edge], -1)
        if capture is not None:
            capture.update(phore_enc=hp, h_all=h_all, pos_all=pos_all, batch_all=batch_all, mask=mask_l,
                           bond_index=bond_index, h_bond0=h_bond)
        h, h_bond, x = self.denoiser(h_all, pos_all, bond_index, h_bond, mask_l, batch_all, phore_norm, capture)
        v = self.lin('v_inference.2', shifted_softplus(self.lin('v_inference.0', h[mask_l])))
        bond = self.lin('bond_inference.2', shifted_softplus(self.lin('bond_inference.0', h_bond)))
        return v, x[mask_l], bond, self.atom_count(hp, batch_phore, h_phore, B)

    # ---- training objective (diffusion.py:249-352) with an explicit noise source ----
    def compute_loss(self, b, rng, loss_weight=(1., 100., 100.), count_factor=1., bond_len_loss=False):
        """`b`: dict with the Appendix-G fields (ligand_x [N] int64, ligand_pos, ligand_batch, ligand_ptr,
        f_edge_index [2,E] (any order of the complete directed graph), f_edge_attr [E], f_edge_batch, phore_*).
        `rng` provides .randint(high, n), .randn(shape), .rand(shape) in the reference's draw order:
        sample_time (diffusion.py:138-145), pos noise (transition.py:28-41), node Gumbel uniforms, edge Gumbel uniforms
        (transition.py:245-263, common.py:425-431).  `bond_len_loss` (config flag, diffusion.py:286-290,333,341): the MSE of the
        predicted against the true lengths of the molecule's bonds `b['edge_index']` [2, n_bonds] joins the loss.  Returns (loss, dict)."""
        B = int(b['ligand_ptr'].numel() - 1)
        ts = rng.randint(self.T, B // 2 + 1)
        t = torch.cat([ts, self.T - ts - 1])[:B]
        bn, be = b['ligand_batch'], b['f_edge_batch']
        ab = self.tab_pos['alphas_bar'][t][bn].unsqueeze(-1)
        pos0 = b['ligand_pos']
        pos_pert = ab.sqrt() * pos0 + (1 - ab).sqrt() * rng.randn(pos0.shape)

        def noise_cat(tab, v, K, batch):
            log_v0 = torch.log(F.one_hot(v, K).float().clamp(min=1e-30))                     # common.py:398-402
            q = torch.einsum('bi,bij->bj', log_v0.exp(), tab['q_mats'][t[batch]])             # transition.py:265-271
            log_q = torch.log(q + 1e-30).clamp_min(-32.)
            cls = gumbel_argmax(log_q, rng.rand(log_q.shape))
            return F.one_hot(cls, K).float(), torch.log(F.one_hot(cls, K).float().clamp(min=1e-30)), log_v0
        h_node, log_node_t, log_node_0 = noise_cat(self.tab_node, b['ligand_x'], 12, bn)
        h_edge, log_edge_t, log_edge_0 = noise_cat(self.tab_edge, b['f_edge_attr'], 6, be)
        pred_node, pred_pos, pred_edge, (c_l, c_u) = self.forward(
            h_node, pos_pert, bn, h_edge, b['f_edge_index'], be, t, b['phore_x'], b['phore_pos'], b['phore_norm'],
            b['phore_batch'])
        loss_pos = F.mse_loss(pred_pos, pos0) * loss_weight[0]

        def cat_loss(tab, pred, log_t, log_0, batch):
            log_rec = F.log_softmax(pred, -1)
            post_true = q_v_posterior(tab, log_0, log_t, t, batch)
            post_pred = q_v_posterior(tab, log_rec, log_t, t, batch)
            kl = (post_true.exp() * (post_true - post_pred)).sum(-1)                           # common.py:434-436
            nll = -(log_0.exp() * post_pred).sum(-1)                                           # common.py:439-440
            m = (t == 0).float()[batch]
            return torch.mean(m * nll + (1 - m) * kl)                                          # transition.py:317-329
        loss_node = cat_loss(self.tab_node, pred_node, log_node_t, log_node_0, bn) * loss_weight[1]
        loss_edge = cat_loss(self.tab_edge, pred_edge, log_edge_t, log_edge_0, be) * loss_weight[2]
        true = ((b['ligand_ptr'][1:] - b['ligand_ptr'][:-1]).float() - self.min_atom) / (self.max_atom - self.min_atom)
        loss_count = qd_loss(true.unsqueeze(-1), c_l, c_u, factor=count_factor)
        loss = loss_pos + loss_node + loss_edge + loss_count
        if bond_len_loss:                                                                      # diffusion.py:286-290
            src, dst = b['edge_index']
            true_len = torch.norm(pos0[src] - pos0[dst], dim=-1)
            pred_len = torch.norm(pred_pos[src] - pred_pos[dst], dim=-1)
            loss_len = F.mse_loss(pred_len, true_len)
            loss = loss + loss_len                                                             # diffusion.py:333

        def acc(true_cls, logits, batch):                                                      # common.py:284-297
            bad = torch.zeros(B).index_add(0, batch, (logits.argmax(-1) != true_cls).float())
            return float((bad[batch.unique()] == 0).sum()) / batch.unique().numel()
        info = dict(loss=loss.item(), loss_pos=loss_pos.item(), loss_node=loss_node.item(), loss_count=loss_count.item(),
                    loss_edge=loss_edge.item(), node_acc=acc(b['ligand_x'], pred_node, bn),
                    edge_acc=acc(b['f_edge_attr'], pred_edge, be))
        if bond_len_loss:
            info['loss_len'] = loss_len.item()                                                 # diffusion.py:341
        return loss, info

    # ---- sampler (diffusion.py:391-525) with an explicit noise source ----
    def sample(self, h_phore, pos_phore, phore_norm, center, num_atoms, rng, t_total=None, guidance=None, n_steps=None,
               observer=None, keep_steps=True):
        """`rng` provides .randn(shape), .rand64(shape), .rand(shape) in the reference's draw order
        (SURVEY.md Appendix B 3-5).  Returns the reference's result dict plus per-step records.
        `observer(i, kind, log_posterior, uniform, logits)` sees every categorical draw (tools/match_rate.py: top-2 margins)."""
        T = self.T if t_total is None else t_total
        B = len(num_atoms)
        p = h_phore.size(0)
        batch_node = torch.repeat_interleave(torch.arange(B), num_atoms)
        edge_index, batch_edge = make_edge_data(num_atoms)
        N, E = batch_node.numel(), batch_edge.numel()
        hp, pp, pn = h_phore.repeat(B, 1), pos_phore.repeat(B, 1), phore_norm.repeat(B, 1)
        bp = torch.repeat_interleave(torch.arange(B), p)
        pos = rng.randn((N, 3)) - center                                                     # diffusion.py:406
        log_pn = torch.log(torch.from_numpy(self.tab_node['init_prob']) + 1e-30).clamp_min(-32.)   # float64!
        log_pe = torch.log(torch.from_numpy(self.tab_edge['init_prob']) + 1e-30).clamp_min(-32.)
        node_t = gumbel_argmax(log_pn.unsqueeze(0).repeat(N, 1), rng.rand64((N, 12)))        # transition.py:331-339
        edge_t = gumbel_argmax(log_pe.unsqueeze(0).repeat(E, 1), rng.rand64((E, 6)))
        h_node, h_edge = F.one_hot(node_t, 12).float(), F.one_hot(edge_t, 6).float()
        log_node = torch.log(h_node.clamp(min=1e-30))                                        # common.py:398-402
        log_edge = torch.log(h_edge.clamp(min=1e-30))
        traj = [[h_node], [pos], [h_edge]]                      # diffusion.py:424-426 (no +center at index 0)
        steps = []
        for step in list(range(T)[::-1])[:n_steps]:          # n_steps: only the first steps (tools/match_rate.py)
            t = torch.full((B,), step, dtype=torch.long)
            v, x0, bond, _ = self.forward(h_node, pos, batch_node, h_edge, edge_index, batch_edge, t, hp, pp, pn, bp)
            if keep_steps:
                steps.append((h_node, pos, h_edge, v, x0, bond))
            log_node = q_v_posterior(self.tab_node, F.log_softmax(v, -1), log_node, t, batch_node)
            un = rng.rand((N, 12))
            h_node_prev = F.one_hot(gumbel_argmax(log_node, un), 12).float()
            log_edge = q_v_posterior(self.tab_edge, F.log_softmax(bond, -1), log_edge, t, batch_edge)
            ue = rng.rand((E, 6))
            h_edge_prev = F.one_hot(gumbel_argmax(log_edge, ue), 6).float()
            if observer is not None:
                observer(len(traj[0]) - 1, 'node', log_node, un, v)
                observer(len(traj[0]) - 1, 'edge', log_edge, ue, bond)
            grad = 0.
            if guidance is not None:
                grad = guidance_grad(guidance, pos, batch_node, h_edge_prev, edge_index, batch_edge, B,
                                     pos_phore[h_phore[:, self.ex_col] != 1].mean(0))
            pos = pos_prev_from_recon(self.tab_pos, pos, x0, t, batch_node, rng.randn((N, 3)), grad)
            h_node, h_edge = h_node_prev, h_edge_prev
            traj[0].append(h_node), traj[1].append(pos + center), traj[2].append(h_edge)
        return {'pred': [v, x0 + center, bond], 'traj': [torch.stack(t_) for t_ in traj],
                'lig_info': [num_atoms, batch_node, edge_index, batch_edge], 'steps': steps}


def guidance_grad(opts, x, batch_node, h_edge_prev, edge_index, batch_edge, B, phore_center):
    """diffusion.py:476-502 + utils/sample_utils.py:135-165, closed-form gradients.

    atom_prox: E = (1/B) sum_g mean_{bonds e of g with argmax type > 0} [relu(d-max_d) + relu(min_d-d)]
    center_prox: E = (1/B) sum_g || mean(x_g) - c ||."""
    grad = torch.zeros_like(x)
    for o in opts:
        if o['type'] == 'atom_prox':
            et = h_edge_prev.argmax(-1)
            sel = et > 0
            cnt = seg_sum(sel.float(), batch_edge, B)
            s, d = edge_index[0][sel], edge_index[1][sel]
            diff = x[s] - x[d]
            ln = diff.norm(dim=-1)
            coef = ((ln > o['max_d']).float() - (ln < o['min_d']).float()) / (cnt[batch_edge[sel]] * B)
            unit = diff / ln.unsqueeze(-1)
            g = coef.unsqueeze(-1) * unit
            grad.index_add_(0, s, g)
            grad.index_add_(0, d, -g)
        elif o['type'] == 'center_prox':
            n = torch.bincount(batch_node, minlength=B).to(x.dtype)
            mean = seg_sum(x, batch_node, B) / n.unsqueeze(-1)
            dv = mean - phore_center
            u = dv / dv.norm(dim=-1, keepdim=True)
            grad += (u / (n.unsqueeze(-1) * B))[batch_node]
    return grad


class TapeRng:
    """Replays recorded CPU draws (tests/golden g5_*): tape = list of arrays in draw order."""

    def __init__(self, tape):
        self.tape, self.i = list(tape), 0

    def _next(self, shape):
        a = self.tape[self.i]
        self.i += 1
        assert tuple(a.shape) == tuple(shape), (a.shape, shape)
        return torch.as_tensor(a)

    randn = rand = rand64 = _next


class TrainTapeRng:
    """Replays the recorded draws of one compute_loss call: randint, normal_, rand_like, rand_like."""

    def __init__(self, time_draw, pos_noise, u_node, u_edge):
        self.time_draw, self.pos_noise, self.u = time_draw, pos_noise, [u_node, u_edge]

    def randint(self, high, n):
        assert self.time_draw.numel() == n
        return self.time_draw

    def randn(self, shape):
        assert tuple(shape) == tuple(self.pos_noise.shape)
        return self.pos_noise

    def rand(self, shape):
        u = self.u.pop(0)
        assert tuple(shape) == tuple(u.shape)
        return u


class TorchCpuRng:
    """Draws from torch's default CPU generator in the reference's order/dtypes (Appendix B 3-5)."""

    def randn(self, shape):
        return torch.randn(shape)

    def rand(self, shape):
        return torch.rand(shape)

    def rand64(self, shape):
        return torch.rand(shape, dtype=torch.float64)
