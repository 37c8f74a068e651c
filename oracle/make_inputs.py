"""Seeded synthetic inputs shared by the golden generator, the tests and the benchmark's CPU leg
(TEST INFRASTRUCTURE; pure torch, no reference import)."""
import torch
import torch.nn.functional as F

from .phoregen_oracle import make_edge_data


def synthetic_phore(gen, p, frac_ex=0.6, spread=3.0):
    """Feature layout of datasets/get_phore_data.py:55-69: 13 type one-hot | alpha | has_norm(2) | EX(2)."""
    n_ex = int(round(p * frac_ex))
    types = torch.cat([torch.randint(0, 12, (p - n_ex,), generator=gen), torch.full((n_ex,), 12)])
    t1 = F.one_hot(types, 13).float()
    ex = F.one_hot((types == 12).long(), 2).float()
    alpha = 0.5 + torch.rand(p, 1, generator=gen)
    has_norm = ((torch.rand(p, generator=gen) < 0.4) & (types != 12)).long()
    hn = F.one_hot(has_norm, 2).float()
    norm = torch.randn(p, 3, generator=gen)
    norm = norm / norm.norm(dim=-1, keepdim=True) * has_norm[:, None].float()
    pos = spread * torch.randn(p, 3, generator=gen)
    pos = pos - pos.mean(0, keepdim=True)
    return torch.cat([t1, alpha, hn, ex], -1), pos, norm


def synthetic_batch(seed, n_atoms, n_phore, t_values):
    """A synthetic PhoreDiff.forward input set (diffusion.py:175-178) for B=len(n_atoms) graphs."""
    gen = torch.Generator().manual_seed(seed)
    na = torch.tensor(n_atoms)
    B = len(n_atoms)
    batch_node = torch.repeat_interleave(torch.arange(B), na)
    edge_index, batch_edge = make_edge_data(na)
    N, E = int(na.sum()), edge_index.size(1)
    h_node = F.one_hot(torch.randint(0, 12, (N,), generator=gen), 12).float()
    h_edge = F.one_hot(torch.randint(0, 6, (E,), generator=gen), 6).float()
    pos = 2.5 * torch.randn(N, 3, generator=gen)
    hp, pp, pn, bp = [], [], [], []
    for gi, p in enumerate(n_phore):
        x, ps, nr = synthetic_phore(gen, p)
        hp.append(x), pp.append(ps), pn.append(nr), bp.append(torch.full((p,), gi))
    return dict(h_node_pert=h_node, pos_pert=pos, batch_node=batch_node, h_edge_pert=h_edge,
                edge_index=edge_index, batch_edge=batch_edge, time_step=torch.tensor(t_values),
                h_phore=torch.cat(hp), pos_phore=torch.cat(pp), phore_norm=torch.cat(pn),
                batch_phore=torch.cat(bp))


def synthetic_train_batch(seed, n_atoms, n_phore):
    """A synthetic `compute_loss` batch with the fields of SURVEY.md Appendix G (datasets/phoregen.py:356-384,
    datasets/transform.py:488-501): atom classes 0..10, bond classes 0..4 on the complete directed graph in the
    dst-major order of FeaturizeLigandBond, ligand coordinates centred on the pharmacophore centre."""
    gen = torch.Generator().manual_seed(seed)
    na = torch.tensor(n_atoms)
    B = len(n_atoms)
    off = torch.cat([torch.zeros(1, dtype=torch.long), na.cumsum(0)])
    srcs, dsts, attrs, ebat = [], [], [], []
    for gi, n in enumerate(n_atoms):
        dst = torch.repeat_interleave(torch.arange(n), n)
        src = torch.arange(n).repeat(n)
        m = dst != src
        src, dst = src[m], dst[m]
        sym = torch.randint(0, 5, (n, n), generator=gen)
        sym = torch.where(torch.rand(n, n, generator=gen) < 0.7, torch.zeros_like(sym), sym)     # mostly "no bond"
        sym = torch.triu(sym, 1)
        sym = sym + sym.t()
        srcs.append(src + off[gi]), dsts.append(dst + off[gi]), attrs.append(sym[src, dst])
        ebat.append(torch.full((src.numel(),), gi))
    N = int(na.sum())
    hp, pp, pn, bp = [], [], [], []
    for gi, p in enumerate(n_phore):
        x, ps, nr = synthetic_phore(gen, p)
        hp.append(x), pp.append(ps), pn.append(nr), bp.append(torch.full((p,), gi))
    return dict(ligand_x=torch.randint(0, 11, (N,), generator=gen), ligand_pos=1.5 * torch.randn(N, 3, generator=gen),
                ligand_batch=torch.repeat_interleave(torch.arange(B), na), ligand_ptr=off,
                f_edge_index=torch.stack([torch.cat(srcs), torch.cat(dsts)]), f_edge_attr=torch.cat(attrs),
                f_edge_batch=torch.cat(ebat), phore_x=torch.cat(hp), phore_pos=torch.cat(pp), phore_norm=torch.cat(pn),
                phore_batch=torch.cat(bp))
