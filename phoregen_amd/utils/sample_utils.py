"""Sampler-side helpers with the reference's names (utils/sample_utils.py:20-54)."""
import torch

from ..plan import make_edge_data  # noqa: F401  (same contract as utils/sample_utils.py:40-54)


def sample_from_interval(lower, upper, batch_size, mode='uniform', scale=4.0):
    """utils/sample_utils.py:28-37 — draws on the CPU default generator, as the reference does."""
    if mode == 'uniform':
        return torch.randint(lower, upper + 1, (batch_size,))
    if mode == 'normal':
        mid, std = (lower + upper) / 2, (upper - lower) / scale
        return torch.normal(mid, std, (batch_size,)).clamp(lower, upper).round().int()
    raise NotImplementedError(f'The sample nodes mode {mode} is not implemented.')


def get_fully_connected_edge(num_nodes):
    """utils/sample_utils.py:20-25 (self pairs kept)."""
    a = torch.arange(num_nodes)
    return torch.stack([torch.repeat_interleave(a, num_nodes), a.repeat(num_nodes)], 0)


# ---- device -> host hand-off (reference: utils/sample_utils.py:57-132, sample_all.py:104-116) ----------------------
ATOM_TYPES = [5, 6, 7, 8, 9, 14, 15, 16, 17, 35, 53]      # class 11 = masked atom (sample_utils.py:17)


def unbatch_data(results, n_graphs, include_bond=True):
    """Same contract as utils/sample_utils.py:57-93: per graph {'pred': [...], 'traj': [...], 'edge_index': local ids}.
    Graphs are contiguous in the sampler's output, so this slices by offsets (one pass) instead of building
    n_graphs boolean masks over every tensor; `traj` entries are views of the [T+1, N, .] trajectory tensors."""
    pred, traj = results['pred'], results['traj']
    num_atoms, edge_index = results['lig_info'][0], results['lig_info'][2]
    na = [int(v) for v in num_atoms.tolist()]
    out, n0, e0 = [], 0, 0
    for n in na[:n_graphs]:
        e = n * (n - 1)
        p = [pred[0][n0:n0 + n], pred[1][n0:n0 + n]]
        t = [traj[0][:, n0:n0 + n], traj[1][:, n0:n0 + n]] if traj[0] is not None else [None, None]
        if include_bond:
            p.append(pred[2][e0:e0 + e])
            t.append(traj[2][:, e0:e0 + e] if traj[2] is not None else None)
        out.append({'pred': p, 'traj': t, 'edge_index': edge_index[:, e0:e0 + e] - n0})
        n0, e0 = n0 + n, e0 + e
    return out


def decode_data(pred_info, edge_index, include_bond=True, num_bond_types=5):
    """Same contract as utils/sample_utils.py:96-132: argmax types, masked atoms (class 11) and their bonds dropped."""
    atom_type = pred_info[0].argmax(dim=-1)                       # softmax is monotone: argmax(logits)
    keep = atom_type < len(ATOM_TYPES)
    remap = torch.full((keep.numel(),), -1, dtype=torch.long, device=keep.device)
    remap[keep] = torch.arange(int(keep.sum()), device=keep.device)
    out = {'element': [ATOM_TYPES[i] for i in atom_type[keep].tolist()], 'atom_pos': pred_info[1][keep],
           'bond_type': None, 'bond_index': None}
    if include_bond:
        edge_type = pred_info[2].argmax(dim=-1)
        is_bond = (edge_type > 0) & (edge_type < num_bond_types)
        bond_index = remap[edge_index[:, is_bond]]
        ok = (bond_index >= 0).all(dim=0)
        out['bond_type'], out['bond_index'] = edge_type[is_bond][ok], bond_index[:, ok]
    return out


def decode_batch(results, include_bond=True, num_bond_types=5):
    """`[decode_data(unbatch_data(results)[g]) for g in graphs]` (sample_all.py:104-116) in one pass: argmax of all atom /
    bond logits on the device the results live on, ONE device->host copy of the compact arrays (types as int8, final
    coordinates, local edge ids), per-graph split by offsets on the host.  The logits and the trajectory stay where they are."""
    import numpy as np
    pred = results['pred']
    num_atoms, edge_index = results['lig_info'][0], results['lig_info'][2]
    na = np.asarray(num_atoms.tolist(), dtype=np.int64)
    at = pred[0].argmax(-1).to(torch.int8)
    et = pred[2].argmax(-1).to(torch.int8) if include_bond else None
    at_h, pos_h = at.cpu().numpy(), pred[1].detach().cpu().numpy()
    et_h = et.cpu().numpy() if include_bond else None
    ei_h = edge_index.cpu().numpy() if include_bond else None
    out, n0, e0 = [], 0, 0
    for n in na.tolist():
        e = n * (n - 1)
        a = at_h[n0:n0 + n]
        keep = a < len(ATOM_TYPES)
        d = {'element': [ATOM_TYPES[i] for i in a[keep].tolist()], 'atom_pos': torch.from_numpy(pos_h[n0:n0 + n][keep]),
             'bond_type': None, 'bond_index': None}
        if include_bond:
            remap = np.full(n, -1, dtype=np.int64)
            remap[keep] = np.arange(int(keep.sum()))
            t = et_h[e0:e0 + e]
            is_bond = (t > 0) & (t < num_bond_types)
            bi = remap[ei_h[:, e0:e0 + e][:, is_bond] - n0]
            ok = (bi >= 0).all(axis=0)
            d['bond_type'] = torch.from_numpy(t[is_bond][ok].astype(np.int64))
            d['bond_index'] = torch.from_numpy(bi[:, ok])
        out.append(d)
        n0, e0 = n0 + n, e0 + e
    return out
