"""Minimal pharmacophore containers + `.phore` parser for the sampler's input side.

`PhoreGraph` quacks like the slice of a PyG HeteroData that PhoreDiff.sample touches
(`data['phore'].x/.pos/.norm`, `data.center`, `data.name`), so a reference HeteroData works too.
"""
import os

import torch
import torch.nn.functional as F

PHORETYPES1 = ['MB', 'HD', 'AR', 'PO', 'HA', 'HY', 'NE', 'CV1', 'CV2', 'CV3', 'CV4', 'XB', 'EX']


class _Store(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__

    @property
    def num_nodes(self):
        return self['x'].size(0)


class PhoreGraph:
    def __init__(self, x, pos, norm, center=None, name=''):
        self._stores = {'phore': _Store(x=x, pos=pos, norm=norm)}
        self.center = center if center is not None else torch.zeros(3)
        self.name = name

    def __getitem__(self, key):
        return self._stores[key]

    def to(self, device):
        st = self._stores['phore']
        for k in list(st):
            st[k] = st[k].to(device)
        self.center = self.center.to(device)
        return self

    def clone(self):
        st = self._stores['phore']
        return PhoreGraph(st.x.clone(), st.pos.clone(), st.norm.clone(), self.center.clone(), self.name)


class TrainBatch:
    """The slice of a collated PyG HeteroData batch that PhoreDiff.compute_loss reads (SURVEY.md Appendix G;
    datasets/phoregen.py:356-384 + DataLoader(follow_batch=['f_edge_attr']), run/run.py:96-101)."""

    def __init__(self, ligand_x, ligand_pos, ligand_batch, ligand_ptr, f_edge_index, f_edge_attr, f_edge_batch,
                 phore_x, phore_pos, phore_norm, phore_batch, edge_index=None):
        """`edge_index` [2, n_bonds]: the molecule's bonds (`data['ligand', 'ligand'].edge_index`), read only by the
        `bond_len_loss` term of compute_loss (diffusion.py:286-290)."""
        self.num_graphs = int(ligand_ptr.numel() - 1)
        bonds = {} if edge_index is None else {'edge_index': edge_index}
        self._stores = {'ligand': _Store(x=ligand_x, pos=ligand_pos, batch=ligand_batch, ptr=ligand_ptr),
                        ('ligand', 'ligand'): _Store(f_edge_index=f_edge_index, f_edge_attr=f_edge_attr,
                                                     f_edge_attr_batch=f_edge_batch, **bonds),
                        'phore': _Store(x=phore_x, pos=phore_pos, norm=phore_norm, batch=phore_batch)}

    def __getitem__(self, key):
        if isinstance(key, tuple) and len(key) == 3:
            key = (key[0], key[2])
        return self._stores[key]

    def to(self, device):
        for st in self._stores.values():
            for k in list(st):
                st[k] = st[k].to(device)
        return self


def parse_phore_file(path, center=True):
    """`.phore` text -> PhoreGraph with the 18-wide feature row of datasets/get_phore_data.py:24-73
    (13 type one-hot | alpha | has_norm one-hot 2 | exclusion one-hot 2), positions centred on the
    pharmacophore centre of mass (:84-88)."""
    idx = {t: i for i, t in enumerate(PHORETYPES1)}
    types, alpha, pos, has_norm, norm = [], [], [], [], []
    with open(path) as f:
        f.readline()
        for line in f:
            line = line.strip()
            if line == '$$$$':
                break
            parts = line.split('\t')
            if len(parts) != 13:
                continue
            t, al, _w, _f, x, y, z, hn, nx, ny, nz, label, _aw = parts
            if t == 'CR':
                continue
            if t == 'CV':
                t += label[0]
            types.append(idx[t]), alpha.append(float(al)), pos.append([float(x), float(y), float(z)])
            has_norm.append(int(hn)), norm.append([float(nx), float(ny), float(nz)])
    tt = F.one_hot(torch.tensor(types), len(PHORETYPES1)).float()
    ex = F.one_hot(tt[:, -1].long(), 2).float()
    nrm = torch.tensor(norm, dtype=torch.float32)
    ln = nrm.norm(dim=-1, keepdim=True)
    unit = torch.where(ln > 0, nrm / ln.clamp(min=1e-30), torch.zeros_like(nrm))
    p = torch.tensor(pos, dtype=torch.float32)
    com = p.mean(0)
    x = torch.cat([tt, torch.tensor(alpha).unsqueeze(-1), F.one_hot(torch.tensor(has_norm), 2).float(), ex], -1)
    return PhoreGraph(x, p - com if center else p, unit, com if center else torch.zeros(3),
                      os.path.splitext(os.path.basename(path))[0])
