"""Minimal HeteroData / Batch / Dataset stand-ins (oracle tooling only; see ../../README.md)."""
import copy
import torch


class _Store(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    @property
    def num_nodes(self):
        for key in ('x', 'pos'):
            if key in self:
                return self[key].size(0)
        raise AttributeError('num_nodes')


class HeteroData:
    def __init__(self):
        object.__setattr__(self, '_stores', {})
        object.__setattr__(self, '_attrs', {})

    def __getitem__(self, key):
        if isinstance(key, tuple) and len(key) == 2:
            key = (key[0], 'to', key[1]) if (key[0], 'to', key[1]) in self._stores or \
                not any(k[0] == key[0] and k[-1] == key[1] for k in self._stores if isinstance(k, tuple)) \
                else next(k for k in self._stores if isinstance(k, tuple) and k[0] == key[0] and k[-1] == key[1])
        if key not in self._stores:
            self._stores[key] = _Store()
        return self._stores[key]

    def __getattr__(self, k):
        attrs = object.__getattribute__(self, '_attrs')
        if k in attrs:
            return attrs[k]
        raise AttributeError(k)

    def __setattr__(self, k, v):
        self._attrs[k] = v

    def clone(self):
        return copy.deepcopy(self)

    def to(self, device):
        for st in self._stores.values():
            for k, v in list(st.items()):
                if torch.is_tensor(v):
                    st[k] = v.to(device)
        for k, v in list(self._attrs.items()):
            if torch.is_tensor(v):
                self._attrs[k] = v.to(device)
        return self

    @property
    def node_types(self):
        return [k for k in self._stores if not isinstance(k, tuple)]


class Batch(HeteroData):
    @classmethod
    def from_data_list(cls, data_list, follow_batch=None, exclude_keys=None):
        out = cls()
        object.__setattr__(out, 'num_graphs', len(data_list))
        node_types = data_list[0].node_types
        offsets = {nt: 0 for nt in node_types}
        acc = {}
        for gi, d in enumerate(data_list):
            for key, st in d._stores.items():
                a = acc.setdefault(key, {})
                for k, v in st.items():
                    if not torch.is_tensor(v):
                        continue
                    if isinstance(key, tuple) and 'index' in k:
                        v = v + torch.tensor([[offsets[key[0]]], [offsets[key[-1]]]])
                    a.setdefault(k, []).append(v)
                if not isinstance(key, tuple):
                    n = st.num_nodes if ('x' in st or 'pos' in st) else 0
                    a.setdefault('batch', []).append(torch.full((n,), gi, dtype=torch.long))
            for nt in node_types:
                st = d._stores[nt]
                offsets[nt] += st.num_nodes if ('x' in st or 'pos' in st) else 0
        for key, a in acc.items():
            st = out[key] if not isinstance(key, tuple) else out._stores.setdefault(key, _Store())
            for k, vs in a.items():
                if isinstance(key, tuple) and 'index' in k:
                    st[k] = torch.cat(vs, dim=1)
                elif vs[0].dim() == 0:
                    st[k] = torch.stack(vs)
                else:
                    st[k] = torch.cat(vs, dim=0)
        for k, v in data_list[0]._attrs.items():
            if torch.is_tensor(v):
                out._attrs[k] = torch.stack([d._attrs[k] for d in data_list]) if v.dim() else v
            else:
                out._attrs[k] = [d._attrs[k] for d in data_list]
        return out


class Dataset:
    def __init__(self, root=None, transform=None, pre_transform=None, pre_filter=None):
        self.transform = transform

    def len(self):
        raise NotImplementedError

    def get(self, idx):
        raise NotImplementedError

    def __len__(self):
        return self.len()

    def __getitem__(self, idx):
        data = self.get(idx)
        return data if self.transform is None else self.transform(data)

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]


class Data(HeteroData):
    pass
