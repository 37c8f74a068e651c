"""Stand-in for torch-sparse 0.6.15 `SparseTensor` (oracle tooling only; see ../README.md).

Only what `models/uni_denoiser.py:106-120` touches: construction from COO (entries kept sorted by
(row, col)), row gather with a 1-D index tensor (duplicates allowed, rows re-numbered in gather
order, columns ascending inside each row), `set_value(None)`, `sum(dim=1)` (entry counts when there
is no value) and `.storage.row()/.col()/.value()`.
"""
import torch


class _Storage:
    def __init__(self, row, col, value):
        self._row, self._col, self._value = row, col, value

    def row(self):
        return self._row

    def col(self):
        return self._col

    def value(self):
        return self._value


class SparseTensor:
    def __init__(self, row, col, value=None, sparse_sizes=None, _sorted=False):
        if not _sorted:
            n_col = int(sparse_sizes[1])
            perm = torch.argsort(row * n_col + col, stable=True)
            row, col = row[perm], col[perm]
            value = value[perm] if value is not None else None
        self.storage = _Storage(row, col, value)
        self._sizes = tuple(int(s) for s in sparse_sizes)

    def sparse_sizes(self):
        return self._sizes

    def __getitem__(self, idx):
        row, col, value = self.storage.row(), self.storage.col(), self.storage.value()
        n_row = self._sizes[0]
        counts = torch.bincount(row, minlength=n_row)
        rowptr = torch.zeros(n_row + 1, dtype=torch.long)
        rowptr[1:] = torch.cumsum(counts, 0)
        cnt = counts[idx]
        start = rowptr[idx]
        new_row = torch.repeat_interleave(torch.arange(idx.numel()), cnt)
        offs = torch.arange(int(cnt.sum())) - torch.repeat_interleave(torch.cumsum(cnt, 0) - cnt, cnt)
        take = torch.repeat_interleave(start, cnt) + offs
        return SparseTensor(new_row, col[take], value[take] if value is not None else None,
                            sparse_sizes=(idx.numel(), self._sizes[1]), _sorted=True)

    def set_value(self, value, layout=None):
        return SparseTensor(self.storage.row(), self.storage.col(), value,
                            sparse_sizes=self._sizes, _sorted=True)

    def sum(self, dim=1):
        assert dim == 1
        row, value = self.storage.row(), self.storage.value()
        if value is None:
            return torch.bincount(row, minlength=self._sizes[0])
        out = torch.zeros(self._sizes[0], dtype=value.dtype)
        return out.index_add_(0, row, value)
