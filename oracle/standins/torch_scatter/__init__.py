"""Stand-in for torch-scatter 2.0.9 (oracle tooling only; see ../README.md).

Semantics restated from the package documentation:
  scatter(src, index, dim, out, dim_size, reduce): out[index[i]] (+)= src[i] along `dim`;
  `reduce='mean'` divides by the per-slot count clamped to >= 1;
  scatter_softmax: exp(src - segmax[index]) / segsum(exp(...))[index]  (no epsilon in 2.0.9).
"""
import torch


def _broadcast(index, src, dim):
    if dim < 0:
        dim = src.dim() + dim
    if index.dim() == 1:
        for _ in range(0, dim):
            index = index.unsqueeze(0)
    for _ in range(index.dim(), src.dim()):
        index = index.unsqueeze(-1)
    return index.expand(src.size()), dim


def scatter_sum(src, index, dim=-1, out=None, dim_size=None):
    index, dim = _broadcast(index, src, dim)
    if out is None:
        size = list(src.size())
        if dim_size is not None:
            size[dim] = dim_size
        elif index.numel() == 0:
            size[dim] = 0
        else:
            size[dim] = int(index.max()) + 1
        out = torch.zeros(size, dtype=src.dtype, device=src.device)
    return out.scatter_add_(dim, index, src)


def scatter_mean(src, index, dim=-1, out=None, dim_size=None):
    out = scatter_sum(src, index, dim, out, dim_size)
    dim_size = out.size(dim)
    index_dim = dim
    if index_dim < 0:
        index_dim = index_dim + src.dim()
    if index.dim() <= index_dim:
        index_dim = index.dim() - 1
    ones = torch.ones(index.size(), dtype=src.dtype, device=src.device)
    count = scatter_sum(ones, index, index_dim, None, dim_size)
    count[count < 1] = 1
    count, _ = _broadcast(count, out, dim)
    if out.is_floating_point():
        out.true_divide_(count)
    else:
        out.div_(count, rounding_mode='floor')
    return out


def scatter_max(src, index, dim=-1, dim_size=None):
    index_b, dim = _broadcast(index, src, dim)
    size = list(src.size())
    size[dim] = dim_size if dim_size is not None else (int(index.max()) + 1 if index.numel() else 0)
    out = torch.full(size, float('-inf'), dtype=src.dtype, device=src.device)
    out = out.scatter_reduce(dim, index_b, src, reduce='amax', include_self=True)
    return out, None


def scatter(src, index, dim=-1, out=None, dim_size=None, reduce='sum'):
    if reduce in ('sum', 'add'):
        return scatter_sum(src, index, dim, out, dim_size)
    if reduce == 'mean':
        return scatter_mean(src, index, dim, out, dim_size)
    raise ValueError(reduce)


def scatter_softmax(src, index, dim=-1, dim_size=None):
    index_b, dim = _broadcast(index, src, dim)
    mx, _ = scatter_max(src, index, dim, dim_size)
    rec = src - mx.gather(dim, index_b)
    ex = rec.exp()
    sm = scatter_sum(ex, index, dim, None, dim_size)
    return ex / sm.gather(dim, index_b)
