"""knn_graph stand-in (torch-cluster 1.6.0 semantics, oracle tooling only; see ../../README.md).

`knn_graph(x, k, batch, loop=False, flow='source_to_target')` finds, for every node, its k nearest
nodes of the same graph (k+1 including itself, then the self pair is removed) and returns
`edge_index` with row 0 = neighbour (source) and row 1 = centre (target), grouped by centre in node
order, neighbours by ascending distance (index breaks exact ties).
"""
import torch


def knn_graph(x, k, batch=None, loop=False, flow='source_to_target', cosine=False, num_workers=1):
    assert flow in ('source_to_target', 'target_to_source')
    n = x.size(0)
    if batch is None:
        batch = torch.zeros(n, dtype=torch.long)
    rows, cols = [], []
    for g in torch.unique(batch).tolist():
        idx = (batch == g).nonzero().squeeze(-1)
        xg = x[idx]
        diff = xg[:, None, :] - xg[None, :, :]
        d2 = (diff * diff).sum(-1)
        kk = min(k if loop else k + 1, idx.numel())
        order = torch.argsort(d2, dim=1, stable=True)[:, :kk]
        centre = idx[:, None].expand(-1, kk)
        neigh = idx[order]
        if not loop:
            # torch_cluster: knn(k+1) then drop row == col; if the node itself is not among its k+1
            # nearest (coincident points), the k+1 found neighbours are all kept.
            keep = neigh != centre
            rows.append(centre[keep])
            cols.append(neigh[keep])
        else:
            rows.append(centre.reshape(-1))
            cols.append(neigh.reshape(-1))
    row = torch.cat(rows) if rows else torch.zeros(0, dtype=torch.long)
    col = torch.cat(cols) if cols else torch.zeros(0, dtype=torch.long)
    if flow == 'source_to_target':
        return torch.stack([col, row], 0)
    return torch.stack([row, col], 0)


def radius_graph(*a, **k):
    raise NotImplementedError


def knn(*a, **k):
    raise NotImplementedError


def radius(*a, **k):
    raise NotImplementedError
