import torch


def remove_self_loops(edge_index, edge_attr=None):
    mask = edge_index[0] != edge_index[1]
    return edge_index[:, mask], (edge_attr[mask] if edge_attr is not None else None)


def k_hop_subgraph(*a, **k):
    raise NotImplementedError


def to_dense_adj(*a, **k):          # imported at module scope by datasets/phoregen.py (training dataset), never called here
    raise NotImplementedError


def dense_to_sparse(*a, **k):
    raise NotImplementedError
