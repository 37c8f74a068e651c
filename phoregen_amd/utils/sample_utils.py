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
