"""Parameter containers and schedule helpers with the reference's names (models/common.py).

Only what the hot path needs.  The arithmetic of the forward pass lives in the HIP kernels
(phoregen_amd/csrc) and is sequenced by phoregen_amd/engine.py; these modules exist so that
``state_dict`` keys/shapes, ``.to()``, ``.eval()``, ``.parameters()`` behave as in the reference.
"""
import math

import numpy as np
import torch
from torch import nn

# models/common.py:18 — the offsets every GaussianSmearing on the path uses (fix_offset=True)
FIXED_OFFSETS = (0, 1, 1.25, 1.5, 1.75, 2, 2.25, 2.5, 2.75, 3, 3.5, 4, 4.5, 5, 5.5, 6, 7, 8, 9, 10)


class GaussianSmearing(nn.Module):
    """Buffer holder for models/common.py:11-31 (`offset`; coeff is a python float there)."""

    def __init__(self, start=0.0, stop=5.0, num_gaussians=50, fix_offset=True):
        super().__init__()
        off = torch.tensor(FIXED_OFFSETS, dtype=torch.float32) if fix_offset else torch.linspace(start, stop, num_gaussians)
        self.num_gaussians = off.numel()
        self.coeff = -0.5 / float(off[1] - off[0]) ** 2
        self.register_buffer('offset', off)


class TimeGaussianSmearing(nn.Module):
    """Buffer holder for models/common.py:34-55 (`coeff`, `offset`, linear spacing)."""

    def __init__(self, stop, num_gaussians):
        super().__init__()
        off = torch.linspace(0.0, float(stop), num_gaussians)
        d = torch.diff(off)
        self.register_buffer('coeff', -0.5 / torch.cat([d[:1], d]) ** 2)
        self.register_buffer('offset', off)


class AngularEncoding(nn.Module):
    """models/common.py:67-87: freq_bands = [1,2,3,1,1/2,1/3]."""

    def __init__(self, num_funcs=3):
        super().__init__()
        self.register_buffer('freq_bands', torch.FloatTensor(
            [i + 1 for i in range(num_funcs)] + [1. / (i + 1) for i in range(num_funcs)]))


class ShiftedSoftplus(nn.Module):
    shift = math.log(2.0)


class MLP(nn.Module):
    """Linear -> LayerNorm -> ReLU -> Linear under `.net.{0,1,3}` (models/common.py:99-119)."""

    def __init__(self, in_dim, out_dim, hidden_dim):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(in_dim, hidden_dim), nn.LayerNorm(hidden_dim), nn.ReLU(),
                                 nn.Linear(hidden_dim, out_dim))


def frozen(x: np.ndarray):
    """models/common.py:386-389: tables are frozen nn.Parameters, so they live in checkpoints."""
    return nn.Parameter(torch.from_numpy(np.ascontiguousarray(x)).float(), requires_grad=False)


# ---- schedules (models/common.py:459-544), float64 numpy as in the reference ----
def _sig(x):
    return 1 / (np.exp(-x) + 1)


def _advance_bar(T, scale_start, scale_end, width):
    a = (scale_end - scale_start) / (_sig(-width) - _sig(width))
    b = 0.5 * (scale_end + scale_start - a)
    return a * _sig(-width * np.linspace(-1, 1, T)) + b


def _betas_of(bar):
    al = np.empty_like(bar)
    al[0] = bar[0]
    al[1:] = bar[1:] / bar[:-1]
    return np.clip(1 - al, 0, 1)


def get_beta_schedule(beta_schedule, num_timesteps, **kw):
    if beta_schedule == 'advance':
        betas = _betas_of(_advance_bar(num_timesteps, kw.get('scale_start', 0.999), kw.get('scale_end', 0.001),
                                       kw.get('width', 2)))
    elif beta_schedule == 'segment':
        bar = []
        for seg, prm in zip(kw['time_segment'], kw['segment_diff']):
            bar.extend(_advance_bar(seg + 1, prm['scale_start'], prm['scale_end'], prm['width'])[1:])
        assert len(bar) == num_timesteps
        betas = _betas_of(np.asarray(bar))
    elif beta_schedule == 'linear':
        betas = np.linspace(kw['beta_start'], kw['beta_end'], num_timesteps, dtype=np.float64)
    elif beta_schedule == 'sigmoid':
        s = kw.get('s', 6)
        betas = _sig(np.linspace(-s, s, num_timesteps)) * (kw['beta_end'] - kw['beta_start']) + kw['beta_start']
    else:
        raise NotImplementedError(beta_schedule)
    assert betas.shape == (num_timesteps,)
    return betas
