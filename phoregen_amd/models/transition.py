"""Diffusion tables with the reference's names/shapes (models/transition.py:9-26,178-215).

The posterior arithmetic itself runs in csrc/posterior.hip; these modules own the frozen tables.
"""
import numpy as np
from torch import nn

from .common import frozen


class ContigousTransition(nn.Module):
    """Gaussian transition tables (models/transition.py:14-26)."""

    def __init__(self, betas, num_classes=None, scaling=1.):
        super().__init__()
        self.num_classes, self.scaling = num_classes, scaling
        alphas = 1. - betas
        bar = np.cumprod(alphas, axis=0)
        bar_prev = np.concatenate([[1.], bar[:-1]])
        self.betas, self.alphas = frozen(betas), frozen(alphas)
        self.alphas_bar, self.alphas_bar_prev = frozen(bar), frozen(bar_prev)
        self.coef_x0 = frozen(np.sqrt(bar_prev) * betas / (1 - bar))
        self.coef_xt = frozen(np.sqrt(alphas) * (1 - bar_prev) / (1 - bar))
        self.std = frozen(np.sqrt((1 - bar_prev) * betas / (1 - bar)))


class GeneralCategoricalTransition(nn.Module):
    """D3PM-style tables with an absorbing prior (models/transition.py:178-243)."""

    def __init__(self, betas, num_classes, init_prob=None):
        super().__init__()
        K = num_classes
        self.eps, self.num_classes, self.num_timesteps, self.betas = 1e-30, K, len(betas), betas
        if init_prob == 'absorb':
            p = 0.01 * np.ones(K)
            p[0] = 1
        elif init_prob == 'tomask':
            p = 0.001 * np.ones(K)
            p[-1] = 1.
        elif init_prob is None or init_prob == 'uniform':
            p = np.ones(K)
        else:
            p = np.asarray(init_prob, dtype=np.float64)
        self.init_prob = p / np.sum(p)
        one = np.stack([b * np.repeat(self.init_prob[None], K, 0) + np.eye(K) * (1. - b) for b in betas])
        cum, cur = [one[0]], one[0]
        for t in range(1, len(betas)):
            cur = np.tensordot(cur, one[t], axes=[[1], [0]])
            cum.append(cur)
        self.q_mats = frozen(np.stack(cum))
        self.transpopse_q_onestep_mats = frozen(np.transpose(one, (0, 2, 1)))
