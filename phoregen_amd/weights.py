"""Deterministic per-key weight generator.

There is no network here, so no checkpoint: both the HIP path and the CPU oracle are run with the
same synthetic parameters, generated independently for every ``state_dict`` key from
``crc32(key) ^ seed``.  Works on any module that follows the reference's naming
(SURVEY.md Appendix C): ``*.net.1.{weight,bias}`` are LayerNorm affine terms, other 2-D ``*.weight``
are Linear matrices, 1-D ``*.bias`` are Linear biases.  Frozen diffusion tables and fixed buffers
(``*_transition.*``, ``*.offset``, ``*.coeff``, ``*.freq_bands``) are left as the constructor made them.
"""
import math
import zlib

import torch

_FIXED_SUFFIX = ('.offset', '.coeff', '.freq_bands')
_FIXED_PREFIX = ('pos_transition.', 'node_transition.', 'edge_transition.')


def is_fixed(key: str) -> bool:
    return key.startswith(_FIXED_PREFIX) or key.endswith(_FIXED_SUFFIX)


def make_tensor(key: str, shape, seed: int = 0, gain: float = 1.0) -> torch.Tensor:
    g = torch.Generator(device='cpu')
    g.manual_seed((zlib.crc32(key.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)
    shape = tuple(shape)
    if key.endswith('net.1.weight'):            # LayerNorm gamma
        return 1.0 + 0.2 * torch.randn(shape, generator=g)
    if key.endswith('net.1.bias'):              # LayerNorm beta
        return 0.2 * torch.randn(shape, generator=g)
    if len(shape) == 2:                         # Linear weight [out, in]
        bound = gain * math.sqrt(3.0 / shape[1])
        return (torch.rand(shape, generator=g) * 2 - 1) * bound
    bound = 0.1                                 # Linear bias
    return (torch.rand(shape, generator=g) * 2 - 1) * bound


@torch.no_grad()
def init_deterministic_(module: torch.nn.Module, seed: int = 0, gain: float = 1.0):
    """Overwrite every learnable entry of ``module.state_dict()`` in place; returns the module."""
    sd = module.state_dict()
    for key, val in sd.items():
        if is_fixed(key) or not val.is_floating_point():
            continue
        val.copy_(make_tensor(key, val.shape, seed, gain).to(val.dtype))
    return module
