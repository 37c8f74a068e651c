"""Deterministic per-key weight generator.

There is no network here, so no checkpoint: both the HIP path and the CPU oracle are run with the
same synthetic parameters, generated independently for every ``state_dict`` key from
``crc32(key) ^ seed``.  Works on any module that follows the reference's naming
(SURVEY.md Appendix C): ``*.net.1.{weight,bias}`` are LayerNorm affine terms, other 2-D ``*.weight``
are Linear matrices, 1-D ``*.bias`` are Linear biases.  Frozen diffusion tables and fixed buffers
(``*_transition.*``, ``*.offset``, ``*.coeff``, ``*.freq_bands``) are left as the constructor made them.

Named profiles (parity must not depend on one benign weight set):
  ``default``       LayerNorm gamma = 1 + 0.2 N(0,1) (always positive), beta = 0.2 N(0,1), Linear ~ U(+-sqrt(3/fan_in)).
  ``gamma_signed``  gamma ~ N(0,1): about half of the channels negative, |gamma| < 1e-3 in ~0.1 % of them, and every
                    LayerNorm gets 3 channels set to exactly 0 and 3 to +-1e-6 (dead / nearly dead channels of a
                    trained checkpoint); beta ~ 0.5 N(0,1).  Exercises the sign fold, the |gamma| division and the
                    dead-channel path of packing._kv_mlp.
  ``trained_like``  Linear gain 2.5, Linear bias U(+-0.5), gamma = 1 + 0.5 N(0,1), beta ~ U(-2, 2): the larger
                    activations, sharper softmaxes and wider LayerNorm shifts of trained weights.
"""
import math
import zlib

import torch

_FIXED_SUFFIX = ('.offset', '.coeff', '.freq_bands')
_FIXED_PREFIX = ('pos_transition.', 'node_transition.', 'edge_transition.')
PROFILES = ('default', 'gamma_signed', 'trained_like')


def is_fixed(key: str) -> bool:
    return key.startswith(_FIXED_PREFIX) or key.endswith(_FIXED_SUFFIX)


def make_tensor(key: str, shape, seed: int = 0, gain: float = 1.0, profile: str = 'default') -> torch.Tensor:
    if profile not in PROFILES:
        raise ValueError(f'unknown weight profile {profile!r} (known: {PROFILES})')
    g = torch.Generator(device='cpu')
    g.manual_seed((zlib.crc32(key.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)
    shape = tuple(shape)
    if key.endswith('net.1.weight'):            # LayerNorm gamma
        if profile == 'gamma_signed':
            gam = torch.randn(shape, generator=g)
            n = gam.numel()
            idx = torch.randperm(n, generator=g)[:6]
            flat = gam.view(-1)
            flat[idx[:3]] = 0.0
            flat[idx[3]], flat[idx[4]], flat[idx[5]] = 1e-6, -1e-6, 1e-6
            return gam
        if profile == 'trained_like':
            return 1.0 + 0.5 * torch.randn(shape, generator=g)
        return 1.0 + 0.2 * torch.randn(shape, generator=g)
    if key.endswith('net.1.bias'):              # LayerNorm beta
        if profile == 'gamma_signed':
            return 0.5 * torch.randn(shape, generator=g)
        if profile == 'trained_like':
            return torch.rand(shape, generator=g) * 4 - 2
        return 0.2 * torch.randn(shape, generator=g)
    if profile == 'trained_like':
        gain = gain * 2.5
    if len(shape) == 2:                         # Linear weight [out, in]
        bound = gain * math.sqrt(3.0 / shape[1])
        return (torch.rand(shape, generator=g) * 2 - 1) * bound
    bound = 0.5 if profile == 'trained_like' else 0.1        # Linear bias
    return (torch.rand(shape, generator=g) * 2 - 1) * bound


@torch.no_grad()
def init_deterministic_(module: torch.nn.Module, seed: int = 0, gain: float = 1.0, profile: str = 'default'):
    """Overwrite every learnable entry of ``module.state_dict()`` in place; returns the module."""
    sd = module.state_dict()
    for key, val in sd.items():
        if is_fixed(key) or not val.is_floating_point():
            continue
        val.copy_(make_tensor(key, val.shape, seed, gain, profile).to(val.dtype))
    return module
