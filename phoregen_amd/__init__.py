"""phoregen_amd — MI355X-native (gfx950) implementation of PhoreGen's diffusion-denoising hot path.

Layout:
  csrc/       hand-written HIP kernels + the C ABI (include/phoregen_hip.h) -> _lib/libphoregen_hip.so
  hip.py      ctypes binding (no torch types cross the boundary)
  plan.py     per-batch topology (constant across the 1000 reverse steps)
  packing.py  weight layouts the kernels want (lane-fixed MFMA operands, fused first-layer blocks)
  engine.py   launch sequence of one denoiser forward / one sampler step
  models/     nn.Module mirror of the reference's models package (same names, same state_dict)
"""
__all__ = ['hip', 'plan', 'packing', 'engine', 'models', 'weights']
