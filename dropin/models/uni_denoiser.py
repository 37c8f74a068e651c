"""`models.uni_denoiser` alias: re-exports phoregen_amd.models.uni_denoiser (reference: models/uni_denoiser.py); anything else the reference's module of
that name defines is looked up there on demand."""
from phoregen_amd.models import uni_denoiser as _impl
from phoregen_amd.models.uni_denoiser import *  # noqa: F401,F403

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith('__')})


def __getattr__(name):
    from . import reference_module
    ref = reference_module('uni_denoiser')
    if ref is not None and hasattr(ref, name):
        return getattr(ref, name)
    raise AttributeError(f"module 'models.uni_denoiser' has no attribute {name!r} (not part of phoregen_amd; no reference checkout on sys.path)")
