"""`models.model_utils` alias: re-exports phoregen_amd.models.model_utils (reference: models/model_utils.py); anything else the reference's module of
that name defines is looked up there on demand."""
from phoregen_amd.models import model_utils as _impl
from phoregen_amd.models.model_utils import *  # noqa: F401,F403

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith('__')})


def __getattr__(name):
    from . import reference_module
    ref = reference_module('model_utils')
    if ref is not None and hasattr(ref, name):
        return getattr(ref, name)
    raise AttributeError(f"module 'models.model_utils' has no attribute {name!r} (not part of phoregen_amd; no reference checkout on sys.path)")
