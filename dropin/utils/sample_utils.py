"""`utils.sample_utils` alias (reference: utils/sample_utils.py): the sampler-side helpers and the device->host hand-off come
from phoregen_amd.utils.sample_utils; the RDKit / OpenBabel molecule reconstruction (`reconstruct_from_generated_with_edges`,
`MolReconsError`, ...) is outside the hot path and is handed through to the reference's own module when it is importable."""
import importlib.util
import os

from phoregen_amd.utils.sample_utils import (ATOM_TYPES, decode_batch, decode_data, get_fully_connected_edge,  # noqa: F401
                                             make_edge_data, sample_from_interval, unbatch_data)

_reference = None
_reference_error = None


def _load_reference():
    """The next `utils/sample_utils.py` on the (extended) package path = the reference's file, loaded under a private name."""
    global _reference, _reference_error
    if _reference is not None or _reference_error is not None:
        return _reference
    import utils
    here = os.path.dirname(os.path.abspath(__file__))
    for d in utils.__path__:
        f = os.path.join(d, 'sample_utils.py')
        if os.path.abspath(d) != here and os.path.isfile(f):
            try:
                spec = importlib.util.spec_from_file_location('utils._reference_sample_utils', f)
                mod = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(mod)
                _reference = mod
            except Exception as e:            # rdkit / openbabel missing: post-processing unavailable, the hot path is not
                _reference_error = e
            return _reference
    _reference_error = ImportError('no reference utils/sample_utils.py on sys.path')
    return None


class MolReconsError(Exception):
    """Placeholder with the reference's name (utils/sample_utils.py); replaced by the reference's class when importable."""


def __getattr__(name):
    ref = _load_reference()
    if ref is not None and hasattr(ref, name):
        return getattr(ref, name)
    if name == 'reconstruct_from_generated_with_edges':
        def unavailable(*a, **k):
            raise ImportError(f'{name} is RDKit/OpenBabel post-processing of the reference (utils/sample_utils.py), not part of '
                              f'phoregen_amd; the reference module could not be loaded: {_reference_error!r}')
        return unavailable
    raise AttributeError(name)


_ref = _load_reference()
if _ref is not None and hasattr(_ref, 'MolReconsError'):
    MolReconsError = _ref.MolReconsError          # noqa: F811
