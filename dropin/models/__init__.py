"""`models` alias of phoregen_amd.models (reference: models/__init__.py:5-35).  Names of the reference's `models.*` modules
that are not part of the hot path (e.g. the dataset featuriser `models.common.get_neib_dist_feat`, used by
datasets/phoregen.py:15) are handed through to the reference's own files when a checkout is further down sys.path."""
import importlib.util
import os
import pkgutil
import sys

_REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _REPO not in sys.path:
    sys.path.append(_REPO)
_HERE = os.path.dirname(os.path.abspath(__file__))
__path__ = pkgutil.extend_path(__path__, __name__)

from phoregen_amd.models import get_denoiser_net, get_phore_encoder  # noqa: E402,F401
from phoregen_amd.models.uni_denoiser import NodeUpdateLayer, UniTransformerO2TwoUpdateGeneralBond  # noqa: E402,F401

_reference_modules = {}


def reference_module(name):
    """The reference's own `models/<name>.py` (the next one on the extended package path), loaded under a private module name;
    None when no checkout is on sys.path or its third-party imports are missing."""
    if name not in _reference_modules:
        mod = None
        for d in __path__:
            f = os.path.join(d, name + '.py')
            if os.path.abspath(d) != _HERE and os.path.isfile(f):
                try:
                    spec = importlib.util.spec_from_file_location(f'models._reference_{name}', f)
                    mod = importlib.util.module_from_spec(spec)
                    spec.loader.exec_module(mod)
                except Exception:
                    mod = None
                break
        _reference_modules[name] = mod
    return _reference_modules[name]
