"""`utils` alias: `utils.sample_utils` is phoregen_amd's; every other `utils.*` module (misc, training_utils, predict_bonds,
phore_utils) is still found in the reference checkout further down sys.path -- the package path is extended, not replaced."""
import os
import pkgutil
import sys

_REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _REPO not in sys.path:
    sys.path.append(_REPO)
__path__ = pkgutil.extend_path(__path__, __name__)
