"""Import the reference's own Python (read-only, /root/reference) with the third-party stand-ins.

TEST INFRASTRUCTURE, build container only: /root/reference does not exist on the GPU box, and
nothing under phoregen_amd/ may import this module.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE = os.environ.get('PHOREGEN_REFERENCE', '/root/reference')


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE, 'models'))


def activate():
    """Put stand-ins + reference on sys.path (idempotent). The reference's top-level packages are
    called ``models``/``utils``/``datasets``; they are imported under those names."""
    if not available():
        raise RuntimeError(f'reference not found at {REFERENCE}')
    for p in (REFERENCE, os.path.join(HERE, 'standins')):
        if p not in sys.path:
            sys.path.insert(0, p)


def load_reference_config(name='train_lig-phore.yml'):
    activate()
    import yaml
    from easydict import EasyDict
    with open(os.path.join(REFERENCE, 'configs', name)) as f:
        cfg = EasyDict(yaml.safe_load(f))
    if cfg.dataset.data_name in ('zinc_300', 'pdbbind'):     # sample_all.py:41-43
        cfg.model.phore_feat_dim += 2
    return cfg
