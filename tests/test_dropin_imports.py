"""The `dropin/` alias packages: the reference's own import lines must resolve to phoregen_amd with ONE sys.path entry
(SURVEY.md 8(b) "Import surface the callers use").  Runs in a fresh interpreter so that this test session's own `models` /
`utils` imports cannot leak in.  With /root/reference present (build container only) the actual import blocks of
sample_all.py and run/run.py are executed unedited; on the GPU box only the alias-side assertions run."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = os.environ.get('PHOREGEN_REFERENCE', '/root/reference')


def _run(code, extra_path=()):
    env = dict(os.environ)
    env.pop('PYTHONPATH', None)
    prog = 'import sys\n' + ''.join(f'sys.path.insert(0, {p!r})\n' for p in reversed([os.path.join(ROOT, 'dropin'), *extra_path])) + \
        textwrap.dedent(code)
    r = subprocess.run([sys.executable, '-c', prog], capture_output=True, text=True, cwd='/tmp', env=env, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout


def test_alias_packages_resolve_to_phoregen_amd():
    out = _run('''
        from models.diffusion import PhoreDiff
        from models.model_utils import EMA
        import models
        from models.uni_denoiser import UniTransformerO2TwoUpdateGeneralBond, NodeUpdateLayer
        from models.common import MLP, GaussianSmearing, get_beta_schedule
        from models.transition import ContigousTransition, GeneralCategoricalTransition
        from utils.sample_utils import unbatch_data, decode_data, make_edge_data, sample_from_interval, MolReconsError
        assert PhoreDiff.__module__ == 'phoregen_amd.models.diffusion'
        assert models.get_denoiser_net.__module__ == 'phoregen_amd.models' and callable(models.get_phore_encoder)
        assert unbatch_data.__module__ == 'phoregen_amd.utils.sample_utils'
        from phoregen_amd.config import default_model_config
        m = PhoreDiff(default_model_config(), 'zinc_300')
        assert len(m.state_dict()) == 641 and callable(m.phore_encoder.forward)
        print('ok')
    ''')
    assert out.strip().endswith('ok')


def test_reference_scripts_import_unchanged():
    import pytest
    if not os.path.isdir(os.path.join(REFERENCE, 'models')):
        pytest.skip('reference checkout not present (GPU box)')
    standins = os.path.join(ROOT, 'oracle', 'standins')       # the third-party wheels this image lacks (test infrastructure)
    out = _run(f'''
        import ast
        def import_block(path, last_line):
            src = open(path).read()
            tree = ast.parse(src)
            keep = [n for n in tree.body if isinstance(n, (ast.Import, ast.ImportFrom)) and n.lineno <= last_line]
            exec(compile(ast.Module(body=keep, type_ignores=[]), path, 'exec'), globals())
        import_block({os.path.join(REFERENCE, 'sample_all.py')!r}, 12)          # sample_all.py:1-12, unedited
        assert PhoreDiff.__module__ == 'phoregen_amd.models.diffusion', PhoreDiff.__module__
        assert unbatch_data.__module__ == 'phoregen_amd.utils.sample_utils' and decode_data.__module__ == unbatch_data.__module__
        assert seed_all.__module__ == 'utils.misc' and 'reference' in sys.modules['utils.misc'].__file__   # the reference's own file
        import_block({os.path.join(REFERENCE, 'run', 'run.py')!r}, 13)           # run/run.py:1-13
        assert EMA.__module__ == 'phoregen_amd.models.model_utils'
        assert MolReconsError.__module__ in ('utils._reference_sample_utils', 'utils.sample_utils')
        print('ok')
    ''', extra_path=(REFERENCE, standins))
    assert out.strip().endswith('ok')
