"""The frozen conditioning-floor table behind the sampler parity bounds (tests/golden/conditioning_floor.json): complete for
every recorded step of every sampler fixture, reproducible from oracle/make_conditioning_floor.py, and the multiplier is the
fixed one.  (CPU: the oracle against itself in float64 -- no HIP code involved, which is the point: the bound cannot move
with a kernel.)"""
import json
import os

import pytest
import torch

from helpers import FLOOR_MULT, GOLDEN, Oracle64, conditioning_floor, golden, make_oracle


def test_floor_table_is_complete_and_multiplier_fixed():
    from oracle.make_conditioning_floor import FIXTURES, K_ENSEMBLE
    assert FLOOR_MULT == 3.0
    with open(os.path.join(GOLDEN, 'conditioning_floor.json')) as f:
        tab = json.load(f)
    assert tab['k_ensemble'] == K_ENSEMBLE >= 8 and tab['outputs'] == ['v', 'x0', 'bond']
    assert sorted(tab['floor']) == sorted(FIXTURES)
    for name in FIXTURES:
        g = golden(name)
        n_rec = sum(1 for k in g.files if k.endswith('_out_v'))
        rows = tab['floor'][name]
        assert len(rows) == n_rec and all(len(r) == 3 and all(0 < v < 0.05 for v in r) for r in rows), name


@pytest.mark.parametrize('name,s', [('g5_sample_head3', 0), ('g5_sample_tail4_trained_like', 2), ('g5_sample_guid3', 1)])
def test_floor_table_reproduces(name, s):
    """Recompute three entries (a benign one, the two worst-conditioned ones) with the committed script."""
    from oracle.make_conditioning_floor import floors_of_step, profile_of, step_inputs
    torch.set_num_threads(8)
    prof = profile_of(name)
    fl = floors_of_step(make_oracle(0, prof), Oracle64(0, prof), step_inputs(golden(name), s))
    for a, b in zip(fl, conditioning_floor(name, s)):
        assert b / 1.5 <= a <= b * 1.5, (name, s, fl, conditioning_floor(name, s))
