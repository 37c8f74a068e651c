"""TEST INFRASTRUCTURE (oracle side).  Conditioning floor of every recorded sampler step of the g5 fixtures.

The sampler fixtures record states on which a randomly initialised network amplifies fp32 rounding 1e3..1e4 x (DESIGN.md
"parity"): there no fp32 implementation can be asked to agree with the reference's fp32 output more closely than fp32
implementations of the SAME dataflow agree with the exact result.  That distance is measured here, once, as an ENSEMBLE:

  floor(fixture, step, output) = max over K = 12 fp32 evaluations of the oracle (the reference's dataflow) of
                                 rel_err(evaluation_k, float64 evaluation of the unperturbed state)

Evaluation 0 is the state as recorded.  Evaluations 1.. permute the ligand atoms inside every graph and the bond rows inside
every graph (a different summation order of every scatter / neighbour list, results permuted back) and move every ligand and
pharmacophore coordinate by -1, 0 or +1 ulp (seeded).  The table is committed (tests/golden/conditioning_floor.json) and the
GPU tests assert  |hip - reference| <= max(5 x TOL, FLOOR_MULT x floor)  with FLOOR_MULT = 3 fixed in tests/helpers.py: the bound
depends on the fixtures only, not on the kernels under test.

Run here (CPU, ~2 min):  python oracle/make_conditioning_floor.py
"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]

K_ENSEMBLE = 12
FIXTURES = ['g5_sample_head3', 'g5_sample_tail4', 'g5_sample_full25', 'g5_sample_guid3',
            'g5_sample_head3_gamma_signed', 'g5_sample_tail4_gamma_signed',
            'g5_sample_head3_trained_like', 'g5_sample_tail4_trained_like']
PROFILES = ('gamma_signed', 'trained_like')
OUT = os.path.join(ROOT, 'tests', 'golden', 'conditioning_floor.json')


def profile_of(name):
    return next((p for p in PROFILES if name.endswith('_' + p)), 'default')


def step_inputs(g, s):
    """The 11 forward inputs of recorded step `s` (diffusion.py:432-447)."""
    from oracle import phoregen_oracle as po
    na = torch.as_tensor(g['n_atoms'])
    B, p = len(na), g['phore_x'].shape[0]
    ei, be = po.make_edge_data(na)
    tt = lambda a: torch.as_tensor(np.asarray(a))
    return dict(h_node_pert=tt(g[f's{s}_h_node']), pos_pert=tt(g[f's{s}_pos']),
                batch_node=torch.repeat_interleave(torch.arange(B), na),
                h_edge_pert=F.one_hot(tt(g[f's{s}_h_edge']).long(), 6).float(), edge_index=ei, batch_edge=be,
                time_step=tt(g[f's{s}_t']), h_phore=tt(g['phore_x']).repeat(B, 1), pos_phore=tt(g['phore_pos']).repeat(B, 1),
                phore_norm=tt(g['phore_norm']).repeat(B, 1), batch_phore=torch.repeat_interleave(torch.arange(B), p))


def _ulp_jitter(x, gen):
    """Every element moved by -1, 0 or +1 ulp."""
    d = torch.randint(-1, 2, x.shape, generator=gen)
    up, dn = torch.nextafter(x, torch.full_like(x, float('inf'))), torch.nextafter(x, torch.full_like(x, -float('inf')))
    return torch.where(d > 0, up, torch.where(d < 0, dn, x))


def _perm_within(batch, gen):
    """Random permutation of the rows that keeps every graph's rows in its (sorted) block."""
    key = batch.double() + torch.rand(batch.numel(), generator=gen, dtype=torch.float64) * 0.5
    return torch.argsort(key)


def variant(inp, k):
    """(inputs of evaluation k, function that maps its outputs back to the recorded row order)."""
    if k == 0:
        return inp, lambda v, x0, bond: (v, x0, bond)
    gen = torch.Generator().manual_seed(7919 * k + 13)
    perm = _perm_within(inp['batch_node'], gen)               # new row r holds old atom perm[r]
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(perm.numel())
    eperm = _perm_within(inp['batch_edge'], gen)
    einv = torch.empty_like(eperm)
    einv[eperm] = torch.arange(eperm.numel())
    out = dict(inp)
    out['h_node_pert'] = inp['h_node_pert'][perm]
    out['pos_pert'] = _ulp_jitter(inp['pos_pert'], gen)[perm]
    out['edge_index'] = inv[inp['edge_index']][:, eperm]
    out['h_edge_pert'] = inp['h_edge_pert'][eperm]
    out['pos_phore'] = _ulp_jitter(inp['pos_phore'], gen)
    return out, lambda v, x0, bond: (v[inv], x0[inv], bond[einv])


def floors_of_step(o32, o64, inp, k_ensemble=K_ENSEMBLE):
    from helpers import rel_err
    r64 = o64.forward(**inp)
    fl = [0.0, 0.0, 0.0]
    with torch.no_grad():
        for k in range(k_ensemble):
            vin, back = variant(inp, k)
            out = back(*o32.forward(**vin)[:3])
            fl = [max(f, rel_err(out[i], r64[i])) for i, f in enumerate(fl)]
    return fl


def main():
    from helpers import Oracle64, golden, make_oracle
    torch.set_num_threads(8)
    table = {'k_ensemble': K_ENSEMBLE, 'outputs': ['v', 'x0', 'bond'], 'floor': {}}
    orc = {}
    for name in FIXTURES:
        prof = profile_of(name)
        if prof not in orc:
            orc[prof] = (make_oracle(0, prof), Oracle64(0, prof))
        g = golden(name)
        n_rec = sum(1 for k in g.files if k.endswith('_out_v'))
        rows = []
        for s in range(n_rec):
            rows.append([float('%.4e' % f) for f in floors_of_step(*orc[prof], step_inputs(g, s))])
            print(name, s, int(g[f's{s}_t'][0]), rows[-1], flush=True)
        table['floor'][name] = rows
    with open(OUT, 'w') as f:
        json.dump(table, f, indent=1)
    print('wrote', OUT)


if __name__ == '__main__':
    main()
