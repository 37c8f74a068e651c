#!/usr/bin/env python3
"""Free-running parity report at scale (SURVEY.md 7 "hard parts"): the HIP sampler and the oracle (CPU) run the SAME S reverse steps
from t = 999 on the same B graphs with the same CPU-generator draws; per step, the number of graphs whose atom / bond types still agree
exactly and the coordinate RMSD over the agreeing graphs.

Two phases, because the oracle needs ~0.15 s per graph-step of CPU and the GPU box's minutes are budgeted:
  python tools/match_rate.py ref S B     (any host, no GPU)  -> tools/_cache/match_rate_ref_S_B.npz  (oracle trajectory + draw checksums;
                                                                 git-ignored, travels to the GPU box with the snapshot)
  python tools/match_rate.py hip S B     (GPU box)           -> one JSON line (kept as profiles/rNN_free_running_match_rate.json)
  python tools/match_rate.py S B         both in one process (small S x B)
"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch

args = sys.argv[1:]
phase = args.pop(0) if args and args[0] in ('ref', 'hip') else 'both'
S = int(args[0]) if len(args) > 0 else 40
B = int(args[1]) if len(args) > 1 else 12
SEED = 77
CACHE = os.path.join(ROOT, 'tools', '_cache', f'match_rate_ref_{S}_{B}.npz')
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'g8_phore_parse.npz'))
t = lambda a: torch.as_tensor(np.asarray(a))
x, pos, nrm, center = t(g['x']), t(g['pos']), t(g['norm']), t(g['center'])
gen = torch.Generator().manual_seed(4)
na = torch.randint(12, 28, (B,), generator=gen)


class SummingRng:
    """TorchCpuRng that also keeps a float64 sum of every draw (a host whose generator stream differs is told apart from a parity failure)."""

    def __init__(self):
        self.sums = []

    def _keep(self, v):
        self.sums.append(float(v.double().sum()))
        return v

    def randn(self, shape):
        return self._keep(torch.randn(shape))

    def rand(self, shape):
        return self._keep(torch.rand(shape))

    def rand64(self, shape):
        return self._keep(torch.rand(shape, dtype=torch.float64))


def run_ref():
    from helpers import make_oracle
    torch.set_num_threads(int(os.environ.get('MATCH_RATE_THREADS', min(16, os.cpu_count() or 1))))
    orc = make_oracle(0)
    rng = SummingRng()
    torch.manual_seed(SEED)
    t0 = time.time()
    with torch.no_grad():
        ref = orc.sample(x, pos, nrm, center, na, rng, n_steps=S)
    out = dict(node=ref['traj'][0].argmax(-1).to(torch.int8).numpy(), edge=ref['traj'][2].argmax(-1).to(torch.int8).numpy(),
               pos=ref['traj'][1].numpy(), sums=np.array(rng.sums), oracle_s=np.array(time.time() - t0), na=na.numpy())
    os.makedirs(os.path.dirname(CACHE), exist_ok=True)
    np.savez_compressed(CACHE, **out)
    return out


def run_hip(ref):
    from phoregen_amd.config import default_model_config
    from phoregen_amd.data import PhoreGraph
    from phoregen_amd.models.diffusion import PhoreDiff
    from phoregen_amd.weights import init_deterministic_
    assert np.array_equal(ref['na'], na.numpy())
    model = init_deterministic_(PhoreDiff(default_model_config(), 'zinc_300'), 0).eval().to('cuda')
    sums, real = [], {n: getattr(torch, n) for n in ('rand', 'randn')}

    def summing(n):
        def f(*a, **k):
            v = real[n](*a, **k)
            sums.append(float(v.double().sum()))
            return v
        return f
    torch.manual_seed(SEED)
    torch.rand, torch.randn = summing('rand'), summing('randn')
    try:
        t0 = time.time()
        res = model.sample(PhoreGraph(x, pos, nrm, center).to('cuda'), B, 'cuda', rng='cpu', num_atoms=na, num_steps=S)
        torch.cuda.synchronize()
        hip_s = time.time() - t0
    finally:
        torch.rand, torch.randn = real['rand'], real['randn']
    sums, rs = np.array(sums), np.asarray(ref['sums'])
    assert len(sums) == len(rs) == 3 + 3 * S
    uniform = np.ones(len(rs), dtype=bool)
    uniform[0] = False
    uniform[5::3] = False                                      # randn draws: the initial positions, then the third draw of every step
    assert np.array_equal(sums[uniform], rs[uniform]), 'torch CPU generator stream differs from the host the reference phase ran on'
    randn_dev = float(np.abs(sums[~uniform] - rs[~uniform]).max())
    bn, be = res['lig_info'][1].cpu(), res['lig_info'][3].cpu()
    tn, te, tp = res['traj'][0].argmax(-1).cpu(), res['traj'][2].argmax(-1).cpu(), res['traj'][1].cpu()
    rn, re_, rp = t(ref['node']).long(), t(ref['edge']).long(), t(ref['pos'])
    cnt = torch.bincount(bn, minlength=B).float()
    ok = torch.ones(B, dtype=torch.bool)
    rows, first_bad = [], {}
    for s in range(S + 1):
        same = (torch.zeros(B).index_add(0, bn, (tn[s] != rn[s]).float()) == 0) & (torch.zeros(B).index_add(0, be, (te[s] != re_[s]).float()) == 0)
        for gi in (ok & ~same).nonzero().flatten().tolist():
            first_bad[gi] = s
        ok &= same                                             # a graph that has differed once stays out (its later states are another trajectory)
        d2 = ((tp[s] - rp[s]) ** 2).sum(-1)
        per_graph = (torch.zeros(B).index_add(0, bn, d2) / cnt).sqrt()
        rows.append((s, int(ok.sum()), float(per_graph[ok].max()) if ok.any() else float('nan')))
    marks = sorted({0, 1, 2, 5, 10, 20, 30, 50, 100, 200, 300, 400, 500, 600, 700, 800, 900, 950, 990, S} & set(range(S + 1)))
    return {'graphs': B, 'steps': S, 'atoms': int(na.sum()), 'bond_rows': int((na * (na - 1)).sum()), 'oracle_s': round(float(ref['oracle_s']), 1),
            'hip_s': round(hip_s, 1), 'graphs_identical_through_step': {r[0]: r[1] for r in rows if r[0] in marks},
            'worst_graph_rmsd_over_identical_graphs': {r[0]: r[2] for r in rows if r[0] in marks and r[0] > 0},
            'worst_rmsd_any_step': max(r[2] for r in rows[1:] if r[2] == r[2]), 'first_differing_step_by_graph': first_bad,
            'match_rate_final': rows[-1][1] / B, 'randn_checksum_max_dev_between_hosts': randn_dev}


if phase == 'ref':
    out = run_ref()
    print(json.dumps({'wrote': os.path.relpath(CACHE, ROOT), 'oracle_s': round(float(out['oracle_s']), 1)}))
elif phase == 'hip':
    print(json.dumps(run_hip(np.load(CACHE))))
else:
    print(json.dumps(run_hip(run_ref())))
