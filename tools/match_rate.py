#!/usr/bin/env python3
"""Free-running parity report (SURVEY.md 7 "hard parts"): the HIP sampler and the oracle (CPU) run the SAME first S
reverse steps from t = 999 on the same graphs with the same CPU-generator draws; per step, the share of graphs whose
atom / bond types still agree exactly and the coordinate RMSD over the agreeing graphs.  GPU box: python tools/match_rate.py"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch
from helpers import make_oracle
from phoregen_amd.config import default_model_config
from phoregen_amd.data import PhoreGraph
from phoregen_amd.models.diffusion import PhoreDiff
from phoregen_amd.weights import init_deterministic_
from oracle import phoregen_oracle as po

S = int(sys.argv[1]) if len(sys.argv) > 1 else 40
B = int(sys.argv[2]) if len(sys.argv) > 2 else 12
torch.set_num_threads(min(16, os.cpu_count() or 1))
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'g8_phore_parse.npz'))
t = lambda a: torch.as_tensor(np.asarray(a))
x, pos, nrm, center = t(g['x']), t(g['pos']), t(g['norm']), t(g['center'])
gen = torch.Generator().manual_seed(4)
na = torch.randint(12, 28, (B,), generator=gen)
model = init_deterministic_(PhoreDiff(default_model_config(), 'zinc_300'), 0).eval().to('cuda')
orc = make_oracle(0)
torch.manual_seed(77)
t0 = time.time()
with torch.no_grad():
    ref = orc.sample(x, pos, nrm, center, na, po.TorchCpuRng(), n_steps=S)
t_ref = time.time() - t0
torch.manual_seed(77)
res = model.sample(PhoreGraph(x, pos, nrm, center).to('cuda'), B, 'cuda', rng='cpu', num_atoms=na, num_steps=S)
bn, be = res['lig_info'][1].cpu(), res['lig_info'][3].cpu()
rows = []
for s in range(S + 1):
    tn, te = res['traj'][0][s].cpu().argmax(-1), res['traj'][2][s].cpu().argmax(-1)
    rn, re_ = ref['traj'][0][s].argmax(-1), ref['traj'][2][s].argmax(-1)
    ok = torch.ones(B, dtype=torch.bool)
    ok &= torch.zeros(B).index_add(0, bn, (tn != rn).float()) == 0
    ok &= torch.zeros(B).index_add(0, be, (te != re_).float()) == 0
    d2 = ((res['traj'][1][s].cpu() - ref['traj'][1][s]) ** 2).sum(-1)
    sel = ok[bn]
    rows.append((s, int(ok.sum()), float(d2[sel].mean().sqrt()) if sel.any() else float('nan')))
print(json.dumps({'graphs': B, 'steps': S, 'atoms': int(na.sum()), 'oracle_s': round(t_ref, 1),
                  'graphs_with_identical_types_after_step': {r[0]: r[1] for r in rows if r[0] in (0, 1, 2, 5, 10, 20, 30, S)},
                  'rmsd_over_identical_graphs': {r[0]: r[2] for r in rows if r[0] in (1, 2, 5, 10, 20, 30, S)}}))
