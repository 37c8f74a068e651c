#!/usr/bin/env python3
"""The Q rows of a large batch behind P on lane 0 (chain_q_from = 150 k bond edges, the default) or on lane 2 at every size
above the v2 regime: ms per sampler step, same box, alternating.
Result (profiles/r06_ab_chain_q.txt): lane 2 is 0.2 - 0.8 % faster per step from 100 graphs up, but only by moving time between co-running kernels -- the
triplet sub-layer then shares the chip with more of the node chain (its span inside the step 1.98 -> 2.08 ms, roofline.frac 0.486 -> 0.463).  Not adopted.   GPU box: python tools/experiments/ab_chain_q.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import torch
from bench import config4_job, ligphore_workload
from fit_schedule import ms_per_step
from phoregen_amd.config import default_model_config
from phoregen_amd.models.diffusion import PhoreDiff
from phoregen_amd.weights import init_deterministic_
model = init_deterministic_(PhoreDiff(default_model_config(), 'zinc_300'), 0).eval().to('cuda')
work = [('headline', G, ligphore_workload(G)) for G in (100, 112, 128, 160)]
job = config4_job(n_phores=16, samples=8)
hp, pp, pn, bp, na, _ = job.batch_inputs(torch.arange(128))
work.append(('config4', 128, dict(h_phore=hp, pos_phore=pp, phore_norm=pn, batch_phore=bp, num_atoms=na)))
for shape, G, w in work:
    acc = {'lane0_from_150k': [], 'lane2': []}
    for rep in range(3):
        for lab, kw in (('lane0_from_150k', dict(chain_q_from=150000)), ('lane2', dict(chain_q_from=10 ** 9))):
            ms, nb = ms_per_step(model, w, K=30, **kw)
            acc[lab].append(ms)
    print(json.dumps(dict(shape=shape, graphs=G, n_bond=nb, ms={k: round(min(v), 3) for k, v in acc.items()}, all=acc)), flush=True)
