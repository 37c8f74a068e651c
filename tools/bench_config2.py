#!/usr/bin/env python3
"""BASELINE.json configs[1]: sample_all.py --num_samples 100 on ONE pharmacophore (the 44-node P03211_merge shape recorded
in tests/golden/g8_phore_parse.npz), 1000 denoise steps, 1x MI355X, guidance off and with sample.sh's guidance options.
Prints one JSON line per variant (wall time of the whole `sample` call, incl. topology, encoder, result tensors)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from phoregen_amd.config import default_model_config
from phoregen_amd.data import PhoreGraph
from phoregen_amd.models.diffusion import PhoreDiff
from phoregen_amd.weights import init_deterministic_

g = np.load(os.path.join(ROOT, 'tests', 'golden', 'g8_phore_parse.npz'))
t = lambda a: torch.as_tensor(np.asarray(a))
data = PhoreGraph(t(g['x']), t(g['pos']), t(g['norm']), t(g['center'])).to('cuda')
model = init_deterministic_(PhoreDiff(default_model_config(), 'zinc_300'), 0).eval().to('cuda')
torch.manual_seed(2032)
guid = [{'type': 'atom_prox', 'min_d': 1.2, 'max_d': 1.9}, {'type': 'center_prox'}]
na = torch.randint(20, 45, (100,))                       # random-init count heads are meaningless: fixed draw of sizes
model.sample(data, 100, 'cuda', num_atoms=na, num_steps=3, return_traj=False)      # warm-up (library load, plan)
for name, opt in (('guidance off', None), ('guidance atom_prox + center_prox', guid)):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = model.sample(data, 100, 'cuda', pos_guidance_opt=opt, num_atoms=na, return_traj=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(json.dumps({'metric': 'sample(100 graphs, 1000 steps) wall time', 'value': dt, 'unit': 's', 'variant': name,
                      'steps_per_sec': 1000 / dt, 'n_lig': int(na.sum()), 'e_bond': int((na * (na - 1)).sum()),
                      'traj_gb': sum(x.numel() * 4 for x in res['traj']) / 2**30}))
