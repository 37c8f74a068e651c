# One triplet launch with the unrolling picked per group vs two launches: needs tools/experiments/triplet2_per_group_unrolling.patch applied and the library rebuilt.
import json, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import torch
from fit_schedule import ms_per_step, workloads
from phoregen_amd.config import default_model_config
from phoregen_amd.models.diffusion import PhoreDiff
from phoregen_amd.weights import init_deterministic_
model = init_deterministic_(PhoreDiff(default_model_config(), 'zinc_300'), 0).eval().to('cuda')
for shape, G, w in workloads():
    if shape == 'config2': continue
    acc = {'split': [], 'one': []}
    for rep in range(3):
        for lab, kw in (('split', dict(tri_split='always')), ('one', dict(tri_split=False))):
            ms, nb = ms_per_step(model, w, K=30, **kw)
            acc[lab].append(ms)
    print(json.dumps(dict(shape=shape, graphs=G, n_bond=nb, split=min(acc['split']), one=min(acc['one']), all=acc)), flush=True)
