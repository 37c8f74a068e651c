#!/usr/bin/env python3
"""The larger ligands' triplet launch BESIDE the other one (options.tri_overlap = side lane) vs behind it on lane 0: ms per sampler step,
alternating, and bit-equality of a 12-step sample.   GPU box: python tools/experiments/ab_tri_overlap.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import torch
from fit_schedule import ms_per_step, workloads
from phoregen_amd import options
from phoregen_amd.config import default_model_config
from phoregen_amd.models.diffusion import PhoreDiff
from phoregen_amd.weights import init_deterministic_
model = init_deterministic_(PhoreDiff(default_model_config(), 'zinc_300'), 0).eval().to('cuda')
VAR = [('one_launch', dict(tri_split=False)), ('behind', dict(tri_split='always', tri_overlap=0)), ('lane3', dict(tri_split='always', tri_overlap=3)), ('lane2', dict(tri_split='always', tri_overlap=2))]


def sample(w, **kw):
    G = int(w['num_atoms'].numel())
    with options.override(**kw):
        model._engine = None
        r = model.sample_batch(w['h_phore'], w['pos_phore'], w['phore_norm'], w['batch_phore'], w['num_atoms'], torch.zeros(G, 3), rng='device', seed=3, num_steps=12)
        model._engine = None
    return [t.clone() for t in r['pred']]


for shape, G, w in workloads():
    if shape == 'config2' or G < 16:
        continue
    acc = {l: [] for l, _ in VAR}
    for rep in range(3):
        for lab, kw in VAR:
            ms, nb = ms_per_step(model, w, K=30, **kw)
            acc[lab].append(ms)
    ref = sample(w, **VAR[0][1])
    same = {lab: all(torch.equal(a, b) for a, b in zip(ref, sample(w, **kw))) for lab, kw in VAR[1:]}
    print(json.dumps(dict(shape=shape, graphs=G, n_bond=nb, ms={l: min(v) for l, v in acc.items()}, identical=same)), flush=True)
