"""Race hunt for the pipelined sampler loop (options.step_ahead): `sample_batch(pipeline=True)` against the plain loop on the same inputs and
seed, S steps, several batch sizes (every schedule regime), with and without guidance, R repetitions of the pipelined run.  Any differing bit in
any trajectory frame is a missing order point.   usage: stress_pipeline.py [steps] [repeats]"""
import sys, torch
import os
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _ROOT)
from bench import ligphore_workload
from phoregen_amd.config import default_model_config
from phoregen_amd.models.diffusion import PhoreDiff
from phoregen_amd.weights import init_deterministic_

S = int(sys.argv[1]) if len(sys.argv) > 1 else 200
R = int(sys.argv[2]) if len(sys.argv) > 2 else 3
GUID = [{'type': 'atom_prox', 'min_d': 1.2, 'max_d': 1.9}, {'type': 'center_prox'}]
model = init_deterministic_(PhoreDiff(default_model_config(), 'zinc_300'), 0).eval().to('cuda')
bad = 0
for G in (2, 6, 16, 24, 48, 72, 128):
    w = ligphore_workload(G, seed=700 + G)
    args = (w['h_phore'], w['pos_phore'], w['phore_norm'], w['batch_phore'], w['num_atoms'], torch.zeros(G, 3))
    for guid in (None, GUID):
        model._engine = None
        ref = model.sample_batch(*args, rng='device', seed=3, num_steps=S, pos_guidance_opt=guid, pipeline=False)
        ref = [t.clone() for t in ref['traj']] + [t.clone() for t in ref['pred']]
        mism = 0
        for r in range(R):
            model._engine = None
            out = model.sample_batch(*args, rng='device', seed=3, num_steps=S, pos_guidance_opt=guid, pipeline=True)
            assert model._engine.prog_step is not None
            # (a pharmacophore of exclusion spheres only has no centre: 0 / 0 = nan, as in the reference, and its graph's guided coordinates
            #  are nan in both loops -- compared as equal)
            same = lambda a, b: torch.equal(torch.isnan(a), torch.isnan(b)) and torch.equal(torch.nan_to_num(a), torch.nan_to_num(b))
            if not all(same(a, b) for a, b in zip(out['traj'] + out['pred'], ref)):
                mism += 1
        bad += mism
        print(f'G={G:4d} guidance={"on " if guid else "off"}: {R} pipelined runs of {S} steps, {mism} differ from the plain loop', flush=True)
print('RACE-FREE' if bad == 0 else f'MISMATCHES: {bad}')
sys.exit(1 if bad else 0)
