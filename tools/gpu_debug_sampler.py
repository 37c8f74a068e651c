"""Per-step error report of the HIP sampler against the recorded reference trajectories (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from helpers import golden, rel_err, t
from test_gpu_parity import _ReplayCpuRng, _tape
from phoregen_amd.config import default_model_config
from phoregen_amd.models.diffusion import PhoreDiff
from phoregen_amd.weights import init_deterministic_
from phoregen_amd.data import PhoreGraph

model = init_deterministic_(PhoreDiff(default_model_config(), 'zinc_300'), 0).eval().to('cuda')
for name in ['g5_sample_head3']:
    g = golden(name)
    data = PhoreGraph(t(g['phore_x']), t(g['phore_pos']), t(g['phore_norm']), t(g['center'])).to('cuda')
    t_total = int(g['t_total']); n_rec = sum(1 for k in g.files if k.endswith('_out_v'))
    guid = [{'type': 'atom_prox', 'min_d': 1.2, 'max_d': 1.9}, {'type': 'center_prox'}] if 'guid' in name else None
    recs = []
    def on_step(i, step, v, x0, bond):
        recs.append((v.cpu().clone(), x0.cpu().clone(), bond.cpu().clone()))
    old = model.num_timesteps
    if t_total != 1000: model.num_timesteps = t_total
    with _ReplayCpuRng(_tape(g)):
        res = model.sample(data, len(g['n_atoms']), 'cuda', pos_guidance_opt=guid, rng='cpu', num_atoms=t(g['n_atoms']),
                           num_steps=n_rec if t_total == 1000 else None, on_step=on_step)
    model.num_timesteps = old
    tn, tp, te = (a.cpu() for a in res['traj'])
    print(name)
    for s in range(n_rec):
        pos_in = tp[s] - (t(g['center']) if s > 0 else 0)
        v, x0, bond = recs[s]
        x0ref = t(g[f's{s}_out_x0'])
        print(f"  s{s}: node_eq={np.array_equal(tn[s].numpy(), g[f's{s}_h_node'])} edge_eq={np.array_equal(te[s].argmax(-1).numpy(), g[f's{s}_h_edge'])} "
              f"pos_in={rel_err(pos_in, g[f's{s}_pos']):.2e} abs_pos={float((pos_in - t(g[f's{s}_pos'])).abs().max()):.2e} "
              f"v={rel_err(v, g[f's{s}_out_v']):.2e} x0={rel_err(x0, x0ref):.2e} abs_x0={float((x0-x0ref).abs().max()):.2e} rmsd_x0={float(((x0-x0ref)**2).sum(-1).mean().sqrt()):.2e} bond={rel_err(bond, g[f's{s}_out_bond']):.2e}")
