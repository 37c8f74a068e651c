"""Loss and parameter gradients of compute_loss with one option switched against the default: usage ab_option.py name=value [name=value ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from oracle.make_inputs import synthetic_train_batch
from phoregen_amd import options
from phoregen_amd.config import default_model_config
from phoregen_amd.data import TrainBatch
from phoregen_amd.models.diffusion import PhoreDiff
from phoregen_amd.weights import init_deterministic_

kw = {}
for a in sys.argv[1:]:
    k, v = a.split('=')
    kw[k] = {'True': True, 'False': False}.get(v, int(v) if v.lstrip('-').isdigit() else v)
model = init_deterministic_(PhoreDiff(default_model_config(), 'zinc_300'), 0).to('cuda')
keys = ('ligand_x', 'ligand_pos', 'ligand_batch', 'ligand_ptr', 'f_edge_index', 'f_edge_attr', 'f_edge_batch',
        'phore_x', 'phore_pos', 'phore_norm', 'phore_batch')
for rep, sizes in enumerate(([2, 21, 3, 30, 9, 16, 17], [5, 33, 48, 2, 40])):
    b = synthetic_train_batch(80 + len(sizes), sizes, [5 + 3 * i for i in range(len(sizes))])
    gen = torch.Generator().manual_seed(7)
    N, E = b['ligand_x'].numel(), b['f_edge_attr'].numel()
    draws = dict(time_draw=torch.randint(10, 990, (len(sizes),), generator=gen), pos_noise=torch.randn(N, 3, generator=gen),
                 u_node=torch.rand(N, 12, generator=gen), u_edge=torch.rand(E, 6, generator=gen))
    out = {}
    for name, o in (('default', {}), ('variant', kw), ('default again', {})):
        with options.override(**o):
            model._plan = None
            model.zero_grad()
            loss, _ = model.compute_loss(TrainBatch(*[b[k] for k in keys]), draws=draws)
            loss.backward()
            torch.cuda.synchronize()
        out[name] = (float(loss.detach()), {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None})
    ref = out['default'][1]
    gmax = max(float(v.norm()) for v in ref.values())
    for name in ('variant', 'default again'):
        var = out[name][1]
        errs = sorted(((float((var[k].double() - r.double()).norm() / r.double().norm()), k) for k, r in ref.items()
                       if float(r.norm()) >= 1e-6 * gmax), reverse=True)
        print('ligands', sizes, name, kw if name == 'variant' else '', 'loss', out[name][0], 'vs', out['default'][0], 'worst', errs[:3],
              'missing', sorted(set(ref) ^ set(var))[:3], flush=True)
    # a parameter update between two steps must reach the replayed layouts
    with torch.no_grad():
        for p in model.parameters():
            p.mul_(1.001)
