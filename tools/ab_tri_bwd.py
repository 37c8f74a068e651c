"""Channel-split triplet adjoint (csrc/triplet_bwd2.hip, options.tri_bwd_form = 1 / 2) against the one-wave-per-tile adjoint: loss and every
parameter gradient on ragged batches whose largest ligand needs 2 / 3 / 4 row tiles.  GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from oracle.make_inputs import synthetic_train_batch
from phoregen_amd import options
from phoregen_amd.config import default_model_config
from phoregen_amd.data import TrainBatch
from phoregen_amd.models.diffusion import PhoreDiff
from phoregen_amd.weights import init_deterministic_

model = init_deterministic_(PhoreDiff(default_model_config(), 'zinc_300'), 0).to('cuda')
keys = ('ligand_x', 'ligand_pos', 'ligand_batch', 'ligand_ptr', 'f_edge_index', 'f_edge_attr', 'f_edge_batch',
        'phore_x', 'phore_pos', 'phore_norm', 'phore_batch')
BATCHES = ([2, 21, 3, 30, 9, 16, 17], [5, 33, 48, 2, 40], [2, 21, 3, 50, 64, 9])
FORMS = tuple(int(v) for v in sys.argv[1].split(',')) if len(sys.argv) > 1 else (1, 2)
WHICH = tuple(int(v) for v in sys.argv[2].split(',')) if len(sys.argv) > 2 else (0, 1, 2)
GRID = int(sys.argv[3]) if len(sys.argv) > 3 else 256
for sizes in [BATCHES[i] for i in WHICH]:
    b = synthetic_train_batch(80 + len(sizes), sizes, [5 + 3 * i for i in range(len(sizes))])
    gen = torch.Generator().manual_seed(7)
    N, E = b['ligand_x'].numel(), b['f_edge_attr'].numel()
    draws = dict(time_draw=torch.randint(10, 990, (len(sizes),), generator=gen), pos_noise=torch.randn(N, 3, generator=gen),
                 u_node=torch.rand(N, 12, generator=gen), u_edge=torch.rand(E, 6, generator=gen))
    out = {}
    for form in (0,) + FORMS:
        with options.override(tri_bwd_form=form, tri_bwd_grid=GRID):
            model._plan = None
            model.zero_grad()
            loss, _ = model.compute_loss(TrainBatch(*[b[k] for k in keys]), draws=draws)
            loss.backward()
            torch.cuda.synchronize()
        out[form] = (float(loss), {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None})
    ref = out[0][1]
    gmax = max(float(v.norm()) for v in ref.values())
    for form in FORMS:
        var = out[form][1]
        errs = []
        for k, r in ref.items():
            if float(r.norm()) < 1e-6 * gmax:
                continue
            errs.append((float((var[k].double() - r.double()).norm() / r.double().norm()), k))
        errs.sort(reverse=True)
        nan = [k for k, v in var.items() if not torch.isfinite(v).all()]
        print('ligands', sizes, 'form', form, 'loss', out[form][0], 'vs', out[0][0], 'worst', errs[:6], 'non-finite', nan[:5], flush=True)
