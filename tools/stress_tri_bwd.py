"""Race hunt for the channel-split triplet adjoint (csrc/triplet_bwd2.hip): the same training step many times; the outputs of the
FIRST triplet adjoint of every backward (layer 5: its inputs come through deterministic kernels only) that do not go through atomics --
d P rows, d Q rows, d U -- must repeat bit for bit.  Config-5 batch (many source atoms per workgroup, both launches) and a ragged
batch on a few workgroups.   usage: stress_tri_bwd.py [repetitions]"""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import torch
from bench_train import train_workload
from oracle.make_inputs import synthetic_train_batch
from phoregen_amd import hip, options
import phoregen_amd.training as tr
from phoregen_amd.config import default_model_config
from phoregen_amd.data import TrainBatch
from phoregen_amd.models.diffusion import PhoreDiff
from phoregen_amd.weights import init_deterministic_

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
model = init_deterministic_(PhoreDiff(default_model_config(), 'zinc_300'), 0).to('cuda')
orig = tr.SegCoreFn.backward
seen = []


def wrapped(ctx, *g):
    out = orig(ctx, *g)
    if ctx.cfg['mode'] == hip.SEG_TRIPLET and not seen:
        torch.cuda.synchronize()
        h = hashlib.sha256()
        for t in out[1:4]:                       # gYdst (d Q), gYsrc (d P), gU
            h.update(t.detach().cpu().numpy().tobytes())
        seen.append(h.hexdigest())
    return out


tr.SegCoreFn.backward = staticmethod(wrapped)
keys = ('ligand_x', 'ligand_pos', 'ligand_batch', 'ligand_ptr', 'f_edge_index', 'f_edge_attr', 'f_edge_batch',
        'phore_x', 'phore_pos', 'phore_norm', 'phore_batch')


def run(name, batch, draws, **kw):
    hashes = set()
    for _ in range(reps):
        seen.clear()
        with options.override(**kw):
            model.zero_grad()
            loss, _ = model.compute_loss(batch, draws=draws)
            loss.backward()
        hashes.add(seen[0])
    print(f'{name}: {reps} steps, {len(hashes)} distinct result(s) of the layer-5 triplet adjoint {kw}', flush=True)
    return len(hashes) == 1


ok = True
b = synthetic_train_batch(83, [2, 21, 3, 50, 64, 9, 33, 17, 40], [5, 8, 11, 14, 17, 20, 23, 26, 29])
gen = torch.Generator().manual_seed(7)
N, E = b['ligand_x'].numel(), b['f_edge_attr'].numel()
draws = dict(time_draw=torch.randint(10, 990, (9,), generator=gen), pos_noise=torch.randn(N, 3, generator=gen),
             u_node=torch.rand(N, 12, generator=gen), u_edge=torch.rand(E, 6, generator=gen))
rb = TrainBatch(*[b[k] for k in keys])
for kw in (dict(tri_bwd_form=2, tri_bwd_grid=256), dict(tri_bwd_form=2, tri_bwd_grid=5), dict(tri_bwd_form=1, tri_bwd_grid=3)):
    ok &= run('ragged batch (2 .. 64 atoms)', rb, draws, **kw)
batch, na = train_workload(256); batch.to('cuda')
gen = torch.Generator().manual_seed(11)
N, E = int(na.sum()), int((na * (na - 1)).sum())
draws5 = dict(time_draw=torch.randint(10, 990, (256,), generator=gen), pos_noise=torch.randn(N, 3, generator=gen),
              u_node=torch.rand(N, 12, generator=gen), u_edge=torch.rand(E, 6, generator=gen))
for kw in (dict(tri_bwd_form=2), dict(tri_bwd_form=1)):
    ok &= run('config-5 batch (256 pairs)', batch, draws5, **kw)
print('OK' if ok else 'MISMATCH')
sys.exit(0 if ok else 1)
