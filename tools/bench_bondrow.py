"""Isolated timing of the fused bond-row launch (csrc/bondrow.hip) against the separate pg_gemm launches it replaces."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from bench import ligphore_workload
from phoregen_amd import hip
from phoregen_amd.config import default_model_config
from phoregen_amd.models.diffusion import PhoreDiff
from phoregen_amd.weights import init_deterministic_

graphs = int(sys.argv[1]) if len(sys.argv) > 1 else 128
model = init_deterministic_(PhoreDiff(default_model_config(), 'zinc_300'), 0).eval().to('cuda')
work = ligphore_workload(graphs)
st = model.begin_sampling(work['h_phore'], work['pos_phore'], work['phore_norm'], work['batch_phore'], work['num_atoms'],
                          torch.zeros(graphs, 3), rng='device', seed=0, return_traj=False, num_steps=2)
model.reverse_step(st, 0, 999)
eng = st.eng
s = hip.stream_ptr()
names = {}
for fn, args, lane in eng.prog_fwd:
    if lane < 0:
        continue
    n = fn.__name__
    if n in ('pg_bond_rows', 'pg_gemm'):
        names.setdefault(n, []).append((fn, args))
for n, calls in names.items():
    for idx, (fn, args) in enumerate(calls[:6] if n == 'pg_bond_rows' else []):
        for _ in range(2):
            fn(*args, s)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn(*args, s)
        e1.record(); torch.cuda.synchronize()
        print(f'{n}[{idx}] jobs={args[0]._obj.n_jobs}: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us')
