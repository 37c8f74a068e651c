"""Print the pipelined step's launch list (prog_step) of a G-graph batch: entry, lane, order-point ops -- to see what a launch waits for."""
import sys, torch
import os
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, _ROOT)
from bench import ligphore_workload
from phoregen_amd.config import default_model_config
from phoregen_amd.models.diffusion import PhoreDiff
from phoregen_amd.weights import init_deterministic_
G = int(sys.argv[1]) if len(sys.argv) > 1 else 16
model = init_deterministic_(PhoreDiff(default_model_config(), 'zinc_300'), 0).eval().to('cuda')
w = ligphore_workload(G)
st = model.begin_sampling(w['h_phore'], w['pos_phore'], w['phore_norm'], w['batch_phore'], w['num_atoms'], torch.zeros(G, 3), rng='device', seed=0,
                          return_traj=False, num_steps=4, pipeline=True)
eng = model._engine
progs = eng.pipelined_programs()
for name, prog in zip(('prog_step', 'prog_ahead'), progs):
    print('==', name, len(prog), 'entries')
    for i, (fn, args, lane) in enumerate(prog[:int(sys.argv[2]) if len(sys.argv) > 2 else 90]):
        if hasattr(fn, 'kind'):
            if fn.kind == 'order':
                print(f'{i:4d}        {fn.name:8s} {fn.ops}')
        else:
            extra = ''
            if fn.__name__ == 'pg_gemm':
                g = args[0]._obj
                extra = f'M={g.M} N={g.N} K1={g.K1} K2={g.K2} add1={"y" if g.add1 else "n"} ln={"y" if g.ln_gamma else "n"}'
            elif fn.__name__ == 'pg_seg_attn':
                extra = f'mode={args[1]._obj.mode}'
            print(f'{i:4d} lane {lane} {fn.__name__:28s} {extra}')
