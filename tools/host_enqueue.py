import sys, time, torch
import os
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _ROOT)
from bench import ligphore_workload
from phoregen_amd.config import default_model_config
from phoregen_amd.models.diffusion import PhoreDiff
from phoregen_amd.weights import init_deterministic_
model = init_deterministic_(PhoreDiff(default_model_config(), 'zinc_300'), 0).eval().to('cuda')
for G in (16, 128):
    w = ligphore_workload(G)
    st = model.begin_sampling(w['h_phore'], w['pos_phore'], w['phore_norm'], w['batch_phore'], w['num_atoms'], torch.zeros(G, 3), rng='device', seed=0, return_traj=True, num_steps=70)
    for i in range(10):
        model.reverse_step(st, i, 999 - i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(10, 60):
        model.reverse_step(st, i, 999 - i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(G, 'graphs: host enqueue %.2f ms/step, total %.2f ms/step' % ((t1 - t0) / 50 * 1e3, (t2 - t0) / 50 * 1e3))
