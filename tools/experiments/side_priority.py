"""Side lanes with HIP stream priorities (one process per variant: streams map onto hardware queues when they are created).
usage: side_priority.py <graphs> <prio lane 1> <prio lane 2> <prio lane 3>     (0 normal, 1 low, -1 high)"""
import sys, time, torch
import os
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, _ROOT)
from bench import ligphore_workload
from phoregen_amd import engine
from phoregen_amd.config import default_model_config
from phoregen_amd.models.diffusion import PhoreDiff
from phoregen_amd.weights import init_deterministic_

G = int(sys.argv[1]); pr = [int(v) for v in sys.argv[2:5]]
torch.cuda.init()
engine._SIDE_STREAMS[(torch.cuda.current_device(), 0)] = [torch.cuda.Stream(priority=p) for p in pr]
model = init_deterministic_(PhoreDiff(default_model_config(), 'zinc_300'), 0).eval().to('cuda')
w = ligphore_workload(G)
W, K, R = 8, 30, 3
st = model.begin_sampling(w['h_phore'], w['pos_phore'], w['phore_norm'], w['batch_phore'], w['num_atoms'], torch.zeros(G, 3),
                          rng='device', seed=0, return_traj=True, num_steps=W + R * K)
for i in range(W):
    model.reverse_step(st, i, 999 - i)
ts = []
for r in range(R):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(W + r * K, W + (r + 1) * K):
        model.reverse_step(st, i, 999 - i)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / K * 1e3)
print(f'G={G:4d} side lane priorities {pr}  {sorted(ts)[1]:7.3f} ms/step   blocks {["%.3f" % t for t in ts]}', flush=True)
