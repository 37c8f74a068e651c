"""Where a small-batch step spends its time, without a profiler attached: the sampler step with the denoiser cut to its first k
layers (k = 0..6).  T(k) - T(k-1) = what layer k costs inside the step, T(0) = embedding + knn + gate + heads + posterior."""
import sys, time, torch
import os
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _ROOT)
from bench import ligphore_workload
from phoregen_amd.config import default_model_config
from phoregen_amd.models.diffusion import PhoreDiff
from phoregen_amd.weights import init_deterministic_

Gs = [int(g) for g in (sys.argv[1] if len(sys.argv) > 1 else '16').split(',')]
model = init_deterministic_(PhoreDiff(default_model_config(), 'zinc_300'), 0).eval().to('cuda')
pack = model.packed()
all_layers = list(pack.layers)
W, K = 8, 30
for G in Gs:
    w = ligphore_workload(G)
    prev = None
    for k in range(0, 7):
        pack.layers = all_layers[:k]
        model._engine = None
        st = model.begin_sampling(w['h_phore'], w['pos_phore'], w['phore_norm'], w['batch_phore'], w['num_atoms'], torch.zeros(G, 3),
                                  rng='device', seed=0, return_traj=True, num_steps=W + 3 * K)
        for i in range(W):
            model.reverse_step(st, i, 999 - i)
        ts = []
        for r in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(W + r * K, W + (r + 1) * K):
                model.reverse_step(st, i, 999 - i)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / K * 1e3)
        t = sorted(ts)[1]
        print(f'G={G:4d} layers={k}: {t:7.3f} ms/step' + ('' if prev is None else f'   (+{(t - prev) * 1e3:6.0f} us)'), flush=True)
        prev = t
        del st
pack.layers = all_layers
