import os, sys, time, torch
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _ROOT); sys.path.insert(0, os.path.join(_ROOT, 'tests'))
from oracle.make_inputs import synthetic_phore
from phoregen_amd.config import default_model_config
from phoregen_amd.models.diffusion import PhoreDiff
from phoregen_amd.weights import init_deterministic_
model = init_deterministic_(PhoreDiff(default_model_config(), 'zinc_300'), 0).eval().to('cuda')
gen = torch.Generator().manual_seed(8)
hp, pp, pn = synthetic_phore(gen, 44)
for B in (1, 10, 30):
    na = torch.randint(20, 40, (B,), generator=gen)
    bp = torch.repeat_interleave(torch.arange(B), 44)
    args = (hp.repeat(B, 1), pp.repeat(B, 1), pn.repeat(B, 1), bp, na, torch.zeros(B, 3))
    out = {}
    for mode in ('0', '1'):
        os.environ['PHOREGEN_DEBUG'], os.environ['PG_GRAPH'] = '1', mode
        model._engine = None
        r = model.sample_batch(*args, rng='device', seed=11, num_steps=20)      # warm
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = model.sample_batch(*args, rng='device', seed=11, num_steps=200)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        out[mode] = (dt / 200 * 1e3, r)
    same = all(torch.equal(a, b) for a, b in zip(out['0'][1]['traj'], out['1'][1]['traj']))
    print('B=%d  eager %.2f ms/step  graph %.2f ms/step  identical=%s' % (B, out['0'][0], out['1'][0], same))
