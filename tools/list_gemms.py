"""Every pg_gemm launch of one denoiser forward on the headline workload: shape, options, time of the isolated launch (GPU box).
usage: tools/list_gemms.py [graphs]"""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
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
lib = hip.lib()
s = hip.stream_ptr()
acc = collections.OrderedDict()
for fn, args, lane in eng.prog_fwd:
    if lane < 0 or fn is not lib.pg_gemm:
        continue
    g = args[0]._obj
    key = (g.M, g.N, g.K1, g.K2, bool(g.ln_gamma), bool(g.rows), bool(g.add1), bool(g.idx1), bool(g.add2), g.act, g.out_scale != 1.0,
           bool(g.bias))
    for _ in range(2):
        fn(*args, s)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn(*args, s)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100.0
    c, t = acc.get(key, (0, 0.0))
    acc[key] = (c + 1, t + us)
print('   M      N   K1  K2  ln rows add1 idx1 add2 act scale bias | calls   us/call   us/step')
tot = 0.0
for k, (c, t) in acc.items():
    tot += t
    print('%7d %5d %4d %3d %3d %4d %4d %4d %4d %3d %5d %4d | %5d %9.1f %9.1f' % (*[int(v) for v in k], c, t / c, t))
print('all pg_gemm launches: %.2f ms per step' % (tot / 1e3))
