"""Cycle counters per section of pg_seg_attn_bwd (library built with -DPG_BWD_PROF; see tools/prof_bwd.sh)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from phoregen_amd import hip
from tools.bench_train import train_workload
from phoregen_amd.config import default_model_config
from phoregen_amd.models.diffusion import PhoreDiff
from phoregen_amd.weights import init_deterministic_
import phoregen_amd.training as tr
lib = hip.lib()
lib.pg_debug_bwd_prof.restype = C.c_int
lib.pg_debug_bwd_prof.argtypes = [C.c_void_p, C.c_int]
names = ['setup', 'pass1', 'softmax', 'tile head', 'recompute', 'sums+z tile', 'dM', 'dz', 'LN adj+db', 'dfeat+dWf', 'scatter', 'geometry', 'outputs']
model = init_deterministic_(PhoreDiff(default_model_config(), 'zinc_300'), 0).to('cuda')
batch, na = train_workload(256); batch.to('cuda')
buf = (C.c_ulonglong * 16)()
orig = tr.SegCoreFn.backward
per_mode = {}
def wrapped(ctx, *g):
    torch.cuda.synchronize(); lib.pg_debug_bwd_prof(buf, 1)
    out = orig(ctx, *g)
    torch.cuda.synchronize(); lib.pg_debug_bwd_prof(buf, 1)
    acc = per_mode.setdefault(ctx.cfg['mode'], [0] * 16)
    for i in range(16): acc[i] += buf[i]
    return out
tr.SegCoreFn.backward = staticmethod(wrapped)
loss, _ = model.compute_loss(batch); loss.backward(); torch.cuda.synchronize()
for mode, acc in sorted(per_mode.items()):
    tot = sum(acc)
    print('mode', mode, 'total wave-cycles %.3g' % tot)
    for n, v in zip(names, acc):
        print('   %-12s %5.1f %%' % (n, 100.0 * v / max(tot, 1)))
