"""Clock counters per section of the channel-split triplet adjoint (library built with -DPG_TB2_PROF; see tools/prof_tb2.sh)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from phoregen_amd import hip, options
from tools.bench_train import train_workload
from phoregen_amd.config import default_model_config
from phoregen_amd.models.diffusion import PhoreDiff
from phoregen_amd.weights import init_deterministic_
lib = hip.lib()
lib.pg_debug_tb2_prof.restype = C.c_int
lib.pg_debug_tb2_prof.argtypes = [C.c_void_p, C.c_int]
names = ['atom prologue', 'segment operands (wait)', 'A1 hidden + stats', 'barrier 1', 'geometry', 'A2 stats + y', 'barrier 2',
         'features (next)', 'softmax + d logit', 'B1 dU dz', 'barrier 3', 'B2 dhid dWf dfeat', 'segment outputs', 'atom epilogue', 'barrier 2b']
model = init_deterministic_(PhoreDiff(default_model_config(), 'zinc_300'), 0).to('cuda')
batch, na = train_workload(256); batch.to('cuda')
buf = (C.c_ulonglong * 16)()
for form in [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else '1,2').split(',')]:
    with options.override(tri_bwd_form=form):
        loss, _ = model.compute_loss(batch); loss.backward(); torch.cuda.synchronize()
        lib.pg_debug_tb2_prof(buf, 1)
        model.zero_grad()
        loss, _ = model.compute_loss(batch); loss.backward(); torch.cuda.synchronize()
        lib.pg_debug_tb2_prof(buf, 1)
    tot = sum(buf)
    print('form', form, 'total wave-clocks %.3g' % tot)
    for n, v in zip(names, buf):
        print('   %-26s %5.1f %%' % (n, 100.0 * v / max(tot, 1)))
