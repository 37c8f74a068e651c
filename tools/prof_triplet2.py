"""Section cycle counters of the staged triplet kernel (-DPG_T2_PROF build, tools/prof_triplet2.sh)."""
import ctypes as C, os, sys
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
hip.lib().pg_debug_force_generic_seg(int(os.environ.get('PG_SEG_DEBUG', '0')))
fn, args, _ = eng.prog_fwd[eng.tri_calls[0]]
seg = C.cast(args[1], C.POINTER(hip.PgSegAttn)).contents
prof = torch.zeros(8, dtype=torch.int64, device='cuda')
seg.alpha = prof.data_ptr()
s = hip.stream_ptr()
for _ in range(2):
    fn(*args, s)
torch.cuda.synchronize()
prof.zero_()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); fn(*args, s); e1.record(); torch.cuda.synchronize()
names = ['queue+barriers', 'staging', 'Q', 'theta+fold+passA', 'softmax', 'passB', 'unfold+store', 'loop exit']
v = prof.cpu().tolist()
tot = sum(v)
print(f'{e0.elapsed_time(e1):.3f} ms; wave-cycles by section (s_memtime ticks, all waves):')
for n, c in zip(names, v):
    print(f'  {n:18s} {c / 1e6:10.1f} M  {100.0 * c / tot:5.1f} %')
