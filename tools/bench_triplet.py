"""Micro-benchmark of the triplet kernel alone on the headline workload (one layer's launch, repeated)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import ligphore_workload, algorithmic_counts
from phoregen_amd import hip
from phoregen_amd.config import default_model_config
from phoregen_amd.models.diffusion import PhoreDiff
from phoregen_amd.weights import init_deterministic_

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
graphs = int(sys.argv[2]) if len(sys.argv) > 2 else 128
model = init_deterministic_(PhoreDiff(default_model_config(), 'zinc_300'), 0).eval().to('cuda')
work = ligphore_workload(graphs)
counts = algorithmic_counts(work['num_atoms'], work['n_phore'])
st = model.begin_sampling(work['h_phore'], work['pos_phore'], work['phore_norm'], work['batch_phore'], work['num_atoms'],
                          torch.zeros(graphs, 3), rng='device', seed=0, return_traj=False, num_steps=2)
model.reverse_step(st, 0, 999)
eng = st.eng
fn, args, _lane = eng.prog_fwd[eng.tri_calls[0]]
s = hip.stream_ptr()
torch.cuda.synchronize()
for _ in range(3):
    fn(*args, s)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    fn(*args, s)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print(f'triplet kernel: {ms:.3f} ms/launch, {counts["flops_triplet_kernel"] / ms / 1e9:.1f} TFLOP/s algorithmic, '
      f'E_bond={counts["e_bond"]} E3={counts["e3"]}')
