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
hip.lib().pg_debug_force_generic_seg(int(os.environ.get('PG_SEG_DEBUG', '0')))
# the triplet sub-layer of layer 0: one launch, or two when the batch's 50+-atom ligands run on their own queue (BatchPlan.tri_split)
per_layer = len(eng.tri_calls) // len(eng.pack.layers)
calls = [eng.prog_fwd[i] for i in eng.tri_calls[:per_layer]]
s = hip.stream_ptr()
torch.cuda.synchronize()
for _ in range(3):
    for fn, args, _lane in calls:
        fn(*args, s)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    for fn, args, _lane in calls:
        fn(*args, s)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print(f'triplet kernel ({per_layer} launch(es) per layer): {ms:.3f} ms/launch, {counts["flops_triplet_kernel"] / ms / 1e9:.1f} TFLOP/s algorithmic, '
      f'E_bond={counts["e_bond"]} E3={counts["e3"]}')
