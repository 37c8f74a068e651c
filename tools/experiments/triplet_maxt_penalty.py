"""What does the 4-tile instantiation of the staged triplet kernel cost the segments that need three tiles?  The headline batch (largest
ligand 56 atoms -> triplet2_kernel<768,4>) against the same batch with its 50+-atom ligands cut to 49 atoms (-> <768,3>): time per
12-segment round of the isolated kernel.  usage: triplet_maxt_penalty.py [graphs]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
from bench import ligphore_workload
from phoregen_amd import hip
from phoregen_amd.config import default_model_config
from phoregen_amd.models.diffusion import PhoreDiff
from phoregen_amd.weights import init_deterministic_

graphs = int(sys.argv[1]) if len(sys.argv) > 1 else 128
model = init_deterministic_(PhoreDiff(default_model_config(), 'zinc_300'), 0).eval().to('cuda')


def run(work, tag):
    model._engine = None
    st = model.begin_sampling(work['h_phore'], work['pos_phore'], work['phore_norm'], work['batch_phore'], work['num_atoms'],
                              torch.zeros(graphs, 3), rng='device', seed=0, return_traj=False, num_steps=2)
    model.reverse_step(st, 0, 999)
    eng = st.eng
    fn, args, _ = eng.prog_fwd[eng.tri_calls[0]]
    s = hip.stream_ptr()
    torch.cuda.synchronize()
    for _ in range(3):
        fn(*args, s)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        fn(*args, s)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 30
    n = work['num_atoms']
    segs = int((n * (n - 1)).sum())
    tiles = int((n * (n - 1) * ((n - 1 + 15) // 16)).sum())
    print(f'{tag}: largest ligand {int(n.max())}, {segs} segments, {tiles} segment-tiles: {ms:.3f} ms per launch = {ms * 1e6 / segs:.2f} ns per segment, '
          f'{ms * 1e6 / tiles:.2f} ns per segment-tile')


w = ligphore_workload(graphs)
run(w, 'headline batch            ')
w2 = dict(w); w2['num_atoms'] = w['num_atoms'].clamp(max=49)
run(w2, 'ligands cut to <= 49 atoms')
w3 = dict(w); keep = w['num_atoms'].clone(); keep[keep >= 50] = 49; keep[0] = 50      # ONE 4-tile ligand among 3-tile ones
w3['num_atoms'] = keep
run(w3, 'one 50-atom ligand        ')
