"""Race hunt: the four-lane launch list against the one-stream list on the SAME inputs, many times, at several batch sizes (each size runs
the schedule variant the engine picks for it: v2 / per-chain closing launch / merged knn launch / Q rows on the side lane / two triplet
launches).  Any mismatch of a single bit is a missing order point.   usage: stress_bits.py [repeats]"""
import sys, torch
import os
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _ROOT)
from bench import ligphore_workload
from phoregen_amd import options
from phoregen_amd.config import default_model_config
from phoregen_amd.models.diffusion import PhoreDiff
from phoregen_amd.weights import init_deterministic_

R = int(sys.argv[1]) if len(sys.argv) > 1 else 100
model = init_deterministic_(PhoreDiff(default_model_config(), 'zinc_300'), 0).eval().to('cuda')
bad = 0
for G in (3, 6, 10, 16, 24, 40, 56, 72, 100, 128):
    w = ligphore_workload(G, seed=100 + G)

    def state(**kw):
        with options.override(tune_grid=False, **kw):
            model._engine = None
            st = model.begin_sampling(w['h_phore'], w['pos_phore'], w['phore_norm'], w['batch_phore'], w['num_atoms'], torch.zeros(G, 3),
                                      rng='device', seed=1, return_traj=False, num_steps=4)
            model.reverse_step(st, 0, 999)          # fills the engine's input buffers with a real state
        return st
    ref_st = state(streams=False)
    ref = [t.clone() for t in ref_st.eng.forward_inplace()]
    torch.cuda.synchronize()
    st = state()
    e = st.eng
    for name in ('in_h_node', 'in_pos', 'in_h_edge', 'in_t'):
        getattr(e.ws, name).copy_(getattr(ref_st.eng.ws, name))
    mism = 0
    for r in range(R):
        out = e.forward_inplace()
        if not all(torch.equal(a, b) for a, b in zip(out, ref)):
            mism += 1
    torch.cuda.synchronize()
    bad += mism
    print(f'G={G:4d} ({int((w["num_atoms"] * (w["num_atoms"] - 1)).sum()):6d} bond edges): {R} forwards on four lanes, {mism} differ from the one-stream result'
          f'   [v2={bool(e.ahead_v2 and e.layer_ahead)}, triplet launches per layer={len(e.tri_calls) // 6}]', flush=True)
    model._engine = None
print('RACE-FREE' if bad == 0 else f'MISMATCHES: {bad}')
sys.exit(1 if bad else 0)
