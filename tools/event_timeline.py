"""Timeline of ONE sampler step from HIP events around every launch of the engine's list (Engine.trace): start (us after the
step's first launch), duration, lane, entry point.  No profiler attached: a kernel-trace profiler adds device-side latency to
every dispatch and stretches exactly the small-batch chains this is for.   usage: event_timeline.py [graphs] [step-count]"""
import sys, torch
import os
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _ROOT)
from bench import ligphore_workload
from phoregen_amd.config import default_model_config
from phoregen_amd.models.diffusion import PhoreDiff
from phoregen_amd.weights import init_deterministic_

G = int(sys.argv[1]) if len(sys.argv) > 1 else 16
model = init_deterministic_(PhoreDiff(default_model_config(), 'zinc_300'), 0).eval().to('cuda')
w = ligphore_workload(G)
st = model.begin_sampling(w['h_phore'], w['pos_phore'], w['phore_norm'], w['batch_phore'], w['num_atoms'], torch.zeros(G, 3),
                          rng='device', seed=0, return_traj=True, num_steps=40)
for i in range(20):
    model.reverse_step(st, i, 999 - i)
torch.cuda.synchronize()
t_end = []
for i in range(20, 24):
    st.eng.trace = [] if i == 22 else None
    if i == 22:
        tr = st.eng.trace
    model.reverse_step(st, i, 999 - i)
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    t_end.append(e)
torch.cuda.synchronize()
t0 = tr[0][2]
rows = sorted(((t0.elapsed_time(a) * 1e3, a.elapsed_time(b) * 1e3, lane, what) for what, lane, a, b in tr))
print(f'# one sampler step of {G} graphs, HIP events around every launch: {len(rows)} launches; '
      f'denoiser span {max(r[0] + r[1] for r in rows):.0f} us, step (to the end of the posterior kernels) {t0.elapsed_time(t_end[2]) * 1e3:.0f} us, '
      f'next untraced step {t_end[2].elapsed_time(t_end[3]) * 1e3:.0f} us')
print('# start_us  dur_us(incl. event latency)  lane  entry')
for s, d, lane, what in rows:
    print('%9.1f %8.1f  L%d  %s' % (s, d, lane, what))
