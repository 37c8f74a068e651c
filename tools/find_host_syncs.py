"""Where does a training step wait for the GPU?  Times every Tensor.to / .cpu / .item / .tolist / bool() / int() / float() of one
step (tools/bench_train.py workload) and prints the call sites that took more than 0.5 ms (i.e. that blocked on the device)."""
import os, sys, time, traceback, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import torch
from bench_train import train_workload
from phoregen_amd.config import default_model_config
from phoregen_amd.models.diffusion import PhoreDiff
from phoregen_amd.weights import init_deterministic_

model = init_deterministic_(PhoreDiff(default_model_config(), 'zinc_300'), 0).to('cuda')
batch, na = train_workload(256)
batch.to('cuda')
opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-5)


def step():
    opt.zero_grad(set_to_none=True)
    loss, _ = model.compute_loss(batch)
    loss.backward()
    opt.step()


for _ in range(2):
    step()
torch.cuda.synchronize()
slow = collections.defaultdict(lambda: [0, 0.0])
for name in ('to', 'cpu', 'item', 'tolist', '__bool__', '__int__', '__float__', 'nonzero', '__index__'):
    orig = getattr(torch.Tensor, name)

    def make(orig, name):
        def wrapped(self, *a, **k):
            t0 = time.perf_counter()
            r = orig(self, *a, **k)
            dt = time.perf_counter() - t0
            if dt > 5e-4:
                fr = [f for f in traceback.extract_stack()[:-1] if 'phoregen_amd' in f.filename or 'tools' in f.filename][-1]
                key = f'{name} @ {os.path.relpath(fr.filename, ROOT)}:{fr.lineno} {fr.line}'
                slow[key][0] += 1
                slow[key][1] += dt
            return r
        return wrapped
    setattr(torch.Tensor, name, make(orig, name))
t0 = time.perf_counter()
step()
t1 = time.perf_counter()
torch.cuda.synchronize()
print('step enqueue time %.1f ms, + %.1f ms until the device is idle' % ((t1 - t0) * 1e3, (time.perf_counter() - t1) * 1e3))
# the same step by phase: host time to enqueue each phase, and how far behind the device is when the phase has been enqueued
ph = []
torch.cuda.synchronize()
t0 = time.perf_counter()
opt.zero_grad(set_to_none=True)
loss, _ = model.compute_loss(batch)
ph.append(('forward', time.perf_counter()))
loss.backward()
ph.append(('backward', time.perf_counter()))
opt.step()
ph.append(('adam', time.perf_counter()))
torch.cuda.synchronize()
t_end = time.perf_counter()
prev = t0
for name, t in ph:
    print('  %-9s enqueued in %6.1f ms (host)' % (name, (t - prev) * 1e3))
    prev = t
print('  device idle %.1f ms after the last enqueue' % ((t_end - prev) * 1e3))
for k, (n, dt) in sorted(slow.items(), key=lambda kv: -kv[1][1]):
    print('%8.1f ms  %3d x  %s' % (dt * 1e3, n, k))
