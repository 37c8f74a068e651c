"""Which LINES of the training path launch the small torch kernels?  torch.profiler over one step of tools/bench_train.py's
workload; every top-level aten op is attributed to the innermost phoregen_amd / torch.optim frame of its Python stack and
summed (launch count, device time) -- the list says where a fused kernel or a shared zero-filled arena pays."""
import os, sys, collections
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
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()

sites = collections.defaultdict(lambda: [0, 0.0, collections.Counter()])
for ev in prof.events():
    if ev.device_type != torch.autograd.DeviceType.CPU or not ev.name.startswith('aten::') or ev.cpu_parent is not None and \
            ev.cpu_parent.name.startswith('aten::'):
        continue
    dev = sum(k.duration for k in ev.kernels) if ev.kernels else 0.0
    n_k = len(ev.kernels)
    def walk(e):
        d, n = 0.0, 0
        for c in e.cpu_children:
            d += sum(k.duration for k in c.kernels); n += len(c.kernels)
            dd, nn = walk(c); d += dd; n += nn
        return d, n
    dd, nn = walk(ev)
    dev += dd; n_k += nn
    if n_k == 0:
        continue
    frame = next((s for s in (ev.stack or []) if 'phoregen_amd' in s or 'tools' in s or 'optim' in s),
                 'autograd engine (no Python frame): ' + ev.name if not ev.stack else 'other: ' + ev.stack[0])
    frame = frame.replace(ROOT + '/', '')
    s = sites[frame]
    s[0] += n_k; s[1] += dev; s[2][ev.name] += 1
print('sample stacks:', [e.stack[:3] for e in prof.events() if e.stack][:3])
tot_n = sum(s[0] for s in sites.values()); tot_d = sum(s[1] for s in sites.values())
print('torch-launched kernels in one step: %d, %.1f ms of device time' % (tot_n, tot_d / 1e3))
for frame, (n, d, ops) in sorted(sites.items(), key=lambda kv: -kv[1][1])[:60]:
    print('%6d kernels %8.2f ms  %-70s %s' % (n, d / 1e3, frame[:70], ', '.join('%s x%d' % (k.replace('aten::', ''), v) for k, v in ops.most_common(4))))
