"""Which torch ops make up the autograd glue of a training step?  torch.profiler over one step of tools/bench_train.py's workload,
grouped by operator (count, device time); the HIP library's own launches go through ctypes and show up only as kernels."""
import os, sys
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
print(prof.key_averages().table(sort_by='count', row_limit=45, max_name_column_width=60))
print(prof.key_averages().table(sort_by='self_cpu_time_total', row_limit=25, max_name_column_width=60))
print(prof.key_averages(group_by_stack_n=6).table(sort_by='cpu_time_total', row_limit=12, max_name_column_width=50)) if False else None

# the blocking copies: every aten::to / aten::copy_ event longer than 1 ms with its Python stack
for ev in prof.events():
    if ev.name in ('aten::copy_', 'aten::_to_copy') and ev.cpu_time_total > 1000:
        print('%.1f ms' % (ev.cpu_time_total / 1e3), ev.name, [s for s in (ev.stack or []) if 'phoregen_amd' in s or 'tools' in s][:4])
