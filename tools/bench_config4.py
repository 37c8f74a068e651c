#!/usr/bin/env python3
"""BASELINE config 4 as a JOB, one GPU's slice: `run_sampling_job` on P pharmacophores x S samples (default 32 x 32 = 1 024 graphs =
8 batches of 128, the full 1000 steps, return_traj=False) -- graphs/hour including everything between the batches (batch inputs,
BatchPlan, Engine + workspace, pharmacophore encoder, allocator), which the step benchmark does not see.  The reference serves such a
job with a serial loop over pharmacophores (sample_all.py:69-183).  GPU box: python tools/bench_config4.py [P S steps]"""
import gc, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import config4_job
from phoregen_amd.config import default_model_config
from phoregen_amd.models.diffusion import PhoreDiff
from phoregen_amd.parallel import run_sampling_job
from phoregen_amd.weights import init_deterministic_

P = int(sys.argv[1]) if len(sys.argv) > 1 else 32
S = int(sys.argv[2]) if len(sys.argv) > 2 else 32
steps = int(sys.argv[3]) if len(sys.argv) > 3 else None
model = init_deterministic_(PhoreDiff(default_model_config(), 'zinc_300'), 0).eval().to('cuda')
job = config4_job(n_phores=P, samples=S)
# a first small batch pays the one-time costs (library load, kernel attributes, side streams)
run_sampling_job(model, config4_job(n_phores=2, samples=4, seed=1), batch_size=8, num_steps=3)
torch.cuda.synchronize()

turn, mem = [], []
begin, finish = model.begin_sampling, model.finish_sampling


def timed_begin(*a, **k):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    st = begin(*a, **k)
    torch.cuda.synchronize()
    turn.append(time.perf_counter() - t0)
    return st


def watched_finish(st):
    out = finish(st)
    mem.append(torch.cuda.memory_allocated())
    return out


model.begin_sampling, model.finish_sampling = timed_begin, watched_finish
gc.disable()                                   # memory must come back by reference counting alone
torch.cuda.reset_peak_memory_stats()
t0 = time.perf_counter()
pred, na = run_sampling_job(model, job, batch_size=128, num_steps=steps)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
gc.enable()
n = job.n_graphs
print(json.dumps({
    'workload': f'BASELINE.json configs[3], one GPU\'s slice: {P} synthetic CpxPhore/DockPhore-shaped pharmacophores x {S} samples = {n} graphs, '
                f'batches of 128, {steps or 1000} steps, return_traj=False, device Philox noise',
    'graphs': n, 'batches': len(turn), 'wall_s': dt, 'graphs_per_hour': n / dt * 3600, 'graph_steps_per_sec': n * (steps or 1000) / dt,
    'turnover_ms_per_batch': [round(t * 1e3, 1) for t in turn], 'turnover_share': sum(turn) / dt,
    'turnover_note': 'begin_sampling per batch, synchronised: batch inputs -> BatchPlan (topology, triplet queue) -> Engine (workspace, launch '
                     'list) -> pharmacophore encoder -> initial state',
    'memory_allocated_after_each_batch_MB': [round(m / 2 ** 20, 1) for m in mem], 'peak_memory_MB': round(torch.cuda.max_memory_allocated() / 2 ** 20, 1),
    'gc': 'disabled during the job (memory returns by reference counting)', 'atoms': int(na.sum()), 'pred_finite': bool(torch.isfinite(pred[1]).all())}))
