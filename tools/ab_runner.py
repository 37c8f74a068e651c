#!/usr/bin/env python3
"""Round-5 A/B on one GPU: the launch list walked inside the library (pg_program_run) vs from Python, and the order points' event flags
(device-scope release = product, vs the round-4 fence-free form, measurement only).  Per variant and batch size: ms per step (median of 3
blocks) and the host time to ENQUEUE a step.  GPU box: python tools/ab_runner.py [graphs ...]"""
import json, os, sys, time
os.environ.setdefault('PHOREGEN_DEBUG', '1')      # (the fence-free order points are a measurement switch: the library refuses it otherwise)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import ligphore_workload, subset_workload
from phoregen_amd import hip, options
from phoregen_amd.config import default_model_config
from phoregen_amd.models.diffusion import PhoreDiff
from phoregen_amd.parallel import partition_graphs
from phoregen_amd.weights import init_deterministic_

sizes = [int(a) for a in sys.argv[1:]] or [16, 128]
model = init_deterministic_(PhoreDiff(default_model_config(), 'zinc_300'), 0).eval().to('cuda')
full = ligphore_workload(128, seed=1234)
W, K = 30, 30


def measure(work, gids, fence_free=False, **opt):
    hip.check(hip.lib().pg_debug_order_point_fence_free(int(fence_free)), 'pg_debug_order_point_fence_free')
    G = int(work['num_atoms'].numel())
    with options.override(**opt):
        model._engine = None
        st = model.begin_sampling(work['h_phore'], work['pos_phore'], work['phore_norm'], work['batch_phore'], work['num_atoms'],
                                  torch.zeros(G, 3), rng='device', seed=0, return_traj=True, num_steps=W + 4 * K, graph_ids=gids, pipeline=True)
    for i in range(W):
        model.reverse_step(st, i, 999 - i)
    ts, host = [], []
    for r in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(W + r * K, W + (r + 1) * K):
            model.reverse_step(st, i, 999 - i)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / K * 1e3)
        host.append((t1 - t0) / K * 1e3)
    # the host's own cost of a step: a short burst right after a synchronise (nothing queued, no back-pressure from the device)
    burst = []
    for r in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(W + (3 + 0) * K + r * 6, W + 3 * K + (r + 1) * 6):
            model.reverse_step(st, i, 999 - i)
        burst.append((time.perf_counter() - t0) / 6 * 1e3)
    host = [min(burst)]
    hip.lib().pg_debug_order_point_fence_free(0)
    out = [t.clone() for t in model.finish_sampling(st)['pred']]
    return sorted(ts)[1], min(host), out


res = {}
for G in sizes:
    if G == 128:
        work, ids = full, torch.arange(128)
    else:
        ids = partition_graphs(full['num_atoms'], 128 // G, full['n_phore'])[0]
        work = subset_workload(full, ids)
    ref = None
    for rep in range(2):                                  # alternating, twice: drift of the box shows as a difference between the two passes
        for name, kw in (('library walk + device-scope release (product)', {}),
                         ('python walk', dict(c_program=False)),
                         ('library walk + fence-free events (round 4 flags)', dict(fence_free=True)),
                         ('python walk + fence-free events (= round 4)', dict(c_program=False, fence_free=True))):
            ms, host, out = measure(work, ids, **kw)
            if ref is None:
                ref = out
            same = all(torch.equal(a, b) for a, b in zip(out, ref))
            res.setdefault(f'{G} graphs', {}).setdefault(name, []).append(dict(ms_per_step=round(ms, 4), host_enqueue_ms=round(host, 4), identical=same))
            print(G, name, f'{ms:.3f} ms/step, host enqueue {host:.3f} ms/step, identical to the first variant: {same}', flush=True)
print(json.dumps(res))
