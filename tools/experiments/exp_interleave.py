"""Experiment: a batch of G graphs run as S independent sub-batches whose reverse steps are enqueued alternately on S
stream sets (graphs are independent, the noise is keyed by the global graph id).  Prints ms per step of the whole batch and the
host enqueue time, for S = 1, 2, 4.   usage: exp_interleave.py [G ...]"""
import sys, time, torch
import os
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, _ROOT)
from bench import ligphore_workload, subset_workload
from phoregen_amd.config import default_model_config
from phoregen_amd.engine import Engine
from phoregen_amd.models.diffusion import PhoreDiff
from phoregen_amd.parallel import partition_graphs
from phoregen_amd.weights import init_deterministic_

model = init_deterministic_(PhoreDiff(default_model_config(), 'zinc_300'), 0).eval().to('cuda')
engines = {}


def engine_for(plan):
    pack = model.packed()
    e = engines.get(id(plan))
    if e is None:
        e = engines[id(plan)] = Engine(pack, plan, knn_k=model.denoiser.k)
        e._plan_keep = plan
        e.stream_set = (len(engines) - 1) % 4
    return e


model.engine_for = engine_for
Gs = [int(a) for a in sys.argv[1:]] or [16, 32, 64, 128]
W, K = 10, 40
lane0 = [torch.cuda.Stream() for _ in range(4)]      # (created once: streams share a few hardware queues)
for G in Gs:
    full = ligphore_workload(G)
    for S in (1, 2, 4):
        if G // S < 4:
            continue
        parts = partition_graphs(full['num_atoms'], S)
        streams = lane0[:S]
        sts = []
        for k in range(S):
            w = subset_workload(full, parts[k]) if S > 1 else full
            with torch.cuda.stream(streams[k]):
                sts.append(model.begin_sampling(w['h_phore'], w['pos_phore'], w['phore_norm'], w['batch_phore'], w['num_atoms'],
                                                torch.zeros(int(w['num_atoms'].numel()), 3), rng='device', seed=0, return_traj=True,
                                                num_steps=W + 3 * K, graph_ids=parts[k]))
        torch.cuda.synchronize()

        def steps(i0, n):
            for i in range(i0, i0 + n):
                for k in range(S):
                    with torch.cuda.stream(streams[k]):
                        model.reverse_step(sts[k], i, 999 - i)
        steps(0, W)
        res = []
        for r in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            steps(W + r * K, K)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            res.append(((t2 - t0) / K * 1e3, (t1 - t0) / K * 1e3))
        res.sort()
        print(f'G={G:4d} S={S}: {res[1][0]:7.3f} ms/step (host enqueue {res[1][1]:6.3f} ms/step)  blocks', ['%.3f' % r[0] for r in res], flush=True)
        del sts
        engines.clear()
        torch.cuda.empty_cache()
