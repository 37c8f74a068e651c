"""The actual rank shares of the headline batch (partition_graphs, by size) timed with several persistent-grid sizes of the triplet kernel:
where does the best grid flip?  usage: share_tri_grid.py <world> <grid> <grid> ..."""
import sys, time, torch
import os
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, _ROOT)
from bench import ligphore_workload, subset_workload
from phoregen_amd import options
from phoregen_amd.config import default_model_config
from phoregen_amd.models.diffusion import PhoreDiff
from phoregen_amd.parallel import partition_graphs
from phoregen_amd.weights import init_deterministic_

world = int(sys.argv[1]); grids = [int(g) for g in sys.argv[2:]]
model = init_deterministic_(PhoreDiff(default_model_config(), 'zinc_300'), 0).eval().to('cuda')
full = ligphore_workload(128, seed=1234)
W, K = 6, 30


def ms_per_step(work, gids, grid):
    G = int(work['num_atoms'].numel())
    with options.override(tri_grid=grid):
        model._engine = None
        st = model.begin_sampling(work['h_phore'], work['pos_phore'], work['phore_norm'], work['batch_phore'], work['num_atoms'],
                                  torch.zeros(G, 3), rng='device', seed=0, return_traj=True, num_steps=W + 3 * K, graph_ids=gids)
        for i in range(W):
            model.reverse_step(st, i, 999 - i)
        ts = []
        for r in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for i in range(W + r * K, W + (r + 1) * K):
                model.reverse_step(st, i, 999 - i)
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / K * 1e3)
    model._engine = None
    return sorted(ts)[1]


parts = partition_graphs(full['num_atoms'], world, full['n_phore'])
for r, ids in enumerate(parts):
    w = subset_workload(full, ids)
    n = w['num_atoms']
    e, nctx = int((n * (n - 1)).sum()), int(n.sum() + w['n_phore'].sum())
    print(f'rank {r}: {ids.numel():3d} graphs, {e:6d} bond edges, {nctx:5d} context nodes, largest ligand {int(n.max())}: ' +
          '  '.join(f'grid {g}: {ms_per_step(w, ids, g):.3f}' for g in grids), flush=True)
