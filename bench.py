#!/usr/bin/env python3
"""Headline benchmark: denoise-steps/sec of the PhoreDiff sampler on a 128-graph / ~40-atom batch (BASELINE.json
config 3) on N MI355X GPUs.

  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run, one rank per GPU)

A "step" is one iteration of the reverse-diffusion loop (models/diffusion.py:432-517): denoiser forward + categorical
posteriors + Gaussian posterior + trajectory write, on synthetic graphs of the LigPhore shape and deterministic
random-init weights, inputs resident in HBM.  Graphs are independent, so each rank runs its own 128-graph batch
(weak scaling, no collective inside the loop); the only collective is the final gather of `pred` over RCCL, exercised
after the timed region.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


# ---------------------------------------------------------------------------- workload
def ligphore_workload(n_graphs=128, seed=1234, fixed_shape=False):
    """B graphs: n_g = clamp(round(N(40,6^2)),20,60) atoms, p_g = clamp(round(N(107,30^2)),23,203) pharmacophore
    nodes (94 % exclusion spheres), 18-wide features laid out as datasets/get_phore_data.py:55-69."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(seed)
    if fixed_shape:
        n_at, n_ph = torch.full((n_graphs,), 40), torch.full((n_graphs,), 107)
    else:
        n_at = (40 + 6 * torch.randn(n_graphs, generator=g)).round().clamp(20, 60).long()
        n_ph = (107 + 30 * torch.randn(n_graphs, generator=g)).round().clamp(23, 203).long()
    xs, ps, ns = [], [], []
    for p in n_ph.tolist():
        is_ex = torch.rand(p, generator=g) < 0.94
        types = torch.where(is_ex, torch.full((p,), 12), torch.randint(0, 12, (p,), generator=g))
        alpha = 0.5 + torch.rand(p, 1, generator=g)
        has_norm = ((torch.rand(p, generator=g) < 0.3) & ~is_ex).long()
        nrm = torch.randn(p, 3, generator=g)
        nrm = nrm / nrm.norm(dim=-1, keepdim=True) * has_norm[:, None].float()
        pos = 6.0 * torch.randn(p, 3, generator=g)
        xs.append(torch.cat([F.one_hot(types, 13).float(), alpha, F.one_hot(has_norm, 2).float(),
                             F.one_hot(is_ex.long(), 2).float()], -1))
        ps.append(pos - pos.mean(0, keepdim=True))
        ns.append(nrm)
    batch_phore = torch.repeat_interleave(torch.arange(n_graphs), n_ph)
    return dict(h_phore=torch.cat(xs), pos_phore=torch.cat(ps), phore_norm=torch.cat(ns), batch_phore=batch_phore,
                num_atoms=n_at, n_phore=n_ph)


def config4_job(n_phores=1024, samples=100, seed=4321):
    """BASELINE config 4: `n_phores` synthetic CpxPhore/DockPhore-shaped pharmacophores
    (p ~ N(80,25^2) clamp [24,152]) x `samples` graphs each, n ~ N(40,6^2) clamp [20,60] atoms, as a SamplingJob."""
    import torch.nn.functional as F
    from phoregen_amd.parallel import SamplingJob
    g = torch.Generator().manual_seed(seed)
    n_ph = (80 + 25 * torch.randn(n_phores, generator=g)).round().clamp(24, 152).long()
    phores = []
    for p in n_ph.tolist():
        is_ex = torch.rand(p, generator=g) < 0.9
        is_ex[0] = False                                   # at least one feature point (the guidance centre needs one)
        types = torch.where(is_ex, torch.full((p,), 12), torch.randint(0, 12, (p,), generator=g))
        alpha = 0.5 + torch.rand(p, 1, generator=g)
        has_norm = ((torch.rand(p, generator=g) < 0.3) & ~is_ex).long()
        nrm = torch.randn(p, 3, generator=g)
        nrm = nrm / nrm.norm(dim=-1, keepdim=True) * has_norm[:, None].float()
        pos = 6.0 * torch.randn(p, 3, generator=g)
        x = torch.cat([F.one_hot(types, 13).float(), alpha, F.one_hot(has_norm, 2).float(), F.one_hot(is_ex.long(), 2).float()], -1)
        phores.append((x, pos - pos.mean(0, keepdim=True), nrm, torch.randn(3, generator=g)))
    graph_phore = torch.repeat_interleave(torch.arange(n_phores), samples)
    num_atoms = (40 + 6 * torch.randn(n_phores * samples, generator=g)).round().clamp(20, 60).long()
    return SamplingJob(phores, graph_phore, num_atoms)


def subset_workload(w, graph_ids):
    """The graphs `graph_ids` (ascending) of a workload, renumbered 0..len-1."""
    keep = torch.isin(w['batch_phore'], graph_ids)
    remap = torch.full((int(w['num_atoms'].numel()),), -1, dtype=torch.long)
    remap[graph_ids] = torch.arange(graph_ids.numel())
    return dict(h_phore=w['h_phore'][keep], pos_phore=w['pos_phore'][keep], phore_norm=w['phore_norm'][keep],
                batch_phore=remap[w['batch_phore'][keep]], num_atoms=w['num_atoms'][graph_ids], n_phore=w['n_phore'][graph_ids])


def algorithmic_counts(n_at, n_ph, knn=32, H=128):
    """Per-step algorithmic work: sizes, GEMM FLOPs of the factored form, compulsory HBM bytes."""
    n_all = int((n_at + n_ph).sum())
    n_lig = int(n_at.sum())
    e_knn = int(torch.minimum(n_at + n_ph - 1, torch.tensor(knn)).mul(n_at + n_ph).sum())
    e_bond = int((n_at * (n_at - 1)).sum())
    e3 = int((n_at * (n_at - 1) * (n_at - 2)).sum())
    mlp = lambda r, i, o: r * (2 * i * H + 2 * H * o)
    knn_f = 4 * (4 * H * H * n_all + 2 * 93 * H * e_knn) + 3 * 2 * H * H * e_knn + 2 * H * 16 * e_knn + 2 * mlp(n_all, 128, 128)
    bond_f = 4 * (4 * H * H * n_all + 2 * H * H * e_bond) + 3 * 2 * H * H * e_bond + 2 * H * 16 * e_bond + 2 * mlp(n_all, 128, 128)
    tri_kernel = 2 * (2 * 13 * H * e3) + 2 * 2 * H * H * e3 + 2 * 2 * 20 * H * e_bond   # inside the triplet kernel
    tri_f = 2 * (2 * (H + 20 + 2 * H) * H * e_bond) + tri_kernel + mlp(e_bond, 256, 128)
    layer_f = knn_f + bond_f + tri_f + 2 * H * H * n_all
    layer_b = 2 * 4 * (128 + 3) * n_all + 2 * 512 * e_bond + 16 * e_knn + 16 * e_bond + 20 * e_knn + 3.31e6
    step_b = 6 * layer_b + 4 * (81 * n_lig + 36 * e_bond)
    # what the triplet kernel actually executes after folding the second key/value layers:
    # 112 MFMA 16x16x4 (2048 FLOP) per 16-row tile + per-segment fold/unfold (2 x 128x128 MACs) + Q (2 x 20x128 MACs)
    tiles = int((((n_at + 15) // 16) * n_at * (n_at - 1)).sum())
    tri_exec = tiles * 112 * 2048 + e_bond * (2 * 2 * H * H + 2 * 2 * 20 * H)
    return dict(n_all=n_all, n_lig=n_lig, e_knn=e_knn, e_bond=e_bond, e3=e3, flops_step=6 * layer_f,
                flops_triplet_kernel=tri_kernel, flops_triplet_executed=tri_exec, tri_tiles=tiles, bytes_step=step_b)


# ---------------------------------------------------------------------------- CPU baseline (oracle, bounded sample)
def cpu_baseline(work, n_sample_graphs=4, n_steps=2, max_threads=16):
    """Reference-dataflow CPU path (oracle/phoregen_oracle.py, pinned to the reference by tests/golden) timed on the
    host cores for a few graphs of the same workload, scaled linearly in graphs to the 128-graph batch."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from helpers import make_oracle
    from oracle import phoregen_oracle as po
    import torch.nn.functional as F
    # a 256-thread intra-op pool on these small tensors is pathological (measured: 300x slower than 8 threads),
    # so the baseline uses at most `max_threads` host cores and says so
    cores = min(os.cpu_count() or 1, max_threads)
    torch.set_num_threads(cores)
    o = make_oracle(0)
    sel = torch.arange(n_sample_graphs)
    na = work['num_atoms'][sel]
    keep = work['batch_phore'] < n_sample_graphs
    hp, pp, pn, bp = (work[k][keep] for k in ('h_phore', 'pos_phore', 'phore_norm', 'batch_phore'))
    g = torch.Generator().manual_seed(7)
    bn = torch.repeat_interleave(torch.arange(n_sample_graphs), na)
    ei, be = po.make_edge_data(na)
    hn = F.one_hot(torch.full((int(na.sum()),), 11), 12).float()
    he = F.one_hot(torch.zeros(ei.size(1), dtype=torch.long), 6).float()
    pos = torch.randn(int(na.sum()), 3, generator=g)
    t0 = None
    with torch.no_grad():
        for s in range(n_steps + 1):
            if s == 1:
                t0 = time.perf_counter()
            tt = torch.full((n_sample_graphs,), 999 - s)
            o.forward(hn, pos, bn, he, ei, be, tt, hp, pp, pn, bp)
    dt = (time.perf_counter() - t0) / n_steps
    graph_steps = n_sample_graphs / dt
    return dict(value=graph_steps / 128.0, unit='denoise-steps/s (128-graph batch)', cores=cores, kind='port',
                sample=f'{n_steps} timed forwards (+1 warm-up) of oracle/phoregen_oracle.py on the first {n_sample_graphs} graphs '
                       f'({int(na.sum())} atoms) of the workload, {dt:.2f} s/forward, scaled linearly to 128 graphs')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--graphs', type=int, default=128)
    ap.add_argument('--fixed-shape', action='store_true', help='n=40, p=107 for every graph (closed-form counts)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--strong', action='store_true',
                    help='strong scaling (SURVEY.md 8d): ONE batch of --graphs graphs partitioned over the ranks by n^3 cost')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    local = local % max(torch.cuda.device_count(), 1)     # (a 2-rank gloo dry run can share one GPU)
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(os.environ.get('PG_DIST_BACKEND', 'nccl'))    # nccl == RCCL on ROCm

    from phoregen_amd.config import default_model_config
    from phoregen_amd.models.diffusion import PhoreDiff
    from phoregen_amd.weights import init_deterministic_

    model = init_deterministic_(PhoreDiff(default_model_config(), 'zinc_300'), 0).eval().to(dev)
    total_graphs = args.graphs if args.strong else world * args.graphs
    if args.strong and world > 1:
        from phoregen_amd.parallel import partition_graphs
        full = ligphore_workload(args.graphs, seed=1234, fixed_shape=args.fixed_shape)
        mine = partition_graphs(full['num_atoms'], world)[rank]
        work = subset_workload(full, mine)
        args.graphs = int(mine.numel())
    else:
        work = ligphore_workload(args.graphs, seed=1234 + rank, fixed_shape=args.fixed_shape)   # each rank: its own graphs
    counts = algorithmic_counts(work['num_atoms'], work['n_phore'])
    K, W = args.steps, args.warmup
    st = model.begin_sampling(work['h_phore'], work['pos_phore'], work['phore_norm'], work['batch_phore'],
                              work['num_atoms'], torch.zeros(args.graphs, 3), rng='device', seed=rank, return_traj=True,
                              num_steps=K + W)
    T = model.num_timesteps

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(W):
        model.reverse_step(st, i, T - 1 - i)
    barrier()
    st.eng.timers = {}
    t0 = time.perf_counter()
    for i in range(W, W + K):
        model.reverse_step(st, i, T - 1 - i)
    barrier()
    dt = time.perf_counter() - t0
    tri_ms = st.eng.kernel_ms('triplet')
    st.eng.timers = None
    res = model.finish_sampling(st)

    # the one collective of the path: re-assemble `pred` of all ranks (phoregen_amd/parallel.py), after the timed loop
    gather_ms = None
    if world > 1:
        from phoregen_amd.parallel import gather_predictions
        tg = time.perf_counter()
        gids = mine if (args.strong and world > 1) else torch.arange(args.graphs) + rank * args.graphs
        gather_predictions(res['pred'], work['num_atoms'], gids)
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - tg) * 1e3
    tmax = torch.tensor([dt], device=dev if (world == 1 or dist.get_backend() != 'gloo') else 'cpu')
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    if rank == 0:
        peak_tf = 157.3                      # fp32 MFMA dense peak, MI355X_MICROARCH.md "Chip-level parameters"
        tri_avg_ms = sum(tri_ms) / max(len(tri_ms), 1)
        achieved = counts['flops_triplet_kernel'] / (tri_avg_ms * 1e-3) / 1e12 if tri_ms else None
        traffic = None
        tpath = os.path.join(ROOT, 'profiles', 'triplet_traffic.json')
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get('hbm_bytes_per_launch')
            except Exception:
                traffic = None
        line = {
            'metric': 'denoise-steps/sec (batch=128, ~40-atom graphs)', 'value': (1 if args.strong else world) * K / dt, 'unit': 'steps/s',
            'n_gpus': world, 'steps': K, 'warmup': W, 'ms_per_step': dt / K * 1e3, 'higher_is_better': True,
            'scaling': 'strong' if args.strong else 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'BASELINE.json configs[2]: 128 LigPhore-shaped graphs per GPU '
                                   '(n~N(40,6) atoms, p~N(107,30) pharmacophore nodes), steps t=999.. of the 1000-step sampler, '
                                   'device Philox noise, trajectory written',
                       'graphs_per_gpu': args.graphs, 'fixed_shape': args.fixed_shape, 'n_ctx': counts['n_all'],
                       'n_lig': counts['n_lig'], 'e_knn': counts['e_knn'], 'e_bond': counts['e_bond'], 'e3': counts['e3'],
                       'parallelism': f'graph-sharded x{world}, final RCCL gather only'},
            'graph_steps_per_sec': K * total_graphs / dt,
            'roofline': {'kernel': 'triplet_kernel (pg_seg_attn PG_SEG_TRIPLET = BondUpdateLayer, 6 launches/step)', 'bound': 'mfma',
                         'achieved': achieved, 'peak': peak_tf, 'unit': 'TFLOP/s',
                         'frac': (achieved / peak_tf) if achieved else None, 'traffic': traffic,
                         'avg_launch_ms': tri_avg_ms, 'launches_timed': len(tri_ms),
                         'algorithmic_flops_per_launch': counts['flops_triplet_kernel'],
                         'note': 'achieved/frac use SURVEY 8d algorithmic FLOPs (unfolded second layers); the kernel executes '
                                 'executed_flops_per_launch after folding them, see executed_frac',
                         'executed_flops_per_launch': counts['flops_triplet_executed'],
                         'executed_frac': (counts['flops_triplet_executed'] / (tri_avg_ms * 1e-3) / 1e12 / peak_tf) if tri_ms else None,
                         'share_of_step': (6 * tri_avg_ms) / (dt / K * 1e3) if tri_ms else None},
            'step_roofline': {'flops_alg': counts['flops_step'], 'bytes_alg': counts['bytes_step'],
                              'mfma_frac': counts['flops_step'] / (dt / K) / (peak_tf * 1e12),
                              'hbm_frac': counts['bytes_step'] / (dt / K) / 8e12},
            'final_gather_ms': gather_ms,
        }
        if not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(ligphore_workload(args.graphs, seed=1234, fixed_shape=args.fixed_shape))
        else:
            line['cpu_baseline'] = None
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
