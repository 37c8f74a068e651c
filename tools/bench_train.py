#!/usr/bin/env python3
"""BASELINE.json config 5: one forward + backward of PhoreDiff.compute_loss on 256 synthetic ligand-pharmacophore
pairs (n ~ N(25,5^2) clamp [8,60] atoms, pharmacophores of the LigPhore shape, random bond labels), 1x MI355X.
Prints one JSON line (ms per training step, and the share of the adjoint kernels when --profile is given)."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import ligphore_workload


def train_workload(n_graphs=256, seed=4321, n_mean=25.0, n_std=5.0, n_max=60):
    from phoregen_amd.data import TrainBatch
    g = torch.Generator().manual_seed(seed)
    w = ligphore_workload(n_graphs, seed)
    na = (n_mean + n_std * torch.randn(n_graphs, generator=g)).round().clamp(8, n_max).long()
    off = torch.cat([torch.zeros(1, dtype=torch.long), na.cumsum(0)])
    N = int(na.sum())
    srcs, dsts, attrs, eb = [], [], [], []
    for gi, n in enumerate(na.tolist()):                      # datasets/transform.py:488-501 (dst-major complete graph)
        dst = torch.repeat_interleave(torch.arange(n), n)
        src = torch.arange(n).repeat(n)
        m = dst != src
        src, dst = src[m], dst[m]
        sym = torch.randint(1, 5, (n, n), generator=g) * (torch.rand(n, n, generator=g) < 2.2 / n).long()
        sym = torch.triu(sym, 1)
        sym = sym + sym.t()
        srcs.append(src + off[gi]); dsts.append(dst + off[gi]); attrs.append(sym[src, dst]); eb.append(torch.full((src.numel(),), gi))
    batch = TrainBatch(torch.randint(0, 11, (N,), generator=g), 1.5 * torch.randn(N, 3, generator=g),
                       torch.repeat_interleave(torch.arange(n_graphs), na), off,
                       torch.stack([torch.cat(srcs), torch.cat(dsts)]), torch.cat(attrs), torch.cat(eb),
                       w['h_phore'], w['pos_phore'], w['phore_norm'], w['batch_phore'])
    return batch, na


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--graphs', type=int, default=256)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--n-mean', type=float, default=25.0, help='mean ligand size (config 5: 25; the sampler headline shape: 40)')
    ap.add_argument('--n-max', type=int, default=60)
    ap.add_argument('--buckets', action='store_true', help='one GPU: also time the step with GradientBuckets attached and a 1-rank '
                    'nccl (RCCL) process group, i.e. the hook-launched bucket all-reduces of the data-parallel path (f-4)')
    a = ap.parse_args(argv)
    from phoregen_amd.config import default_model_config
    from phoregen_amd.models.diffusion import PhoreDiff
    from phoregen_amd.weights import init_deterministic_
    import torch.distributed as dist
    from phoregen_amd.parallel import GradientBuckets
    world, rank = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0')) % max(torch.cuda.device_count(), 1)    # (a 2-rank dry run can share one GPU)
    if world > 1:       # data parallel: one process per GPU, each its own 256-pair batch, one flat gradient bucket over RCCL
        torch.cuda.set_device(local)
        dist.init_process_group(os.environ.get('PG_DIST_BACKEND', 'nccl'), rank=rank, world_size=world)
    dev = f'cuda:{local}'
    model = init_deterministic_(PhoreDiff(default_model_config(), 'zinc_300'), 0).to(dev)
    batch, na = train_workload(a.graphs, seed=4321 + rank, n_mean=a.n_mean, n_std=a.n_mean / 5.0, n_max=a.n_max)
    batch.to(dev)
    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-5)
    # data parallel: ~4 MB gradient buckets, all-reduced asynchronously from gradient hooks while the backward continues
    buckets = GradientBuckets(model.parameters(), bucket_mb=4.0) if world > 1 else None
    torch.manual_seed(0)
    ms_plain = None

    def step():
        opt.zero_grad(set_to_none=True)
        loss, info = model.compute_loss(batch)
        loss.backward()
        if buckets is not None:
            buckets.finish()
        opt.step()
        return info
    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        info = step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / a.steps
    # roofline of the dominant kernel of the step, the triplet adjoint (pg_seg_attn_bwd PG_SEG_TRIPLET, 6 launches per step): HIP
    # events around its launches during two more steps; executed FLOPs = its MFMA count per launch from the committed PMC pass
    # (profiles/train_adjoint_mfma.json: SQ_INSTS_MFMA x 2 048, every MFMA of the kernel is a 16x16x4) scaled by the triplet rows
    from phoregen_amd import training as _tr
    _tr.bwd_timers, _tr.lib_timers = {}, []
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    tri_ms = [e0.elapsed_time(e1) for e0, e1 in _tr.bwd_timers.get(4, [])]
    knn_ms = [e0.elapsed_time(e1) for e0, e1 in _tr.bwd_timers.get(0, [])]
    lib_ms = [e0.elapsed_time(e1) for e0, e1 in _tr.lib_timers]
    _tr.bwd_timers = _tr.lib_timers = None
    roof = None
    try:
        rec = json.load(open(os.path.join(ROOT, 'profiles', 'train_adjoint_mfma.json')))
        e3_now = int((na * (na - 1) * ((na + 15) // 16 * 16)).sum())      # rows the adjoint walks: n (n-1) segments x 16-row tiles of n rows
        mfma = rec['mfma_per_launch'] * e3_now / rec['padded_rows']
        avg = sum(tri_ms) / max(len(tri_ms), 1)
        roof = {'kernel': 'triplet adjoint (pg_seg_attn_bwd PG_SEG_TRIPLET, ' + rec['kernel'] + '), 6 launches per step', 'bound': 'mfma',
                'achieved': mfma * 2048 / (avg * 1e-3) / 1e12, 'peak': 157.3, 'unit': 'TFLOP/s', 'frac': mfma * 2048 / (avg * 1e-3) / 1e12 / 157.3,
                'avg_launch_ms': avg, 'launches_timed': len(tri_ms), 'flops_per_launch': mfma * 2048, 'traffic': rec.get('hbm_bytes_per_launch'),
                'share_of_step': 6 * avg / ms, 'knn_node_adjoint_avg_launch_ms': sum(knn_ms) / max(len(knn_ms), 1),
                'source': {k: rec.get(k) for k in ('commit', 'method', 'workload')},
                'note': 'achieved = fp32 FLOPs the kernel EXECUTES (16x16x4 MFMAs counted by SQ_INSTS_MFMA in a PMC pass of this benchmark, '
                        'padding rows included) / mean launch duration (HIP events on the launch stream)'}
    except Exception as ex:
        roof = {'error': f'no PMC record of the adjoint ({ex})'}
    if a.buckets and world == 1:      # the same step again with the bucketed all-reduce path on a 1-rank RCCL group
        ms_plain = ms
        torch.cuda.set_device(local)
        dist.init_process_group('nccl', init_method='tcp://127.0.0.1:29541', rank=0, world_size=1)
        buckets = GradientBuckets(model.parameters(), bucket_mb=4.0)
        for _ in range(a.warmup):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            info = step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / a.steps
        hooked = sum(1 for _, h in buckets.launch_log[-len(buckets.buckets):] if h)
        dist.destroy_process_group()
    if world > 1:
        tmax = torch.tensor([ms], device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        ms = float(tmax)
        dist.destroy_process_group()
        if rank != 0:
            return
    e_bond = int((na * (na - 1)).sum()); e3 = int((na * (na - 1) * (na - 2)).sum())
    print(json.dumps({'metric': 'train step (compute_loss forward + backward + Adam), batch=%d' % a.graphs, 'value': ms, 'unit': 'ms/step',
                      'higher_is_better': False, 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'dtype': 'f32',
                      'data': 'synthetic', 'config': {'workload': 'BASELINE.json configs[4] shape: %d synthetic ligand-phore pairs, n~N(%g,%g) clamp [8,%d]' % (a.graphs, a.n_mean, a.n_mean / 5.0, a.n_max),
                                                      'graphs': a.graphs, 'n_lig': int(na.sum()), 'e_bond': e_bond, 'e3': e3},
                      **({'ms_without_buckets': ms_plain, 'buckets': len(buckets.buckets), 'buckets_launched_from_hooks': hooked,
                          'note': 'value = with GradientBuckets on a 1-rank nccl group'} if ms_plain is not None else {}),
                      'library_ms': sum(lib_ms) / 2, 'library_calls_per_step': len(lib_ms) // 2,
                      'library_note': 'the input-gradient GEMMs of the adjoint pass (gX = gY W, K = 256 ... 1 920) run on the library GEMM through torch.mm '
                                      '(rocBLAS / hipBLASLt; phoregen_amd/training.py _dgrad) -- the one place where the training path leaves the in-tree '
                                      'kernels; HIP events around every call, mean of two steps',
                      'roofline': roof, 'peak_mem_gb': torch.cuda.max_memory_allocated() / 2**30, 'last_loss': info['loss']}))


if __name__ == '__main__':
    main()
