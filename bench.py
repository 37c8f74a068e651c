#!/usr/bin/env python3
"""Headline benchmark: denoise-steps/sec of the PhoreDiff sampler on a 128-graph / ~40-atom batch (BASELINE.json
config 3) on N MI355X GPUs.

  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run, one rank per GPU)

A "step" is one iteration of the reverse-diffusion loop (models/diffusion.py:432-517): denoiser forward + categorical
posteriors + Gaussian posterior + trajectory write, on synthetic graphs of the LigPhore shape and deterministic
random-init weights, inputs resident in HBM.  Graphs are independent: by default ONE 128-graph batch is partitioned over the
N ranks by the fitted step cost (strong scaling, the figure SURVEY.md 8(d) and the ">= 6x at 8 GPUs" target refer to; `--weak` gives every rank
its own 128-graph batch instead), no collective inside the loop; the only collective is the final gather of `pred` over RCCL,
exercised after the timed region.  Prints ONE JSON line on rank 0.
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


def triplet_tiles(n_at):
    """16-row tiles the staged triplet kernel (csrc/triplet2.hip) walks per segment of a ligand with n atoms: its rows are
    the n-2 atoms that are neither the segment's source nor its target (round 6; before: n-1 rows, the target's masked)."""
    return (n_at - 2 + 15) // 16


def knn_node_mfma(eng):
    """16x16x4 MFMAs one knn-node sub-layer (csrc/node_attn.hip: ligand + pharmacophore targets, as one merged launch or two) EXECUTES on the
    engine's current neighbour lists: per 16-row tile and pass 8 x (2 + 5 [tile has ligand sources] + 5 [tile has pharmacophore
    sources]) first-layer steps + 32 for the logits / the value aggregate; two passes.  (The neighbour slots are partitioned by
    source kind, ligand atoms first: pg_knn_group_by_kind.)"""
    w, p = eng.ws, eng.plan
    deg = w.deg.long()
    k = w.nbr.size(1)
    valid = torch.arange(k, device=deg.device)[None, :] < deg[:, None]
    is_lig = torch.zeros(p.n_ctx, dtype=torch.bool, device=deg.device)
    is_lig[p.lig2ctx_long] = True
    n_l = (is_lig[w.nbr.long().clamp(min=0)] & valid).sum(1)                # ligand sources of every node (they come first)
    total = 0
    for t0 in range(0, k, 16):
        in_tile = (deg > t0)
        has_l = in_tile & (n_l > t0)
        has_p = in_tile & (deg > n_l) & (torch.minimum(deg, torch.tensor(t0 + 16, device=deg.device)) > n_l)
        total += int((in_tile.long() * (8 * 2 + 32) + has_l.long() * 40 + has_p.long() * 40).sum())
    return 2 * total


def executed_flops(eng):
    """fp32 FLOPs the launches of ONE denoiser forward EXECUTE on the matrix / vector pipes, by kernel class, from the engine's own
    launch list and neighbour lists (the numerator of `step_roofline.exec_frac`):
      gemm      2 M N K of every pg_gemm / pg_rows_linear launch (their fused epilogues are not counted);
      triplet   112 MFMA 16x16x4 per 16-row tile + query fold / value unfold per segment, padding rows included;
      knn_node  knn_node_mfma() x 2048 + fold / unfold per node;  knn_pos: both MLP paths in the key layout + the 16-column
                value product per tile, fold per ligand atom;  bond_node / bond_pos: 64 MFMA per 16-row tile of the ligand
                (no feature columns) + fold (/ unfold);
      gate      80 MFMA per context node (2 tiles x 5 steps x 8).
    Elementwise kernels (embeddings, knn search, geometry, posteriors) execute no matrix work and are left out."""
    import ctypes as C
    from phoregen_amd import hip
    p, w, lib = eng.plan, eng.ws, eng.lib
    out = dict(gemm=0, triplet=0, knn_node=0, knn_pos=0, bond_node=0, bond_pos=0, gate=80 * 2048 * p.n_ctx)
    n_at = p.num_atoms
    tri_tiles = int((triplet_tiles(n_at) * n_at * (n_at - 1)).sum())
    node_tiles = int((((n_at + 15) // 16) * n_at).sum())             # bond modes: a ligand atom walks ceil(n / 16) tiles
    fold = 2 * 128 * 128
    # knn-pos tiles: ligand targets only, first-layer steps per tile as in knn_node_mfma
    deg = w.deg.long()
    k = w.nbr.size(1)
    valid = torch.arange(k, device=deg.device)[None, :] < deg[:, None]
    is_lig = torch.zeros(p.n_ctx, dtype=torch.bool, device=deg.device)
    is_lig[p.lig2ctx_long] = True
    n_l = (is_lig[w.nbr.long().clamp(min=0)] & valid).sum(1)
    pos_mfma = 0
    for t0 in range(0, k, 16):
        in_tile = (deg > t0) & is_lig
        has_l = in_tile & (n_l > t0)
        has_p = in_tile & (deg > n_l) & (torch.minimum(deg, torch.tensor(t0 + 16, device=deg.device)) > n_l)
        pos_mfma += int((in_tile.long() * 2 * (16 + 32) + has_l.long() * 80 + has_p.long() * 80).sum())
    knn_mf = knn_node_mfma(eng)
    for fn, args, lane in eng.prog_fwd:
        if lane < 0:
            continue
        if fn is lib.pg_gemm:
            g = args[0]._obj
            out['gemm'] += 2 * g.M * g.N * (g.K1 + g.K2)
        elif fn is lib.pg_rows_linear:
            out['gemm'] += 2 * args[6] * args[5] * args[2]
        elif fn is lib.pg_seg_attn:
            sa = args[1]._obj
            if sa.mode == hip.SEG_TRIPLET:               # (one or two launches per layer -- by row tiles of the ligand --: counted once per layer below)
                pass
            elif sa.mode == hip.SEG_KNN_NODE:            # (one or two launches per sub-layer: counted once per sub-layer below)
                out['knn_node'] += (sa.n_seg + sa.n_seg2) * 2 * fold
            elif sa.mode == hip.SEG_KNN_POS:
                out['knn_pos'] += pos_mfma * 2048 + sa.n_seg * fold
            elif sa.mode == hip.SEG_BOND_NODE:
                out['bond_node'] += node_tiles * 64 * 2048 + sa.n_seg * 2 * fold
            elif sa.mode == hip.SEG_BOND_POS:
                out['bond_pos'] += node_tiles * 64 * 2048 + sa.n_seg * fold
    out['knn_node'] += len(eng.pack.layers) * knn_mf * 2048
    out['triplet'] += len(eng.pack.layers) * (tri_tiles * 112 * 2048 + p.n_bond * 2 * fold)
    out['total'] = sum(out.values())
    return out


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
    # 112 MFMA 16x16x4 (2048 FLOP) per 16-row tile + per-segment fold/unfold (2 x 128x128 MACs).  (Q = Wg2 . smear(d_ji) is NOT
    # computed in the staged kernel: it arrives as a row of a [E,20]x[20,256] GEMM, so its FLOPs are not this kernel's)
    tiles = int((triplet_tiles(n_at) * n_at * (n_at - 1)).sum())
    tri_exec = tiles * 112 * 2048 + e_bond * (2 * 2 * H * H)
    return dict(n_all=n_all, n_lig=n_lig, e_knn=e_knn, e_bond=e_bond, e3=e3, flops_step=6 * layer_f,
                flops_triplet_kernel=tri_kernel, flops_triplet_executed=tri_exec, flops_triplet_mfma=tiles * 112 * 2048, tri_tiles=tiles, bytes_step=step_b,
                tri_useful_rows=e3, tri_padded_rows=16 * tiles)      # (e3 = n(n-1)(n-2) useful rows)


# ---------------------------------------------------------------------------- CPU baseline (oracle, bounded sample)
def cpu_baseline(work, n_sample_graphs=8, n_steps=3, max_threads=16):
    """Reference-dataflow CPU path (oracle/phoregen_oracle.py: unfactored, materialised [E3,437] tensors, fp32; pinned to
    the reference by tests/golden) timed on the host cores: SURVEY 8(d)'s K = 3 steps on a sub-batch, scaled linearly in
    graphs to the 128-graph batch.  The sub-batch is bounded by TIME, not by RAM: 8 graphs x 3 steps is ~20-30 s of host
    work, the budget a default bench run can afford (B = 32 would fit host RAM but takes minutes)."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from helpers import make_oracle
    from oracle import phoregen_oracle as po
    import torch.nn.functional as F
    # a 256-thread intra-op pool on these small tensors is pathological (measured: 300x slower than 8 threads),
    # so the baseline uses at most `max_threads` host cores and says so
    cores = min(os.cpu_count() or 1, max_threads)
    torch.set_num_threads(cores)
    o = make_oracle(0)
    n_sample_graphs = min(n_sample_graphs, int(work['num_atoms'].numel()))
    sel = torch.arange(n_sample_graphs)
    na = work['num_atoms'][sel]
    keep = work['batch_phore'] < n_sample_graphs
    hp, pp, pn, bp = (work[k][keep] for k in ('h_phore', 'pos_phore', 'phore_norm', 'batch_phore'))
    g = torch.Generator().manual_seed(7)
    bn = torch.repeat_interleave(torch.arange(n_sample_graphs), na)
    ei, be = po.make_edge_data(na)
    N, E = int(na.sum()), ei.size(1)
    hn = F.one_hot(torch.full((N,), 11), 12).float()
    he = F.one_hot(torch.zeros(E, dtype=torch.long), 6).float()
    pos = torch.randn(N, 3, generator=g)
    log_n, log_e = torch.log(hn.clamp(min=1e-30)), torch.log(he.clamp(min=1e-30))
    t0 = None
    with torch.no_grad():
        for s in range(n_steps + 1):                       # one reverse step = forward + the three posteriors
            if s == 1:
                t0 = time.perf_counter()
            tt = torch.full((n_sample_graphs,), 999 - s)
            v, x0, bond, _ = o.forward(hn, pos, bn, he, ei, be, tt, hp, pp, pn, bp)
            log_n = po.q_v_posterior(o.tab_node, F.log_softmax(v, -1), log_n, tt, bn)
            hn = F.one_hot(po.gumbel_argmax(log_n, torch.rand(N, 12, generator=g)), 12).float()
            log_e = po.q_v_posterior(o.tab_edge, F.log_softmax(bond, -1), log_e, tt, be)
            he = F.one_hot(po.gumbel_argmax(log_e, torch.rand(E, 6, generator=g)), 6).float()
            pos = po.pos_prev_from_recon(o.tab_pos, pos, x0, tt, bn, torch.randn(N, 3, generator=g))
    dt = (time.perf_counter() - t0) / n_steps
    graph_steps = n_sample_graphs / dt
    return dict(value=graph_steps / 128.0, unit='denoise-steps/s (128-graph batch)', cores=cores, kind='port',
                sample=f'K = {n_steps} timed reverse steps (+1 warm-up) of oracle/phoregen_oracle.py on the first {n_sample_graphs} '
                       f'graphs ({N} atoms, {E} bond edges) of the workload, {dt:.2f} s/step, scaled linearly to 128 graphs')


def measured_mfma_peak(dev):
    """fp32 matrix rate this GPU sustains now (TFLOP/s): pg_micro_mfma_f32 (csrc/micro.hip: 4 waves per SIMD issuing nothing but
    v_mfma_f32_16x16x4_f32), best of 3 launches of ~1.7 ms, HIP events on the launch stream.  After the timed region."""
    import ctypes as C
    from phoregen_amd import hip
    lib = hip.lib()
    sink = torch.zeros(16, device=dev)
    flops = C.c_double()
    best = 0.0
    for it in range(4):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        hip.check(lib.pg_micro_mfma_f32(1024, 200 if it == 0 else 2000, sink.data_ptr(), C.byref(flops), hip.stream_ptr()), 'pg_micro_mfma_f32')
        b.record()
        b.synchronize()
        if it:
            best = max(best, flops.value / (a.elapsed_time(b) * 1e-3) / 1e12)
    return best


def secondary_configs(model, dev, budget_s=150.0):
    """BASELINE.json's other single-GPU configurations, run AFTER the headline's timed region in the same process (rank 0, N = 1) so that
    the driver's own bench run carries them: configs[1] (one 100-graph `sample`, 1000 steps, sample.sh's guidance), configs[4] (the
    training step, 256 pairs) and configs[3] as a job on this GPU (2 batches of 128 graphs, 1000 steps each, graphs / hour).  Each entry
    states its `steps` and the arithmetic that ties its value to the wall time measured here; an entry that would not fit the time
    budget is listed as skipped.  The headline fields are untouched."""
    import numpy as np
    out, t_start = {}, time.perf_counter()
    left = lambda: budget_s - (time.perf_counter() - t_start)
    # ---- configs[1]: sample_all.py --num_samples 100 on one pharmacophore (tests/golden/g8_phore_parse.npz = P03211_merge.phore parsed) ----
    try:
        from phoregen_amd.data import PhoreGraph
        g = np.load(os.path.join(ROOT, 'tests', 'golden', 'g8_phore_parse.npz'))
        tt = lambda a: torch.as_tensor(np.asarray(a))
        data = PhoreGraph(tt(g['x']), tt(g['pos']), tt(g['norm']), tt(g['center'])).to(dev)
        guid = [{'type': 'atom_prox', 'min_d': 1.2, 'max_d': 1.9}, {'type': 'center_prox'}]            # sample.sh:21,31
        na = torch.randint(20, 45, (100,), generator=torch.Generator().manual_seed(2032))
        model.sample(data, 100, dev, pos_guidance_opt=guid, num_atoms=na, num_steps=3, return_traj=False)   # (plan / engine of this shape)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = model.sample(data, 100, dev, pos_guidance_opt=guid, num_atoms=na, return_traj=True)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        out['config2_s'] = {'value': dt, 'unit': 's', 'steps': 1000, 'graphs': 100, 'guidance': 'atom_prox(1.2, 1.9) + center_prox',
                            'what': 'BASELINE.json configs[1]: wall time of ONE PhoreDiff.sample call (100 graphs on the 44-node pharmacophore, '
                                    '1000 steps, trajectory written), topology + encoder + result tensors included',
                            'ms_per_step': dt, 'steps_per_sec': 1000 / dt, 'n_lig': int(na.sum()), 'e_bond': int((na * (na - 1)).sum()),
                            'finite': bool(torch.isfinite(res['pred'][1]).all())}
        del res
    except Exception as ex:                                                  # a secondary figure never takes the headline line down
        out['config2_s'] = {'error': repr(ex)[:300]}
    # ---- configs[4]: one training step (compute_loss forward + backward + Adam) on 256 synthetic pairs ----
    try:
        if left() < 20:
            raise TimeoutError(f'skipped: {left():.0f} s of the {budget_s:.0f} s budget left')
        sys.path.insert(0, os.path.join(ROOT, 'tools'))
        from bench_train import train_workload
        batch, na_t = train_workload(256, seed=4321)
        batch.to(dev)
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()
        model.train()
        opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-5)

        def step():
            opt.zero_grad(set_to_none=True)
            loss, info = model.compute_loss(batch)
            loss.backward()
            opt.step()
            return info
        for _ in range(2):
            step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            info = step()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        from phoregen_amd import training as _tr
        _tr.lib_timers = []                       # one more step with HIP events around the library GEMM calls (the input gradients)
        step()
        torch.cuda.synchronize()
        lib_ms = [a.elapsed_time(b) for a, b in _tr.lib_timers]
        _tr.lib_timers = None
        out['train_ms_per_step'] = {'value': dt / 3 * 1e3, 'unit': 'ms/step', 'steps': 3, 'warmup': 2, 'graphs': 256,
                                    'library_ms': sum(lib_ms), 'library_calls': len(lib_ms),
                                    'library_note': 'input-gradient GEMMs on rocBLAS / hipBLASLt through torch.mm (training.py _dgrad), part of `value`',
                                    'peak_mem_gb': torch.cuda.max_memory_allocated() / 2 ** 30,
                                    'what': 'BASELINE.json configs[4]: compute_loss forward + backward + Adam on 256 synthetic ligand-pharmacophore '
                                            'pairs (n ~ N(25, 5) clamp [8, 60]); tools/bench_train.py is the full benchmark',
                                    'wall_s': dt, 'n_lig': int(na_t.sum()), 'e_bond': int((na_t * (na_t - 1)).sum()), 'last_loss': float(info['loss'])}
        del opt, batch
        model.zero_grad(set_to_none=True)
        model.eval()
        model.invalidate_pack()            # (Adam moved the weights: the sampler below packs them again)
        torch.cuda.empty_cache()
    except Exception as ex:
        model.eval()
        out['train_ms_per_step'] = {'error': repr(ex)[:300]}
    # ---- configs[3] as a job on ONE GPU: 2 batches of 128 graphs of several pharmacophores, 1000 steps each, return_traj=False ----
    try:
        if left() < 45:
            raise TimeoutError(f'skipped: {left():.0f} s of the {budget_s:.0f} s budget left')
        from phoregen_amd.parallel import run_sampling_job
        job = config4_job(n_phores=8, samples=32)
        run_sampling_job(model, config4_job(n_phores=2, samples=4, seed=1), batch_size=8, num_steps=3)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pred, na4 = run_sampling_job(model, job, batch_size=128)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        out['config4_graphs_per_hour'] = {'value': job.n_graphs / dt * 3600, 'unit': 'graphs/hour (one GPU)', 'steps': 1000, 'batches': 2, 'graphs': job.n_graphs,
                                          'what': 'BASELINE.json configs[3], one GPU\'s slice: 8 synthetic CpxPhore/DockPhore-shaped pharmacophores x 32 samples, '
                                                  'run_sampling_job in batches of 128, everything between the batches included',
                                          'wall_s': dt, 'ms_per_step_per_batch': dt / 2, 'graph_steps_per_sec': job.n_graphs * 1000 / dt,
                                          'finite': bool(torch.isfinite(pred[1]).all())}
    except Exception as ex:
        out['config4_graphs_per_hour'] = {'error': repr(ex)[:300]}
    out['secondary_wall_s'] = time.perf_counter() - t_start
    return out


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a torch.distributed.run child (one process per
    GPU) from this still GPU-free parent and hand its exit code on.  Nothing here may initialise HIP."""
    import subprocess
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    return subprocess.run(cmd, env=env).returncode


def _traffic_record(kernel_prefix):
    """HBM bytes per launch of the dominant kernel from the committed PMC profile (rocprofv3 --pmc passes, gfx950 corrections
    of MI355X_MICROARCH.md applied by tools/pmc_summary.py).  A bench run cannot read PMC counters itself, so the figure is
    labelled with where it came from and dropped when it was recorded for another kernel."""
    path = os.path.join(ROOT, 'profiles', 'triplet_traffic.json')
    try:
        rec = json.load(open(path))
    except Exception:
        return None, None
    if not str(rec.get('kernel', '')).startswith(kernel_prefix):
        return None, {'file': 'profiles/triplet_traffic.json', 'note': f"recorded for {rec.get('kernel')!r}, not for the kernel timed here"}
    src = {k: rec.get(k) for k in ('kernel', 'commit', 'workload', 'fetch_bytes_x2', 'write_bytes', 'method') if k in rec}
    src['file'] = 'profiles/triplet_traffic.json'
    return rec.get('hbm_bytes_per_launch'), src


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=40)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--repeats', type=int, default=3, help='timed blocks of --steps steps each; the median block is reported')
    ap.add_argument('--graphs', type=int, default=128)
    ap.add_argument('--fixed-shape', action='store_true', help='n=40, p=107 for every graph (closed-form counts)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-secondary', action='store_true', help='skip BASELINE configs[1] / [3] / [4] after the timed region (they run at N = 1 in a full default-style run only)')
    ap.add_argument('--weak', action='store_true',
                    help='weak scaling only: --graphs graphs PER GPU (default: strong scaling, SURVEY.md 8d: ONE batch of '
                         '--graphs graphs partitioned over the ranks by the fitted step cost; a weak-scaling figure is added for N > 1)')
    ap.add_argument('--strong', action='store_true', help='(default) kept for compatibility')
    ap.add_argument('--train', action='store_true',
                    help='BASELINE config 5 instead: one compute_loss forward + backward + Adam step on 256 synthetic pairs '
                         '(tools/bench_train.py prints its own JSON line, ms/step)')
    args = ap.parse_args()
    if args.train:
        sys.path.insert(0, os.path.join(ROOT, 'tools'))
        import bench_train
        return bench_train.main(['--steps', str(max(args.steps // 8, 3)), '--warmup', '2'])

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        sys.exit(_spawn_ranks(args.gpus))

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    n_dev = max(torch.cuda.device_count(), 1)
    local = local % n_dev                                  # (a 2-rank gloo dry run can share one GPU)
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    backend = None
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        backend = os.environ.get('PG_DIST_BACKEND', 'nccl')                   # nccl == RCCL on ROCm
        dist.init_process_group(backend)

    from phoregen_amd.config import default_model_config
    from phoregen_amd.models.diffusion import PhoreDiff
    from phoregen_amd.parallel import gather_predictions, partition_graphs
    from phoregen_amd.weights import init_deterministic_

    model = init_deterministic_(PhoreDiff(default_model_config(), 'zinc_300'), 0).eval().to(dev)
    K, W, R = args.steps, args.warmup, max(args.repeats, 1)
    T = model.num_timesteps

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device='cpu' if backend == 'gloo' else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def run(work, gids, time_triplet):
        """W warm-up steps, then R blocks of exactly K steps, each bracketed by barrier + synchronize; per-block wall time
        is the MAX over ranks.  Returns (block times [R], triplet launch durations, final result dict)."""
        n_total = min(W + 32 + R * K, max(T, W + R * K))       # room for up to 32 more untimed steps (grid tuning of small batches, below)
        st = model.begin_sampling(work['h_phore'], work['pos_phore'], work['phore_norm'], work['batch_phore'],
                                  work['num_atoms'], torch.zeros(int(work['num_atoms'].numel()), 3), rng='device', seed=0,
                                  return_traj=True, num_steps=n_total, graph_ids=gids, pipeline=True)
        for i in range(W):
            model.reverse_step(st, i, T - 1 - i)
        # (small batches choose their persistent triplet grid inside begin_sampling since round 6 -- Engine.calibrate_tri_grid -- so the loop
        #  below runs a fixed launch list; nothing of the choice falls into the timed region)
        extra = 0
        if time_triplet:                       # (the headline run; the weak-scaling figure below reports its own)
            run.warmup_steps = W + extra
        times, tri, knn = [], [], []
        i = W + extra
        for r in range(R):
            barrier()
            # HIP events around the triplet / knn-node launches (on their lanes) during the LAST timed block only: 24 event records
            # per step cost a small batch ~2 % (16 graphs: 3.36 vs 3.30 ms per step), so with R = 3 the median block -- the
            # headline -- is one without them, and the instrumented block stays listed in `repeats_ms_per_step`
            if time_triplet and r == R - 1:
                st.eng.timers = {}
            t0 = time.perf_counter()
            for _ in range(K):
                model.reverse_step(st, i, T - 1 - i)
                i += 1
            barrier()
            times.append(max_over_ranks(time.perf_counter() - t0))
            if time_triplet and r == R - 1:
                tri += st.eng.kernel_ms('triplet')
                knn += st.eng.kernel_ms('knn_node')
                st.eng.timers = None
        if time_triplet:
            run.knn_ms, run.knn_mfma = knn, knn_node_mfma(st.eng)
            run.exec_flops = executed_flops(st.eng)
            run.tri_launches = len(st.eng.tri_calls)
            run.tri_grid = st.eng.prog_fwd[st.eng.tri_calls[0]][1][-1]._obj.tri_grid or 256      # persistent triplet workgroups in use (small batches: tuned online)
            run.knn_launches = 1 if any(a[1]._obj.mode == 0 and a[1]._obj.n_seg2 > 0 for f, a, l in st.eng.prog_fwd if l >= 0 and f is st.eng.lib.pg_seg_attn) else 2
        return times, tri, model.finish_sampling(st)

    # ---- strong scaling (headline): ONE batch of --graphs graphs, partitioned over the ranks by phoregen_amd.parallel.graph_cost ----
    full = ligphore_workload(args.graphs, seed=1234, fixed_shape=args.fixed_shape)
    if args.weak:
        work = ligphore_workload(args.graphs, seed=1234 + rank, fixed_shape=args.fixed_shape)   # each rank: its own graphs
        mine = torch.arange(args.graphs) + rank * args.graphs
        total_graphs = world * args.graphs
    else:
        mine = partition_graphs(full['num_atoms'], world, full['n_phore'])[rank]
        work = subset_workload(full, mine) if world > 1 else full
        total_graphs = args.graphs
    counts = algorithmic_counts(work['num_atoms'], work['n_phore'])
    times, tri_ms, res = run(work, mine, time_triplet=True)
    order = sorted(range(R), key=lambda r: times[r])
    dt = times[order[R // 2]]                              # median block

    # the one collective of the path: re-assemble `pred` of all ranks (phoregen_amd/parallel.py), after the timed loop
    gather_ms = None
    if world > 1:
        tg = time.perf_counter()
        gather_predictions(res['pred'], work['num_atoms'], mine)
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - tg) * 1e3

    # ---- secondary figure for N > 1: weak scaling, --graphs graphs per GPU ----
    weak = None
    if world > 1 and not args.weak:
        wwork = ligphore_workload(args.graphs, seed=1234 + rank, fixed_shape=args.fixed_shape)
        wt, _, _ = run(wwork, torch.arange(args.graphs) + rank * args.graphs, time_triplet=False)
        wdt = sorted(wt)[R // 2]
        weak = {'graphs_per_gpu': args.graphs, 'ms_per_step': wdt / K * 1e3, 'graph_steps_per_sec': K * world * args.graphs / wdt,
                'batch_steps_per_sec': world * K / wdt}

    if rank == 0:
        peak_tf = 157.3                      # fp32 MFMA dense peak, MI355X_MICROARCH.md "Chip-level parameters"
        tri_avg_ms = sum(tri_ms) / max(len(tri_ms), 1)
        exec_tf = counts['flops_triplet_executed'] / (tri_avg_ms * 1e-3) / 1e12 if tri_ms else None
        alg_tf = counts['flops_triplet_kernel'] / (tri_avg_ms * 1e-3) / 1e12 if tri_ms else None
        traffic, traffic_src = _traffic_record('pg::triplet')
        meas_tf = measured_mfma_peak(dev)
        mfma_only_tf = counts['flops_triplet_mfma'] / (tri_avg_ms * 1e-3) / 1e12 if tri_ms else None
        scale = 1.0 if not args.weak else float(world)
        line = {
            'metric': 'denoise-steps/sec (batch=128, ~40-atom graphs)', 'value': scale * K / dt, 'unit': 'steps/s',
            'n_gpus': world, 'steps': K, 'warmup': getattr(run, 'warmup_steps', W), 'warmup_requested': W, 'ms_per_step': dt / K * 1e3, 'higher_is_better': True,
            'scaling': 'weak' if args.weak else 'strong', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'BASELINE.json configs[2]: ONE batch of {args.graphs} LigPhore-shaped graphs '
                                   '(n~N(40,6) atoms, p~N(107,30) pharmacophore nodes)' +
                                   (' per GPU' if args.weak else f', partitioned over {world} GPU(s) by the fitted step cost') +
                                   ', steps t=999.. of the 1000-step sampler, device Philox noise, trajectory written',
                       'graphs_total': total_graphs, 'graphs_rank0': int(work['num_atoms'].numel()), 'fixed_shape': args.fixed_shape,
                       'n_ctx': counts['n_all'], 'n_lig': counts['n_lig'], 'e_knn': counts['e_knn'], 'e_bond': counts['e_bond'],
                       'e3': counts['e3'], 'parallelism': f'graph-sharded x{world}, final RCCL gather only',
                       'triplet_workgroups_rank0': getattr(run, 'tri_grid', None)},
            'repeats': R, 'repeats_ms_per_step': [t / K * 1e3 for t in times], 'statistic': 'median block of `repeats` blocks of `steps` steps',
            'graph_steps_per_sec': K * total_graphs / dt,
            'ranks': world, 'distinct_devices': min(world, n_dev), 'dist_backend': backend,
            'roofline': {'kernel': 'triplet kernel (pg_seg_attn PG_SEG_TRIPLET = BondUpdateLayer, 6 sub-layers/step' + (': each as two launches, ligands of up to 50 atoms on the 3-tile instance and the larger ones beside them on another lane; a launch = the pair' if getattr(run, 'tri_launches', 6) > 6 else '') + '), rank 0',
                         'bound': 'mfma', 'achieved': exec_tf, 'peak': peak_tf, 'unit': 'TFLOP/s',
                         'frac': (exec_tf / peak_tf) if exec_tf else None, 'traffic': traffic, 'traffic_source': traffic_src,
                         'frac_mfma_only': (mfma_only_tf / peak_tf) if mfma_only_tf else None,
                         'peak_measured': meas_tf, 'frac_vs_measured_peak': (exec_tf / meas_tf) if exec_tf else None,
                         'frac_mfma_only_vs_measured_peak': (mfma_only_tf / meas_tf) if mfma_only_tf else None,
                         'peak_note': '`peak` = nominal dense fp32 MFMA rate (MI355X_MICROARCH.md); `peak_measured` = pg_micro_mfma_f32 in this '
                                      'process after the timed region (4 waves per SIMD issuing only v_mfma_f32_16x16x4_f32, best of 3 launches); '
                                      '`frac_mfma_only` leaves the query fold / value unfold (vector-ALU FLOPs) out of the numerator',
                         'avg_launch_ms': tri_avg_ms, 'launches_timed': len(tri_ms),
                         'timed_in': f'block {R} of {R} of the timed region (every launch of its {K} steps; HIP events on the launch stream)',
                         'flops_per_launch': counts['flops_triplet_executed'],
                         'note': 'achieved = fp32 FLOPs the kernel EXECUTES per launch (112 MFMA 16x16x4 per 16-row tile + query fold / '
                                 'value unfold, padding rows included; Q arrives as a GEMM row) / mean launch duration (HIP events on the launch stream). '
                                 'SURVEY 8d counts the unfolded second layers the kernel never runs: see survey_* (can exceed 1).',
                         'useful_frac': (exec_tf / peak_tf * counts['tri_useful_rows'] / counts['tri_padded_rows']) if exec_tf else None,
                         'survey_flops_per_launch': counts['flops_triplet_kernel'], 'survey_achieved': alg_tf,
                         'survey_frac': (alg_tf / peak_tf) if alg_tf else None,
                         'share_of_step': (6 * tri_avg_ms) / (dt / K * 1e3) if tri_ms else None},
            # the kernel furthest below its roofline (round-2 review): the knn-node attention sub-layer, both launches of a layer
            'roofline_knn_node': (lambda ms, mf: {
                'kernel': 'knn-node attention (pg_seg_attn PG_SEG_KNN_NODE fused form, node_attn_kernel<true,false,2,768,true>), '
                          + ('ONE launch per layer serving the ligand and the pharmacophore target lists' if run.knn_launches == 1 else
                             '2 launches per layer: ligand targets, pharmacophore targets') + ', rank 0',
                'bound': 'mfma', 'peak': peak_tf, 'unit': 'TFLOP/s', 'avg_sublayer_ms': ms, 'sublayers_timed': len(run.knn_ms),
                'flops_per_sublayer': mf * 2048 + 2 * 2 * 128 * 128 * counts['n_all'],
                'achieved': (mf * 2048 + 2 * 2 * 128 * 128 * counts['n_all']) / (ms * 1e-3) / 1e12,
                'frac': (mf * 2048 + 2 * 2 * 128 * 128 * counts['n_all']) / (ms * 1e-3) / 1e12 / peak_tf,
                'note': 'achieved = fp32 FLOPs EXECUTED (16x16x4 MFMAs counted from the current neighbour lists, kind-uniform tiles '
                        'skip the other kind\'s distance columns, + query fold / value unfold per node) / mean duration of the '
                        'sub-layer (HIP events on its lane around its launch(es)); inside the four-lane step, other lanes run beside it'})(
                sum(run.knn_ms) / max(len(run.knn_ms), 1), run.knn_mfma) if getattr(run, 'knn_ms', None) else None,
            'step_roofline': {'flops_executed': run.exec_flops['total'], 'flops_executed_by_kernel': run.exec_flops,
                              'exec_frac': run.exec_flops['total'] / (dt / K) / (peak_tf * 1e12),
                              'exec_note': 'fp32 FLOPs EXECUTED by all launches of a step (bench.executed_flops: GEMMs 2MNK from the launch '
                                           'list, attention kernels by MFMA count incl. padding rows + fold / unfold) / step time / fp32 '
                                           'MFMA peak: the utilisation of the whole step, rank 0',
                              'flops_alg_survey': counts['flops_step'], 'bytes_alg': counts['bytes_step'],
                              'survey_mfma_frac': counts['flops_step'] / (dt / K) / (peak_tf * 1e12),
                              'hbm_frac': counts['bytes_step'] / (dt / K) / 8e12,
                              'note': 'rank-0 share of the batch; survey_mfma_frac uses SURVEY 8d FLOPs (unfolded form) and is not a '
                                      'utilisation; the step is fp32-issue-bound, not HBM-bound (AI ~ 3000 FLOP/B)'},
            'weak_scaling': weak,
            'final_gather_ms': gather_ms,
        }
        # the other single-GPU configurations of BASELINE.json, after the timed region (N = 1, full runs only: the quick runs of tools/ pass
        # --no-cpu-baseline or another --graphs)
        if world == 1 and not (args.no_secondary or args.no_cpu_baseline or args.weak) and args.graphs == 128:
            del res
            torch.cuda.empty_cache()
            line.update(secondary_configs(model, dev))
        if not args.no_cpu_baseline:           # rank 0, after the timed region, at every world size (the other ranks wait at the
            line['cpu_baseline'] = cpu_baseline(full)      # final barrier below)
        else:
            line['cpu_baseline'] = None
        print(json.dumps(line))
    if world > 1:
        dist.barrier()                  # (rank 0 times the CPU baseline after the timed region; the others wait here)
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
