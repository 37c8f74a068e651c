#!/usr/bin/env python3
"""Where the engine's size thresholds come from (round-5 review, item 8): every schedule switch of phoregen_amd/options.py that is chosen by
batch size, forced both ways on three workload SHAPES at several batch sizes, in one process with alternating runs:
  headline  BASELINE configs[2]: n ~ N(40, 6) atoms, p ~ N(107, 30) pharmacophore nodes per graph   (bench.ligphore_workload)
  config2   BASELINE configs[1]: G samples of ONE 44-node pharmacophore, n uniform in 20 .. 44            (tests/golden/g8_phore_parse.npz)
  config4   BASELINE configs[3]: p ~ N(80, 25), several pharmacophores per batch                        (bench.config4_job)
Per (shape, graphs): ms per sampler step of the DEFAULT and of each forced setting, and the default's distance from the best.  The table is
kept as profiles/r06_schedule_fit.md; tests/test_gpu_schedule.py::test_default_schedule_is_near_the_best_variant asserts the distance on
one size per shape.        GPU box: python tools/fit_schedule.py [quick]
"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

VARIANTS = {                    # switch -> the forced settings that bracket its threshold
    'ahead_v2': [dict(ahead_v2='never'), dict(ahead_v2='always')],
    'chain_q_from': [dict(chain_q_from=0), dict(chain_q_from=10 ** 9)],
    'tri_split': [dict(tri_split=False), dict(tri_split='always')],
    'tri_overlap': [dict(tri_overlap=0)],
    'pos_tiled': [dict(pos_tiled='never'), dict(pos_tiled='always')],
    'geom_split': [dict(geom_split='never'), dict(geom_split='always')],
}


def workloads(quick=False):
    from bench import config4_job, ligphore_workload
    out = []
    for G in ((16, 128) if quick else (8, 16, 32, 64, 128)):
        out.append(('headline', G, ligphore_workload(G)))
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'g8_phore_parse.npz'))
    t = lambda a: torch.as_tensor(np.asarray(a))
    for G in ((100,) if quick else (25, 100)):
        na = torch.randint(20, 45, (G,), generator=torch.Generator().manual_seed(2032))
        p = g['x'].shape[0]
        out.append(('config2', G, dict(h_phore=t(g['x']).repeat(G, 1), pos_phore=t(g['pos']).repeat(G, 1), phore_norm=t(g['norm']).repeat(G, 1),
                                       batch_phore=torch.repeat_interleave(torch.arange(G), p), num_atoms=na)))
    for G in ((128,) if quick else (16, 128)):
        job = config4_job(n_phores=max(G // 8, 2), samples=8)
        hp, pp, pn, bp, na, _ = job.batch_inputs(torch.arange(G))
        out.append(('config4', G, dict(h_phore=hp, pos_phore=pp, phore_norm=pn, batch_phore=bp, num_atoms=na)))
    return out


def ms_per_step(model, w, W=6, K=24, R=3, **kw):
    from phoregen_amd import options
    G = int(w['num_atoms'].numel())
    with options.override(**kw):
        model._engine = None
        st = model.begin_sampling(w['h_phore'], w['pos_phore'], w['phore_norm'], w['batch_phore'], w['num_atoms'], torch.zeros(G, 3),
                                  rng='device', seed=0, return_traj=False, num_steps=W + R * K, pipeline=True)
        for i in range(W):
            model.reverse_step(st, i, 999 - i)
        ts = []
        for r in range(R):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(W + r * K, W + (r + 1) * K):
                model.reverse_step(st, i, 999 - i)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / K * 1e3)
        n_bond = st.eng.plan.n_bond
        del st
        model._engine = None
    return sorted(ts)[R // 2], n_bond


def measure(model, w, reps=2):
    """{variant label: median ms} with the default first; every variant `reps` times, alternating."""
    labels = [('default', {})] + [(f'{k}={list(v.values())[0]}', v) for k, vs in VARIANTS.items() for v in vs]
    acc = {l: [] for l, _ in labels}
    n_bond = 0
    for _ in range(reps):
        for l, kw in labels:
            ms, n_bond = ms_per_step(model, w, **kw)
            acc[l].append(ms)
    return {l: min(v) for l, v in acc.items()}, n_bond


if __name__ == '__main__':
    from phoregen_amd.config import default_model_config
    from phoregen_amd.models.diffusion import PhoreDiff
    from phoregen_amd.weights import init_deterministic_
    quick = 'quick' in sys.argv[1:]
    model = init_deterministic_(PhoreDiff(default_model_config(), 'zinc_300'), 0).eval().to('cuda')
    rows = []
    for shape, G, w in workloads(quick):
        res, n_bond = measure(model, w)
        best = min(res, key=res.get)
        rows.append(dict(shape=shape, graphs=G, n_bond=n_bond, ms=res, best=best, default_over_best=res['default'] / res[best]))
        print(json.dumps(rows[-1]), flush=True)
    labels = list(rows[0]['ms'])
    print('\n| shape | graphs | bond rows | ' + ' | '.join(labels) + ' | default / best |')
    print('|---|---|---|' + '---|' * (len(labels) + 1))
    for r in rows:
        print(f"| {r['shape']} | {r['graphs']} | {r['n_bond']} | " + ' | '.join(('**%.3f**' if l == r['best'] else '%.3f') % r['ms'][l] for l in labels) + f" | {r['default_over_best']:.3f} |")
