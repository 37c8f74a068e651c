"""The engine's launch lists are ORDERED by construction, not by timing (tools/check_schedule.py): every two launches that touch the same
buffer region, one of them writing, are connected through lane order and the record / wait points -- in the plain forward list and across
three consecutive steps of the pipelined sampler loop, for batch sizes that cover every schedule regime (phoregen_amd/options.py).
(The lists hold device pointers, so an engine must be built on the GPU; nothing is launched.)"""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def model():
    from phoregen_amd.config import default_model_config
    from phoregen_amd.models.diffusion import PhoreDiff
    from phoregen_amd.weights import init_deterministic_
    return init_deterministic_(PhoreDiff(default_model_config(), 'zinc_300'), 0).eval().to('cuda')


@pytest.mark.parametrize('graphs', [2, 8, 16, 48, 72, 128])
@pytest.mark.parametrize('guided', [False, True])
def test_launch_lists_are_ordered(model, graphs, guided):
    from bench import ligphore_workload
    import check_schedule as cs
    fwd, pipe, info = cs.check_engine(model, ligphore_workload(graphs), guided)
    model._engine = None
    assert info['pipelined'] and pipe is not None
    assert not fwd, [cs.fmt(h) for h in fwd]
    assert not pipe, [cs.fmt(h) for h in pipe]


def test_checker_notices_a_missing_order_point(model):
    """The check is sensitive to the order points it verifies: with any single wait of `prog_step` removed a hazard appears for most of them (the
    list carries few redundant waits) -- among them the class round 5's advisor found by reading the code (a side lane released too early
    overwriting / reading h' of the last layer): lane 2's wait for `hn_done` in front of the node head, without which the head reads h[0] while
    `lin_node` on lane 1 is still writing it.  (Round 5's own hazard -- the next step's embedding into h[0] racing the position phase's reads --
    is gone by construction since round 6: the embedded features live in `ws.h_in`.)"""
    from bench import ligphore_workload
    import check_schedule as cs
    work = ligphore_workload(16)
    _, pipe, info = cs.check_engine(model, work)
    eng = info['eng']
    assert info['v2'] and not pipe
    waits = [(k, pt, ln) for k, (fn, a, lane) in enumerate(eng.prog_step) if lane < 0 and fn.kind == 'order'
             for what, pt, ln in fn.ops if what == 'wait']
    found_h0, n_sensitive = False, 0
    for k0, pt0, ln0 in waits:
        _, pipe, _ = cs.check_engine(model, work, drop_step=lambda k, what, pt, ln: what == 'wait' and (k, pt, ln) == (k0, pt0, ln0))
        n_sensitive += bool(pipe)
        found_h0 = found_h0 or (ln0 == 2 and any(h[6] == 'ws.h[0]' and h[1] == 1 and h[4] == 2 for h in pipe))
    model._engine = None
    assert found_h0, 'a lane-2 wait on lane 1 (hn_done) must surface as a hazard on ws.h[0] when removed'
    assert n_sensitive >= len(waits) // 2, (n_sensitive, len(waits))


def test_triplet_launches_side_by_side_write_disjoint_ligands(model, monkeypatch):
    """From 82 k bond edges the bond-triplet sub-layer is two launches, the larger ligands' on lane 3 beside the other (options.tri_overlap).
    The checker carries the set of ligands of the launch's queue on the region it writes: the pair is not a conflict BECAUSE the queues are
    disjoint (and together cover the batch) -- told that both queues hold the same ligands, it reports the pair."""
    from bench import ligphore_workload
    import check_schedule as cs
    work = ligphore_workload(72)
    fwd, pipe, info = cs.check_engine(model, work)
    eng = info['eng']
    assert not fwd and not pipe and len(eng.tri_calls) == 12 and eng.plan.tri_split is not None
    assert [eng.prog_fwd[k][2] for k in eng.tri_calls] == [3, 0] * 6
    small, big = (set(int(v) for v in q[0][:q[1], 2].cpu().tolist()) for q in (eng.plan.tri_split['small'], eng.plan.tri_split['big']))
    assert not (small & big) and (small | big) == set(int(v) for v in eng.plan.tri_iters[:eng.plan.n_tri_iters, 2].cpu().tolist())
    real = cs.Buffers.add_queue

    def same_ligands(self, iters, n):
        real(self, iters, n)
        self.queues[iters.data_ptr()] = frozenset([0])
    monkeypatch.setattr(cs.Buffers, 'add_queue', same_ligands)
    fwd, pipe, _ = cs.check_engine(model, work)
    model._engine = None
    assert any(h[6].startswith('ws.hb') and {h[1], h[4]} == {0, 3} and h[2] == h[5] == 'w' for h in fwd), [cs.fmt(h) for h in fwd]


@pytest.mark.parametrize('shape,graphs', [('headline', 16), ('config2', 100), ('config4', 128)])
def test_default_schedule_is_near_the_best_variant(model, shape, graphs):
    """The size thresholds of phoregen_amd/options.py were fitted on the headline shape; here every size-dependent switch is forced both ways on
    three shapes (tools/fit_schedule.py; BASELINE configs[2], [1], [3]) and the DEFAULT choice must stay within 4 % of the fastest variant
    (best of two alternating runs each; the committed table profiles/r06_schedule_fit.md has it within 2 % at every size measured)."""
    import fit_schedule as fs
    w = next(w for s, g, w in fs.workloads(quick=True) if s == shape and g == graphs)
    labels = [('default', {})] + [(f'{k}={list(v.values())[0]}', v) for k, vs in fs.VARIANTS.items() for v in vs]
    best = {l: 1e9 for l, _ in labels}
    for _ in range(2):
        for l, kw in labels:
            ms, _ = fs.ms_per_step(model, w, W=4, K=16, R=3, **kw)
            best[l] = min(best[l], ms)
    fastest = min(best, key=best.get)
    assert best['default'] <= 1.04 * best[fastest], (shape, graphs, fastest, best)
