"""-m gpu: BASELINE config 4 on one GPU -- a multi-pharmacophore job partitioned over 8 (virtual) ranks must reproduce the
unsharded run (SURVEY.md 4 item 6, 8(d) "Config 4", 8(e)): graphs are independent, the device noise is keyed by the global
graph id, and the only collective is the final gather (covered with real processes in tests/test_parallel_gloo.py)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.fixture(scope='module')
def model():
    from phoregen_amd.config import default_model_config
    from phoregen_amd.models.diffusion import PhoreDiff
    from phoregen_amd.weights import init_deterministic_
    return init_deterministic_(PhoreDiff(default_model_config(), 'zinc_300'), 0).eval().to(DEV)


def _job(n_phores=6, samples=4, seed=5):
    from bench import config4_job
    return config4_job(n_phores, samples, seed)


def _reassemble(parts, job):
    """What gather_predictions does with the gathered buffers: per-rank results back into global graph order."""
    per = {}
    for gids, pred in parts:
        n_off = e_off = 0
        for g in gids.tolist():
            n = int(job.num_atoms[g])
            e = n * (n - 1)
            per[g] = (pred[0][n_off:n_off + n], pred[1][n_off:n_off + n], pred[2][e_off:e_off + e])
            n_off, e_off = n_off + n, e_off + e
    return [torch.cat([per[g][i] for g in sorted(per)]) for i in range(3)]


@pytest.mark.parametrize('guidance', [False, True])
def test_sharded_equals_unsharded(model, guidance):
    from phoregen_amd.parallel import partition_graphs, sample_job_shard
    job = _job()
    G = job.n_graphs
    opt = [{'type': 'atom_prox', 'min_d': 1.2, 'max_d': 1.9}, {'type': 'center_prox'}] if guidance else None
    steps = 8
    # unsharded: one batch of all G graphs (sample_job_shard normalises the guidance energies by its batch_size = G)
    whole = sample_job_shard(model, job, torch.arange(G), batch_size=G, seed=99, num_steps=steps, pos_guidance_opt=opt)
    # 8 ranks, run one after the other on this GPU, each in batches of 2 graphs (different batch composition everywhere);
    # the guidance energies average over the logical batch (G graphs) in both runs
    parts = []
    shards = partition_graphs(job.num_atoms, 8, job.n_phore)
    assert sorted(torch.cat(shards).tolist()) == list(range(G))
    for mine in shards:
        res = []
        for b0 in range(0, int(mine.numel()), 2):
            gids = mine[b0:b0 + 2]
            hp, pp, pn, bp, na, centers = job.batch_inputs(gids)
            r = model.sample_batch(hp, pp, pn, bp, na, centers, pos_guidance_opt=opt, rng='device', seed=99, return_traj=False,
                                   num_steps=steps, graph_ids=gids, guidance_batch=G)
            res.append(r['pred'])
        parts.append((mine, [torch.cat([r[i] for r in res]) for i in range(3)]))
    got = _reassemble(parts, job)
    torch.cuda.synchronize()
    assert all(torch.isfinite(t).all() for t in whole)
    # atom / bond types bit-exact, coordinates <= 1e-6 (relative to the coordinate scale)
    assert torch.equal(got[0].argmax(-1), whole[0].argmax(-1)) and torch.equal(got[2].argmax(-1), whole[2].argmax(-1))
    scale = float(whole[1].abs().max())
    assert float((got[1] - whole[1]).abs().max()) <= 1e-6 * max(scale, 1.0)
    assert float((got[0] - whole[0]).abs().max()) <= 1e-5 * float(whole[0].abs().max())


def test_job_driver_matches_manual_shards(model):
    """run_sampling_job without torch.distributed = this rank's shard; the union over ranks covers the job exactly once."""
    from phoregen_amd.parallel import partition_graphs, run_sampling_job
    job = _job(n_phores=3, samples=3, seed=8)
    seen = []
    for rank in range(4):
        pred, na = run_sampling_job(model, job, world=4, rank=rank, batch_size=4, seed=3, num_steps=3)
        mine = partition_graphs(job.num_atoms, 4, job.n_phore)[rank]
        assert torch.equal(na, job.num_atoms[mine]) and pred[0].size(0) == int(na.sum())
        assert pred[2].size(0) == int((na * (na - 1)).sum())
        seen += mine.tolist()
    assert sorted(seen) == list(range(job.n_graphs))


def test_consecutive_sample_calls_draw_fresh_noise(model):
    """`model.sample(data, n)` without a seed (sample_all.py's while loop calls it once per batch): a fresh Philox key per
    call, reproducible under torch.manual_seed."""
    from oracle.make_inputs import synthetic_phore
    from phoregen_amd.data import PhoreGraph
    x, pos, norm = synthetic_phore(torch.Generator().manual_seed(2), 24)
    data = PhoreGraph(x, pos, norm, torch.zeros(3)).to(DEV)
    na = torch.tensor([9, 9, 9])
    torch.manual_seed(2032)
    a = model.sample(data, 3, DEV, num_atoms=na, num_steps=3)
    b = model.sample(data, 3, DEV, num_atoms=na, num_steps=3)
    torch.manual_seed(2032)
    c = model.sample(data, 3, DEV, num_atoms=na, num_steps=3)
    assert not torch.equal(a['traj'][1][0], b['traj'][1][0])            # different initial noise
    assert not torch.equal(a['pred'][1], b['pred'][1])
    assert torch.equal(a['traj'][1], c['traj'][1]) and torch.equal(a['pred'][0], c['pred'][0])
    # graphs of one call differ from each other as well
    assert not torch.equal(a['traj'][1][0][:9], a['traj'][1][0][9:18])


def test_config4_full_batch_equals_single_graph_runs(model):
    """BASELINE config 4 at its real batch size: ONE 128-graph batch of a config-4-shaped job -- 32 different pharmacophores
    (p ~ N(80, 25^2), CpxPhore / DockPhore statistics) x 4 samples, n ~ N(40, 6^2) -- as `run_sampling_job` cuts them
    (sample_all.py:69-175 serves such a job one pharmacophore at a time).  Size-independent properties: (1) the sampler's result
    for a graph inside the 128-graph mixed batch equals that graph sampled alone (types bit-exact, coordinates <= 1e-6), which is
    what makes any shard / batch cut of the job exact; (2) one forward of the whole batch == the same graphs alone, and one
    of them against the oracle."""
    import torch.nn.functional as F
    from helpers import make_oracle, rel_err
    from phoregen_amd.parallel import sample_job_shard
    from phoregen_amd.plan import make_edge_data
    job = _job(n_phores=32, samples=4, seed=17)
    G = job.n_graphs
    assert G == 128 and len({int(p[0].size(0)) for p in job.phores}) > 10          # really mixed pharmacophore sizes
    steps = 4
    whole = sample_job_shard(model, job, torch.arange(G), batch_size=128, seed=21, num_steps=steps)
    n_off = torch.cat([torch.zeros(1, dtype=torch.long), job.num_atoms.cumsum(0)])
    e_off = torch.cat([torch.zeros(1, dtype=torch.long), (job.num_atoms * (job.num_atoms - 1)).cumsum(0)])
    for gi in (0, 41, 86, 127):
        alone = sample_job_shard(model, job, torch.tensor([gi]), batch_size=128, seed=21, num_steps=steps)
        sl_n, sl_e = slice(int(n_off[gi]), int(n_off[gi + 1])), slice(int(e_off[gi]), int(e_off[gi + 1]))
        assert torch.equal(alone[0].argmax(-1), whole[0][sl_n].argmax(-1)) and torch.equal(alone[2].argmax(-1), whole[2][sl_e].argmax(-1))
        assert float((alone[1] - whole[1][sl_n]).abs().max()) <= 1e-6 * max(1.0, float(whole[1].abs().max()))
        assert float((alone[0] - whole[0][sl_n]).abs().max()) <= 1e-5 * float(whole[0].abs().max())
    assert all(torch.isfinite(t).all() for t in whole)
    # ---- one forward of the full mixed batch, graph slices against single-graph forwards and the oracle ----
    hp, pp, pn, bp, na, _ = job.batch_inputs(torch.arange(G))
    g = torch.Generator().manual_seed(23)
    N = int(na.sum())
    ei, be = make_edge_data(na)
    inp = dict(h_node_pert=F.one_hot(torch.randint(0, 12, (N,), generator=g), 12).float(), pos_pert=3.0 * torch.randn(N, 3, generator=g),
               batch_node=torch.repeat_interleave(torch.arange(G), na), h_edge_pert=F.one_hot(torch.randint(0, 6, (ei.size(1),), generator=g), 6).float(),
               edge_index=ei, batch_edge=be, time_step=torch.randint(0, 1000, (G,), generator=g), h_phore=hp, pos_phore=pp,
               phore_norm=pn, batch_phore=bp)
    nph = torch.bincount(bp, minlength=G)
    p_off = torch.cat([torch.zeros(1, dtype=torch.long), nph.cumsum(0)])
    with torch.no_grad():
        out = [o.cpu() for o in model(**{k: v.to(DEV) for k, v in inp.items()})[:3]]
        for gi in (3, 100):
            n0, n1, e0, e1, p0, p1 = (int(v) for v in (n_off[gi], n_off[gi + 1], e_off[gi], e_off[gi + 1], p_off[gi], p_off[gi + 1]))
            one = dict(h_node_pert=inp['h_node_pert'][n0:n1], pos_pert=inp['pos_pert'][n0:n1], batch_node=torch.zeros(n1 - n0, dtype=torch.long),
                       h_edge_pert=inp['h_edge_pert'][e0:e1], edge_index=ei[:, e0:e1] - n0, batch_edge=torch.zeros(e1 - e0, dtype=torch.long),
                       time_step=inp['time_step'][gi:gi + 1], h_phore=hp[p0:p1], pos_phore=pp[p0:p1], phore_norm=pn[p0:p1],
                       batch_phore=torch.zeros(p1 - p0, dtype=torch.long))
            alone = [o.cpu() for o in model(**{k: v.to(DEV) for k, v in one.items()})[:3]]
            assert rel_err(out[0][n0:n1], alone[0]) <= 1e-6 and rel_err(out[1][n0:n1], alone[1]) <= 1e-6
            assert rel_err(out[2][e0:e1], alone[2]) <= 1e-6
            if gi == 100:
                torch.set_num_threads(8)
                ref = make_oracle(0).forward(**one)
                errs = [rel_err(alone[i], ref[i]) for i in range(3)]
                assert max(errs) <= 2e-5, errs


def test_guidance_normalisation_of_a_tail_batch(model):
    """sample_job_shard(guidance_norm=...): the guidance energies are means over the graphs of a call.  5 graphs in batches of 4
    leave a tail batch of one graph: with 'batch_size' (default, partition-invariant) that graph is sampled as
    `guidance_batch=4`, with 'actual' (the reference loop's behaviour for its last batch, sample_all.py:88) as `guidance_batch=1`
    -- and the two differ (a stronger drift in the second)."""
    from phoregen_amd.parallel import sample_job_shard
    job = _job(n_phores=2, samples=3, seed=11)
    opt = [{'type': 'atom_prox', 'min_d': 1.2, 'max_d': 1.9}, {'type': 'center_prox'}]
    ids = torch.arange(5)
    steps = 6
    by_size = sample_job_shard(model, job, ids, batch_size=4, seed=7, num_steps=steps, pos_guidance_opt=opt)
    actual = sample_job_shard(model, job, ids, batch_size=4, seed=7, num_steps=steps, pos_guidance_opt=opt, guidance_norm='actual')
    n4 = int(job.num_atoms[:4].sum())
    # the first (full) batch is the same in both (the guidance kernel's per-graph means are atomic sums: equal up to their order)
    assert float((by_size[1][:n4] - actual[1][:n4]).abs().max()) <= 1e-6 * max(1.0, float(actual[1].abs().max()))
    gid = torch.tensor([4])
    hp, pp, pn, bp, na, centers = job.batch_inputs(gid)
    for norm, got in ((4, by_size), (1, actual)):
        r = model.sample_batch(hp, pp, pn, bp, na, centers, pos_guidance_opt=opt, rng='device', seed=7, return_traj=False,
                               num_steps=steps, graph_ids=gid, guidance_batch=norm)
        assert torch.equal(r['pred'][0].argmax(-1), got[0][n4:].argmax(-1))
        assert float((r['pred'][1] - got[1][n4:]).abs().max()) <= 1e-6 * max(1.0, float(got[1].abs().max()))
    assert float((by_size[1][n4:] - actual[1][n4:]).abs().max()) > 1e-6
    with pytest.raises(ValueError):
        sample_job_shard(model, job, ids, batch_size=4, guidance_norm='mean')


def test_gather_predictions_over_rccl_one_rank():
    """The path's one collective on the device: a 1-rank `nccl` (= RCCL) group on cuda:0, `gather_predictions` on device tensors (the counts /
    tables / payload collectives run through RCCL on device memory, the re-assembly on the device), graph ids NOT in order and with gaps: the
    result is `pred` in ascending graph id, for the rank-0 form and the every-rank form.  (World size > 1 over gloo: tests/test_parallel_gloo.py.)"""
    import socket
    import torch.distributed as dist
    from phoregen_amd.parallel import gather_predictions
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1)
    try:
        gids = torch.tensor([7, 2, 11, 3])
        na = torch.tensor([5, 9, 3, 6])
        gen = torch.Generator().manual_seed(0)
        N, E = int(na.sum()), int((na * (na - 1)).sum())
        pred = [torch.randn(N, 12, generator=gen).cuda(), torch.randn(N, 3, generator=gen).cuda(), torch.randn(E, 6, generator=gen).cuda()]
        order = torch.argsort(gids)
        n_off = torch.cat([torch.zeros(1, dtype=torch.long), na.cumsum(0)])
        e_off = torch.cat([torch.zeros(1, dtype=torch.long), (na * (na - 1)).cumsum(0)])
        exp = [torch.cat([pred[k][(n_off if k < 2 else e_off)[g]:(n_off if k < 2 else e_off)[g + 1]] for g in order.tolist()]) for k in range(3)]
        for dst in (0, None):
            out, nat = gather_predictions(pred, na, gids, dst=dst)
            torch.cuda.synchronize()
            assert torch.equal(nat, na[order]) and all(o.is_cuda and torch.equal(o, e) for o, e in zip(out, exp))
    finally:
        dist.destroy_process_group()
