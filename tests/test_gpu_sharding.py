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
    """What gather_predictions does after the all_gather: per-rank results back into global graph order."""
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
    shards = partition_graphs(job.num_atoms, 8)
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
        mine = partition_graphs(job.num_atoms, 4)[rank]
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
