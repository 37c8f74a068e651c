"""-m gpu: the drop-in boundary as the reference's callers use it: the `sample_all.py:79-116` sequence
(`sample()` with its own atom-count draw, `.cpu()`, `unbatch_data`, `decode_data`), the callable pharmacophore encoder,
per-graph centres in the trajectory, and the out-of-memory message contract (`sample_all.py:95-99`)."""
import os

import numpy as np
import pytest
import torch

from helpers import golden, make_oracle, rel_err, t

pytestmark = pytest.mark.gpu
DEV = 'cuda'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def model():
    from phoregen_amd.config import default_model_config
    from phoregen_amd.models.diffusion import PhoreDiff
    from phoregen_amd.weights import init_deterministic_
    return init_deterministic_(PhoreDiff(default_model_config(), 'zinc_300'), 0).eval().to(DEV)


def test_sample_all_caller_sequence(model):
    """sample_all.py:79-116 on tests/data/synthetic_test.phore: `model.sample(data, n_graphs, device, ...)` WITHOUT
    num_atoms (sample_nodes + sample_from_interval draw, diffusion.py:356-387), everything `.cpu()`, unbatch_data ->
    decode_data per graph; and decode_batch (device argmax + one copy) must give exactly those molecules."""
    from phoregen_amd.data import parse_phore_file
    from phoregen_amd.utils.sample_utils import decode_batch, decode_data, unbatch_data
    data = parse_phore_file(os.path.join(ROOT, 'tests', 'data', 'synthetic_test.phore')).to(DEV)
    n_graphs = 5
    # the count heads against the oracle: the interval sample_nodes draws from
    o = make_oracle(0)
    ph = data['phore']
    p = ph.x.size(0)
    from oracle import phoregen_oracle as po
    import torch.nn.functional as F
    ei, be = po.make_edge_data(torch.tensor([2]))
    with torch.no_grad():      # the count heads see only the pharmacophore (diffusion.py:148-163): any ligand will do
        _, _, _, (cl, cu) = o.forward(F.one_hot(torch.tensor([0, 1]), 12).float(), torch.zeros(2, 3), torch.zeros(2, dtype=torch.long),
                                      F.one_hot(torch.tensor([0, 0]), 6).float(), ei, be, torch.tensor([500]), ph.x.cpu(),
                                      ph.pos.cpu(), ph.norm.cpu(), torch.zeros(p, dtype=torch.long))
    torch.manual_seed(2032)                                            # seed_all(args.seed), sample_all.py:33
    na_draw = model.sample_nodes(data, 64, DEV)
    eng = model._count_engine
    lo = int((eng.ws.count_l * 74 + 4).round().item())
    hi = int((eng.ws.count_u * 74 + 4).round().item())
    assert 4 <= lo <= hi <= 78 and int(na_draw.min()) >= lo and int(na_draw.max()) <= hi
    assert model.sample_nodes(data, 3, DEV) is not None and model._count_engine is eng          # one engine, reused
    assert abs(float(cl) - float(eng.ws.count_l)) <= 2e-5 and abs(float(cu) - float(eng.ws.count_u)) <= 2e-5
    assert lo == int((cl * 74 + 4).round().item()) and hi == int((cu * 74 + 4).round().item())
    torch.manual_seed(2032)
    results = model.sample(data, n_graphs, DEV, pos_guidance_opt=None, sample_mode='uniform', normal_scale=4.0, num_steps=12)
    assert set(results) == {'pred', 'traj', 'lig_info'}
    na = results['lig_info'][0]
    assert na.numel() == n_graphs and int(na.min()) >= lo and int(na.max()) <= hi
    dev_mols = decode_batch(results, include_bond=True)                 # on the device tensors
    results = {key: [v.cpu() for v in value if v is not None] for key, value in results.items()}     # sample_all.py:102
    outs = unbatch_data(results, n_graphs, include_bond=True)
    assert len(outs) == n_graphs
    N = int(na.sum())
    assert results['traj'][1].shape == (13, N, 3) and results['pred'][0].shape == (N, 12)
    for o_g, n, dm in zip(outs, na.tolist(), dev_mols):
        assert o_g['pred'][0].shape == (n, 12) and o_g['edge_index'].shape == (2, n * (n - 1))
        assert int(o_g['edge_index'].min()) == 0 and int(o_g['edge_index'].max()) == n - 1
        mol = decode_data(pred_info=o_g['pred'], edge_index=o_g['edge_index'], include_bond=True)   # sample_all.py:109-113
        assert mol['element'] == dm['element']
        assert torch.equal(mol['atom_pos'], dm['atom_pos'])
        assert torch.equal(mol['bond_type'], dm['bond_type']) and torch.equal(mol['bond_index'], dm['bond_index'])
        assert np.isfinite(mol['atom_pos'].numpy()).all()


def test_phore_encoder_is_callable_like_the_reference(model):
    """diffusion.py:185-191: `self.phore_encoder(h_phore_emb, dist_feat, f_edge_index_p)` with the edges of
    fully_connect_two_graphs; result against the reference's recorded encoder output (G3 fixtures)."""
    for name in ('g3_forward_a', 'g3_forward_b'):
        g = golden(name)
        h_phore, pos, bp = (t(g[k]).to(DEV) for k in ('in_h_phore', 'in_pos_phore', 'in_batch_phore'))
        with torch.no_grad():
            h_emb = model.phore_embedding(h_phore)
            same = bp[:, None] == bp[None, :]
            src, dst = same.nonzero(as_tuple=True)                     # fully_connect_two_graphs(batch, batch)
            dist = torch.norm(pos[dst] - pos[src], p=2, dim=-1, keepdim=True)
            out = model.phore_encoder(h_emb, dist, torch.stack([src, dst]))
        assert out.shape == h_emb.shape
        assert rel_err(out.cpu(), g['phore_enc']) <= 2e-5, name
    with pytest.raises(NotImplementedError):
        model.denoiser.base_block[0].node_layer_with_bond(h_emb, dist, torch.stack([src, dst]))


@pytest.mark.parametrize('name', ['g3_forward_a', 'g3_forward_b', 'g3_forward_a_gamma_signed', 'g3_forward_a_trained_like'])
def test_denoiser_module_forward_is_callable_like_the_reference(name):
    """uni_denoiser.py:396-430 through the plugin seam `models.get_denoiser_net(cfg)(h, x, group_idx, bond_index, h_bond,
    mask_ligand, mask_ligand_atom, batch, phore_norm)` (SURVEY.md 8(b) "Denoiser module"): the reference's recorded inputs of
    layer 0 in, its recorded outputs of layer 5 out (G3 fixtures, four weight profiles)."""
    from phoregen_amd.config import default_model_config
    from phoregen_amd.models.diffusion import PhoreDiff
    from phoregen_amd.weights import init_deterministic_
    prof = next((p for p in ('gamma_signed', 'trained_like') if name.endswith(p)), 'default')
    m = init_deterministic_(PhoreDiff(default_model_config(), 'zinc_300'), 0, profile=prof).eval().to(DEV)
    g = golden(name)
    h, x, bi, hb, mask, batch, pn = (t(g['L0_in_' + k]).to(DEV) for k in ('h', 'x', 'bond_index', 'h_bond', 'mask_ligand', 'batch',
                                                                          'phore_norm'))
    with torch.no_grad():
        out = m.denoiser(h, x, None, bi, hb, mask, mask, batch, phore_norm=pn, return_all=True)
    assert set(out) == {'x', 'h', 'h_bond', 'all_x', 'all_h', 'all_h_bond'}
    errs = {k: rel_err(out[k].cpu(), g['L5_out_' + k]) for k in ('h', 'h_bond', 'x')}
    assert max(errs.values()) <= 2e-5, (name, errs)
    assert torch.equal(out['all_h'][0], h) and torch.equal(out['all_x'][-1], out['x'])
    # pharmacophore rows do not move (uni_denoiser.py:295-296)
    assert torch.equal(out['x'][~mask], x[~mask])
    # the reference grows the edge-type one-hot by two columns for a group index (:386-393): not silently ignored
    with pytest.raises(NotImplementedError):
        m.denoiser(h, x, torch.zeros_like(batch), bi, hb, mask, mask, batch, phore_norm=pn)
    with pytest.raises(NotImplementedError):
        m.denoiser(h, x, None, bi, hb, mask, ~mask, batch, phore_norm=pn)


def test_sample_nodes_normal_mode(model):
    """diffusion.py:356-387 with sample_mode='normal' (sample_utils.py:28-37: clamp(round(N(mid, std)), lo, hi))."""
    from phoregen_amd.data import parse_phore_file
    data = parse_phore_file(os.path.join(ROOT, 'tests', 'data', 'synthetic_test.phore')).to(DEV)
    torch.manual_seed(7)
    na = model.sample_nodes(data, 256, DEV, sample_mode='normal', normal_scale=4.0)
    eng = model._count_engine
    lo = int((eng.ws.count_l * 74 + 4).round().item())
    hi = int((eng.ws.count_u * 74 + 4).round().item())
    assert na.shape == (256,) and na.dtype == torch.int32 and int(na.min()) >= lo and int(na.max()) <= hi
    # same generator state -> the reference's own draw: torch.normal(mid, std, (B,)) clamped and rounded
    torch.manual_seed(7)
    mid, std = (lo + hi) / 2, (hi - lo) / 4.0
    ref = torch.normal(mid, std, (256,)).clamp(lo, hi).round().int()
    assert torch.equal(na.cpu(), ref)
    res = model.sample(data, 3, DEV, sample_mode='normal', normal_scale=4.0, num_steps=2)
    assert int(res['lig_info'][0].min()) >= lo and int(res['lig_info'][0].max()) <= hi


def test_trajectory_with_per_graph_centres(model):
    """Multi-pharmacophore batches (f-2) return trajectories: frame k of graph g = its ligand-frame state + centre g."""
    from oracle.make_inputs import synthetic_phore
    gen = torch.Generator().manual_seed(12)
    phs = [synthetic_phore(gen, p) for p in (20, 31, 25)]
    na = torch.tensor([7, 10, 6])
    bp = torch.cat([torch.full((x.size(0),), i) for i, (x, _, _) in enumerate(phs)])
    centers = torch.tensor([[1.0, -2.0, 3.0], [10.0, 0.5, -4.0], [-6.0, 6.0, 0.0]])
    st = model.begin_sampling(torch.cat([x for x, _, _ in phs]), torch.cat([p for _, p, _ in phs]), torch.cat([n for _, _, n in phs]),
                              bp, na, centers, rng='device', seed=5, return_traj=True, num_steps=4)
    states = []
    for i, step in enumerate((999, 998, 997, 996)):
        model.reverse_step(st, i, step)
        states.append(st.eng.ws.in_pos.clone())
    res = model.finish_sampling(st)
    rows = centers.to(DEV)[st.plan.batch_node]
    for k, x in enumerate(states):
        assert torch.allclose(res['traj'][1][k + 1], x + rows, atol=1e-6)
    assert torch.allclose(res['pred'][1], st.x0 + rows, atol=1e-6)
    assert torch.equal(res['traj'][1][0], st.pos_traj[0])                # frame 0: the initial state (no centre, diffusion.py:424-426)


def test_out_of_memory_surfaces_with_the_reference_message_contract(model):
    """sample_all.py:95-99 / run/run.py:144-151 branch on `'out of memory' in str(e)`: an allocation failure inside
    `model.sample` must be an exception carrying that text, and the model must keep working afterwards."""
    from oracle.make_inputs import synthetic_phore
    from phoregen_amd.data import PhoreGraph
    x, pos, norm = synthetic_phore(torch.Generator().manual_seed(3), 40)
    data = PhoreGraph(x, pos, norm, torch.zeros(3)).to(DEV)
    import gc
    gc.collect()                 # engines of earlier tests sit in reference cycles (launch-list closures): without this their
    torch.cuda.empty_cache()     # workspaces would be collected DURING sample() and satisfy its allocations from the cache
    free, _total = torch.cuda.mem_get_info()
    # leave 512 MB: room for what the HIP runtime itself allocates on first use of a new stream (kernel scratch, signals -- it
    # ABORTS the process when those fail), but not for the workspace of the batch below
    hog = torch.empty(max(free - (512 << 20), 0), dtype=torch.uint8, device=DEV)
    try:
        with pytest.raises(Exception) as ei:
            model.sample(data, 96, DEV, num_atoms=torch.full((96,), 60), num_steps=2)   # needs ~2 GB of workspace
        assert 'out of memory' in str(ei.value), str(ei.value)[:300]
    finally:
        del hog
        torch.cuda.empty_cache()
    res = model.sample(data, 2, DEV, num_atoms=torch.tensor([8, 9]), num_steps=2)
    assert torch.isfinite(res['pred'][1]).all()


def test_cpu_rng_mode_consumes_the_generator_exactly_like_the_reference(model):
    """rng='cpu': `torch.manual_seed(s); model.sample(...)` must draw from torch's CPU generator what the reference's sampler
    draws, in its order and NOTHING else -- checked against the oracle's sampler under the same seed: identical
    initial state, types bit-exact over free-running steps from t = 999."""
    from oracle import phoregen_oracle as po
    from oracle.make_inputs import synthetic_phore
    from phoregen_amd.data import PhoreGraph
    x, pos, nrm = synthetic_phore(torch.Generator().manual_seed(21), 30)
    center = torch.tensor([1.0, -2.0, 0.5])
    na = torch.tensor([7, 11, 9])
    S = 6
    torch.manual_seed(77)
    with torch.no_grad():
        ref = make_oracle(0).sample(x, pos, nrm, center, na, po.TorchCpuRng(), n_steps=S)
    torch.manual_seed(77)
    res = model.sample(PhoreGraph(x, pos, nrm, center).to(DEV), 3, DEV, rng='cpu', num_atoms=na, num_steps=S)
    for s in range(S + 1):
        assert torch.equal(res['traj'][0][s].cpu().argmax(-1), ref['traj'][0][s].argmax(-1)), s
        assert torch.equal(res['traj'][2][s].cpu().argmax(-1), ref['traj'][2][s].argmax(-1)), s
        rmsd = float(((res['traj'][1][s].cpu() - ref['traj'][1][s]) ** 2).sum(-1).mean().sqrt())
        assert rmsd <= 1e-4, (s, rmsd)


def test_config2_full_size_sample_call(model):
    """BASELINE.json configs[1] at its real size through the reference's entry point: `model.sample(data, 100, device)` on ONE pharmacophore
    (the 44-node P03211 shape recorded in g8_phore_parse), all 1000 reverse steps, guidance on as in sample.sh.  Contract of the result
    (diffusion.py:505-525): shapes, finite coordinates, one-hot discrete trajectories, atom counts honoured; same seed -> same sample;
    the first graphs equal the same graphs sampled in a smaller call (noise is keyed by graph id, not by the batch)."""
    from phoregen_amd.data import PhoreGraph
    g = golden('g8_phore_parse')
    data = PhoreGraph(t(g['x']), t(g['pos']), t(g['norm']), t(g['center'])).to(DEV)
    gen = torch.Generator().manual_seed(2032)
    na = torch.randint(20, 45, (100,), generator=gen)
    guid = [{'type': 'atom_prox', 'min_d': 1.2, 'max_d': 1.9}, {'type': 'center_prox'}]
    res = model.sample(data, 100, DEV, pos_guidance_opt=guid, num_atoms=na, seed=77, return_traj=True)
    N, E = int(na.sum()), int((na * (na - 1)).sum())
    tn, tp, te = res['traj']
    assert tn.shape == (1001, N, 12) and tp.shape == (1001, N, 3) and te.shape == (1001, E, 6)
    assert torch.isfinite(tp).all() and all(torch.isfinite(x).all() for x in res['pred'])
    assert tn.sum(-1).eq(1).all() and te.sum(-1).eq(1).all()
    assert torch.equal(res['lig_info'][0].cpu(), na)
    again = model.sample(data, 100, DEV, pos_guidance_opt=guid, num_atoms=na, seed=77, return_traj=False)
    assert torch.equal(again['pred'][0], res['pred'][0]) and torch.equal(again['pred'][1], res['pred'][1])
    # the first 10 graphs in a 10-graph call: same types, coordinates to 1e-6 of the coordinate scale (guidance energies are means over
    # the call's graphs, so the smaller call passes the full batch size on)
    n10, e10 = int(na[:10].sum()), int((na[:10] * (na[:10] - 1)).sum())
    p = g['x'].shape[0]
    small = model.sample_batch(t(g['x']).repeat(10, 1), t(g['pos']).repeat(10, 1), t(g['norm']).repeat(10, 1),
                               torch.repeat_interleave(torch.arange(10), p), na[:10], t(g['center']).unsqueeze(0).expand(10, 3),
                               pos_guidance_opt=guid, rng='device', seed=77, return_traj=False, guidance_batch=100,
                               guidance_center=t(g['pos'])[t(g['x'])[:, 12] != 1].mean(0))
    assert torch.equal(small['pred'][0].argmax(-1), res['pred'][0][:n10].argmax(-1))
    assert torch.equal(small['pred'][2].argmax(-1), res['pred'][2][:e10].argmax(-1))
    scale = max(1.0, float(res['pred'][1].abs().max()))
    assert float((small['pred'][1] - res['pred'][1][:n10]).abs().max()) <= 1e-6 * scale


def test_engines_of_consecutive_batches_are_freed_by_reference_counting(model):
    """A config-4 shard builds a new BatchPlan + Engine (workspace, launch lists, programs inside the library) per batch
    (sample_all.py:69-183 serves one pharmacophore after the other).  With the cyclic collector OFF, twelve batches of different plans
    must leave `torch.cuda.memory_allocated()` where it was after the first: an Engine is released when the next one replaces it, not
    whenever the collector happens to run (round 4: the launch-list closures held every Engine in a reference cycle)."""
    import gc
    import weakref
    from bench import config4_job
    from phoregen_amd.parallel import sample_job_shard
    job = config4_job(n_phores=12, samples=6, seed=99)
    gc.collect()
    gc.disable()
    try:
        levels, engines = [], []
        for b in range(12):
            ids = torch.arange(b * 6, b * 6 + 6)
            pred = sample_job_shard(model, job, ids, batch_size=6, seed=b, num_steps=2)
            engines.append(weakref.ref(model._engine))
            del pred
            torch.cuda.synchronize()
            levels.append(torch.cuda.memory_allocated())
        assert all(e() is None for e in engines[:-1]), [e() is None for e in engines]      # only the current engine is alive
        # the batches differ in size (p ~ N(80, 25), n ~ N(40, 6)): the current engine's own workspace is all that varies; a leak would
        # add one workspace per batch (12 x by the end)
        assert max(levels) < 2.0 * levels[0] and levels[-1] < 2.0 * min(levels), levels
        model._engine = model._plan = None
        torch.cuda.synchronize()
        assert torch.cuda.memory_allocated() < min(levels), (torch.cuda.memory_allocated(), levels)
    finally:
        gc.enable()
