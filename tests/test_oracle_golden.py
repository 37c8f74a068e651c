"""The CPU oracle (oracle/phoregen_oracle.py) against vectors recorded from the reference itself."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import DIFF_CFG, golden, make_oracle, oracle_state_dict, rel_err, t
from oracle import phoregen_oracle as po

TOL = 2e-5   # max-abs error <= 2e-5 x max-abs(reference), fp32 re-association


@pytest.fixture(scope='module')
def oracle():
    torch.set_num_threads(4)
    return make_oracle(0)


_ORACLES = {}


def _oracle_for(name):
    """Fixtures named *_<profile> were recorded from the reference with that weight profile (phoregen_amd/weights.py)."""
    prof = next((p for p in ('gamma_signed', 'trained_like') if name.endswith('_' + p)), 'default')
    if prof not in _ORACLES:
        torch.set_num_threads(4)
        _ORACLES[prof] = make_oracle(0, prof)
    return _ORACLES[prof]


def test_g1_small_ops():
    g = golden('g1_ops')
    assert np.array_equal(po.gaussian_smearing(t(g['dist'])).numpy(), g['smear'])
    assert np.allclose(po.time_smearing(t(g['tvals']).float()).numpy(), g['tsmear'], rtol=0, atol=1e-7)
    assert np.allclose(po.angular_encoding(t(g['ang'])).numpy(), g['ang_code'], rtol=0, atol=1e-7)
    assert np.allclose(po.shifted_softplus(t(g['ssp_in'])).numpy(), g['ssp'], rtol=0, atol=1e-7)
    et, sm = t(g['edge_type']).float(), t(g['smear'])
    assert np.array_equal((et.unsqueeze(-1) * sm.unsqueeze(1)).reshape(64, -1).numpy(), g['outer'])


def test_g1_edge_data_and_triplets():
    g = golden('g1_ops')
    ei, eb = po.make_edge_data(t(g['med_num_atoms']))
    assert np.array_equal(ei.numpy(), g['med_edge_index']) and np.array_equal(eb.numpy(), g['med_edge_batch'])
    i, j, k, kj, ji = po.triplets(ei, int(g['med_num_atoms'].sum()))
    for mine, ref in zip((i, j, k, kj, ji), ('tri_i', 'tri_j', 'tri_k', 'tri_kj', 'tri_ji')):
        assert np.array_equal(mine.numpy(), g[ref]), ref


def test_g1_compose_context():
    g = golden('g1_ops')
    h, pos, batch, mask, pidx, lidx = po.compose_context(t(g['cc_hp']), t(g['cc_hl']), t(g['cc_pp']), t(g['cc_pl']),
                                                         t(g['cc_bp']), t(g['cc_bl']))
    for mine, ref in zip((h, pos, batch, mask, pidx, lidx), ('cc_h', 'cc_pos', 'cc_batch', 'cc_mask', 'cc_pidx', 'cc_lidx')):
        assert np.array_equal(mine.numpy(), g[ref]), ref
    same = t(g['cc_bp'])[:, None] == t(g['cc_bp'])[None, :]
    assert np.array_equal(torch.stack(same.nonzero(as_tuple=True)).numpy(), g['fc_index'])


def test_g4_tables_bit_exact():
    g = golden('g4_tables')
    sd = oracle_state_dict(0)
    rows = g['rows']
    n = 0
    for key in g.files:
        if key.endswith('|rows'):
            k = key[:-5]
            assert np.array_equal(sd[k].numpy()[rows], g[key]), k
            s = sd[k].numpy().astype(np.float64)
            assert np.allclose([s.sum(), np.abs(s).sum()], g[k + '|sum64'], rtol=1e-12, atol=0), k
            n += 1
    assert n == 11
    assert np.array_equal(po.categorical_init_prob(12, 'tomask'), g['node_init_prob'])
    assert np.array_equal(po.categorical_init_prob(6, 'absorb'), g['edge_init_prob'])


def test_posterior_kats(oracle):
    g = golden('g_posterior')
    batch, tt = t(g['batch']), t(g['t'])
    for tag, tab in (('node', oracle.tab_node), ('edge', oracle.tab_edge)):
        post = po.q_v_posterior(tab, t(g[f'{tag}_log_v0']), t(g[f'{tag}_log_vt']), tt, batch)
        assert np.allclose(post.numpy(), g[f'{tag}_post'], rtol=0, atol=1e-6), tag
        assert np.array_equal(po.gumbel_argmax(t(g[f'{tag}_post']), t(g[f'{tag}_u'])).numpy(), g[f'{tag}_sample'])
    prev = po.pos_prev_from_recon(oracle.tab_pos, t(g['pos_xt']), t(g['pos_x0']), tt, batch, t(g['pos_eps']))
    assert np.allclose(prev.numpy(), g['pos_prev'], rtol=0, atol=1e-7)


@pytest.mark.parametrize('name', ['g3_forward_a', 'g3_forward_b', 'g3_forward_a_gamma_signed', 'g3_forward_a_trained_like'])
def test_g23_forward_and_layers(name):
    g = golden(name)
    oracle = _oracle_for(name)
    inp = {k[3:]: t(g[k]) for k in g.files if k.startswith('in_')}
    cap = {}
    with torch.no_grad():
        v, x0, bond, (cl, cu) = oracle.forward(**inp, capture=cap)
    assert np.array_equal(cap['edge_index'].numpy(), g['L0_in_edge_index'])       # knn graph, identical order
    assert np.array_equal(cap['edge_type'].numpy(), g['L0_in_edge_attr'])
    assert np.array_equal(cap['bond_index'].numpy(), g['L0_in_bond_index'])
    assert np.array_equal(cap['mask'].numpy(), g['L0_in_mask_ligand'])
    errs = dict(phore=rel_err(cap['phore_enc'], g['phore_enc']), h_all=rel_err(cap['h_all'], g['L0_in_h']),
                e_w=rel_err(cap['e_w'], g['L0_in_e_w']),
                L0_h=rel_err(cap['L0_out'][0], g['L0_out_h']), L0_hb=rel_err(cap['L0_out'][1], g['L0_out_h_bond']),
                L0_x=rel_err(cap['L0_out'][2], g['L0_out_x']),
                L5_h=rel_err(cap['L5_out'][0], g['L5_out_h']), L5_hb=rel_err(cap['L5_out'][1], g['L5_out_h_bond']),
                L5_x=rel_err(cap['L5_out'][2], g['L5_out_x']),
                v=rel_err(v, g['out_v']), x0=rel_err(x0, g['out_x0']), bond=rel_err(bond, g['out_bond']),
                cl=rel_err(cl, g['out_count_l']), cu=rel_err(cu, g['out_count_u']))
    print(name, errs)
    assert max(errs.values()) <= TOL, errs


def _tape(g):
    keys = sorted(k for k in g.files if k.startswith('rng'))
    return [(k.split('_', 1)[1], g[k]) for k in keys]


@pytest.mark.parametrize('name', ['g5_sample_head3', 'g5_sample_tail4', 'g5_sample_full25', 'g5_sample_guid3',
                                  'g5_sample_head3_gamma_signed', 'g5_sample_tail4_gamma_signed',
                                  'g5_sample_head3_trained_like', 'g5_sample_tail4_trained_like'])
def test_g5_sampler(name):
    g = golden(name)
    oracle = _oracle_for(name)
    tape = _tape(g)
    assert tape[0][0] == 'randint'          # sample_from_interval draw (replaced by the forced atom counts)
    draws = [a for _, a in tape[1:]]
    t_total = int(g['t_total'])
    n_atoms = t(g['n_atoms'])
    n_steps = sum(1 for k in g.files if k.endswith('_out_v'))
    guid = [{'type': 'atom_prox', 'min_d': 1.2, 'max_d': 1.9}, {'type': 'center_prox'}] if 'guid' in name else None
    run_T = t_total if t_total != 1000 else 1000
    # head fixtures stop early: run the oracle sampler for the recorded number of steps only
    o = oracle
    if t_total == 1000:
        class Stop(Exception):
            pass
        rng = po.TapeRng(draws + [np.zeros_like(draws[-3]), np.zeros_like(draws[-2]), np.zeros_like(draws[-1])])
        fwd, count = o.forward, [0]
        recs = []

        def limited(*a, **k):
            out = fwd(*a, **k)
            recs.append((a, out))
            count[0] += 1
            if count[0] == n_steps:
                raise Stop()
            return out
        o.forward = limited
        try:
            o.sample(t(g['phore_x']), t(g['phore_pos']), t(g['phore_norm']), t(g['center']), n_atoms, rng)
        except Stop:
            pass
        finally:
            del o.forward
        steps = [(a[0], a[1], a[3], out[0], out[1], out[2]) for a, out in recs]
        res = None
    else:
        res = o.sample(t(g['phore_x']), t(g['phore_pos']), t(g['phore_norm']), t(g['center']), n_atoms,
                       po.TapeRng(draws), t_total=run_T, guidance=guid)
        steps = res['steps']
    assert len(steps) == n_steps
    flips = 0
    for s, (h_node, pos, h_edge, v, x0, bond) in enumerate(steps):
        # discrete state bit-exact, coordinates / logits within tolerance, at every recorded step
        assert np.array_equal(h_node.numpy(), g[f's{s}_h_node']), (name, s)
        assert np.array_equal(h_edge.argmax(-1).numpy(), g[f's{s}_h_edge']), (name, s)
        assert rel_err(pos, g[f's{s}_pos']) <= TOL, (name, s)
        assert rel_err(v, g[f's{s}_out_v']) <= 5 * TOL and rel_err(bond, g[f's{s}_out_bond']) <= 5 * TOL, (name, s)
        assert rel_err(x0, g[f's{s}_out_x0']) <= 5 * TOL, (name, s)
    if res is not None:
        assert np.array_equal(res['traj'][0].argmax(-1).numpy(), g['traj_node'])
        assert np.array_equal(res['traj'][2].argmax(-1).numpy(), g['traj_edge'])
        rmsd = float(np.sqrt(((res['traj'][1].numpy() - g['traj_pos']) ** 2).sum(-1).mean()))
        assert rmsd <= 1e-4, rmsd
        assert rel_err(res['pred'][1], g['pred_pos']) <= 5 * TOL
        assert np.array_equal(res['lig_info'][2].numpy(), g['lig_edge_index'])


@pytest.mark.parametrize('name', ['g5_sample_full1000_a', 'g5_sample_full1000_guid', 'g5_sample_full1000_n34', 'g5_sample_full1000_headline'])
def test_g5_full_length_fixture_windows(name):
    """The 1000-step fixtures of the reference's own `sample()` (oracle/make_golden.py g5_sample_full1000): the draws are re-created by
    seeding torch's CPU generator like the reference run (their float64 sums are in the fixture), the initial state must come out bit
    for bit, and from three of the stored checkpoints (t = 949, 499, 49) the oracle free-runs 10 steps: types bit-exact, coordinates
    <= 1e-5 A from the reference's trajectory.  (The whole 1000-step run of the oracle -- 5 min of CPU -- equals the reference's bit
    for bit without guidance: 0 type mismatches, RMSD 0.0, profiles/r05_oracle_vs_reference_full1000.txt.)"""
    import torch.nn.functional as F
    g = golden(name)
    o = _oracle_for(name)
    guid = [{'type': 'atom_prox', 'min_d': 1.2, 'max_d': 1.9}, {'type': 'center_prox'}] if 'guid' in name else None
    na = t(g['n_atoms'])
    B, p, T = len(na), g['phore_x'].shape[0], 1000
    bn = torch.repeat_interleave(torch.arange(B), na)
    ei, be = po.make_edge_data(na)
    N, E = bn.numel(), be.numel()
    center = t(g['center'])
    state = torch.get_rng_state()
    try:
        torch.manual_seed(int(g['sample_seed']))
        pos0 = torch.randn(N, 3) - center                                               # Appendix B items 3-4
        un0, ue0 = torch.rand(N, 12, dtype=torch.float64), torch.rand(E, 6, dtype=torch.float64)
        draws = [(torch.rand(N, 12), torch.rand(E, 6), torch.randn(N, 3)) for _ in range(T)]
    finally:
        torch.set_rng_state(state)
    assert all(float(d[0].double().sum()) == g['u_node_sum'][i] and float(d[1].double().sum()) == g['u_edge_sum'][i]
               for i, d in enumerate(draws)), 'torch CPU generator stream differs from the one the fixture was recorded with'
    assert np.abs(np.array([float(d[2].double().sum()) for d in draws]) - g['eps_sum']).max() < 1e-4
    assert np.allclose(pos0.numpy(), g['pos_init'], rtol=0, atol=2e-6) and np.array_equal(g['traj_pos'][0], g['pos_init'])   # (randn: last ulp may differ between CPU kinds)
    log_pn = torch.log(torch.from_numpy(o.tab_node['init_prob']) + 1e-30).clamp_min(-32.)
    log_pe = torch.log(torch.from_numpy(o.tab_edge['init_prob']) + 1e-30).clamp_min(-32.)
    assert np.array_equal(po.gumbel_argmax(log_pn.unsqueeze(0).repeat(N, 1), un0).numpy(), g['traj_node'][0])
    assert np.array_equal(po.gumbel_argmax(log_pe.unsqueeze(0).repeat(E, 1), ue0).numpy(), g['traj_edge'][0])
    hp, pp, pn = t(g['phore_x']).repeat(B, 1), t(g['phore_pos']).repeat(B, 1), t(g['phore_norm']).repeat(B, 1)
    bp = torch.repeat_interleave(torch.arange(B), p)
    pc = t(g['phore_pos'])[t(g['phore_x'])[:, 12] != 1].mean(0)
    ck = list(g['ck_steps'])
    worst = 0.0
    with torch.no_grad():
        # (`n34` -- ligands of 34 / 21 atoms -- stores no checkpoints: its window is the first 6 steps from the initial state; `headline` -- 38 / 40 /
        #  43 / 52 atoms on a 107-node pharmacophore, round 6 -- the first 3: ~10 s of CPU per step)
        for after in ((49, 499, 949) if ck else (-1,)):
            if after < 0:
                pos = pos0
                h_node = F.one_hot(t(g['traj_node'][0]).long(), 12).float()
                h_edge = F.one_hot(t(g['traj_edge'][0]).long(), 6).float()
                log_node, log_edge = torch.log(h_node.clamp(min=1e-30)), torch.log(h_edge.clamp(min=1e-30))
            else:
                k = ck.index(after)
                pos, log_node, log_edge = t(g['ck_pos'][k]), t(g['ck_log_node'][k]), t(g['ck_log_edge'][k])
                h_node = F.one_hot(t(g['traj_node'][after + 1]).long(), 12).float()
                h_edge = F.one_hot(t(g['traj_edge'][after + 1]).long(), 6).float()
            for i in range(after + 1, after + (11 if ck else (4 if 'headline' in name else 7))):
                tt = torch.full((B,), T - 1 - i)
                v, x0, bond, _ = o.forward(h_node, pos, bn, h_edge, ei, be, tt, hp, pp, pn, bp)
                un, ue, eps = draws[i]
                log_node = po.q_v_posterior(o.tab_node, F.log_softmax(v, -1), log_node, tt, bn)
                h_node = F.one_hot(po.gumbel_argmax(log_node, un), 12).float()
                log_edge = po.q_v_posterior(o.tab_edge, F.log_softmax(bond, -1), log_edge, tt, be)
                h_edge = F.one_hot(po.gumbel_argmax(log_edge, ue), 6).float()
                grad = po.guidance_grad(guid, pos, bn, h_edge, ei, be, B, pc) if guid else 0.
                pos = po.pos_prev_from_recon(o.tab_pos, pos, x0, tt, bn, eps, grad)
                assert np.array_equal(h_node.argmax(-1).numpy(), g['traj_node'][i + 1]), (name, i)
                assert np.array_equal(h_edge.argmax(-1).numpy(), g['traj_edge'][i + 1]), (name, i)
                worst = max(worst, float((pos + center - t(g['traj_pos'][i + 1])).norm(dim=-1).max()))
    assert worst <= 1e-5, worst


def _train_batch(g):
    keys = ('ligand_x', 'ligand_pos', 'ligand_batch', 'ligand_ptr', 'f_edge_index', 'f_edge_attr', 'f_edge_batch',
            'phore_x', 'phore_pos', 'phore_norm', 'phore_batch') + (('edge_index',) if 'edge_index' in g.files else ())
    return {k: t(g[k]) for k in keys}


@pytest.mark.parametrize('name', ['g6_loss_a', 'g6_loss_b', 'g6_loss_a_gamma_signed', 'g6_loss_a_trained_like', 'g6_loss_len'])
def test_compute_loss_and_gradients_match_reference(name):
    """G6: loss terms and every parameter-gradient norm of the reference's compute_loss + backward
    (diffusion.py:249-352), replaying its recorded draws through the oracle's autograd."""
    from oracle import phoregen_oracle as po
    g = golden(name)
    orc = make_oracle(0, next((p for p in ('gamma_signed', 'trained_like') if name.endswith('_' + p)), 'default'))
    names = [str(k) for k in g['param_names']]
    for k in names:
        orc.sd[k].requires_grad_(True)
    rng = po.TrainTapeRng(t(g['time_draw']), t(g['pos_noise']), t(g['u_node']), t(g['u_edge']))
    loss, info = orc.compute_loss(_train_batch(g), rng, bond_len_loss='edge_index' in g.files)   # g6_loss_len: config flag on
    ref = dict(zip([str(k) for k in g['info_keys']], g['info_vals']))
    assert set(info) == set(ref)
    assert abs(float(loss) - float(g['loss'])) <= 1e-5 * abs(float(g['loss']))
    for k, v in ref.items():
        assert abs(info[k] - v) <= 2e-5 * max(abs(v), 1e-3), (k, info[k], v)
    loss.backward()
    gn = np.array([float(orc.sd[k].grad.norm()) if orc.sd[k].grad is not None else 0.0 for k in names])
    big = g['grad_norm'] > 1e-4 * g['grad_norm'].max()
    assert np.abs(gn - g['grad_norm'])[big].max() / g['grad_norm'][big].max() < 1e-4
    assert (np.abs(gn - g['grad_norm'])[big] / g['grad_norm'][big]).max() < 2e-3
    for key in g.files:
        if key.startswith('grad::'):
            assert rel_err(orc.sd[key[6:]].grad, g[key]) < 2e-3, key
