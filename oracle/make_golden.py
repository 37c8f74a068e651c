#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own Python (/root/reference, read-only).

TEST INFRASTRUCTURE, build container only.  The reference is imported unmodified (see
oracle/ref_import.py + oracle/standins/README.md for the absent third-party wheels), loaded with
the deterministic synthetic weights of phoregen_amd/weights.py (there are no checkpoints offline)
and driven on small seeded inputs.  Only *data* is written: inputs, recorded RNG draws, expected
outputs.  Re-run with:  python oracle/make_golden.py            (everything; the full-length fixtures alone: `full1000`, `full1000_n34`,
`full1000_headline` -- the last one is ~3 h of CPU at the script's 4 threads: the reference's own 1000-step run at the bench's shape)
"""
import os
import random
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_import  # noqa: E402

ref_import.activate()
OUT = os.path.join(ROOT, 'tests', 'golden')
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(4)

from phoregen_amd.weights import init_deterministic_  # noqa: E402

import models.common as rc  # noqa: E402  (reference)
import models.transition as rt  # noqa: E402
import models.uni_denoiser as ru  # noqa: E402
from models.diffusion import PhoreDiff  # noqa: E402
from utils.sample_utils import make_edge_data, get_fully_connected_edge  # noqa: E402
from datasets.get_phore_data import PhoreData_New  # noqa: E402
from datasets.transform import AddPhoreNoise, FeaturizeLigandBond  # noqa: E402
from torch_geometric.transforms import Compose  # noqa: E402


def seed_all(seed):  # utils/misc.py:29-32
    torch.manual_seed(seed)
    np.random.seed(seed)
    random.seed(seed)


def npy(t):
    if torch.is_tensor(t):
        return t.detach().cpu().numpy()
    return np.asarray(t)


def save(name, **arrays):
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **{k: npy(v) for k, v in arrays.items()})
    print(f'{name}: {os.path.getsize(path) / 1024:.1f} KiB, {len(arrays)} arrays')


# --------------------------------------------------------------------------------------------
# G1  op-level known-answer vectors
# --------------------------------------------------------------------------------------------
def g1_ops():
    g = torch.Generator().manual_seed(11)
    dist = torch.cat([torch.linspace(0, 12, 40), torch.rand(24, generator=g) * 4])
    smear = rc.GaussianSmearing(0., 10., num_gaussians=20)(dist)
    tvals = torch.tensor([0, 1, 2, 57, 111, 500, 998, 999], dtype=torch.long)
    tsmear = rc.TimeGaussianSmearing(stop=1000, num_gaussians=10, type_='linear')(tvals)
    ang = torch.cat([torch.linspace(0, np.pi, 17), torch.rand(15, generator=g) * np.pi]).float()
    ang_code = rc.AngularEncoding()(ang)
    et = torch.nn.functional.one_hot(torch.randint(0, 4, (64,), generator=g), 4)
    outer = rc.outer_product(et, smear)
    ssp_in = torch.linspace(-6, 6, 25)
    ssp = rc.ShiftedSoftplus()(ssp_in)
    arrays = dict(dist=dist, smear=smear, tvals=tvals, tsmear=tsmear, ang=ang, ang_code=ang_code,
                  edge_type=et, outer=outer, ssp_in=ssp_in, ssp=ssp)
    # make_edge_data + triplets, n = 3, 4, 5 in one batch (utils/sample_utils.py:40-54, uni_denoiser.py:101-121)
    na = torch.tensor([3, 4, 5])
    ei, eb = make_edge_data(na)
    col, row, idx_i, idx_j, idx_k, idx_kj, idx_ji = ru.BondUpdateLayer.triplets(ei, num_nodes=int(na.sum()))
    arrays.update(med_num_atoms=na, med_edge_index=ei, med_edge_batch=eb, tri_i=idx_i, tri_j=idx_j,
                  tri_k=idx_k, tri_kj=idx_kj, tri_ji=idx_ji)
    # compose_context (common.py:180-208)
    bp = torch.tensor([0, 0, 0, 1, 1, 2, 2, 2, 2])
    bl = torch.tensor([0, 0, 1, 1, 1, 2])
    hp = torch.randn(9, 4, generator=g)
    hl = torch.randn(6, 4, generator=g)
    pp = torch.randn(9, 3, generator=g)
    pl = torch.randn(6, 3, generator=g)
    h_ctx, pos_ctx, batch_ctx, mask_l, mask_la, p_idx, l_idx = rc.compose_context(hp, hl, pp, pl, bp, bl)
    arrays.update(cc_bp=bp, cc_bl=bl, cc_hp=hp, cc_hl=hl, cc_pp=pp, cc_pl=pl, cc_h=h_ctx, cc_pos=pos_ctx,
                  cc_batch=batch_ctx, cc_mask=mask_l, cc_pidx=p_idx, cc_lidx=l_idx)
    arrays.update(fc_index=rc.fully_connect_two_graphs(bp, bp), fce_index=get_fully_connected_edge(4))
    save('g1_ops', **arrays)


# --------------------------------------------------------------------------------------------
# shared: model + synthetic / real pharmacophores
# --------------------------------------------------------------------------------------------
def build_model(seed=0, profile='default'):
    cfg = ref_import.load_reference_config('train_lig-phore.yml')
    m = PhoreDiff(cfg.model, cfg.dataset.data_name)
    init_deterministic_(m, seed, profile=profile)
    m.eval()
    return m, cfg


from oracle.make_inputs import synthetic_batch, synthetic_phore  # noqa: E402  (shared with tests / bench)


# --------------------------------------------------------------------------------------------
# G2 + G3  one denoiser layer / full PhoreDiff.forward
# --------------------------------------------------------------------------------------------
def g23_forward(model, name, seed, n_atoms, n_phore, t_values):
    inp = synthetic_batch(seed, n_atoms, n_phore, t_values)
    cap = {}

    def layer_hook(idx):
        def hook(mod, args, kwargs, out):
            names = ['h', 'x', 'edge_attr', 'edge_index', 'h_bond', 'bond_index', 'mask_ligand']
            d = {f'L{idx}_in_{n}': a for n, a in zip(names, args)}
            d.update({f'L{idx}_in_{k}': v for k, v in kwargs.items()})
            d.update({f'L{idx}_out_h': out[0], f'L{idx}_out_h_bond': out[1], f'L{idx}_out_x': out[2]})
            cap.update(d)
        return hook

    def sub_hook(tag):
        def hook(mod, args, out):
            cap[tag] = out
        return hook

    hs = [model.denoiser.base_block[i].register_forward_hook(layer_hook(i), with_kwargs=True) for i in (0, 5)]
    blk = model.denoiser.base_block[0]
    hs += [blk.node_layer_with_edge.register_forward_hook(sub_hook('L0_node_edge')),
           blk.node_layer_with_bond.register_forward_hook(sub_hook('L0_node_bond')),
           blk.bond_layer.register_forward_hook(sub_hook('L0_bond_upd')),
           blk.pos_layer_with_edge.register_forward_hook(sub_hook('L0_pos_edge')),
           blk.pos_layer_with_bond.register_forward_hook(sub_hook('L0_pos_bond')),
           model.phore_encoder.register_forward_hook(sub_hook('phore_enc'))]
    with torch.no_grad():
        v, x0, bond, (cl, cu) = model(**inp)
    for h in hs:
        h.remove()
    arrays = {f'in_{k}': t for k, t in inp.items()}
    arrays.update(out_v=v, out_x0=x0, out_bond=bond, out_count_l=cl, out_count_u=cu)
    arrays.update(cap)
    save(name, **arrays)


# --------------------------------------------------------------------------------------------
# G4  schedule / transition tables (diffusion.py:89-135, transition.py:14-26,200-215)
# --------------------------------------------------------------------------------------------
def g4_tables(model):
    rows = np.r_[0:5, 498:503, 995:1000]
    arrays = dict(rows=rows)
    sd = model.state_dict()
    for k, v in sd.items():
        if k.startswith(('pos_transition.', 'node_transition.', 'edge_transition.')):
            a = npy(v)
            arrays[k + '|rows'] = a[rows]
            arrays[k + '|sum64'] = np.array([a.astype(np.float64).sum(), np.abs(a.astype(np.float64)).sum()])
    arrays['node_init_prob'] = model.node_transition.init_prob
    arrays['edge_init_prob'] = model.edge_transition.init_prob
    save('g4_tables', **arrays)


# --------------------------------------------------------------------------------------------
# G5  sampler trajectories with the RNG draws recorded
# --------------------------------------------------------------------------------------------
class RngTape:
    """Record every CPU draw the sampler makes, in order."""
    NAMES = ('rand_like', 'randn_like', 'randn', 'randint', 'normal', 'rand')

    def __init__(self):
        self.tape, self._orig = [], {}

    def __enter__(self):
        for n in self.NAMES:
            self._orig[n] = getattr(torch, n)
            setattr(torch, n, self._wrap(n))
        return self

    def _wrap(self, n):
        def f(*a, **k):
            out = self._orig[n](*a, **k)
            self.tape.append((n, out.detach().clone()))
            return out
        return f

    def __exit__(self, *exc):
        for n, f in self._orig.items():
            setattr(torch, n, f)


class StopAfter(Exception):
    pass


def real_phore_data(seed):
    seed_all(seed)
    tr = Compose([FeaturizeLigandBond(), AddPhoreNoise(noise_std=0.1, angle=5.0)])  # training_utils.py:86-91
    ds = PhoreData_New([os.path.join(ref_import.REFERENCE, 'data/phores_for_sampling/P03211_merge.phore')],
                       transform=tr)
    return ds[0]


def g5_sample(model, name, seed, n_atoms, n_steps, t_total, guidance=None):
    """Run the reference sampler. `t_total`=1000 -> first `n_steps` steps from t=999 (stopped by a
    hook after the forward of step n_steps+1, so posteriors of n_steps steps are observed);
    `t_total`<1000 -> `model.num_timesteps` is set so the loop runs t = t_total-1 .. 0 to the end."""
    data = real_phore_data(seed)
    fwd_in, fwd_out = [], []

    ref_forward = model.forward        # sample() calls self.forward directly (diffusion.py:436): wrap it

    def recording_forward(**kwargs):
        out = ref_forward(**kwargs)
        fwd_in.append({k: v.detach().clone() for k, v in kwargs.items()})
        fwd_out.append([out[0].detach().clone(), out[1].detach().clone(), out[2].detach().clone()])
        if t_total == 1000 and len(fwd_in) == n_steps + 1:
            raise StopAfter()
        return out

    # n forced small: the reference's atom-count head runs, its draw is replaced
    orig_sample_nodes = model.sample_nodes
    counts = {}

    def forced_nodes(data_, batch_size, device, sample_mode='uniform', normal_scale=4.0):
        counts['model'] = orig_sample_nodes(data_, batch_size, device, sample_mode, normal_scale)
        return torch.tensor(n_atoms)

    model.sample_nodes = forced_nodes
    model.forward = recording_forward
    old_T = model.num_timesteps
    res = None
    with RngTape() as tape:
        try:
            if t_total != 1000:
                model.num_timesteps = t_total
            res = model.sample(data, len(n_atoms), 'cpu', pos_guidance_opt=guidance)
        except StopAfter:
            pass
        finally:
            model.num_timesteps = old_T
            del model.forward
            model.sample_nodes = orig_sample_nodes
    arrays = dict(phore_x=data['phore'].x, phore_pos=data['phore'].pos, phore_norm=data['phore'].norm,
                  center=data.center, n_atoms=np.array(n_atoms), t_total=np.array(t_total),
                  model_count_draw=counts['model'])
    for i, (n, t) in enumerate(tape.tape):
        arrays[f'rng{i:03d}_{n}'] = t
    for s, (fi, fo) in enumerate(zip(fwd_in, fwd_out)):
        arrays[f's{s}_h_node'] = fi['h_node_pert']
        arrays[f's{s}_pos'] = fi['pos_pert']
        arrays[f's{s}_h_edge'] = fi['h_edge_pert'].argmax(-1).to(torch.int8)
        arrays[f's{s}_t'] = fi['time_step']
        arrays[f's{s}_out_v'], arrays[f's{s}_out_x0'], arrays[f's{s}_out_bond'] = fo
    if res is not None:
        arrays.update(pred_node=res['pred'][0], pred_pos=res['pred'][1], pred_edge=res['pred'][2],
                      traj_node=res['traj'][0].argmax(-1).to(torch.int8), traj_pos=res['traj'][1],
                      traj_edge=res['traj'][2].argmax(-1).to(torch.int8),
                      lig_batch=res['lig_info'][1], lig_edge_index=res['lig_info'][2],
                      lig_edge_batch=res['lig_info'][3])
    save(name, **arrays)


FULL_CK = 50          # g5_sample_full1000*: the carried state is stored after every FULL_CK-th step


def synthetic_phore_data(seed, p, frac_ex=0.94, spread=6.0):
    """A `HeteroData` of the reference's shape (datasets/get_phore_data.py:55-105) holding a synthetic pharmacophore of the headline
    statistics (p ~ 107 nodes, 94 % exclusion spheres, positions 6 randn, centred): the parsed real file's
    object with its `phore` store and `center` replaced, so every other field the reference touches is the reference's own."""
    data = real_phore_data(seed)
    x, pos, norm = synthetic_phore(torch.Generator().manual_seed(seed), p, frac_ex=frac_ex, spread=spread)
    st = data['phore']
    st.x, st.pos, st.norm = x, pos, norm
    st.center_of_mass = torch.zeros(3)
    data.center = torch.zeros(3)
    return data


def g5_sample_full1000(model, name, seed, n_atoms, guidance=None, ck_every=FULL_CK, sparse_edge_gaps=False, data=None):
    """The reference's own `sample()` for ALL 1000 steps (diffusion.py:391-525), CPU generator seeded right in front of the
    sampler's first draw (`seed_all(seed + 1)` inside the forced atom-count call: a test re-creates the draws by seeding the same
    generator and drawing in the reference's order, SURVEY.md Appendix B -- 3.3 MB of recorded noise stay out of the fixture).
    Stored, thinned: the types of every step (int8), the positions of every step, the final `pred`; per step and row the top-2 gap
    of (Gumbel + log-posterior) of the reference's categorical draw (float16: how close each draw was to flipping); after every
    50th step the carried state the trajectory does not hold (log-posteriors, un-centred positions); float64 sums of every draw
    (a host whose generator stream differs is told apart from a parity failure); per-step max |logit| (scale of the bounds)."""
    import models.diffusion as rd
    data = real_phore_data(seed) if data is None else data
    T = model.num_timesteps
    rec = dict(gap=[[], []], scale=[[], []], usum=[[], []], post=[[], []], pos_in=[], eps_sum=[])
    orig_lsc, orig_fwd, orig_randn_like = rd.log_sample_categorical, model.forward, torch.randn_like
    calls = {'n': 0}

    def lsc(logits):
        """The reference's function makes the draw; the same uniform numbers are then re-drawn from the saved generator state
        to measure the margin of that draw."""
        state = torch.get_rng_state()
        idx = orig_lsc(logits)
        after = torch.get_rng_state()
        torch.set_rng_state(state)
        u = torch.rand_like(logits)
        assert torch.equal(torch.get_rng_state(), after)
        top = (-torch.log(-torch.log(u + 1e-30) + 1e-30) + logits).topk(2, dim=-1)
        assert torch.equal(top.indices[:, 0], idx)
        k = calls['n'] % 2                                   # node, edge, node, edge, ... (diffusion.py:457,465)
        rec['gap'][k].append((top.values[:, 0] - top.values[:, 1]).to(torch.float16))
        rec['usum'][k].append(float(u.double().sum()))
        rec['post'][k].append(logits.detach().clone())
        calls['n'] += 1
        return idx

    def fwd(**kw):
        out = orig_fwd(**kw)
        rec['pos_in'].append(kw['pos_pert'].detach().clone())
        rec['scale'][0].append(float(out[0].abs().max()))
        rec['scale'][1].append(float(out[2].abs().max()))
        return out

    def randn_like(x, *a, **k):                              # transition.py:55 (the position noise of every step)
        out = orig_randn_like(x, *a, **k)
        rec['eps_sum'].append(float(out.double().sum()))
        return out

    orig_sample_nodes = model.sample_nodes
    init = {}

    def forced_nodes(data_, batch_size, device, sample_mode='uniform', normal_scale=4.0):
        orig_sample_nodes(data_, batch_size, device, sample_mode, normal_scale)
        seed_all(seed + 1)
        return torch.tensor(n_atoms)

    model.sample_nodes, model.forward = forced_nodes, fwd
    rd.log_sample_categorical, torch.randn_like = lsc, randn_like
    try:
        res = model.sample(data, len(n_atoms), 'cpu', pos_guidance_opt=guidance)
    finally:
        rd.log_sample_categorical, torch.randn_like = orig_lsc, orig_randn_like
        del model.forward
        model.sample_nodes = orig_sample_nodes
    assert calls['n'] == 2 * T and len(rec['pos_in']) == T and len(rec['eps_sum']) == T
    # loop indices i: the state after step i = the input of step i + 1  (ck_every=None: no checkpoints -- a large fixture stays small)
    ck = list(range(ck_every - 1, T - 1, ck_every)) if ck_every else []
    arrays = dict(phore_x=data['phore'].x, phore_pos=data['phore'].pos, phore_norm=data['phore'].norm, center=data.center,
                  n_atoms=np.array(n_atoms), sample_seed=np.array(seed + 1), ck_steps=np.array(ck),
                  traj_node=res['traj'][0].argmax(-1).to(torch.int8), traj_pos=res['traj'][1],
                  traj_edge=res['traj'][2].argmax(-1).to(torch.int8),
                  pred_node=res['pred'][0], pred_pos=res['pred'][1], pred_edge=res['pred'][2],
                  gap_node=torch.stack(rec['gap'][0]),
                  scale_node=np.array(rec['scale'][0], dtype=np.float32), scale_edge=np.array(rec['scale'][1], dtype=np.float32),
                  u_node_sum=np.array(rec['usum'][0]), u_edge_sum=np.array(rec['usum'][1]), eps_sum=np.array(rec['eps_sum']),
                  pos_init=rec['pos_in'][0])
    if ck:
        arrays.update(ck_log_node=torch.stack([rec['post'][0][i] for i in ck]), ck_log_edge=torch.stack([rec['post'][1][i] for i in ck]),
                      ck_pos=torch.stack([rec['pos_in'][i + 1] for i in ck]))
    ge = torch.stack(rec['gap'][1])
    if sparse_edge_gaps:          # thousands of bond rows: only the margins below 0.02 are kept (a row that is not listed had a larger one)
        st_, row_ = (ge < 0.02).nonzero(as_tuple=True)
        arrays.update(gap_edge_step=st_.to(torch.int16), gap_edge_row=row_.to(torch.int16), gap_edge_val=ge[st_, row_], gap_edge_floor=np.array(0.02))
    else:
        arrays['gap_edge'] = ge
    save(name, **arrays)


# --------------------------------------------------------------------------------------------
# G7  state_dict manifest
# --------------------------------------------------------------------------------------------
def g7_manifest(model):
    with open(os.path.join(OUT, 'g7_state_dict_manifest.txt'), 'w') as f:
        for k, v in model.state_dict().items():
            f.write(f'{k}\t{tuple(v.shape)}\t{str(v.dtype).replace("torch.", "")}\n')
    print('g7 manifest:', len(model.state_dict()), 'entries')


def g_posterior(model):
    """Posterior-step KATs straight from transition.py:44-63,285-315 and common.py:425-431."""
    g = torch.Generator().manual_seed(5)
    batch = torch.tensor([0, 0, 0, 1, 1, 2, 2, 2, 2, 2])
    t = torch.tensor([999, 500, 0])
    arrays = dict(batch=batch, t=t)
    for tag, tr, K in (('node', model.node_transition, 12), ('edge', model.edge_transition, 6)):
        log_v0 = torch.log_softmax(2 * torch.randn(10, K, generator=g), -1)
        log_vt = torch.log_softmax(3 * torch.randn(10, K, generator=g), -1)
        post = tr.q_v_posterior(log_v0, log_vt, t, batch, v0_prob=True)
        u = torch.rand(10, K, generator=g)
        gumbel = -torch.log(-torch.log(u + 1e-30) + 1e-30)
        arrays.update({f'{tag}_log_v0': log_v0, f'{tag}_log_vt': log_vt, f'{tag}_post': post,
                       f'{tag}_u': u, f'{tag}_sample': (gumbel + post).argmax(-1)})
    x_t = torch.randn(10, 3, generator=g)
    x0 = torch.randn(10, 3, generator=g)
    eps = torch.randn(10, 3, generator=g)
    orig = torch.randn_like
    torch.randn_like = lambda m: eps
    try:
        prev = model.pos_transition.get_prev_from_recon(x_t, x0, t, batch)
    finally:
        torch.randn_like = orig
    arrays.update(pos_xt=x_t, pos_x0=x0, pos_eps=eps, pos_prev=prev)
    save('g_posterior', **arrays)


def g6_compute_loss(model, name, seed, n_atoms, n_phore, bond_len_loss=False):
    """G6: the reference's compute_loss (diffusion.py:249-352) + backward on a synthetic HeteroData batch; the draws of
    sample_time / add_noise are recorded (torch.randint, Tensor.normal_, torch.rand_like).  `bond_len_loss`: the run has the
    config flag on (diffusion.py:286-290,333,341) and the molecule's bonds `edge_index` = the directed pairs with a bond class > 0."""
    from torch_geometric.data import Batch
    from oracle.make_inputs import synthetic_train_batch
    d = synthetic_train_batch(seed, n_atoms, n_phore)
    if bond_len_loss:
        d['edge_index'] = d['f_edge_index'][:, d['f_edge_attr'] > 0]
        assert d['edge_index'].size(1) >= 8
    data = Batch()
    object.__setattr__(data, 'num_graphs', len(n_atoms))
    data['ligand'].x, data['ligand'].pos = d['ligand_x'], d['ligand_pos']
    data['ligand'].batch, data['ligand'].ptr = d['ligand_batch'], d['ligand_ptr']
    e = data['ligand', 'ligand']
    e.f_edge_index, e.f_edge_attr, e.f_edge_attr_batch = d['f_edge_index'], d['f_edge_attr'], d['f_edge_batch']
    if bond_len_loss:
        e.edge_index = d['edge_index']
    flag_before = model.bond_len_loss
    model.bond_len_loss = bond_len_loss
    ph = data['phore']
    ph.x, ph.pos, ph.norm, ph.batch = d['phore_x'], d['phore_pos'], d['phore_norm'], d['phore_batch']
    seed_all(seed)
    normal_draws = []
    orig_normal = torch.Tensor.normal_

    def rec_normal(self, *a, **k):
        out = orig_normal(self, *a, **k)
        normal_draws.append(out.detach().clone())
        return out
    model.zero_grad()
    torch.Tensor.normal_ = rec_normal
    try:
        with RngTape() as tape:
            loss, info = model.compute_loss(data)
    finally:
        torch.Tensor.normal_ = orig_normal
        model.bond_len_loss = flag_before
    loss.backward()
    draws = {n: [t for (nn_, t) in tape.tape if nn_ == n] for n in ('randint', 'rand_like')}
    assert len(draws['randint']) == 1 and len(draws['rand_like']) == 2 and len(normal_draws) == 1, \
        (len(draws['randint']), len(draws['rand_like']), len(normal_draws))
    names = [k for k, p in model.named_parameters()]
    gn = np.array([float(p.grad.norm()) if p.grad is not None else 0.0 for k, p in model.named_parameters()])
    arrays = {k: v for k, v in d.items()}
    arrays.update(time_draw=draws['randint'][0], pos_noise=normal_draws[0], u_node=draws['rand_like'][0],
                  u_edge=draws['rand_like'][1], loss=loss.detach(),
                  info_keys=np.array(sorted(info)), info_vals=np.array([info[k] for k in sorted(info)], dtype=np.float64),
                  param_names=np.array(names), grad_norm=gn)
    params = dict(model.named_parameters())
    for k in ('v_inference.2.bias', 'bond_inference.0.bias', 'node_embedder.weight', 'edge_embedder.weight',
              'atom_mlp.2.weight', 'phore_encoder.hq_func.net.0.bias', 'denoiser.edge_pred_layer.net.1.weight',
              'denoiser.base_block.0.dire_embedding.weight', 'denoiser.base_block.0.bond_layer.hk_func.net.1.weight',
              'denoiser.base_block.2.pos_layer_with_edge.xv_func.net.3.weight',
              'denoiser.base_block.3.node_layer_with_bond.hv_func.net.3.bias',
              'denoiser.base_block.5.pos_layer_with_bond.xq_func.net.3.bias', 'denoiser.base_block.5.lin_node.bias'):
        arrays['grad::' + k] = params[k].grad
    model.zero_grad()
    save(name, **arrays)


def g8_phore_parse():
    """datasets/get_phore_data.py:24-105 on a shipped pharmacophore file, no transform: the input side of `sample`."""
    path = os.path.join(ref_import.REFERENCE, 'data/phores_for_sampling/P03211_merge.phore')
    d = PhoreData_New([path])[0]
    save('g8_phore_parse', file_text=np.frombuffer(open(path, 'rb').read(), dtype=np.uint8), x=d['phore'].x,
         pos=d['phore'].pos, norm=d['phore'].norm, center=d.center)


def g9_unbatch_decode():
    """utils/sample_utils.py:57-132 (unbatch_data / decode_data) on a seeded synthetic result dict with two masked atoms."""
    from utils.sample_utils import decode_data, unbatch_data
    g = torch.Generator().manual_seed(9)
    na = torch.tensor([5, 3, 6])
    N = int(na.sum())
    ei, eb = make_edge_data(na)
    E = ei.size(1)
    bn = torch.repeat_interleave(torch.arange(3), na)
    pred = [torch.randn(N, 12, generator=g), torch.randn(N, 3, generator=g), torch.randn(E, 6, generator=g)]
    pred[0][2, 11] = 9.0
    pred[0][9, 11] = 9.0
    traj = [torch.randn(4, N, 12, generator=g), torch.randn(4, N, 3, generator=g), torch.randn(4, E, 6, generator=g)]
    outs = unbatch_data({'pred': pred, 'traj': traj, 'lig_info': [na, bn, ei, eb]}, 3, include_bond=True)
    arr = dict(na=na, ei=ei, eb=eb, bn=bn)
    arr.update({f'pred{i}': t for i, t in enumerate(pred)})
    arr.update({f'traj{i}': t for i, t in enumerate(traj)})
    for gi, o in enumerate(outs):
        d = decode_data(o['pred'], o['edge_index'], include_bond=True)
        arr.update({f'g{gi}_edge_index': o['edge_index'], f'g{gi}_traj1': o['traj'][1], f'g{gi}_element': np.array(d['element']),
                    f'g{gi}_atom_pos': d['atom_pos'], f'g{gi}_bond_type': d['bond_type'], f'g{gi}_bond_index': d['bond_index']})
    save('g9_unbatch_decode', **arr)


def profile_fixtures(profile):
    """Adversarial weight sets (phoregen_amd/weights.py PROFILES): the reference itself run with negative / tiny / zero
    LayerNorm gammas and with trained-like scales -- forward + layer captures, a 3-step sampler head and one loss/gradient
    fixture per profile (small graphs: the fixtures stay < 0.5 MB each)."""
    model, _ = build_model(seed=0, profile=profile)
    g23_forward(model, f'g3_forward_a_{profile}', seed=101, n_atoms=[5, 9], n_phore=[6, 11], t_values=[700, 30])
    g5_sample(model, f'g5_sample_head3_{profile}', seed=2032, n_atoms=[6, 9], n_steps=3, t_total=1000)
    g5_sample(model, f'g5_sample_tail4_{profile}', seed=2033, n_atoms=[7, 5, 8], n_steps=4, t_total=4)
    g6_compute_loss(model, f'g6_loss_a_{profile}', seed=61, n_atoms=[6, 9], n_phore=[7, 12])


GUIDANCE = [{'type': 'atom_prox', 'min_d': 1.2, 'max_d': 1.9}, {'type': 'center_prox'}]


def full1000_fixtures(model):
    g5_sample_full1000(model, 'g5_sample_full1000_a', seed=2040, n_atoms=[6, 9, 7])
    g5_sample_full1000(model, 'g5_sample_full1000_guid', seed=2041, n_atoms=[8, 6], guidance=GUIDANCE)
    # ligands of realistic size: 34 atoms = 3 row tiles of the triplet kernel, 21 atoms = 2 (1 542 bond rows; ~15 min of CPU)
    g5_sample_full1000(model, 'g5_sample_full1000_n34', seed=2042, n_atoms=[34, 21], ck_every=None, sparse_edge_gaps=True)


def full1000_headline(model):
    """The headline shape (BASELINE.json configs[2]; /root/reference/models/diffusion.py:432-517): 4 ligands of 38 / 40 / 43 / 52 atoms
    (3 / 3 / 3 / 4 row tiles of the triplet kernel; 7 424 bond rows, 316 542 triplets) on ONE synthetic pharmacophore of 107 nodes,
    unguided, all 1000 steps of the reference's own `sample()` (~2.5 h of CPU at 4 threads)."""
    g5_sample_full1000(model, 'g5_sample_full1000_headline', seed=2043, n_atoms=[38, 40, 43, 52], ck_every=None, sparse_edge_gaps=True,
                       data=synthetic_phore_data(2043, 107))


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'full1000_headline':
        model, cfg = build_model(seed=0)
        if len(sys.argv) > 2:                                  # plumbing check: `full1000_headline 3` runs 3 steps only, writes nothing kept
            model.num_timesteps = int(sys.argv[2])
            OUT = '/tmp'
        full1000_headline(model)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'len':        # only the bond_len_loss fixture (added in round 3)
        model, cfg = build_model(seed=0)
        g6_compute_loss(model, 'g6_loss_len', seed=67, n_atoms=[7, 10, 5], n_phore=[9, 14, 6], bond_len_loss=True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'full1000':   # only the full-length free-running fixtures (added in round 5; ~20 min)
        model, cfg = build_model(seed=0)
        full1000_fixtures(model)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'full1000_n34':
        model, cfg = build_model(seed=0)
        g5_sample_full1000(model, 'g5_sample_full1000_n34', seed=2042, n_atoms=[34, 21], ck_every=None, sparse_edge_gaps=True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'profiles':
        for prof in ('gamma_signed', 'trained_like'):
            profile_fixtures(prof)
        sys.exit(0)
    g1_ops()
    g8_phore_parse()
    g9_unbatch_decode()
    model, cfg = build_model(seed=0)
    g7_manifest(model)
    g4_tables(model)
    g_posterior(model)
    g23_forward(model, 'g3_forward_a', seed=101, n_atoms=[5, 9], n_phore=[6, 11], t_values=[700, 30])
    g23_forward(model, 'g3_forward_b', seed=202, n_atoms=[19, 4, 12], n_phore=[40, 23, 37], t_values=[999, 0, 412])
    g5_sample(model, 'g5_sample_head3', seed=2032, n_atoms=[6, 9], n_steps=3, t_total=1000)
    g5_sample(model, 'g5_sample_tail4', seed=2033, n_atoms=[7, 5, 8], n_steps=4, t_total=4)
    g5_sample(model, 'g5_sample_full25', seed=2034, n_atoms=[8, 6], n_steps=25, t_total=25)
    g6_compute_loss(model, 'g6_loss_a', seed=61, n_atoms=[6, 9], n_phore=[7, 12])
    g6_compute_loss(model, 'g6_loss_b', seed=64, n_atoms=[11, 4, 8], n_phore=[23, 9, 15])
    g6_compute_loss(model, 'g6_loss_len', seed=67, n_atoms=[7, 10, 5], n_phore=[9, 14, 6], bond_len_loss=True)
    g5_sample(model, 'g5_sample_guid3', seed=2035, n_atoms=[6, 7], n_steps=3, t_total=3,
              guidance=[{'type': 'atom_prox', 'min_d': 1.2, 'max_d': 1.9}, {'type': 'center_prox'}])
    for prof in ('gamma_signed', 'trained_like'):
        profile_fixtures(prof)
    full1000_fixtures(model)
    full1000_headline(model)
