#!/usr/bin/env python3
"""Free-running parity report at scale (SURVEY.md 7 "hard parts"): the HIP sampler and the oracle (CPU) run the SAME S reverse steps
from t = 999 on the same B graphs with the same CPU-generator draws; per step, the number of graphs whose atom / bond types still agree
exactly and the coordinate RMSD over the agreeing graphs.  Round 6: every departure is LABELLED (top-2 margin of the flipped draw in both
implementations against the test suite's tie rule, tests/test_gpu_parity.py FLIP_GAP_MULT), the final-frame RMSD is given per graph,
and a CONTROL run says what two fp32 CPU evaluations of the reference's own dataflow do to each other over the same run: the oracle
against the oracle with the atoms (hence the bond rows and the triplet order) of every ligand permuted -- same draws per atom / per bond,
only the fp32 summation order inside the bond / triplet segments differs.

Phases (the oracle needs ~0.15 s per graph-step of CPU and the GPU box's minutes are budgeted; caches are git-ignored and travel to the
GPU box with the snapshot):
  python tools/match_rate.py ref S B          (any host)  -> tools/_cache/match_rate_ref_S_B.npz   oracle trajectory + draw checksums
  python tools/match_rate.py perm S B         (any host)  -> tools/_cache/match_rate_perm_S_B.npz  permuted-oracle trajectory (rows put back
                                                             in the caller's order) + the top-2 margin of every categorical draw
  python tools/match_rate.py sub S B 14,21    (any host)  -> tools/_cache/match_rate_sub_S_B.npz   the oracle re-run on a few graphs alone with
                                                             their slice of the draws (bit-identical to `ref`, checked) + margins
  python tools/match_rate.py hip S B          (GPU box)   -> one JSON line + gpurun_out/match_rate_hip_S_B.npz (margins of its departures)
  python tools/match_rate.py report S B       (any host)  -> one JSON line: hip-vs-oracle and control side by side, every departure labelled
  python tools/match_rate.py hipperm S B [headline]  (GPU box) -> one JSON line: the HIP sampler against ITSELF with every ligand's atoms permuted (same
                                                             draws per atom / bond): the same control on the device, affordable at the HEADLINE shape
                                                             (`headline`: the first B graphs of bench.ligphore_workload(128): n ~ 40, p ~ 107 per graph)
  python tools/match_rate.py S B              ref + hip in one process (small S x B)
"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch

args = sys.argv[1:]
phase = args.pop(0) if args and args[0] in ('ref', 'hip', 'perm', 'sub', 'report', 'hipperm') else 'both'
HEADLINE = 'headline' in args
args = [a for a in args if a != 'headline']
S = int(args[0]) if len(args) > 0 else 40
B = int(args[1]) if len(args) > 1 else 12
SUBSET = [int(v) for v in args[2].split(',')] if len(args) > 2 else []
SEED = 77
PERM_SEED = 911
TOL = 2e-5                              # tests/helpers.py: forward outputs vs oracle, x max |reference|
FLIP_GAP_MULT = 2 * 5 * TOL             # tests/test_gpu_parity.py: x max |logit| of the step -> a categorical draw below this margin is a tie
CACHE = lambda kind: os.path.join(ROOT, 'tools', '_cache', f'match_rate_{kind}_{S}_{B}.npz')
HIP_OUT = os.path.join(ROOT, 'gpurun_out', f'match_rate_hip_{S}_{B}.npz')
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'g8_phore_parse.npz'))
t = lambda a: torch.as_tensor(np.asarray(a))
x, pos, nrm, center = t(g['x']), t(g['pos']), t(g['norm']), t(g['center'])
gen = torch.Generator().manual_seed(4)
na = torch.randint(12, 28, (B,), generator=gen)
WORK = None
if HEADLINE:                                  # the bench's workload (BASELINE configs[2]): its first B graphs, each with its own pharmacophore
    from bench import ligphore_workload, subset_workload
    WORK = subset_workload(ligphore_workload(128), torch.arange(B))
    na = WORK['num_atoms']
N, E = int(na.sum()), int((na * (na - 1)).sum())
MARKS = sorted({0, 1, 2, 5, 10, 20, 30, 50, 100, 200, 300, 400, 500, 600, 700, 800, 900, 950, 990, S} & set(range(S + 1)))


def topology():
    from oracle.phoregen_oracle import make_edge_data
    bn = torch.repeat_interleave(torch.arange(B), na)
    ei, be = make_edge_data(na)
    return bn, ei, be


class SummingRng:
    """TorchCpuRng that also keeps a float64 sum of every draw (a host whose generator stream differs is told apart from a parity failure).
    `rows_node` / `rows_edge`: the draw of a [N, .] / [E, .] shape is taken whole from the generator and THEN indexed (a permutation of
    the rows, or the rows of a subset of graphs): every atom / bond keeps the numbers the plain run gave it."""

    def __init__(self, rows_node=None, rows_edge=None):
        self.sums, self.rows = [], {N: rows_node, E: rows_edge}

    def _keep(self, v):
        self.sums.append(float(v.double().sum()))
        r = self.rows.get(v.shape[0])
        return v if r is None else v[r]

    def _full(self, shape):
        n = {len(r): full for full, r in self.rows.items() if r is not None}.get(shape[0], shape[0])
        return (n,) + tuple(shape[1:])

    def randn(self, shape):
        return self._keep(torch.randn(self._full(shape)))

    def rand(self, shape):
        return self._keep(torch.rand(self._full(shape)))

    def rand64(self, shape):
        return self._keep(torch.rand(self._full(shape), dtype=torch.float64))


class Margins:
    """Top-2 margin of (Gumbel + log-posterior) of every categorical draw (float16, as the fixtures keep it) and the step's max |logit|."""

    def __init__(self, n_node, n_edge):
        self.gap = {'node': torch.zeros(S, n_node, dtype=torch.float16), 'edge': torch.zeros(S, n_edge, dtype=torch.float16)}
        self.scale = torch.zeros(S, 2)

    def __call__(self, i, kind, logp, u, logits):
        top = (-torch.log(-torch.log(u + 1e-30) + 1e-30) + logp).topk(2, dim=-1).values
        self.gap[kind][i] = (top[:, 0] - top[:, 1]).to(torch.float16)
        self.scale[i, 0 if kind == 'node' else 1] = float(logits.abs().max())


def oracle_run(num_atoms, rng, observer=None):
    from helpers import make_oracle
    torch.set_num_threads(int(os.environ.get('MATCH_RATE_THREADS', min(16, os.cpu_count() or 1))))
    orc = make_oracle(0)
    torch.manual_seed(SEED)
    t0 = time.time()
    with torch.no_grad():
        ref = orc.sample(x, pos, nrm, center, num_atoms, rng, n_steps=S, observer=observer, keep_steps=False)
    return ref, time.time() - t0


def run_ref():
    rng = SummingRng()
    ref, secs = oracle_run(na, rng)
    out = dict(node=ref['traj'][0].argmax(-1).to(torch.int8).numpy(), edge=ref['traj'][2].argmax(-1).to(torch.int8).numpy(),
               pos=ref['traj'][1].numpy(), sums=np.array(rng.sums), oracle_s=np.array(secs), na=na.numpy())
    os.makedirs(os.path.dirname(CACHE('ref')), exist_ok=True)
    np.savez_compressed(CACHE('ref'), **out)
    return out


def graph_permutations():
    """perm_atom[r'] = the caller's atom row that sits at row r' of the permuted run; perm_edge likewise for bond rows (the permuted run's
    bond list is make_edge_data's list over the permuted atoms, utils/sample_utils.py:40-54)."""
    bn, ei, be = topology()
    gp = torch.Generator().manual_seed(PERM_SEED)
    off = torch.cat([torch.zeros(1, dtype=torch.long), na.cumsum(0)])
    perm_atom = torch.cat([off[gi] + torch.randperm(int(n), generator=gp) for gi, n in enumerate(na.tolist())])
    row_of = torch.full((N, N), -1, dtype=torch.long)
    row_of[ei[0], ei[1]] = torch.arange(E)
    perm_edge = row_of[perm_atom[ei[0]], perm_atom[ei[1]]]
    assert (perm_edge >= 0).all() and torch.equal(torch.sort(perm_edge).values, torch.arange(E)) and torch.equal(be[perm_edge], be)
    return perm_atom, perm_edge


def run_perm():
    perm_atom, perm_edge = graph_permutations()
    rng, obs = SummingRng(perm_atom, perm_edge), Margins(N, E)
    res, secs = oracle_run(na, rng, obs)

    def back(v, perm):                      # row r' of the permuted run is the caller's row perm[r']
        out = torch.empty_like(v)
        out[:, perm] = v
        return out
    out = dict(node=back(res['traj'][0].argmax(-1).to(torch.int8), perm_atom).numpy(), edge=back(res['traj'][2].argmax(-1).to(torch.int8), perm_edge).numpy(),
               pos=back(res['traj'][1], perm_atom).numpy(), sums=np.array(rng.sums), oracle_s=np.array(secs), na=na.numpy(),
               gap_node=back(obs.gap['node'], perm_atom).numpy(), gap_edge=back(obs.gap['edge'], perm_edge).numpy(), scale=obs.scale.numpy())
    np.savez_compressed(CACHE('perm'), **out)
    return out


def run_sub(graphs):
    """The oracle on a few graphs alone, each atom / bond with the numbers the full run gave it.  The graphs of a batch are independent in the
    reference (block-diagonal batch), so this reproduces the full run's rows bit for bit -- checked against the `ref` cache."""
    ref = np.load(CACHE('ref'))
    bn, ei, be = topology()
    sel = torch.tensor(sorted(graphs))
    rows_n, rows_e = torch.isin(bn, sel).nonzero().flatten(), torch.isin(be, sel).nonzero().flatten()
    rng, obs = SummingRng(rows_n, rows_e), Margins(len(rows_n), len(rows_e))
    res, secs = oracle_run(na[sel], rng, obs)
    assert np.allclose(np.array(rng.sums), ref['sums'], rtol=1e-12, atol=1e-9), 'generator stream differs from the host the `ref` phase ran on'   # (the float64 checksum's own summation order depends on the thread count)
    tn, te, tp = res['traj'][0].argmax(-1).numpy(), res['traj'][2].argmax(-1).numpy(), res['traj'][1].numpy()
    same = dict(node=bool(np.array_equal(tn, ref['node'][:, rows_n.numpy()])), edge=bool(np.array_equal(te, ref['edge'][:, rows_e.numpy()])),
                pos_max_abs=float(np.abs(tp - ref['pos'][:, rows_n.numpy()]).max()))
    # per graph: the first frame whose types or coordinates are not the full run's bit for bit (S + 1: none) -- a graph alone is NOT always the
    # graph inside its batch: the library GEMM blocks its rows by the batch's size, and the late dynamics amplify that last-bit difference too
    first_diff = {}
    for gi in sel.tolist():
        mn, me = (bn[rows_n] == gi).numpy(), (be[rows_e] == gi).numpy()
        bad = (tn[:, mn] != ref['node'][:, rows_n.numpy()][:, mn]).any(1) | (te[:, me] != ref['edge'][:, rows_e.numpy()][:, me]).any(1) | \
              (tp[:, mn] != ref['pos'][:, rows_n.numpy()][:, mn]).any((1, 2))
        first_diff[gi] = int(np.argmax(bad)) if bad.any() else S + 1
    same['first_differing_frame_by_graph'] = first_diff
    out = dict(graphs=sel.numpy(), rows_node=rows_n.numpy(), rows_edge=rows_e.numpy(), gap_node=obs.gap['node'].numpy(), gap_edge=obs.gap['edge'].numpy(),
               scale=obs.scale.numpy(), identical_to_ref=np.array([same['node'], same['edge']]), pos_max_abs=np.array(same['pos_max_abs']), oracle_s=np.array(secs),
               first_diff=np.array([first_diff[gi] for gi in sel.tolist()]))
    if os.path.exists(CACHE('sub')):         # keep the graphs of earlier calls
        old = dict(np.load(CACHE('sub')))
        keep = ~np.isin(old['graphs'], out['graphs'])
        if keep.any():
            kn, ke = np.isin(bn[old['rows_node']].numpy(), old['graphs'][keep]), np.isin(be[old['rows_edge']].numpy(), old['graphs'][keep])
            out.update(graphs=np.concatenate([old['graphs'][keep], out['graphs']]), rows_node=np.concatenate([old['rows_node'][kn], out['rows_node']]),
                       rows_edge=np.concatenate([old['rows_edge'][ke], out['rows_edge']]), gap_node=np.concatenate([old['gap_node'][:, kn], out['gap_node']], 1),
                       gap_edge=np.concatenate([old['gap_edge'][:, ke], out['gap_edge']], 1), scale=np.maximum(old['scale'], out['scale']),
                       identical_to_ref=old['identical_to_ref'] & out['identical_to_ref'], pos_max_abs=np.maximum(old['pos_max_abs'], out['pos_max_abs']),
                       first_diff=np.concatenate([old['first_diff'][keep], out['first_diff']]))
    np.savez_compressed(CACHE('sub'), **out)
    return same, secs


def compare(a, b):
    """Per step: which graphs still hold identical atom AND bond types in trajectories a, b (dicts with node / edge / pos), the per-graph RMSD, and
    for every graph that leaves: the step and the rows whose type differs there."""
    bn, ei, be = topology()
    cnt = torch.bincount(bn, minlength=B).float()
    an, ae, ap = t(a['node']).long(), t(a['edge']).long(), t(a['pos'])
    rn, re_, rp = t(b['node']).long(), t(b['edge']).long(), t(b['pos'])
    ok = torch.ones(B, dtype=torch.bool)
    rows, departures, rmsd_all = [], [], torch.zeros(S + 1, B)
    for s in range(S + 1):
        dn, de = an[s] != rn[s], ae[s] != re_[s]
        same = (torch.zeros(B).index_add(0, bn, dn.float()) == 0) & (torch.zeros(B).index_add(0, be, de.float()) == 0)
        for gi in (ok & ~same).nonzero().flatten().tolist():
            departures.append(dict(graph=gi, frame=s, draw_step=s - 1, node_rows=(dn & (bn == gi)).nonzero().flatten().tolist(),
                                   edge_rows=(de & (be == gi)).nonzero().flatten().tolist()))
        ok &= same                                             # a graph that has differed once stays out (its later states are another trajectory)
        rmsd_all[s] = (torch.zeros(B).index_add(0, bn, ((ap[s] - rp[s]) ** 2).sum(-1)) / cnt).sqrt()
        rows.append((s, int(ok.sum()), float(rmsd_all[s][ok].max()) if ok.any() else float('nan')))
    final = rmsd_all[S][ok]
    edges = [0, 1e-6, 1e-5, 1e-4, 1e-3, 1e-2, float('inf')]
    hist = {f'<= {hi:g}': int(((final > lo) & (final <= hi)).sum()) for lo, hi in zip(edges[:-1], edges[1:])}
    for d in departures:                                       # RMSD of the departing graph in the frame BEFORE it left (was it still on the trajectory?)
        d['rmsd_frame_before'] = float(rmsd_all[max(d['frame'] - 1, 0), d['graph']])
    on_traj = torch.ones(S + 1, B, dtype=torch.bool)
    for d in departures:
        on_traj[d['frame']:, d['graph']] = False
    worst_step_by_graph = rmsd_all.masked_fill(~on_traj, 0).max(0).values
    return dict(graphs_identical_through_step={r[0]: r[1] for r in rows if r[0] in MARKS},
                worst_graph_rmsd_over_identical_graphs={r[0]: r[2] for r in rows if r[0] in MARKS and r[0] > 0},
                worst_rmsd_any_step=max(r[2] for r in rows[1:] if r[2] == r[2]),
                final_frame_rmsd_by_graph={gi: float(rmsd_all[S, gi]) for gi in ok.nonzero().flatten().tolist()},
                final_frame_rmsd_histogram_identical_graphs=hist, graphs_final_rmsd_within_1e4=int((final <= 1e-4).sum()),
                graphs_within_1e4_at_every_step=int(((worst_step_by_graph <= 1e-4) & ok).sum()),
                frame_before_final_worst_rmsd=float(rmsd_all[S - 1][ok].max()) if ok.any() else float('nan'),
                match_rate_final=rows[-1][1] / B, departures=departures)


def label(departures, margins_a, margins_b, name_a, name_b):
    """Tie / not-a-tie by the test suite's rule: every flipped row's top-2 margin below FLIP_GAP_MULT x max |logit| of the step in BOTH runs, and the
    graph still within 1e-4 of the other trajectory in the frame before."""
    for d in departures:
        s = d['draw_step']
        flips = []
        for kind, rows in (('node', d.pop('node_rows')), ('edge', d.pop('edge_rows'))):
            for r in rows:
                flips.append({'kind': kind, 'row': r, f'gap_{name_a}': margins_a(kind, s, r), f'gap_{name_b}': margins_b(kind, s, r)})
        sc = [m('scale', s, 0) for m in (margins_a, margins_b)]
        sc = [v for v in sc if v is not None]
        bound = FLIP_GAP_MULT * max(sc) if sc else None
        d.update(flips=flips, tie_bound=bound)
        known = bound is not None and all(f[f'gap_{name_a}'] is not None and f[f'gap_{name_b}'] is not None for f in flips)
        tie = known and all(f[f'gap_{name_a}'] <= bound and f[f'gap_{name_b}'] <= bound for f in flips)
        one_sided = bound is not None and any(v is not None and v > bound for f in flips for k_, v in f.items() if k_.startswith('gap_'))
        d['label'] = ('tie' if tie and d['rmsd_frame_before'] <= 1e-4 else 'tie, graph already > 1e-4 off' if tie else 'NOT A TIE') if known else \
            ('NOT A TIE (the one recorded margin is above the bound)' if one_sided else 'margin not recorded')
    return departures


def margins_from(npz, rows_key=None):
    """Accessor (kind, step, caller's row) -> margin, over a cache that holds all rows (`perm`) or the rows of some graphs (`sub`, hip departures)."""
    if npz is None:
        return lambda kind, s, r: None
    idx, valid_before = {}, {}
    if rows_key:
        idx = {'node': {int(r): k for k, r in enumerate(npz['rows_node'])}, 'edge': {int(r): k for k, r in enumerate(npz['rows_edge'])}}
        if 'first_diff' in getattr(npz, 'files', npz):           # (`sub`: a margin counts only while the graph alone still IS the graph of the full run)
            bn, ei, be = topology()
            fd = {int(g_): int(f_) for g_, f_ in zip(npz['graphs'], npz['first_diff'])}
            valid_before = {'node': lambda r: fd[int(bn[r])], 'edge': lambda r: fd[int(be[r])]}

    def f(kind, s, r):
        if kind == 'scale':
            return float(npz['scale'][s].max())
        if rows_key:
            k = idx[kind].get(r)
            if k is None or (valid_before and s + 1 >= valid_before[kind](r)):
                return None
            return float(npz['gap_' + kind][s, k])
        return float(npz['gap_' + kind][s, r])
    return f


def run_hip(ref):
    import torch.nn.functional as F
    from phoregen_amd.config import default_model_config
    from phoregen_amd.models.diffusion import PhoreDiff
    from phoregen_amd.weights import init_deterministic_
    assert np.array_equal(ref['na'], na.numpy())
    DEV = 'cuda'
    model = init_deterministic_(PhoreDiff(default_model_config(), 'zinc_300'), 0).eval().to(DEV)
    p = x.shape[0]
    bp = torch.repeat_interleave(torch.arange(B), p)
    sums = []

    def keep(v):
        sums.append(float(v.double().sum()))
        return v
    real = {n: getattr(torch, n) for n in ('rand', 'randn')}
    torch.manual_seed(SEED)
    torch.rand, torch.randn = (lambda *a, **k: keep(real['rand'](*a, **k))), (lambda *a, **k: keep(real['randn'](*a, **k)))
    try:
        t0 = time.time()
        st = model.begin_sampling(x.repeat(B, 1), pos.repeat(B, 1), nrm.repeat(B, 1), bp, na, center.unsqueeze(0).expand(B, 3), rng='cpu', seed=0)
    finally:
        torch.rand, torch.randn = real['rand'], real['randn']
    assert st.N == N and st.E == E
    gap_n, gap_e = torch.zeros(S, N, device=DEV), torch.zeros(S, E, device=DEV)
    scale = torch.zeros(S, 2, device=DEV)

    def margins(u, logp):
        top = (-torch.log(-torch.log(u + 1e-30) + 1e-30) + logp).topk(2, dim=-1).values
        return top[:, 0] - top[:, 1]
    for i in range(S):
        un, ue, eps = keep(torch.rand(N, 12)), keep(torch.rand(E, 6)), keep(torch.randn(N, 3))
        un, ue = un.to(DEV), ue.to(DEV)
        model.reverse_step(st, i, 999 - i, None, draws=(un, ue, eps))
        gap_n[i], gap_e[i] = margins(un, st.log_node[st.cur]), margins(ue, st.log_edge[st.cur])
        scale[i, 0], scale[i, 1] = st.eng.ws.out_v.abs().max(), st.eng.ws.out_bond.abs().max()
    torch.cuda.synchronize()
    hip_s = time.time() - t0
    sums, rs = np.array(sums), np.asarray(ref['sums'])
    assert len(sums) == len(rs) == 3 + 3 * S
    uniform = np.ones(len(rs), dtype=bool)
    uniform[0] = False
    uniform[5::3] = False                                      # randn draws: the initial positions, then the third draw of every step
    assert np.array_equal(sums[uniform], rs[uniform]), 'torch CPU generator stream differs from the host the reference phase ran on'
    randn_dev = float(np.abs(sums[~uniform] - rs[~uniform]).max())
    hip = dict(node=st.node_traj[:S + 1].argmax(-1).cpu().numpy(), edge=st.edge_traj[:S + 1].argmax(-1).cpu().numpy(), pos=st.pos_traj[:S + 1].cpu().numpy())
    cmp_ = compare(hip, ref)
    bn, ei, be = topology()
    dep = sorted({d['graph'] for d in cmp_['departures']})
    rows_n, rows_e = torch.isin(bn, torch.tensor(dep, dtype=torch.long)).nonzero().flatten(), torch.isin(be, torch.tensor(dep, dtype=torch.long)).nonzero().flatten()
    os.makedirs(os.path.dirname(HIP_OUT), exist_ok=True)
    np.savez_compressed(HIP_OUT, graphs=np.array(dep), rows_node=rows_n.numpy(), rows_edge=rows_e.numpy(), gap_node=gap_n[:, rows_n.to(DEV)].cpu().numpy(),
                        gap_edge=gap_e[:, rows_e.to(DEV)].cpu().numpy(), scale=scale.cpu().numpy(), min_margin=np.array([float(gap_n.min()), float(gap_e.min())]),
                        compare=np.array(json.dumps(cmp_)), hip_s=np.array(hip_s), randn_dev=np.array(randn_dev), pos=hip['pos'], node=hip['node'].astype(np.int8), edge=hip['edge'].astype(np.int8))
    sub = np.load(CACHE('sub')) if os.path.exists(CACHE('sub')) else None
    label(cmp_['departures'], margins_from(np.load(HIP_OUT), 'rows'), margins_from(sub, 'rows'), 'hip', 'oracle')
    return {'graphs': B, 'steps': S, 'atoms': N, 'bond_rows': E, 'oracle_s': round(float(ref['oracle_s']), 1), 'hip_s': round(hip_s, 1), **cmp_,
            'randn_checksum_max_dev_between_hosts': randn_dev}


def run_hipperm():
    """HIP vs HIP with the atoms of every ligand permuted (the control of `perm`, on the device): both runs take the same CPU-generator draws,
    row-permuted for the second one; its trajectory is put back into the caller's order and compared like `hip` vs `ref`."""
    from phoregen_amd.config import default_model_config
    from phoregen_amd.models.diffusion import PhoreDiff
    from phoregen_amd.weights import init_deterministic_
    import torch.nn.functional as F
    DEV = 'cuda'
    model = init_deterministic_(PhoreDiff(default_model_config(), 'zinc_300'), 0).eval().to(DEV)
    perm_atom, perm_edge = graph_permutations()
    if WORK is not None:
        hp, pp, pn, bp = WORK['h_phore'], WORK['pos_phore'], WORK['phore_norm'], WORK['batch_phore']
        centers = torch.zeros(B, 3)
    else:
        p = x.shape[0]
        hp, pp, pn, bp = x.repeat(B, 1), pos.repeat(B, 1), nrm.repeat(B, 1), torch.repeat_interleave(torch.arange(B), p)
        centers = center.unsqueeze(0).expand(B, 3)
    def margins(u, logp):
        top = (-torch.log(-torch.log(u + 1e-30) + 1e-30) + logp).topk(2, dim=-1).values
        return top[:, 0] - top[:, 1]

    def run(pa, pe):
        """One HIP run whose row r holds the caller's atom pa[r] / bond pe[r] (None: the caller's order); everything returned in the caller's order."""
        st = model.begin_sampling(hp, pp, pn, bp, na, centers, rng='cpu', seed=0)           # (its own init draws are overwritten below)
        w = st.eng.ws
        torch.manual_seed(SEED)                        # both runs take the same stream, in the reference's order
        pos0 = torch.randn(N, 3)
        un0, ue0 = torch.rand(N, 12, dtype=torch.float64), torch.rand(E, 6, dtype=torch.float64)
        ia = torch.arange(N) if pa is None else pa
        ie = torch.arange(E) if pe is None else pe
        lp_n = torch.log(torch.from_numpy(model.node_transition.init_prob) + model.node_transition.eps).clamp_min(-32.)
        lp_e = torch.log(torch.from_numpy(model.edge_transition.init_prob) + model.edge_transition.eps).clamp_min(-32.)
        gum = lambda u: -torch.log(-torch.log(u + 1e-30) + 1e-30)
        h_node = F.one_hot((gum(un0[ia]) + lp_n.unsqueeze(0)).argmax(-1), 12).float().to(DEV)
        h_edge = F.one_hot((gum(ue0[ie]) + lp_e.unsqueeze(0)).argmax(-1), 6).float().to(DEV)
        p0 = (pos0[ia]).to(DEV) - st.center_rows
        w.in_h_node.copy_(h_node), w.in_pos.copy_(p0), w.in_h_edge.copy_(h_edge)
        st.log_node[0].copy_(torch.log(h_node.clamp(min=1e-30))), st.log_edge[0].copy_(torch.log(h_edge.clamp(min=1e-30)))
        st.node_traj[0], st.pos_traj[0], st.edge_traj[0] = h_node, p0, h_edge
        gn, ge, sc = torch.zeros(S, N, device=DEV), torch.zeros(S, E, device=DEV), torch.zeros(S, 2, device=DEV)
        for i in range(S):
            un, ue, eps = torch.rand(N, 12), torch.rand(E, 6), torch.randn(N, 3)
            un, ue = un[ia].to(DEV), ue[ie].to(DEV)
            model.reverse_step(st, i, 999 - i, None, draws=(un, ue, eps[ia]))
            gn[i], ge[i] = margins(un, st.log_node[st.cur]), margins(ue, st.log_edge[st.cur])
            sc[i, 0], sc[i, 1] = w.out_v.abs().max(), w.out_bond.abs().max()
        torch.cuda.synchronize()

        def back(v, perm):
            if perm is None:
                return v
            out = torch.empty_like(v)
            out[:, perm.to(v.device)] = v
            return out
        return dict(node=back(st.node_traj[:S + 1].argmax(-1), pa).cpu().numpy(), edge=back(st.edge_traj[:S + 1].argmax(-1), pe).cpu().numpy(),
                    pos=back(st.pos_traj[:S + 1], pa).cpu().numpy(), gap_node=back(gn, pa).cpu().numpy(), gap_edge=back(ge, pe).cpu().numpy(), scale=sc.cpu().numpy())
    t0 = time.time()
    a = run(None, None)
    b = run(perm_atom, perm_edge)
    secs = time.time() - t0
    cmp_ = compare(b, a)
    label(cmp_['departures'], margins_from(b), margins_from(a), 'hip_permuted', 'hip')
    return dict(what='the HIP sampler against itself with the atoms (hence bond rows and triplet order) of every ligand permuted, same draws per atom / bond',
                workload='headline shape: first %d graphs of bench.ligphore_workload(128)' % B if HEADLINE else 'P03211 pharmacophore x %d ligands of 12-27 atoms' % B,
                graphs=B, steps=S, atoms=N, bond_rows=E, both_runs_s=round(secs, 1), **cmp_)


def run_report():
    ref = np.load(CACHE('ref'))
    out = {'graphs': B, 'steps': S, 'atoms': N, 'bond_rows': E, 'tie_rule': f'every flipped row: top-2 margin <= {FLIP_GAP_MULT:g} x max |logit| of the step in both runs'}
    sub = np.load(CACHE('sub')) if os.path.exists(CACHE('sub')) else None
    if sub is not None:
        out['oracle_subset_rerun'] = dict(graphs=sub['graphs'].tolist(), first_frame_not_bit_identical_to_the_full_run=dict(zip(map(int, sub['graphs']), map(int, sub['first_diff']))),
                                          note=f'{S + 1} = identical through the whole run; margins of a graph are used only before that frame')
    if os.path.exists(HIP_OUT):
        hip = np.load(HIP_OUT)
        cmp_ = compare(hip, ref)
        label(cmp_['departures'], margins_from(hip, 'rows'), margins_from(sub, 'rows'), 'hip', 'oracle')
        out['hip_vs_oracle'] = dict(hip_s=round(float(hip['hip_s']), 1), oracle_s=round(float(ref['oracle_s']), 1), **cmp_, min_margin_hip=hip['min_margin'].tolist(),
                                    randn_checksum_max_dev_between_hosts=float(hip['randn_dev']))
    if os.path.exists(CACHE('perm')):
        perm = np.load(CACHE('perm'))
        assert np.allclose(perm['sums'], ref['sums'], rtol=1e-12, atol=1e-9), 'generator stream differs between the `ref` and `perm` hosts'
        cmp_ = compare(perm, ref)
        label(cmp_['departures'], margins_from(perm), margins_from(sub, 'rows'), 'permuted_oracle', 'oracle')
        out['control_permuted_oracle_vs_oracle'] = dict(what='the oracle (fp32, CPU) with the atoms of every ligand permuted -- same draws per atom / bond -- against the plain oracle: '
                                                        'two fp32 evaluations of the reference dataflow that differ only in the summation order inside bond / triplet segments',
                                                        oracle_s=round(float(perm['oracle_s']), 1), **cmp_)
    return out


if phase == 'ref':
    out = run_ref()
    print(json.dumps({'wrote': os.path.relpath(CACHE('ref'), ROOT), 'oracle_s': round(float(out['oracle_s']), 1)}))
elif phase == 'perm':
    out = run_perm()
    print(json.dumps({'wrote': os.path.relpath(CACHE('perm'), ROOT), 'oracle_s': round(float(out['oracle_s']), 1)}))
elif phase == 'sub':
    same, secs = run_sub(SUBSET)
    print(json.dumps({'wrote': os.path.relpath(CACHE('sub'), ROOT), 'graphs': SUBSET, 'identical_to_full_run': same, 'oracle_s': round(secs, 1)}))
elif phase == 'hip':
    print(json.dumps(run_hip(np.load(CACHE('ref')))))
elif phase == 'report':
    print(json.dumps(run_report()))
elif phase == 'hipperm':
    print(json.dumps(run_hipperm()))
else:
    print(json.dumps(run_hip(run_ref())))
