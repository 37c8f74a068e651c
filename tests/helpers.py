"""Shared test helpers: golden loading, deterministic state_dict, default config."""
import os

import numpy as np
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')

# the `model.diff` block of configs/train_lig-phore.yml:15-41 (identical in both shipped configs)
DIFF_CFG = yaml.safe_load("""
num_timesteps: 1000
time_dim: 10
categorical_space: discrete
diff_pos: {beta_schedule: advance, scale_start: 0.9999, scale_end: 0.0001, width: 3}
diff_atom: {init_prob: tomask, beta_schedule: advance, scale_start: 0.9999, scale_end: 0.0001, width: 3}
diff_bond:
  init_prob: absorb
  beta_schedule: segment
  time_segment: [600, 400]
  segment_diff:
    - {scale_start: 0.9999, scale_end: 0.001, width: 3}
    - {scale_start: 0.001, scale_end: 0.0001, width: 2}
""")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'))


def manifest():
    out = []
    with open(os.path.join(GOLDEN, 'g7_state_dict_manifest.txt')) as f:
        for line in f:
            k, shape, dt = line.rstrip('\n').split('\t')
            out.append((k, tuple(int(s) for s in shape.strip('()').split(',') if s.strip()), dt))
    return out


def oracle_state_dict(seed=0, profile='default'):
    """state_dict built WITHOUT any model class: manifest shapes + weights.make_tensor + oracle tables."""
    from oracle import phoregen_oracle as po
    from phoregen_amd.weights import is_fixed, make_tensor
    T = DIFF_CFG['num_timesteps']
    fixed = {}
    for k, v in po.continuous_tables(po.beta_schedule(T, DIFF_CFG['diff_pos'])).items():
        fixed['pos_transition.' + k] = v
    for tag, K, c in (('node', 12, DIFF_CFG['diff_atom']), ('edge', 6, DIFF_CFG['diff_bond'])):
        tb = po.categorical_tables(po.beta_schedule(T, c), K, c['init_prob'])
        fixed[f'{tag}_transition.q_mats'] = tb['q_mats']
        fixed[f'{tag}_transition.transpopse_q_onestep_mats'] = tb['transpopse_q_onestep_mats']
    sd = {}
    for k, shape, dt in manifest():
        if k in fixed:
            sd[k] = fixed[k]
        elif k.endswith('.freq_bands'):
            sd[k] = torch.tensor([1., 2., 3., 1., 1. / 2, 1. / 3])
        elif k == 'distance_expansion.offset':
            sd[k] = torch.linspace(0., 5., 20)
        elif k.endswith('distance_expansion.offset'):
            sd[k] = torch.tensor(po.SMEAR_OFFSETS, dtype=torch.float32)
        elif k == 'time_emb.0.offset':
            sd[k] = torch.linspace(0., 1000., 10)
        elif k == 'time_emb.0.coeff':
            d = torch.diff(torch.linspace(0., 1000., 10))
            sd[k] = -0.5 / torch.cat([d[:1], d]) ** 2
        else:
            assert not is_fixed(k), k
            sd[k] = make_tensor(k, shape, seed, profile=profile)
        assert tuple(sd[k].shape) == shape, (k, sd[k].shape, shape)
    return sd


def make_oracle(seed=0, profile='default'):
    from oracle import phoregen_oracle as po
    return po.Oracle(oracle_state_dict(seed, profile), diff_cfg=DIFF_CFG)


def t(a):
    return torch.as_tensor(np.asarray(a))


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


# ---- conditioning floor of the recorded sampler steps (oracle/make_conditioning_floor.py; frozen, kernel-independent) ----
FLOOR_MULT = 3.0       # fixed: |hip - reference| <= max(5 x TOL, FLOOR_MULT x floor); never retuned for a kernel
_FLOOR = None


def conditioning_floor(name, s):
    """[v, x0, bond] ensemble floor of recorded step `s` of sampler fixture `name`: the largest distance of K = 12 fp32
    evaluations of the reference's dataflow (atom / bond rows permuted, coordinates moved by <= 1 ulp) from the float64 result."""
    global _FLOOR
    if _FLOOR is None:
        import json
        with open(os.path.join(GOLDEN, 'conditioning_floor.json')) as f:
            _FLOOR = json.load(f)
    return _FLOOR['floor'][name][s]


# ---- aggregate guard over a fixture's checked steps (round 4): the per-step bound max(5 x TOL, 3 x floor) is loose wherever
# 3 x floor > 5 x TOL, so a UNIFORM loss of precision in a kernel (say 10 x) would pass every single step.  The median of
# err / max(floor, 1e-6) over all (step, output) pairs of a fixture must stay below the value frozen here: 4 x the median measured
# in round 3 (profiles/r03_parity_ratio_table.md: 0.04 .. 0.61), at least 0.3, at most 1.0.  Frozen like the floor table: derived
# from recorded measurements once, not retuned for a kernel.  tools/degraded_build_check.sh shows a build whose triplet kernel
# drops 9 mantissa bits of its activations failing this guard while passing most per-step bounds.
AGG_RATIO_EPS = 1e-6
AGG_MEDIAN_BOUND = {
    ('teacher_forced', 'g5_sample_head3'): 0.75, ('teacher_forced', 'g5_sample_tail4'): 0.30,
    ('teacher_forced', 'g5_sample_full25'): 0.31, ('teacher_forced', 'g5_sample_guid3'): 0.42,
    ('teacher_forced', 'g5_sample_head3_gamma_signed'): 1.0, ('teacher_forced', 'g5_sample_tail4_gamma_signed'): 0.64,
    ('teacher_forced', 'g5_sample_head3_trained_like'): 1.0, ('teacher_forced', 'g5_sample_tail4_trained_like'): 1.0,
    ('closed_loop', 'g5_sample_head3'): 0.91, ('closed_loop', 'g5_sample_head3_gamma_signed'): 0.97,
    ('closed_loop', 'g5_sample_head3_trained_like'): 1.0,
}


def assert_aggregate_parity(test, name, ratios):
    """`ratios` = err / max(floor, AGG_RATIO_EPS) of every (step, output) pair checked for fixture `name`."""
    med = float(np.median(np.asarray(ratios, dtype=np.float64)))
    bound = AGG_MEDIAN_BOUND[(test, name)]
    assert med <= bound, f'{test} {name}: median(err / floor) = {med:.3f} over {len(ratios)} (step, output) pairs exceeds the frozen ' \
                         f'{bound} -- a uniform precision regression (per-step bounds may all still hold)'
    return med


def record_parity_ratio(test, name, s, errs, floors, tol):
    """Evidence trail (gpurun_out/ is scratch; the table judged is copied to profiles/): one JSON line per checked step."""
    import json
    d = os.path.join(ROOT, 'gpurun_out')
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, 'parity_ratios.jsonl'), 'a') as f:
            f.write(json.dumps(dict(test=test, fixture=name, step=s, err=[float(e) for e in errs], floor=list(floors),
                                    tol=[float(x) for x in tol])) + '\n')
    except OSError:
        pass


class Oracle64:
    """The oracle's dataflow in float64 (same fp32-generated weights, exactly converted): the "exact" result against which
    the conditioning of a weight set is measured -- rel_err(oracle fp32, oracle fp64) is the error ANY fp32 implementation
    of the reference's dataflow carries on that input, and bounds what can be asked of the HIP path there."""

    def __init__(self, seed=0, profile='default'):
        from oracle import phoregen_oracle as po
        sd = oracle_state_dict(seed, profile)                 # drawn in float32 (default dtype), then widened
        self.o = po.Oracle(sd, diff_cfg=DIFF_CFG, dtype=torch.float64)

    def forward(self, **inp):
        old = torch.get_default_dtype()
        torch.set_default_dtype(torch.float64)
        try:
            with torch.no_grad():
                return self.o.forward(**{k: (v.double() if v.is_floating_point() else v) for k, v in inp.items()})
        finally:
            torch.set_default_dtype(old)
