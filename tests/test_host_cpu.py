"""CPU-side checks of the host mirror: state_dict schema, tables, packing layouts, C-ABI exports, plan."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from helpers import ROOT, golden, manifest, oracle_state_dict
from oracle import phoregen_oracle as po
from phoregen_amd import hip, packing
from phoregen_amd.config import default_model_config
from phoregen_amd.models.diffusion import PhoreDiff
from phoregen_amd.plan import BatchPlan, make_edge_data
from phoregen_amd.weights import init_deterministic_


@pytest.fixture(scope='module')
def model():
    return init_deterministic_(PhoreDiff(default_model_config(), 'zinc_300'), 0).eval()


def test_state_dict_matches_reference_manifest(model):
    sd = model.state_dict()
    ref = manifest()
    assert [k for k, _, _ in ref] == list(sd.keys())                 # same names, same order (641 entries)
    for k, shape, dt in ref:
        assert tuple(sd[k].shape) == shape and str(sd[k].dtype).replace('torch.', '') == dt, k


def test_tables_and_weights_equal_oracle_state_dict(model):
    sd, osd = model.state_dict(), oracle_state_dict(0)
    for k in sd:
        assert torch.equal(sd[k], osd[k]), k                          # tables bit-exact vs the (golden-pinned) oracle


def test_frozen_tables_not_trainable(model):
    n_train = sum(p.numel() for p in model.parameters() if p.requires_grad)
    n_all = sum(p.numel() for p in model.parameters())
    assert (n_all, n_train) == (5568785, 5201785)


def test_product_fails_loudly_without_gpu(model):
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        model.packed()
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        model.compute_loss(None)


def test_library_exports_every_declared_symbol():
    lib = hip.load_library()
    header = open(os.path.join(ROOT, 'include', 'phoregen_hip.h')).read()
    declared = set(re.findall(r'\b(pg_[a-z_0-9]+)\s*\(', header))
    assert declared == set(hip.EXPORTS), declared ^ set(hip.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.pg_abi_version() == 11 == hip.ABI_VERSION
    import ctypes as _C
    sizes = (_C.c_int * 5)()
    assert lib.pg_abi_struct_sizes(sizes, 5) == 5      # (load_library has already held the ctypes mirrors against these)
    assert list(sizes) == [_C.sizeof(x) for x in (hip.PgGemm, hip.PgTopo, hip.PgSegAttn, hip.PgSegAttnGrad, hip.PgLaunch)]
    assert isinstance(lib.pg_last_error(), bytes)


def test_driver_build_entry_point_runs():
    """__graft_entry__.build() is what the driver calls on a GPU-less host every round: make (a no-op on an up-to-date tree), load the
    library, check its ABI version against the binding, import the package.  (It once carried a literal version number and would have
    failed the round's build check after an ABI bump.)"""
    import __graft_entry__ as entry
    entry.build()


def test_streaming_gemm_isa_keeps_its_counted_wait_valid(tmp_path):
    """csrc/gemm_stream.hip retires its LDS-DMA with `s_waitcnt vmcnt(32)`: correct only while the 32 result stores of a tile are
    the ONLY younger vector-memory operations of a wave (fewer would let the wait pass before the DMA lands).  Cross-compile
    the file and check, for every variant that uses the DMA: no scratch (spill traffic would be uncounted VMEM), exactly 32
    buffer_store_dword per tile body (three bodies per kernel: first tile, stage 1, stage 0), 8 or 4 DMA pieces per body."""
    import shutil, subprocess
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        pytest.skip('no hipcc')
    out = tmp_path / 'gs.s'
    subprocess.run([hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=on', '--cuda-device-only', '-S',
                    os.path.join(ROOT, 'phoregen_amd', 'csrc', 'gemm_stream.hip'), '-o', str(out)], check=True,
                   stderr=subprocess.DEVNULL)
    lines = out.read_text().split('\n')
    starts = [i for i, l in enumerate(lines) if re.match(r'_ZN2pg18gemm_stream_kernelI.*PgGemmi:', l)]
    assert len(starts) >= 12
    for i in starts:
        nw, nadd, k1, k2, ln = re.match(r'_ZN2pg18gemm_stream_kernelILi(\d)ELi(\d)ELi(\d+)ELi(\d+)ELb([01])E', lines[i]).groups()
        end = next(j for j in range(i, len(lines)) if 's_endpgm' in lines[j])
        scratch = next(l for l in lines[end:] if '; ScratchSize:' in l)
        body = '\n'.join(lines[i:end])
        assert scratch.split(':')[1].strip() == '0', (lines[i], scratch)
        if k1 == '128':
            assert len(re.findall(r'buffer_store_dword ', body)) == 96, lines[i]
            assert len(re.findall(r'buffer_load_dwordx4 .* lds', body)) == 4 * (8 if nw == '4' else 4), lines[i]
            assert len(re.findall(r's_waitcnt vmcnt\(32\)\n\ts_barrier', body)) == 2, lines[i]


def test_product_never_imports_oracle():
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'phoregen_amd')):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                src = open(os.path.join(dirpath, f)).read()
                if re.search(r'^\s*(from|import)\s+oracle\b', src, re.M) or 'root/reference' in src:
                    bad.append(f)
    assert not bad, bad


def test_make_edge_data_matches_reference_order():
    g = golden('g1_ops')
    ei, eb = make_edge_data(torch.as_tensor(g['med_num_atoms']))
    assert np.array_equal(ei.numpy(), g['med_edge_index']) and np.array_equal(eb.numpy(), g['med_edge_batch'])
    na = torch.tensor([1, 7, 2, 12, 3])
    a, b = make_edge_data(na)
    c, d = po.make_edge_data(na)
    assert torch.equal(a, c) and torch.equal(b, d)


def test_plan_topology_matches_compose_context():
    g = golden('g3_forward_b')
    bn, bp = torch.as_tensor(g['in_batch_node']), torch.as_tensor(g['in_batch_phore'])
    ei, be = torch.as_tensor(g['in_edge_index']), torch.as_tensor(g['in_batch_edge'])
    plan = BatchPlan(bn, bp, ei, be, 3, torch.device('cpu'))
    assert np.array_equal(plan.ctx_is_lig.numpy().astype(bool), g['L0_in_mask_ligand'])
    assert np.array_equal(plan.ctx_graph.numpy(), g['L0_in_batch'])
    # bond rows: the plan's internal order is target-major per graph; edge_ref / edge_int translate to the caller's rows
    ref, inv = plan.edge_ref.long(), plan.edge_int.long()
    assert torch.equal(ref[inv], torch.arange(plan.n_bond)) and not plan.edge_identity
    assert np.array_equal(torch.stack([plan.bond_src, plan.bond_dst])[:, inv].numpy(), g['L0_in_bond_index'])
    d, s_ = plan.bond_dst.long(), plan.bond_src.long()
    key = (d[1:] > d[:-1]) | ((d[1:] == d[:-1]) & (s_[1:] > s_[:-1]))
    assert bool(key.all())                                             # sorted by (target, source): one block per target
    # edge-id table enumerates exactly the reference triplets (uni_denoiser.py:101-121)
    i, j, k, kj, ji = po.triplets(torch.as_tensor(g['L0_in_bond_index']), plan.n_ctx)
    eid, off, nl = plan.eid.numpy(), plan.g_eid_off.numpy(), plan.g_nlig.numpy()
    lig0 = (plan.g_ctx_off[:-1] + plan.g_nph).numpy()
    gr = plan.ctx_graph.numpy()
    mine = []
    for e_ref in range(plan.n_bond):                                  # segments in the caller's edge order, like the reference
        e = int(inv[e_ref])
        cj, ci = int(plan.bond_src[e]), int(plan.bond_dst[e])
        gi = gr[cj]
        n, lj, li = nl[gi], cj - lig0[gi], ci - lig0[gi]
        for kk in range(n):
            if kk != lj and kk != li:
                mine.append((lig0[gi] + kk, int(ref[eid[off[gi] + kk * n + lj]]), e_ref))
    assert mine == list(zip(k.tolist(), kj.tolist(), ji.tolist()))
    # FeaturizeLigandBond's order (datasets/transform.py:488-501) already is the internal one
    from oracle.make_inputs import synthetic_train_batch
    tb = synthetic_train_batch(3, [5, 7], [6, 9])
    assert BatchPlan(tb['ligand_batch'], tb['phore_batch'], tb['f_edge_index'], tb['f_edge_batch'], 2, torch.device('cpu')).edge_identity
    with pytest.raises(ValueError):
        BatchPlan(bn, bp, ei[:, :-1], be[:-1], 3, torch.device('cpu'))


def test_lane_fixed_layouts():
    W = torch.arange(128 * 128, dtype=torch.float32).view(128, 128)
    L = packing.lane_fixed_w2(W)
    assert L.shape == (64, 64, 4)
    for (i, lane, j) in [(0, 0, 0), (5, 17, 3), (63, 63, 3), (34, 40, 1)]:
        n, g, h = i * 4 + j, lane >> 4, lane & 15
        tau, r, d = n >> 5, (n >> 3) & 3, n & 7
        assert L[i, lane, j] == W[8 * h + d, 16 * tau + 4 * g + r]
    F = torch.arange(128 * 48, dtype=torch.float32).view(128, 48)
    LF = packing.lane_fixed_feat(F)
    assert LF.shape == (12, 8, 64)
    for (st, tau, lane) in [(0, 0, 0), (11, 7, 63), (4, 2, 37)]:
        assert LF[st, tau, lane] == F[16 * tau + (lane & 15), 4 * st + (lane >> 4)]
    X = torch.arange(16 * 128, dtype=torch.float32).view(16, 128)
    LX = packing.lane_fixed_xv(X)
    for (i, lane) in [(0, 0), (31, 63), (9, 21)]:
        assert LX[i, lane] == X[lane & 15, 16 * (i >> 2) + 4 * (lane >> 4) + (i & 3)]


def test_factored_first_layer_algebra(model):
    """packing's column maps: factored pieces re-assemble the reference's concatenated first layer."""
    sd = {k: v.double() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(3)
    p = 'denoiser.base_block.2'
    # triplet k: [h_bond_kj | smear(d_kj) | smear(d_ji) | ang13 | h_k | h_j]
    W1, b1 = sd[p + '.bond_layer.hk_func.net.0.weight'], sd[p + '.bond_layer.hk_func.net.0.bias']
    hb, gkj, gji, hk, hj = (torch.randn(n, generator=g, dtype=torch.float64) for n in (128, 20, 20, 128, 128))
    th = torch.rand(1, generator=g, dtype=torch.float64) * 3.1
    ang = po.angular_encoding(th)[0]
    ref = W1 @ torch.cat([hb, gkj, gji, ang, hk, hj]) + b1
    f = torch.tensor([1., 2., 3., .5, 1 / 3], dtype=torch.float64)
    feat = torch.cat([th, torch.sin(th * f), torch.cos(th * f), torch.zeros(1, dtype=torch.float64)])
    Wf = packing._tri_feat(W1)
    mine = (W1[:, 0:148] @ torch.cat([hb, gkj]) + W1[:, 181:309] @ hk + (W1[:, 309:437] @ hj + b1)
            + W1[:, 148:168] @ gji + Wf @ feat)
    assert torch.allclose(mine, ref, atol=1e-10)
    # knn k, dst ligand, src phore (type 2)
    W1, b1 = sd[p + '.node_layer_with_edge.hk_func.net.0.weight'], sd[p + '.node_layer_with_edge.hk_func.net.0.bias']
    Wd, bd = sd[p + '.dire_embedding.weight'], sd[p + '.dire_embedding.bias']
    sm, dots, hd, hs = (torch.randn(n, generator=g, dtype=torch.float64) for n in (20, 3, 128, 128))
    et = torch.zeros(4, dtype=torch.float64)
    et[2] = 1
    e = torch.cat([(et[:, None] * sm[None]).reshape(-1), et, Wd @ dots + bd])
    ref = W1 @ torch.cat([e, hd, hs]) + b1
    Wfe = packing._knn_feat(W1, W1[:, 84:93] @ Wd, True)
    feat = torch.cat([torch.zeros(20, dtype=torch.float64), sm, dots, torch.tensor([0., 1., 0., 0., 0.], dtype=torch.float64)])
    mine = Wfe @ feat + W1[:, 93:221] @ hd + b1 + W1[:, 84:93] @ bd + W1[:, 221:349] @ hs
    assert torch.allclose(mine, ref, atol=1e-10)


@pytest.mark.parametrize('profile', ['default', 'gamma_signed', 'trained_like'])
def test_folded_layernorm_rewrite_is_exact_for_every_weight_profile(profile):
    """packing._kv_mlp (centred first layer, sign fold, |gamma| into the second Linear, per-row rstd, dead channels) against
    the plain MLP of models/common.py:99-119 (Linear -> LayerNorm -> ReLU -> Linear) in float64, on LayerNorms with negative,
    tiny and exactly-zero gamma (`gamma_signed`) and on trained-like scales."""
    from phoregen_amd.weights import make_tensor
    p = 'denoiser.base_block.1.bond_layer.hv_func'
    shapes = {'.net.0.weight': (128, 437), '.net.0.bias': (128,), '.net.1.weight': (128,), '.net.1.bias': (128,),
              '.net.3.weight': (128, 128), '.net.3.bias': (128,)}
    sd = {p + k: make_tensor(p + k, s, 0, profile=profile).double() for k, s in shapes.items()}
    gam = sd[p + '.net.1.weight']
    if profile == 'gamma_signed':
        assert (gam == 0).sum() >= 3 and (gam < 0).sum() > 30 and (gam.abs() < 1e-5).sum() >= 6
    x = torch.randn(50, 437, generator=torch.Generator().manual_seed(1), dtype=torch.float64) * 3
    hid = x @ sd[p + '.net.0.weight'].t() + sd[p + '.net.0.bias']
    ref = torch.relu(torch.nn.functional.layer_norm(hid, (128,), gam, sd[p + '.net.1.bias'])) @ sd[p + '.net.3.weight'].t() \
        + sd[p + '.net.3.bias']
    m = packing._kv_mlp(sd, p)
    h = x @ m['W1'].t() + m['b1']                                   # centred, sign-normalised
    sgn = torch.where(gam < 0, -1.0, 1.0).double()
    assert float((h * sgn).mean(-1).abs().max()) < 1e-12        # centred before the sign fold: no mean pass in the kernels
    var = (h * h).mean(-1, keepdim=True) + 1e-5
    sigma, rstd = var.sqrt(), var.rsqrt()
    z = torch.relu(h + m['bp'] * sigma)                             # what the kernels evaluate per element
    mine = rstd * (z @ m['W2'].t()) + m['b2']
    assert torch.isfinite(mine).all()
    assert float((mine - ref).abs().max()) <= 1e-9 * float(ref.abs().max())


@pytest.mark.parametrize('profile', ['default', 'gamma_signed'])
def test_stacked_packing_equals_layer_by_layer_packing(profile):
    """The training path packs the six layers at once (packing.LayerPack on a list of prefixes: stacked parameters, one pass of
    tensor ops) and differentiates through the packing; the sampler packs layer by layer.  Both give the same kernel-layout
    tensors (to rounding: a batched product replaces a matrix-vector one), the per-layer views are what the kernels can read
    (unit inner stride, contiguous 128-row slabs), and the gradient of a random functional of ALL packed tensors reaches the
    parameters identically through either route."""
    m = init_deterministic_(PhoreDiff(default_model_config(), 'zinc_300'), 0, profile=profile).double()
    sd = {k: v.clone().requires_grad_(v.is_floating_point() and 'transition' not in k) for k, v in m.state_dict().items()}
    ref = packing.ModelPack({k: v.detach() for k, v in sd.items()}, 6)
    one = packing.ModelPack(sd, 6, detach=False)
    gen = torch.Generator().manual_seed(0)
    pairs = []

    def walk(a, b, path):
        if torch.is_tensor(a):
            assert a.shape == b.shape, path
            assert torch.allclose(a, b.detach(), rtol=1e-12, atol=1e-14), path
            assert b.stride(-1) == 1 or b.numel() <= 1, path
            pairs.append((path, b))
        elif isinstance(a, dict):
            for k in a:
                walk(a[k], b[k], f'{path}[{k}]')
        elif isinstance(a, (tuple, list)):
            for i, (x, y) in enumerate(zip(a, b)):
                walk(x, y, f'{path}[{i}]')
        elif hasattr(a, '__dict__'):
            for k, v in vars(a).items():
                if hasattr(b, k) and getattr(b, k) is not None:
                    walk(v, getattr(b, k), f'{path}.{k}')
    ref.gate = None
    walk(ref, one, 'pk')
    assert len(pairs) > 300
    for L in one.layers:
        assert L.W_node1.shape == (1920, 128) and L.W_node1.is_contiguous() and L.W_node2.shape == (1280, 128)
        for (c0, c1), (w, b) in list(L.node1_parts.items()) + list(L.node2_parts.items()):
            assert w.shape == (c1 - c0, 128) and w.is_contiguous() and b.shape == (c1 - c0,)
    # gradient through the stacked route == gradient through the layer-by-layer route (same random cotangents)
    cot = {path: torch.randn(t.shape, generator=gen, dtype=t.dtype) for path, t in pairs if t.requires_grad}
    names = [k for k, v in sd.items() if v.requires_grad]
    sd2 = {k: v.detach().clone().requires_grad_(v.requires_grad) for k, v in sd.items()}
    two = packing.ModelPack.__new__(packing.ModelPack)
    two.layers = [packing.LayerPack(sd2, f'denoiser.base_block.{l}') for l in range(6)]
    two.PH, b_ph = packing.pack_phore(sd2)
    two.W_ph, two.b_ph = packing.fuse_blocks(b_ph)
    two.ph_parts = packing._parts(two.W_ph, two.b_ph, packing.PHORE_PARTS)
    pairs2 = {}

    def walk2(b, path):
        if torch.is_tensor(b):
            pairs2[path] = b
        elif isinstance(b, dict):
            for k in b:
                walk2(b[k], f'{path}[{k}]')
        elif isinstance(b, (tuple, list)):
            for i, y in enumerate(b):
                walk2(y, f'{path}[{i}]')
        elif hasattr(b, '__dict__'):
            for k, v in vars(b).items():
                walk2(v, f'{path}.{k}')
    walk2(two, 'pk')
    loss2 = sum((pairs2[path] * c).sum() for path, c in cot.items() if path in pairs2)
    missing = [path for path in cot if path not in pairs2]
    assert all(not path.startswith('pk.layers') and not path.startswith('pk.PH') for path in missing), missing[:5]
    loss1 = sum((t * cot[path]).sum() for path, t in pairs if t.requires_grad and path in pairs2)
    g_one = torch.autograd.grad(loss1, [sd[k] for k in names], allow_unused=True)
    g_two = torch.autograd.grad(loss2, [sd2[k] for k in names], allow_unused=True)
    n_checked = 0
    for k, a, b in zip(names, g_one, g_two):
        assert (a is None) == (b is None), k
        if a is not None:
            assert torch.allclose(a, b, rtol=1e-9, atol=1e-12), k
            n_checked += 1
    assert n_checked > 400


def test_phore_parser_matches_reference(tmp_path):
    """phoregen_amd.data.parse_phore_file against datasets/get_phore_data.py run on a shipped .phore file."""
    from phoregen_amd.data import parse_phore_file
    g = golden('g8_phore_parse')
    f = tmp_path / 'P03211_merge.phore'
    f.write_bytes(g['file_text'].tobytes())
    d = parse_phore_file(str(f))
    assert np.array_equal(d['phore'].x.numpy(), g['x']) and np.array_equal(d['phore'].norm.numpy(), g['norm'])
    assert np.allclose(d['phore'].pos.numpy(), g['pos'], rtol=0, atol=1e-6) and np.allclose(d.center.numpy(), g['center'], atol=1e-6)
    assert d.name == 'P03211_merge'


def test_unbatch_and_decode_match_reference():
    from phoregen_amd.utils.sample_utils import decode_data, unbatch_data
    g = golden('g9_unbatch_decode')
    res = {'pred': [torch.as_tensor(g[f'pred{i}']) for i in range(3)],
           'traj': [torch.as_tensor(g[f'traj{i}']) for i in range(3)],
           'lig_info': [torch.as_tensor(g['na']), torch.as_tensor(g['bn']), torch.as_tensor(g['ei']), torch.as_tensor(g['eb'])]}
    outs = unbatch_data(res, 3)
    for gi, o in enumerate(outs):
        assert np.array_equal(o['edge_index'].numpy(), g[f'g{gi}_edge_index'])
        assert np.array_equal(o['traj'][1].numpy(), g[f'g{gi}_traj1'])
        d = decode_data(o['pred'], o['edge_index'])
        assert d['element'] == g[f'g{gi}_element'].tolist()
        assert np.array_equal(d['atom_pos'].numpy(), g[f'g{gi}_atom_pos'])
        assert np.array_equal(d['bond_type'].numpy(), g[f'g{gi}_bond_type'])
        assert np.array_equal(d['bond_index'].numpy(), g[f'g{gi}_bond_index'])
    # the one-pass batch decode gives the same molecules as the reference's per-graph unbatch + decode
    from phoregen_amd.utils.sample_utils import decode_batch
    for gi, d in enumerate(decode_batch(res)):
        assert d['element'] == g[f'g{gi}_element'].tolist()
        assert np.array_equal(d['atom_pos'].numpy(), g[f'g{gi}_atom_pos'])
        assert np.array_equal(d['bond_type'].numpy(), g[f'g{gi}_bond_type'])
        assert np.array_equal(d['bond_index'].numpy(), g[f'g{gi}_bond_index'])


def test_ema_matches_reference_formula():
    """models/model_utils.py:21-42: shadow = beta * shadow + (1 - beta) * current, state round trip."""
    from phoregen_amd.models.model_utils import EMA
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(5, 4), torch.nn.Linear(4, 2))
    ema = EMA(0.9, net.parameters())
    ref = [p.detach().clone() for p in net.parameters()]
    for _ in range(3):
        with torch.no_grad():
            for p in net.parameters():
                p.add_(torch.randn_like(p))
        ema.update_model_average(net)
        ref = [r * 0.9 + 0.1 * p.detach() for r, p in zip(ref, net.parameters())]
    assert all(torch.allclose(a, b, atol=1e-6) for a, b in zip(ema.shadow_params, ref))
    e2 = EMA(0.5, net.parameters())
    e2.load_state_dict(ema.state_dict(), 'cpu')
    assert e2.beta == 0.9 and all(torch.equal(a, b) for a, b in zip(e2.shadow_params, ema.shadow_params))


def test_loss_dict_reads_as_python_floats():
    """compute_loss returns its metrics as 0-dim tensors that convert on first read (the reference returns floats,
    diffusion.py:333-350): every access path of a dict must yield floats."""
    from phoregen_amd.models.diffusion import _LazyFloats
    d = _LazyFloats({'loss': torch.tensor(1.5), 'node_acc': torch.tensor(0.25), 'n': 3.0})
    assert isinstance(d['loss'], float) and d['loss'] == 1.5
    assert d.get('node_acc') == 0.25 and d.get('missing', -1.0) == -1.0
    assert dict(d.items()) == {'loss': 1.5, 'node_acc': 0.25, 'n': 3.0}
    assert sorted(d.values()) == [0.25, 1.5, 3.0]
    assert 'loss' in repr(d) and 'tensor' not in repr(d)
    assert '%.3f' % d['loss'] == '1.500'


def test_plan_and_partition_properties_random_sizes():
    """Property checks over random batches (hypothesis): edge-id table, source-ordered triplet visiting order, chunk cover,
    make_edge_data equals the oracle's reference-order construction, partition completeness."""
    from hypothesis import given, settings, strategies as st
    from oracle import phoregen_oracle as po
    from phoregen_amd.parallel import partition_graphs
    from phoregen_amd.plan import make_edge_data

    @settings(max_examples=25, deadline=None)
    @given(st.lists(st.integers(min_value=1, max_value=23), min_size=1, max_size=6),
           st.lists(st.integers(min_value=1, max_value=9), min_size=6, max_size=6), st.integers(min_value=1, max_value=5))
    def check(n_atoms, n_ph, world):
        na = torch.tensor(n_atoms)
        ei, be = make_edge_data(na)
        ei_ref, be_ref = po.make_edge_data(na)
        assert torch.equal(ei, ei_ref) and torch.equal(be, be_ref)
        B = na.numel()
        bn = torch.repeat_interleave(torch.arange(B), na)
        bp = torch.repeat_interleave(torch.arange(B), torch.tensor(n_ph[:B]))
        plan = BatchPlan(bn, bp, ei, be, B, 'cpu')
        E = ei.size(1)
        off = torch.zeros(B + 1, dtype=torch.long)
        off[1:] = na.cumsum(0)
        eo = plan.g_eid_off.long()
        inv, ref = plan.edge_int.long(), plan.edge_ref.long()
        assert sorted(inv.tolist()) == list(range(E)) and (E == 0 or torch.equal(ref[inv], torch.arange(E)))
        for e in range(0, E, max(1, E // 50)):          # eid[src, dst] is the (internal) id of the caller's edge e
            g = int(be[e])
            n = int(na[g])
            ls, ld = int(ei[0, e] - off[g]), int(ei[1, e] - off[g])
            assert int(plan.eid[eo[g] + ls * n + ld]) == int(inv[e])
            # internal rows: per graph target-major, source ascending -> closed form
            assert int(inv[e]) == int(plan.g_bond_off[g]) + ld * (n - 1) + (ls if ls < ld else ls - 1)
        order = plan.tri_order.long()
        assert sorted(order.tolist()) == list(range(E))
        src_sorted = plan.bond_src.long()[order]
        assert bool((src_sorted[1:] >= src_sorted[:-1]).all()) if E > 1 else True
        ch = plan.tri_chunks.long()
        assert int(ch[0]) == 0 and int(ch[-1]) == E and bool((ch[1:] >= ch[:-1]).all())
        parts = partition_graphs(na, world)
        assert sorted(torch.cat(parts).tolist()) == list(range(B))
        # queue entries of the staged triplet kernel: every segment (bond edge j->i) of every ligand is handed out exactly once,
        # whole groups or parts of them; a part starts on a 12-wave round boundary
        if plan.n_tri_iters:
            seen = torch.zeros(E, dtype=torch.int32)
            l2c = plan.lig2ctx.long().tolist()
            first_ctx = {l2c[int(off[g])]: g for g in range(B) if int(na[g]) > 0}
            for lig0, packed, boff, rng in plan.tri_iters.tolist():
                n, j0, A = packed & 0xff, (packed >> 8) & 0xff, packed >> 16
                g = first_ctx[lig0]
                assert n == int(na[g]) and boff == int(plan.g_bond_off[g]) and A * (n - 1) <= 80 and j0 + A <= n
                s0, s1 = rng & 0xffff, rng >> 16
                if rng == 0:
                    s0, s1 = 0, A * (n - 1)
                assert 0 <= s0 < s1 <= A * (n - 1) and s0 % 12 == 0
                for sg in range(s0, s1):
                    a, ip = divmod(sg, n - 1)
                    j = j0 + a
                    i = ip + (1 if ip >= j else 0)
                    seen[boff + i * (n - 1) + (j if j < i else j - 1)] += 1
            assert bool((seen == 1).all())
    check()


def test_philox_checker_reproduces_random123_known_answers():
    """oracle/philox_ref.py (the checker of the device generator) against Random123's published philox4x32-10 vectors."""
    from oracle import philox_ref as pr
    for ctr, key, out in pr.KAT:
        assert tuple(int(v) for v in pr.philox4x32([ctr], [key])[0]) == out
    u = pr.uniform24(np.array([0, 0xffffffff, 0x80000000], dtype=np.uint32))
    assert u[0] == 0.0 and u[1] < 1.0 and u[2] == 0.5


def test_plan_matches_is_safe_for_inference_tensors_and_recycled_storage():
    """BatchPlan.matches: its fast path (same tensor objects, unmodified) must neither crash on tensors that do not track
    versions (created under torch.inference_mode()) nor accept a different batch of the same shapes whose tensors happen to land
    on a freed batch's addresses."""
    import gc
    from phoregen_amd.plan import BatchPlan, make_edge_data

    def batch(n_atoms, n_phore):
        na = torch.tensor(n_atoms)
        bn = torch.repeat_interleave(torch.arange(len(n_atoms)), na)
        bp = torch.repeat_interleave(torch.arange(len(n_phore)), torch.tensor(n_phore))
        ei, be = make_edge_data(na)
        return bn, bp, ei, be
    bn, bp, ei, be = batch([3, 5], [4, 4])
    plan = BatchPlan(bn, bp, ei, be, 2, torch.device('cpu'))
    assert plan.matches(bn, bp, ei) and plan.matches(bn, bp, ei)          # by content, then by identity
    assert plan.matches(bn.clone(), bp.clone(), ei.clone())                # other objects, same contents
    bn[0] = 1                                                              # written in place: the identity answer is void
    assert not plan.matches(bn, bp, ei)
    bn[0] = 0
    assert plan.matches(bn, bp, ei)
    with torch.inference_mode():                                           # no version counters there: contents decide, no crash
        ibn, ibp, iei, _ = batch([3, 5], [4, 4])
        assert plan.matches(ibn, ibp, iei) and plan.matches(ibn, ibp, iei)
        obn, obp, oei, _ = batch([5, 3], [4, 4])
        assert not plan.matches(obn, obp, oei)
    # a freed batch's addresses re-used by a batch of the same shapes and other contents ([3,5] atoms -> [5,3])
    for _ in range(20):
        a = batch([3, 5], [4, 4])
        assert plan.matches(*a[:3])
        ptrs = [x.data_ptr() for x in a[:3]]
        del a
        gc.collect()
        b = batch([5, 3], [4, 4])
        assert not plan.matches(*b[:3]), ([x.data_ptr() for x in b[:3]], ptrs)
        del b


def test_partition_balances_the_fitted_step_cost():
    """parallel.partition_graphs on graph_cost = a tiles n (n-1) + b n (n-1) + c (n + p) + d.  Every graph lands on exactly one rank; plain LPT
    (by_size=False) keeps the heaviest rank within one graph of the mean; the default groups the largest ligands on the first ranks (the
    attention kernels are instantiated for the row tiles of a batch's largest ligand: ligands of 51+ atoms then do not put EVERY rank on the
    4-tile kernels) and gives the ranks that hold them 2.5 % less than the others hold (16 graphs per rank; 8 % at 32); on a batch whose
    pharmacophore sizes are skewed against the atom counts the fitted cost balances better than n^3 alone does."""
    from phoregen_amd.parallel import COST_US, graph_cost, partition_graphs
    g = torch.Generator().manual_seed(5)
    for B, world in ((128, 8), (128, 3), (37, 4), (5, 8), (128, 1)):
        na = (40 + 6 * torch.randn(B, generator=g)).round().clamp(20, 60).long()
        nph = (107 + 30 * torch.randn(B, generator=g)).round().clamp(23, 203).long()
        cost = graph_cost(na, nph)
        mean = float(cost.sum()) / world
        for by_size in (False, True):
            parts = partition_graphs(na, world, nph, by_size=by_size)
            assert sorted(torch.cat(parts).tolist()) == list(range(B)) and all(bool((p[1:] > p[:-1]).all()) for p in parts if p.numel() > 1)
            loads = torch.stack([cost[p].sum() if p.numel() else cost.new_zeros(()) for p in parts])
            assert float(loads.max()) <= mean * (1.06 if by_size else 1.0) + float(cost.max()) + 1e-9
        if B == 128 and world == 8:
            parts = partition_graphs(na, world, nph)                       # default: by size
            with_big = [r for r, p in enumerate(parts) if int(na[p].max()) >= 51]
            assert with_big and with_big == list(range(len(with_big))) and len(with_big) <= 3       # the 51+-atom ligands sit together on the first ranks
            lpt_big = [r for r, p in enumerate(partition_graphs(na, world, nph, by_size=False)) if int(na[p].max()) >= 51]
            assert len(lpt_big) > len(with_big)
            loads = torch.stack([cost[p].sum() for p in parts])
            assert float(loads[with_big].max()) <= mean * 0.99 and float(loads.max()) <= mean * 1.03         # (2.5 % less than the others hold at 16 graphs per rank)
    # an empty job (a shard of a finished sampling job): empty shards for every rank, at any world size
    for world in (1, 2, 8):
        parts = partition_graphs(torch.tensor([], dtype=torch.long), world)
        assert len(parts) == world and all(p.numel() == 0 and p.dtype == torch.long for p in parts)
    # the model's terms are the ones the kernels scale with: tiles of the triplet kernel, bond edges, context nodes
    one = graph_cost(torch.tensor([40]), torch.tensor([107]))
    assert abs(float(one) - (COST_US['tile'] * 3 * 40 * 39 + COST_US['bond'] * 40 * 39 + COST_US['node'] * 147 + COST_US['graph'])) < 1e-9
    # small ligands with huge pharmacophores vs large ligands with small ones: n^3 alone piles the node work on one rank
    na = torch.tensor([20] * 8 + [30] * 8)
    nph = torch.tensor([203] * 8 + [23] * 8)
    cost = graph_cost(na, nph)
    worst = lambda parts: max(float(cost[p].sum()) for p in parts)
    assert worst(partition_graphs(na, 4, nph, by_size=False)) <= worst(partition_graphs(na, 4, by_size=False))


def test_triplet_adjoint_atom_order_is_a_balanced_permutation():
    """BatchPlan.bwd_atom_order (PgSegAttnGrad.atom_order): every ligand atom exactly once, and the persistent workgroups of the triplet
    adjoint -- workgroup b works off entries b, b + grid, ... -- carry nearly equal cost, unlike in index order."""
    from phoregen_amd.plan import BatchPlan, make_edge_data
    g = torch.Generator().manual_seed(11)
    na = (25 + 5 * torch.randn(256, generator=g)).round().clamp(8, 60).long()
    B = na.numel()
    ei, be = make_edge_data(na)
    plan = BatchPlan(torch.repeat_interleave(torch.arange(B), na), torch.zeros(0, dtype=torch.long), ei, be, B, 'cpu')
    n = torch.repeat_interleave(na, na).double()
    cost = (n - 1) * (torch.ceil(n / 16) + 1)
    for grid in (256, 64, 7):
        order = plan.bwd_atom_order(grid).long()
        assert sorted(order.tolist()) == list(range(int(na.sum())))
        load = lambda seq: torch.stack([cost[seq[b::grid]].sum() for b in range(grid)])
        sorted_load, index_load = load(order), load(torch.arange(order.numel()))
        assert float(sorted_load.max() / sorted_load.mean()) <= 1.03, grid
        assert float(sorted_load.max()) <= float(index_load.max())
    assert float(load(torch.arange(order.numel())).max() / cost.sum() * 7) > 0      # (index order at 256 workgroups: 1.16 x the mean)


def test_triplet_queues_by_row_tiles_cover_the_single_queue():
    """BatchPlan.tri_split: the entries of the staged triplet kernel's queue as two queues, ligands of up to 50 atoms (a segment visits n - 2 <= 48 rows = 3 row tiles of 16) and
    larger ones.  Together they hold exactly the entries of the single queue; the first holds no ligand above 50 atoms and names its largest
    (PgSegAttn.tri_max_nlig: the kernel instance is picked from it); a batch of one class has no split."""
    from phoregen_amd.plan import BatchPlan, make_edge_data

    def plan_of(sizes):
        na = torch.tensor(sizes)
        ei, be = make_edge_data(na)
        return BatchPlan(torch.repeat_interleave(torch.arange(len(sizes)), na), torch.zeros(0, dtype=torch.long), ei, be, len(sizes), 'cpu')
    p = plan_of([20, 50, 33, 51, 56, 8, 2, 41])
    assert p.tri_split is not None
    (it_s, n_s, max_s, ctr_s), (it_b, n_b, max_b, ctr_b) = p.tri_split['small'], p.tri_split['big']
    def segments(t, n):
        """Every (ligand, group, segment) an entry list hands out (entry = {ligand, n | j0 << 8 | A << 16, bond row 0, s0 | s1 << 16}; 0 = the
        whole group).  The tail of a queue is cut into half-groups, so the two queues hold other ENTRIES than the single one, the same segments."""
        out = []
        for lig0, w1, boff, w3 in t[:n].tolist():
            n_, a_ = w1 & 0xff, w1 >> 16
            s0, s1 = (w3 & 0xffff, w3 >> 16) if w3 else (0, a_ * (n_ - 1))
            out += [(lig0, w1, boff, s_) for s_ in range(s0, s1)]
        return sorted(out)
    assert n_s > 0 and n_b > 0
    both, single = segments(it_s, n_s) + segments(it_b, n_b), segments(p.tri_iters, p.n_tri_iters)
    assert sorted(both) == single and len(set(single)) == len(single)            # every segment exactly once, either way
    atoms = lambda t, n: {int(v) & 0xff for v in t[:n, 1].tolist()}            # entry word 1 = n | j0 << 8 | a << 16
    assert max(atoms(it_s, n_s)) == max_s == 50 and min(atoms(it_b, n_b)) == 51 and max_b == 0
    assert ctr_s.numel() == 2 and ctr_b.numel() == 2 and ctr_s.data_ptr() != ctr_b.data_ptr() != p.tri_counter.data_ptr()
    assert plan_of([20, 33, 50]).tri_split is None and plan_of([51, 60]).tri_split is None


def test_option_switches_are_consistent(monkeypatch):
    """phoregen_amd.options: every PG_* variable names a real option and converts; the environment counts only under PHOREGEN_DEBUG=1;
    override() nests and restores."""
    from phoregen_amd import options
    for var, (key, conv) in options._ENV.items():
        assert key in options.DEFAULTS, (var, key)
    monkeypatch.setenv('PG_TRI_GRID', '96')
    monkeypatch.delenv('PHOREGEN_DEBUG', raising=False)
    assert options.get('tri_grid') == options.DEFAULTS['tri_grid']          # ambient environment: ignored
    monkeypatch.setenv('PHOREGEN_DEBUG', '1')
    assert options.get('tri_grid') == 96
    with options.override(tri_grid=128, knn_merge='never'):
        assert options.get('tri_grid') == 128 and options.get('knn_merge') == 'never'
        with options.override(tri_grid=64):
            assert options.get('tri_grid') == 64 and options.get('knn_merge') == 'never'
        assert options.get('tri_grid') == 128
    assert options.get('tri_grid') == 96 and options.get('knn_merge') == options.DEFAULTS['knn_merge']
    assert set(options.snapshot()) == set(options.DEFAULTS)



def test_launch_records_carry_the_arguments_ctypes_would_pass():
    """hip.launch_record: one entry of an engine launch list as the PgLaunch the library walks (include/phoregen_hip.h): pointers and
    integers as 64-bit values, a float as its bit pattern, NULL as 0, byref(struct) as the struct's address; the library validates op,
    argument count, lane and order-point index when a program is created (no GPU needed for either)."""
    import ctypes as C
    import struct
    from phoregen_amd import hip
    lib = hip.load_library()
    topo, g = hip.PgTopo(), hip.PgGemm()
    L = hip.launch_record(lib.pg_edge_gate, (C.byref(topo), 0x7f0000001000, 0x7f0000002000, None, 32, 5, 6, 7, 8, 9, C.c_float(-0.375), 11), 1)
    assert (L.op, L.lane, L.n_arg) == (hip.PROGRAM_OPS['pg_edge_gate'], 1, 12)
    assert L.a[0] == C.addressof(topo) and L.a[1] == 0x7f0000001000 and L.a[3] == 0 and L.a[4] == 32
    assert struct.unpack('<f', struct.pack('<I', L.a[10]))[0] == -0.375
    L2 = hip.launch_record(lib.pg_gemm, (C.byref(g),), 0)
    assert L2.a[0] == C.addressof(g) and L2.n_arg == 1
    L3 = hip.launch_record(lib.pg_knn_ctx, (C.byref(topo), 1, -1, 2, 3), 3)          # a negative int keeps its value through the 64-bit slot
    assert C.c_int64(L3.a[2]).value == -1
    # the library side: validation without touching the GPU (no events asked for)
    h = C.c_void_p()
    arr = (hip.PgLaunch * 2)(L, L2)
    assert lib.pg_program_create(arr, 2, 0, C.byref(h)) == 0 and lib.pg_program_length(h) == 2 and lib.pg_program_destroy(h) == 0
    bad = hip.PgLaunch()
    bad.op, bad.lane, bad.ev, bad.n_arg = hip.OP_WAIT, 0, 3, 0
    assert lib.pg_program_create((hip.PgLaunch * 1)(bad), 1, 2, C.byref(h)) != 0 and b'order point 3 of 2' in lib.pg_last_error()
    bad.op, bad.ev, bad.lane = hip.PROGRAM_OPS['pg_gemm'], -1, 7
    bad.n_arg = 1
    assert lib.pg_program_create((hip.PgLaunch * 1)(bad), 1, 0, C.byref(h)) != 0 and b'lane 7' in lib.pg_last_error()
    bad.lane, bad.n_arg = 0, 3
    assert lib.pg_program_create((hip.PgLaunch * 1)(bad), 1, 0, C.byref(h)) != 0 and b'takes 1 arguments' in lib.pg_last_error()
    assert set(hip.PROGRAM_OPS) <= set(hip.EXPORTS)


def test_failed_program_run_poisons_the_program_and_debug_switch_is_gated(monkeypatch):
    """pg_program_run: an entry that fails leaves the lanes half-enqueued -- the library drains the device, returns the entry's error and
    refuses the program from then on (advisor, round 5).  pg_debug_order_point_fence_free weakens every order point created after it: it is
    refused unless the process runs with PHOREGEN_DEBUG=1.  (Argument errors only: nothing here reaches a GPU.)"""
    import ctypes as C
    from phoregen_amd import hip
    lib = hip.load_library()
    bad = hip.PgLaunch()
    bad.op, bad.lane, bad.ev, bad.n_arg = hip.PROGRAM_OPS['pg_gemm'], 0, -1, 1
    bad.a[0] = 0                                                   # pg_gemm(NULL): refused before any launch
    h = C.c_void_p()
    assert lib.pg_program_create((hip.PgLaunch * 1)(bad), 1, 0, C.byref(h)) == 0
    streams = (C.c_void_p * 4)()
    assert lib.pg_program_run(h, streams) != 0 and b'poisoned' not in lib.pg_last_error()
    assert lib.pg_program_run(h, streams) != 0 and b'poisoned' in lib.pg_last_error()
    assert lib.pg_program_destroy(h) == 0
    monkeypatch.delenv('PHOREGEN_DEBUG', raising=False)
    assert lib.pg_debug_order_point_fence_free(1) != 0 and b'PHOREGEN_DEBUG' in lib.pg_last_error()
    assert lib.pg_debug_order_point_fence_free(0) == 0
