"""-m gpu: the device generator behind rng='device' (Philox4x32-10 in phoregen_amd/csrc/posterior.hip), through the C ABI.

The reference draws `torch.rand_like` / `torch.randn_like` from torch's global generator (models/common.py:425-431,
models/transition.py:60); `sample(..., rng='device')` replaces that stream with a counter-based one, so it is verified
on its own: known answers (Random123's published vectors and an independent numpy Philox on random counters/keys),
the uniform and Box-Muller marginals the transition kernels actually produce, and the Gumbel-argmax sampling frequencies
against the categorical posterior the same kernel reports."""
import numpy as np
import pytest
import torch
from scipy import stats

from oracle import philox_ref as pr

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _device_words(ctr, key):
    from phoregen_amd import hip
    lib = hip.lib()
    ck = np.concatenate([np.asarray(ctr, dtype=np.uint32).reshape(-1, 4), np.asarray(key, dtype=np.uint32).reshape(-1, 2)], 1)
    d_in = torch.from_numpy(ck.view(np.int32).copy()).to(DEV)
    d_out = torch.zeros(ck.shape[0], 4, dtype=torch.int32, device=DEV)
    hip.check(lib.pg_selftest_philox(d_in.data_ptr(), ck.shape[0], d_out.data_ptr(), hip.stream_ptr()), 'selftest_philox')
    torch.cuda.synchronize()
    return d_out.cpu().numpy().view(np.uint32)


def test_philox_known_answers():
    for ctr, key, out in pr.KAT:
        assert tuple(int(v) for v in _device_words([ctr], [key])[0]) == out
    rng = np.random.default_rng(3)
    ctr = rng.integers(0, 2 ** 32, (4096, 4), dtype=np.uint64).astype(np.uint32)
    key = rng.integers(0, 2 ** 32, (4096, 2), dtype=np.uint64).astype(np.uint32)
    assert np.array_equal(_device_words(ctr, key), pr.philox4x32(ctr, key))


def test_uniform_words_are_uniform():
    """Consecutive counters of one key, as the kernels enumerate them (counter = element index, step, stream)."""
    n = 1 << 18
    ctr = np.zeros((n, 4), dtype=np.uint32)
    ctr[:, 0] = np.arange(n, dtype=np.uint32)
    ctr[:, 2], ctr[:, 3] = 999, 1
    u = pr.uniform24(_device_words(ctr, np.tile(np.array([[12345, 0]], dtype=np.uint32), (n, 1)))).astype(np.float64).ravel()
    assert u.min() >= 0.0 and u.max() < 1.0
    assert abs(u.mean() - 0.5) < 4 * np.sqrt(1 / 12 / u.size) and abs(u.var() - 1 / 12) < 2e-3
    assert stats.kstest(u, 'uniform').pvalue > 1e-3
    # the four words of a counter and neighbouring counters are uncorrelated
    w = u.reshape(n, 4)
    for a, b in ((w[:, 0], w[:, 1]), (w[:, 2], w[:, 3]), (w[:-1, 0], w[1:, 0])):
        assert abs(np.corrcoef(a, b)[0, 1]) < 5 / np.sqrt(a.size)


def test_position_noise_is_standard_normal():
    """pg_posterior_position with mu = 0, sigma = 1: x_prev is exactly the kernel's Box-Muller draw."""
    from phoregen_amd import hip
    lib = hip.lib()
    n = 400_000
    z = torch.zeros(n, 3, device=DEV)
    rg = torch.zeros(n, dtype=torch.int32, device=DEV)
    tt = torch.full((1,), 5, dtype=torch.int64, device=DEV)
    zero_tab, one_tab = torch.zeros(1000, device=DEV), torch.ones(1000, device=DEV)
    outs = []
    for step in (999, 998):
        out = torch.empty(n, 3, device=DEV)
        hip.check(lib.pg_posterior_position(z.data_ptr(), z.data_ptr(), rg.data_ptr(), tt.data_ptr(), zero_tab.data_ptr(),
                                            zero_tab.data_ptr(), one_tab.data_ptr(), None, None, 77, 2, step, n, None, None,
                                            None, out.data_ptr(), None, hip.stream_ptr()), 'posterior(pos)')
        outs.append(out.cpu().numpy().astype(np.float64).ravel())
    e = outs[0]
    assert np.isfinite(e).all()
    m = e.size
    assert abs(e.mean()) < 4 / np.sqrt(m) and abs(e.var() - 1) < 4 * np.sqrt(2 / m)
    assert abs(stats.skew(e)) < 4 * np.sqrt(6 / m) and abs(stats.kurtosis(e)) < 4 * np.sqrt(24 / m)
    assert stats.kstest(e, 'norm').pvalue > 1e-3
    assert abs(np.corrcoef(outs[0], outs[1])[0, 1]) < 5 / np.sqrt(m)          # steps are independent streams
    assert abs(np.corrcoef(e[:-1], e[1:])[0, 1]) < 5 / np.sqrt(m)


def test_gumbel_argmax_frequencies_follow_the_posterior():
    """>= 10^6 draws of one 12-class row: the sampled class frequencies against exp(posterior) as reported by the same kernel
    (models/common.py:425-431 Gumbel-max == a categorical draw), chi-square and max deviation."""
    from helpers import make_oracle
    from phoregen_amd import hip
    lib = hip.lib()
    o = make_oracle(0)
    K, n = 12, 1_200_000
    g = torch.Generator().manual_seed(4)
    logits = (1.5 * torch.randn(1, K, generator=g)).expand(n, K).contiguous().to(DEV)
    log_vt = torch.log_softmax(2.0 * torch.randn(1, K, generator=g), -1).expand(n, K).contiguous().to(DEV)
    qm, qt = (o.tab_node[k].to(DEV).contiguous() for k in ('q_mats', 'transpopse_q_onestep_mats'))
    rg = torch.zeros(n, dtype=torch.int32, device=DEV)
    tt = torch.full((1,), 400, dtype=torch.int64, device=DEV)
    post, onehot = torch.empty(n, K, device=DEV), torch.empty(n, K, device=DEV)
    hip.check(lib.pg_posterior_categorical(logits.data_ptr(), log_vt.data_ptr(), rg.data_ptr(), tt.data_ptr(), qm.data_ptr(),
                                           qt.data_ptr(), n, K, None, 2024, 0, 400, None, None, post.data_ptr(), onehot.data_ptr(),
                                           None, hip.stream_ptr()), 'posterior(node)')
    torch.cuda.synchronize()
    p = post[0].double().exp().cpu().numpy()
    assert abs(p.sum() - 1) < 1e-5 and (onehot.sum(-1) == 1).all()
    counts = onehot.sum(0).double().cpu().numpy()
    keep = p * n >= 10                                         # chi-square needs a handful of expected hits per cell
    chi2 = (((counts - p * n) ** 2) / (p * n))[keep].sum()
    assert stats.chi2.sf(chi2, int(keep.sum()) - 1) > 1e-3, (chi2, counts, p * n)
    assert counts[~keep].sum() <= 10 * max(1, (~keep).sum()) + 5 * (p[~keep] * n).sum()
    assert np.abs(counts / n - p).max() < 5 * np.sqrt(0.25 / n)
