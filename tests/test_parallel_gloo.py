"""N>1 path on CPU: graph partition + final gather with world_size 2 over gloo (no HIP compute involved)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from phoregen_amd.parallel import allreduce_gradients, gather_predictions, partition_graphs


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _fake_pred(graph_ids, num_atoms):
    """Deterministic per-graph 'predictions' so the gathered result can be checked exactly."""
    node, pos, edge = [], [], []
    for g, n in zip(graph_ids.tolist(), num_atoms.tolist()):
        gen = torch.Generator().manual_seed(1000 + g)
        node.append(torch.randn(n, 12, generator=gen))
        pos.append(torch.randn(n, 3, generator=gen))
        edge.append(torch.randn(n * (n - 1), 6, generator=gen))
    if not node:                               # a rank without graphs
        return [torch.zeros(0, 12), torch.zeros(0, 3), torch.zeros(0, 6)]
    return [torch.cat(node), torch.cat(pos), torch.cat(edge)]


def _worker(rank, world, port, num_atoms, q):
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        parts = partition_graphs(num_atoms, world)
        mine = parts[rank]
        pred = _fake_pred(mine, num_atoms[mine])
        ref = _fake_pred(torch.arange(num_atoms.numel()), num_atoms)
        same = lambda out: all(torch.equal(a, b) for a, b in zip(out, ref))
        out, nat = gather_predictions(pred, num_atoms[mine], mine)                 # the path's form: to rank 0 only
        ok = torch.equal(nat, num_atoms) and (same(out) if rank == 0 else out is None)
        out, nat = gather_predictions(pred, num_atoms[mine], mine, dst=None)       # on request: every rank
        ok = ok and same(out) and torch.equal(nat, num_atoms)
        out, nat = gather_predictions(pred, num_atoms[mine], mine, dst=world - 1)
        ok = ok and (same(out) if rank == world - 1 else out is None)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def _pattern_pred(graph_ids, num_atoms):
    """Per-graph payload that is cheap to make and to check at job scale: column 0 = graph id, column 1 = row inside the graph."""
    out = []
    for width, rows in ((12, num_atoms), (3, num_atoms), (6, num_atoms * (num_atoms - 1))):
        t = torch.zeros(int(rows.sum()), width)
        start = rows.cumsum(0) - rows
        t[:, 0] = torch.repeat_interleave(graph_ids.float(), rows)
        t[:, 1] = (torch.arange(t.size(0)) - torch.repeat_interleave(start, rows)).float()
        out.append(t)
    return out


def _job_worker(rank, world, port, n_graphs, q):
    """BASELINE config 4's gather (sample_all.py:104-116 at job scale): `n_graphs` graphs' worth of rows, partitioned as
    `run_sampling_job` does, payload without compute."""
    import time
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(4321)
        num_atoms = (40 + 6 * torch.randn(n_graphs, generator=g)).round().clamp(20, 60).long()      # bench.config4_job
        t0 = time.perf_counter()
        mine = partition_graphs(num_atoms, world)[rank]
        t_part = time.perf_counter() - t0
        pred = _pattern_pred(mine, num_atoms[mine])
        calls = {'n': 0}
        real = {n: getattr(dist, n) for n in ('all_gather_into_tensor', 'all_gather', 'gather', 'all_reduce', 'broadcast')}
        for n, f in real.items():
            setattr(dist, n, (lambda f_: lambda *a, **k: (calls.__setitem__('n', calls['n'] + 1), f_(*a, **k))[1])(f))
        dist.barrier()
        t0 = time.perf_counter()
        out, nat = gather_predictions(pred, num_atoms[mine], mine)
        t_gather = time.perf_counter() - t0
        for n, f in real.items():
            setattr(dist, n, f)
        ok = torch.equal(nat, num_atoms)
        if rank == 0:
            exp = _pattern_pred(torch.arange(n_graphs), num_atoms)
            ok = ok and all(torch.equal(a, b) for a, b in zip(out, exp))
            payload = sum(t.numel() for t in out) * 4
        else:
            ok, payload = ok and out is None, 0
        # peak resident set of THIS process (VmHWM: per address space, reset by exec; getrusage's ru_maxrss starts from the parent's value -- in a
        # full pytest run the 3 GB of the test process itself)
        hwm = next(int(l.split()[1]) * 1024 for l in open('/proc/self/status') if l.startswith('VmHWM'))
        q.put((rank, bool(ok), calls['n'], t_part, t_gather, hwm, payload))
    finally:
        dist.destroy_process_group()


def run_job_gather(n_graphs, world=8):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_job_worker, args=(r, world, port, n_graphs, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=900) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    return res


def test_gather_world_size_8_gloo_job_scale():
    """World size 8 over gloo, 12 800 graphs (1 / 8 of BASELINE config 4; the full 102 400 graphs -- 4.1 GB of rows, ~70 s and ~14 GB on
    rank 0 of this 8-core host -- are run by tools/gather_config4_gloo8.py, record in profiles/r06_gather_config4_gloo8.txt): global order,
    at most 3 collectives, and the destination's peak host memory bounded by a small multiple of the payload."""
    res = run_job_gather(12800)
    assert [r[0] for r in res] == list(range(8)) and all(r[1] for r in res), res
    assert all(r[2] <= 3 for r in res), [r[2] for r in res]
    payload = res[0][6]
    assert payload > 400e6                                         # (12 800 graphs ~ 0.5 GB of fp32 rows)
    assert res[0][5] <= 4.0 * payload + 1.5e9, (res[0][5], payload)  # gathered buffers + result + one row index (+ the interpreter and torch)
    assert all(r[5] <= 0.5 * payload + 2.5e9 for r in res[1:]), [r[5] for r in res]       # (the other ranks hold their own share + the interpreter; the job-size run records 1.7 GB each)


def test_partition_is_balanced_and_complete():
    na = torch.tensor([40, 20, 60, 33, 47, 41, 39, 25, 58, 44])
    for world in (1, 2, 4, 8):
        parts = partition_graphs(na, world)
        assert sorted(torch.cat(parts).tolist()) == list(range(10))
        loads = [float((na[p].double() ** 3).sum()) for p in parts]
        if world <= 4:
            assert max(loads) <= 1.6 * (sum(loads) / world)


def test_gather_world_size_2_gloo_with_an_empty_rank():
    """One graph for two ranks (the tail of a job): the rank without graphs takes part in the three collectives with empty payloads."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, torch.tensor([6]), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


def test_gather_world_size_2_gloo():
    na = torch.tensor([5, 9, 3, 7, 4, 8, 6])
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, na, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


def _grad_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3))
        net[1].bias.requires_grad_(False)
        for i, p in enumerate(net.parameters()):
            if p.requires_grad and not (rank == 1 and i == 0):          # rank 1 has no gradient for the first tensor
                p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
        n = allreduce_gradients(net.parameters())
        exp = {0: 0.5 * 1.0, 1: 1.5 * 2.0, 2: 1.5 * 3.0}                # mean over ranks of (rank+1)*(i+1); tensor 0: (1+0)/2
        ok = n == 7 * 5 + 5 + 5 * 3 and all(torch.allclose(p.grad, torch.full_like(p, exp[i]))
                                             for i, p in enumerate(net.parameters()) if p.requires_grad)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_gradient_bucket_allreduce_world_size_2_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    assert res == {0: True, 1: True}


def test_strong_scaling_partition_covers_the_workload():
    """bench.py --strong: the per-rank sub-workloads are a partition of the 128-graph batch (same graphs, same order
    inside a rank, pharmacophore rows follow their graphs)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import ligphore_workload, subset_workload
    w = ligphore_workload(16, seed=1234)
    parts = partition_graphs(w['num_atoms'], 4)
    seen = []
    for ids in parts:
        s = subset_workload(w, ids)
        assert torch.equal(s['num_atoms'], w['num_atoms'][ids]) and torch.equal(s['n_phore'], w['n_phore'][ids])
        assert s['batch_phore'].min() == 0 and s['batch_phore'].max() == ids.numel() - 1
        assert torch.equal(torch.bincount(s['batch_phore']), w['n_phore'][ids])
        first = int((w['batch_phore'] == ids[0]).nonzero()[0])
        assert torch.equal(s['h_phore'][0], w['h_phore'][first])
        seen += ids.tolist()
    assert sorted(seen) == list(range(16))


def _bucket_worker(rank, world, port, q):
    import torch.distributed as dist
    from phoregen_amd.parallel import GradientBuckets
    dist.init_process_group('gloo', init_method=f'tcp://127.0.0.1:{port}', rank=rank, world_size=world)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(40, 300), torch.nn.ReLU(), torch.nn.Linear(300, 300), torch.nn.ReLU(), torch.nn.Linear(300, 7))
    net[4].bias.requires_grad_(False)
    extra = torch.nn.Parameter(torch.zeros(11))                 # never used: no gradient on any rank
    params = list(net.parameters()) + [extra]
    gb = GradientBuckets(params, bucket_mb=0.2)
    assert len(gb.buckets) >= 2
    x = torch.randn(16, 40, generator=torch.Generator().manual_seed(10 + rank))
    out = {}
    for it in range(2):                                          # two steps: counters reset correctly
        for p in params:
            p.grad = None
        net(x).pow(2).sum().backward()
        local = [p.grad.numpy().copy() if p.grad is not None else None for p in params]
        out_hooks_step2 = [h for _, h in gb.launch_log][-len(gb.buckets):]      # (step 2: the unused `extra` no longer holds bucket 0 back)
        n = gb.finish()
        out[it] = (local, [p.grad.numpy().copy() if p.grad is not None else None for p in params], n)      # numpy: plain pickles
    # rank-dependent autograd graph: rank 1 skips the first Linear (its gradient hooks never fire there) -- the buckets must still
    # be launched in the same (index) order on both ranks, early-complete buckets waiting for their predecessors
    for p in params:
        p.grad = None
    gb.launch_log.clear()
    assert out_hooks_step2 == [True] * len(gb.buckets), out_hooks_step2
    if rank == 0:
        net(x).pow(2).sum().backward()
    else:
        net[2:](torch.randn(16, 300, generator=torch.Generator().manual_seed(77))).pow(2).sum().backward()
    local = [p.grad.numpy().copy() if p.grad is not None else None for p in params]
    n = gb.finish()
    out[2] = (local, [p.grad.numpy().copy() if p.grad is not None else None for p in params], n)
    out['launch_order'] = [b for b, _ in gb.launch_log]
    out['from_hook'] = [h for _, h in gb.launch_log]
    # the set of parameters a bucket does not wait for is agreed on by all ranks (MAX over the ranks' used-bitmaps in finish()):
    # the first Linear, unused on rank 1 in the step above, is still waited for there -- its gradients arriving now do not find
    # their bucket launched (per-rank sets made that a RuntimeError on rank 1 and a hang on rank 0)
    out['absent_after_3'] = sorted(len(a) for a in gb.absent)
    for p in params:
        p.grad = None
    gb.launch_log.clear()
    net(x).pow(2).sum().backward()
    gb.finish()
    out['from_hook_4'] = [h for _, h in gb.launch_log]
    # a finish() without a backward says nothing about which parameters are used: the next step still launches from hooks
    gb.launch_log.clear()
    for p in params:
        p.grad = None
    gb.finish()
    gb.launch_log.clear()
    net(x).pow(2).sum().backward()
    gb.finish()
    out['from_hook_5'] = [h for _, h in gb.launch_log]
    q.put((rank, out))
    dist.destroy_process_group()


def test_bucketed_hook_allreduce_world_size_2_gloo():
    """GradientBuckets (f-4): hook-launched asynchronous bucket all-reduces == the mean of the ranks' local gradients."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q, port = ctx.Queue(), _free_port()
    procs = [ctx.Process(target=_bucket_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    n_b = len(res[0]['launch_order'])
    assert res[0]['launch_order'] == res[1]['launch_order'] == list(range(n_b)) and n_b >= 2
    assert all(res[0]['from_hook'])     # rank 0: every bucket from a hook (the never-used parameter is known absent after step 1)
    assert not all(res[1]['from_hook'])                                  # rank 1: the first Linear's bucket only in finish()
    for r in (0, 1):
        assert sum(res[r]['absent_after_3']) == 1, res[r]['absent_after_3']      # only the parameter no rank uses
        assert all(res[r]['from_hook_4']) and all(res[r]['from_hook_5']), (res[r]['from_hook_4'], res[r]['from_hook_5'])
    for it in range(3):
        l0, r0, n0 = res[0][it]
        l1, r1, n1 = res[1][it]
        import numpy as np
        assert n0 == n1 == sum(t.size for t in r0 if t is not None)
        for a, b, ra, rb in zip(l0, l1, r0, r1):
            if ra is None:
                continue
            za = a if a is not None else np.zeros_like(ra)
            zb = b if b is not None else np.zeros_like(rb)
            assert np.allclose(ra, (za + zb) / 2, atol=1e-6) and np.array_equal(ra, rb)


def test_bucket_hooks_refuse_a_second_backward_before_finish():
    """Gradient accumulation over micro-batches would be silently dropped by finish()'s write-back: it raises instead."""
    import pytest
    from phoregen_amd.parallel import GradientBuckets
    net = torch.nn.Linear(5, 3)
    gb = GradientBuckets(net.parameters(), bucket_mb=0.001)
    x = torch.randn(4, 5)
    net(x).sum().backward()
    with pytest.raises(RuntimeError, match='second gradient'):
        net(x).sum().backward()
    gb.remove()
    # after finish() the next backward is fine, and without a process group the gradients are left as they are
    net.zero_grad(set_to_none=True)
    gb = GradientBuckets(net.parameters(), bucket_mb=0.001)
    net(x).sum().backward()
    g0 = net.weight.grad.clone()
    gb.finish()
    assert torch.equal(net.weight.grad, g0)
    net.zero_grad(set_to_none=True)
    net(x).sum().backward()
    gb.finish()
    assert [b for b, _ in gb.launch_log] == list(range(len(gb.buckets))) * 2
