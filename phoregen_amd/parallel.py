"""Graph-level sharding of a sampling job over the GPUs of one node (one process per GPU, RCCL/xGMI).

Graphs are independent (block-diagonal batch: knn is per graph, every scatter index stays inside its graph), so a
job partitions by graph with NO exchange inside the 1000-step loop.  The only collective is the final gather of
the per-graph predictions; it works on any torch.distributed backend (`nccl` = RCCL on ROCm, `gloo` in the CPU tests).
"""
import torch
import torch.distributed as dist


# What a graph of n atoms and p pharmacophore nodes costs inside a sampler step, in microseconds, from the kernel tables of the
# headline batch (profiles/r03_bench_serial_kernel_stats.md, 128 graphs, one stream): the triplet kernel's row tiles (618 440 tiles
# of 16 rows <-> 6 x 1.95 ms), everything that scales with the bond edges n (n - 1) (their GEMMs and the two bond attention
# modes: 203 720 edges <-> 6.2 ms), everything that scales with the context nodes n + p (knn attention, node GEMMs, knn search,
# gate: 18 816 nodes <-> 3.0 ms), and a constant per graph (workgroups that are instantiated per graph, tails of its row tiles: fitted on the
# measured rank shares of the headline batch, tools/predict_scaling.py: 5.6 us per graph at 1.16 us per unit of the other three terms).  Only
# the ratios matter for the partition.
COST_US = dict(tile=11.7e3 / 618440, bond=6.2e3 / 203720, node=3.0e3 / 18816, graph=4.8)


def graph_cost(num_atoms, n_phore=None):
    """Per-graph cost model of a sampler step (see COST_US).  Without pharmacophore sizes the node term counts the atoms only."""
    n = num_atoms.double()
    tiles = torch.div(num_atoms.clamp(min=2) - 2 + 15, 16, rounding_mode='floor').double()       # 16-row tiles per triplet segment (n - 2 rows)
    ctx = n + (n_phore.double() if n_phore is not None else 0.0)
    return COST_US['tile'] * tiles * n * (n - 1) + COST_US['bond'] * n * (n - 1) + COST_US['node'] * ctx + COST_US['graph']


def _level(parts, cl, caps, ok, rounds=256):
    """Level a partition in place: moves and swaps between the fullest rank (load / capacity) and any other, emptiest partner first, while that
    lowers the larger of the two; `ok(graph, destination rank)` vetoes a placement."""
    world_size = len(parts)
    tot = [sum(cl[g] for g in parts[i]) for i in range(world_size)]
    rel = lambda i: tot[i] / caps[i]
    for _ in range(rounds):
        hi = max(range(world_size), key=rel)
        applied = False
        for lo in sorted((i for i in range(world_size) if i != hi), key=rel):      # emptiest partner first
            pair_max = max(rel(hi), rel(lo))
            best = None
            for g in parts[hi]:                            # a move (h = None) or a swap g <-> h
                for h in [None] + parts[lo]:
                    if not ok(g, lo) or (h is not None and not ok(h, hi)):
                        continue
                    c = cl[g] - (cl[h] if h is not None else 0.0)
                    if c <= 0:
                        continue
                    new_max = max(rel(hi) - c / caps[hi], rel(lo) + c / caps[lo])
                    if new_max < pair_max - 1e-9 and (best is None or new_max < best[0]):
                        best = (new_max, g, h)
            if best is None:
                continue
            parts[hi].remove(best[1])
            parts[lo].append(best[1])
            tot[hi] -= cl[best[1]]
            tot[lo] += cl[best[1]]
            if best[2] is not None:
                parts[lo].remove(best[2])
                parts[hi].append(best[2])
                tot[lo] -= cl[best[2]]
                tot[hi] += cl[best[2]]
            applied = True
            break
        if not applied:
            break
    return parts


def partition_graphs(num_atoms, world_size, n_phore=None, by_size=True, slack=0.0, big_discount=None):
    """Greedy (longest-processing-time) balanced partition of independent graphs over the ranks by `graph_cost`: the triplet
    term ~ n^3 dominates a large graph, but at the 16 graphs a rank gets of the headline batch 45 % of a step scales with n^2 and
    n + p (p ranges 23 .. 203), so n^3 alone picks the wrong slowest rank.  Returns a list of LongTensors of graph ids
    (ascending inside each rank, so results can be re-assembled deterministically)."""
    cost = graph_cost(num_atoms, n_phore)
    if num_atoms.numel() == 0 or float(cost.sum()) <= 0.0:           # an empty job: every rank gets an empty shard (no capacities to divide by)
        return [torch.arange(num_atoms.numel(), dtype=torch.long) if r == 0 else torch.empty(0, dtype=torch.long) for r in range(world_size)]
    order = torch.argsort(cost, descending=True, stable=True)
    load = [0.0] * world_size
    parts = [[] for _ in range(world_size)]
    if by_size and world_size > 1:
        # first-fit decreasing into bins of the mean load: the largest ligands end up TOGETHER on the first ranks.  The attention kernels
        # are instantiated for the row tiles of the largest ligand of a batch (n >= 51 atoms: 4 tiles of 16 rows, n <= 50: 3): spread by
        # LPT, the 51+-atom graphs of the headline batch (five; nine of 50+ before the target's own row left the segments, round 6) put all eight ranks on the 4-tile kernels (+ 5 % per step)
        # ... and a rank on the 4-tile kernels gets `big_discount` less than its share (measured on the headline batch: equal cost, + 5 % time)
        cap_mean = float(cost.sum()) / world_size
        if big_discount is None:
            # measured on shares of the headline batch (tools/predict_scaling.py): per unit of cost a rank on the 4-tile kernels is 5.0 / 5.7 %
            # slower at 16 / 32 graphs per rank, 0.7 % at 64 (the kernels are throughput-bound there)
            # (a finer fit -- 3.5 % at 16 graphs per rank, 6 % at 32 -- was tried and lost: the shares scatter by +- 3 % around ANY smooth model,
            #  grid choice and whole rounds of the node kernels on the CUs that are left: profiles/r04_share_tri_grid.txt)
            # round 6: with the target's own row out of the triplet segments only ligands of 51+ atoms take the 4-tile kernels (five of the headline
            # batch, nine before) and the rank that holds them was 7 % light at 16 graphs per rank: 2.5 % there (two workload seeds, 8 ranks: 6.10 /
            # 6.10 x with 5 %, 6.18 / 6.15 with 2.5 %, 6.21 / 6.14 with none; 4 ranks: 8 % stays -- none loses 1.5 - 3 %; profiles/r06_partition_big_discount.txt)
            big_discount = 0.025 if cap_mean < 4000.0 else (0.08 if cap_mean < 8000.0 else 0.0075)      # 16 / 32 / 64+ graphs per rank
        big = (num_atoms >= 51).tolist()
        n_big_cost = float(cost[num_atoms >= 51].sum())
        cap0 = float(cost.sum()) / world_size
        n_big_bins = max(1, -(-int(n_big_cost * 1000) // int(cap0 * (1.0 - big_discount) * 1000))) if n_big_cost > 0 else 0
        # a rank on the 4-tile kernels holds (1 - big_discount) of what the others hold
        cap_small = float(cost.sum()) / (n_big_bins * (1.0 - big_discount) + (world_size - n_big_bins)) if n_big_bins < world_size else cap0
        caps = [cap_small * (1.0 - big_discount) if i < n_big_bins else cap_small for i in range(world_size)]
        rest = []
        for g in order.tolist():
            r = next((i for i in range(world_size) if load[i] + float(cost[g]) <= caps[i] * (1.0 + slack) and (not big[g] or i < max(n_big_bins, 1))), None)
            if r is None:
                rest.append(g)
                continue
            parts[r].append(g)
            load[r] += float(cost[g])
        order = torch.tensor(rest, dtype=torch.long)
        load = [l / c * cap0 for l, c in zip(load, caps)]          # (the leftovers level the ranks relative to their capacities)
    for g in order.tolist():
        r = min(range(world_size), key=lambda i: (load[i], i))
        parts[r].append(g)
        load[r] += float(cost[g])
    if by_size and world_size > 1 and len(num_atoms) <= 2048:      # (a large job is level to a fraction of a per cent already)
        _level(parts, cost.tolist(), caps, lambda g, dst: not (big[g] and dst >= max(n_big_bins, 1)))
    return [torch.tensor(sorted(p), dtype=torch.long) for p in parts]


def _comm_device(t, group):
    """gloo moves host memory; RCCL (backend `nccl` on ROCm) works on device memory."""
    return torch.device('cpu') if dist.get_backend(group) == 'gloo' else t.device


def gather_predictions(pred, num_atoms_local, graph_ids_local, group=None, dst=0):
    """The one collective step of the sampling path (sample_all.py:104-116 hands `pred` of every graph to the host-side decode): re-assemble
    `pred` = [logits_node [N,12], pos [N,3], logits_edge [E,6]] of all ranks in global (ascending graph id) order.

    Three collectives whatever the job size:
      1. all_gather of the per-rank graph count (one int64),
      2. all_gather of the per-rank table [graph id | atoms] padded to the largest count (16 B per graph: 1.6 MB for BASELINE config 4),
      3. ONE flat fp32 buffer per rank -- its node logits, coordinates and bond logits back to back (15 N_r + 6 E_r floats), padded to the
         longest rank -- gathered to rank `dst` only (`dist.gather`; the north star asks for the final gather on rank 0), or to every rank
         with dst=None (`all_gather`).
    Re-assembly is vectorised: per rank and kind one `index_copy_` whose row index comes from two cumulative sums -- O(ranks) Python work,
    nothing per graph.  Peak memory on the destination: the gathered buffers (world x longest rank) + the result + one kind's row index.
    Returns (pred_global, num_atoms_global); pred_global is None on the ranks that are not the destination."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    out_dev = pred[0].device
    cdev = _comm_device(pred[0], group)
    gids = graph_ids_local.to(cdev, torch.long)
    nat = num_atoms_local.to(cdev, torch.long)
    assert gids.numel() == nat.numel() and pred[0].size(0) == int(nat.sum()) and pred[2].size(0) == int((nat * (nat - 1)).sum())
    # 1. graph counts
    cnt = torch.tensor([gids.numel()], device=cdev, dtype=torch.long)
    counts = torch.empty(world, device=cdev, dtype=torch.long)
    dist.all_gather_into_tensor(counts, cnt, group=group)
    counts = counts.tolist()
    gmax = max(max(counts), 1)                        # (a rank, or the whole job, may be empty)
    # 2. the [graph id | atoms] tables
    meta = torch.zeros(2, gmax, device=cdev, dtype=torch.long)
    meta[0, :gids.numel()], meta[1, :gids.numel()] = gids, nat
    metas = torch.empty(world * 2 * gmax, device=cdev, dtype=torch.long)        # (flat: the gloo backend wants the outputs concatenated along dim 0)
    dist.all_gather_into_tensor(metas, meta.view(-1), group=group)
    metas = metas.view(world, 2, gmax).cpu()
    ids_r = [metas[r, 0, :c] for r, c in enumerate(counts)]
    nat_r = [metas[r, 1, :c] for r, c in enumerate(counts)]
    n_r = [int(x.sum()) for x in nat_r]
    e_r = [int((x * (x - 1)).sum()) for x in nat_r]
    lmax = max(max(15 * n + 6 * e for n, e in zip(n_r, e_r)), 1)
    # global order = ascending graph id; where every graph's rows start in it
    ids_all, nat_all = torch.cat(ids_r), torch.cat(nat_r)
    order = torch.argsort(ids_all, stable=True)
    nat_sorted = nat_all[order]
    # 3. the payload
    flat = torch.zeros(lmax, device=cdev, dtype=torch.float32)
    o = 0
    for t in pred:
        flat[o:o + t.numel()] = t.reshape(-1).to(cdev, torch.float32)
        o += t.numel()
    to_all = dst is None
    if to_all:
        bufs = torch.empty(world * lmax, device=cdev, dtype=torch.float32)
        dist.all_gather_into_tensor(bufs, flat, group=group)
        bufs = bufs.view(world, lmax)
    else:
        bufs = [torch.empty(lmax, device=cdev, dtype=torch.float32) for _ in range(world)] if rank == dst else None
        dist.gather(flat, bufs, dst=dst, group=group)
    del flat
    if not (to_all or rank == dst):
        return None, nat_sorted
    out = []
    for kind, (width, rows_of) in enumerate(((12, lambda n: n), (3, lambda n: n), (6, lambda n: n * (n - 1)))):
        rows_sorted = rows_of(nat_sorted)
        start_sorted = torch.zeros(order.numel(), dtype=torch.long)           # first row of the k-th graph of the global order
        start_sorted[1:] = rows_sorted.cumsum(0)[:-1]
        start_of = torch.empty_like(start_sorted)                               # ... indexed by position in the rank-major list
        start_of[order] = start_sorted
        res = torch.empty(int(rows_sorted.sum()), width, device=cdev, dtype=torch.float32)
        g0 = 0
        for r in range(world):
            rows = rows_of(nat_r[r])
            n_rows = int(rows.sum())
            if n_rows:
                src0 = torch.zeros(counts[r], dtype=torch.long)                # first row of each of the rank's graphs in its own block
                src0[1:] = rows.cumsum(0)[:-1]
                idx = torch.repeat_interleave(start_of[g0:g0 + counts[r]] - src0, rows) + torch.arange(n_rows)
                off = (0, 12 * n_r[r], 15 * n_r[r])[kind]
                res.index_copy_(0, idx.to(cdev), bufs[r][off:off + n_rows * width].view(n_rows, width))
            g0 += counts[r]
        out.append(res.to(out_dev))
    return out, nat_sorted


class SamplingJob:
    """A multi-pharmacophore sampling job (BASELINE config 4: P pharmacophores x S samples): `phores` = list of
    (x [p,18], pos [p,3], norm [p,3], center [3]); graph g samples pharmacophore graph_phore[g] with num_atoms[g] atoms.
    The reference serves such a job with a serial loop over pharmacophores (sample_all.py:69-175, one `model.sample` per
    pharmacophore); here the graphs of all pharmacophores are one pool, partitioned over GPUs and cut into batches."""

    def __init__(self, phores, graph_phore, num_atoms):
        self.phores = phores
        self.graph_phore = torch.as_tensor(graph_phore, dtype=torch.long)
        self.num_atoms = torch.as_tensor(num_atoms, dtype=torch.long)
        assert self.graph_phore.numel() == self.num_atoms.numel()

    @property
    def n_graphs(self):
        return int(self.num_atoms.numel())

    @property
    def n_phore(self):
        """Pharmacophore nodes of every graph (the node term of the partition's cost model)."""
        sizes = torch.tensor([int(ph[0].size(0)) for ph in self.phores], dtype=torch.long)
        return sizes[self.graph_phore]

    def batch_inputs(self, gids):
        xs, ps, ns, cs, bp = [], [], [], [], []
        for i, g in enumerate(gids.tolist()):
            x, pos, norm, center = self.phores[int(self.graph_phore[g])]
            xs.append(x), ps.append(pos), ns.append(norm), cs.append(center.view(1, 3))
            bp.append(torch.full((x.size(0),), i, dtype=torch.long))
        return torch.cat(xs), torch.cat(ps), torch.cat(ns), torch.cat(bp), self.num_atoms[gids], torch.cat(cs)


def sample_job_shard(model, job, graph_ids, batch_size=128, seed=0, num_steps=None, pos_guidance_opt=None,
                     guidance_norm='batch_size'):
    """Sample the graphs `graph_ids` (ascending) of `job` on this process's GPU in batches of <= batch_size.  Noise is keyed
    by the GLOBAL graph id, so the result of a graph does not depend on which shard or batch it lands in.
    guidance_norm: the guidance energies are means over the graphs of a call (sample_utils.py:135-165 divide by num_graphs).
      'batch_size' (default, an INTENTIONAL deviation for partitioned jobs): every call divides by `batch_size`, also a shorter
                   tail batch or a shard with fewer graphs -- the drift a graph feels is then the same in every partition of the
                   job (sharded == unsharded, tests/test_gpu_sharding.py);
      'actual'   : divide by the number of graphs in the call, as the reference's loop does for its last batch
                   (sample_all.py:88: n = min(batch_size, remaining)) -- a tail-batch graph then gets the reference's stronger drift.
    Returns pred = [logits_node, pos, logits_edge] in graph_ids order."""
    if guidance_norm not in ('batch_size', 'actual'):
        raise ValueError(f'guidance_norm={guidance_norm!r}')
    parts = ([], [], [])
    for b0 in range(0, int(graph_ids.numel()), batch_size):
        gids = graph_ids[b0:b0 + batch_size]
        hp, pp, pn, bp, na, centers = job.batch_inputs(gids)
        res = model.sample_batch(hp, pp, pn, bp, na, centers, pos_guidance_opt=pos_guidance_opt, rng='device', seed=seed,
                                 return_traj=False, num_steps=num_steps, graph_ids=gids,
                                 guidance_batch=batch_size if guidance_norm == 'batch_size' else int(gids.numel()))
        for acc, t in zip(parts, res['pred']):
            acc.append(t)
    dev = next(model.parameters()).device
    empty = (torch.zeros(0, 12, device=dev), torch.zeros(0, 3, device=dev), torch.zeros(0, 6, device=dev))
    return [torch.cat(a) if a else e for a, e in zip(parts, empty)]


def run_sampling_job(model, job, world=1, rank=0, batch_size=128, seed=0, num_steps=None, pos_guidance_opt=None, group=None):
    """Config 4 end to end on one rank: cost-balanced partition (graph_cost) of ALL graphs of the job, this rank's shard in batches, then
    the one collective of the path (gather of `pred`).  Returns (pred_global, num_atoms_global) in global graph order when
    torch.distributed is initialised (every rank), else this rank's (pred, num_atoms) in its graph_ids order."""
    mine = partition_graphs(job.num_atoms, world, job.n_phore)[rank]
    pred = sample_job_shard(model, job, mine, batch_size, seed, num_steps, pos_guidance_opt)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        return gather_predictions(pred, job.num_atoms[mine], mine, group)
    return pred, job.num_atoms[mine]


def allreduce_gradients(params, group=None, average=True):
    """Data-parallel training step (reference harness run/run.py:160-311 is single-GPU): the gradients
    of all trainable parameters (20.8 MB of fp32 for PhoreDiff) travel as ONE flat bucket, one all-reduce per step.
    xGMI rings are per-link bound (~153 GB/s), so one large message beats per-tensor collectives by the launch latency
    of ~600 small ones.  Parameters that received no gradient on this rank contribute zeros.  Returns the bucket size."""
    params = [p for p in params if p.requires_grad]
    if not params:
        return 0
    dev, dt = params[0].device, params[0].dtype
    flat = torch.zeros(sum(p.numel() for p in params), dtype=dt, device=dev)
    off = 0
    for p in params:
        if p.grad is not None:
            flat[off:off + p.numel()].copy_(p.grad.reshape(-1))
        off += p.numel()
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        if average:
            flat /= dist.get_world_size(group)
    off = 0
    for p in params:
        g = flat[off:off + p.numel()].view_as(p)
        if p.grad is None:
            p.grad = g.clone()
        else:
            p.grad.copy_(g)
        off += p.numel()
    return flat.numel()


class GradientBuckets:
    """Bucketed gradient all-reduce launched from gradient hooks (the reference's harness, run/run.py:160-311,
    is single-GPU).  Parameters are grouped into ~`bucket_mb` MB buckets in reverse registration order (roughly the order their
    gradients are produced); a `post_accumulate_grad` hook counts arrivals and, when a bucket is complete, packs it and starts
    an ASYNCHRONOUS all-reduce (RCCL runs it on its own stream: the collective of bucket k travels over xGMI while the
    backward of the earlier layers is still computing).  Buckets are launched STRICTLY IN INDEX ORDER (bucket k only after
    buckets 0..k-1), as DDP does: every rank then issues the same sequence of collectives even if its autograd order differs
    (a parameter unused on one rank, a data-dependent branch) -- a bucket that completes early waits for its predecessors.
    `finish()` -- before `optimizer.step()` -- launches whatever did not fire (parameters without a gradient contribute zeros),
    waits, averages and writes the results back into `.grad`.  One backward per `finish()`: a second `backward()` before
    `finish()` (gradient accumulation) would be dropped by the write-back, so it raises instead.
    A few-MB bucket is the size at which an xGMI ring (per-link bound, ~153 GB/s) is already bandwidth- rather than
    latency-dominated; PhoreDiff's 20.8 MB of gradients make ~5 buckets.  The collectives run whenever a process group is
    initialised (a 1-rank `nccl` group included: tests/test_gpu_training.py drives the RCCL path that way on one GPU)."""

    def __init__(self, params, bucket_mb=4.0, group=None, average=True):
        self.group, self.average = group, average
        params = [p for p in params if p.requires_grad]
        cap = int(bucket_mb * (1 << 20) / 4)
        self.buckets, cur, n = [], [], 0
        for p in reversed(params):
            cur.append(p)
            n += p.numel()
            if n >= cap:
                self.buckets.append(cur)
                cur, n = [], 0
        if cur:
            self.buckets.append(cur)
        self.owner = {id(p): bi for bi, b in enumerate(self.buckets) for p in b}
        self.flat = [None] * len(self.buckets)
        self.work = [None] * len(self.buckets)
        self.fired = [set() for _ in self.buckets]     # parameters whose gradient has arrived in this backward
        # parameters a bucket does not wait for: those that produced no gradient in the previous step (an unused parameter would
        # otherwise hold its bucket -- and, launches being in order, every later one -- back until finish())
        self.absent = [set() for _ in self.buckets]
        self.next_launch = 0                 # buckets [0, next_launch) have been launched
        self.launch_log = []                 # (bucket, launched from a hook?) in launch order, per backward (tests)
        self.hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in params]

    def _active(self):
        return dist.is_available() and dist.is_initialized()

    def _on_grad(self, p):
        bi = self.owner[id(p)]
        if id(p) in self.fired[bi]:
            raise RuntimeError('GradientBuckets: a parameter received a second gradient before finish() -- call finish() after '
                               'every backward() (gradient accumulation over micro-batches is not supported by the hook path)')
        if self.work[bi] is not None:
            # earlier buckets have collectives in flight: drain them and start over, so that the state is usable after the error
            # (on more than one rank the peers must reach the same decision -- the absent set is agreed on in finish())
            self._abandon()
            raise RuntimeError('GradientBuckets: a parameter that had no gradient in the previous step received one after its '
                               'bucket was launched; call reset_absent() when the set of used parameters changes')
        self.fired[bi].add(id(p))
        while self.next_launch < len(self.buckets) and self._complete(self.next_launch):
            self._launch(self.next_launch, True)

    def _complete(self, bi):
        return len(self.fired[bi] | self.absent[bi]) == len(self.buckets[bi])

    def _abandon(self):
        """Wait for whatever is in flight and forget this backward (error path)."""
        for bi, w in enumerate(self.work):
            if w is not None and w is not True:
                w.wait()
            self.flat[bi], self.work[bi], self.fired[bi] = None, None, set()
        self.next_launch = 0
        self.launch_log = []
        self.reset_absent()

    def reset_absent(self):
        """Forget which parameters were unused in the previous step (every bucket waits for all of its parameters again)."""
        self.absent = [set() for _ in self.buckets]

    def _launch(self, bi, from_hook=False):
        assert bi == self.next_launch and self.work[bi] is None
        b = self.buckets[bi]
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in b])
        self.flat[bi] = flat
        self.work[bi] = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True) if self._active() else True
        self.next_launch = bi + 1
        self.launch_log.append((bi, from_hook))

    def finish(self):
        """Wait for every bucket (launching the ones whose hooks did not all fire) and write the reduced gradients back."""
        world = dist.get_world_size(self.group) if self._active() else 1
        any_fired = any(self.fired)
        while self.next_launch < len(self.buckets):
            self._launch(self.next_launch)
        # which parameters produced a gradient, agreed on by ALL ranks (a parameter used on any rank is waited for on every rank
        # in the next step: per-rank sets would make the ranks launch different bucket sequences)
        used = torch.tensor([1 if id(p) in self.fired[bi] else 0 for bi, b in enumerate(self.buckets) for p in b], dtype=torch.int32)
        if world > 1:
            dev = self.buckets[0][0].device if dist.get_backend(self.group) != 'gloo' else torch.device('cpu')
            used = used.to(dev)
            any_t = torch.tensor([1 if any_fired else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(used, op=dist.ReduceOp.MAX, group=self.group)
            dist.all_reduce(any_t, op=dist.ReduceOp.MAX, group=self.group)
            any_fired = bool(int(any_t.item()))
        used = used.cpu().tolist()
        n_elems = 0
        k = 0
        for bi, b in enumerate(self.buckets):
            if self.work[bi] is not True:
                self.work[bi].wait()
            flat = self.flat[bi]
            if self.average and world > 1:
                flat /= world
            off = 0
            dst, src = [], []
            for p in b:
                g = flat[off:off + p.numel()].view_as(p)
                if p.grad is None:
                    p.grad = g.clone()
                else:
                    dst.append(p.grad)
                    src.append(g)
                off += p.numel()
            if dst:
                torch._foreach_copy_(dst, src)          # one multi-tensor launch per bucket instead of one copy per parameter
            n_elems += off
            if any_fired:                   # (a finish() without a backward says nothing about which parameters are used)
                self.absent[bi] = {id(p) for j, p in enumerate(b) if not used[k + j]}
            k += len(b)
            self.flat[bi], self.work[bi], self.fired[bi] = None, None, set()
        self.next_launch = 0
        return n_elems

    def remove(self):
        for h in self.hooks:
            h.remove()
        self.hooks = []
