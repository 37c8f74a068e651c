"""Batch topology ("plan") of one ligand-pharmacophore batch, constant over the 1000 denoise steps.

Host-side mirror of what the reference recomputes every step: compose_context's stable sort
(models/common.py:180-208), the bond index remap (models/diffusion.py:201) and the triplet
enumeration (models/uni_denoiser.py:101-121, here an n x n edge-id table per ligand).
"""
import ctypes as C

import torch

from . import hip


class BatchPlan:
    def __init__(self, batch_node, batch_phore, edge_index, batch_edge, n_graphs, device):
        """batch_node [N_lig], batch_phore [N_ph] (graph id per ligand atom / phore node, ascending),
        edge_index [2,E] ligand-local atom ids (fully connected directed ligand graphs), batch_edge [E]."""
        bn, bp = batch_node.detach().cpu().long(), batch_phore.detach().cpu().long()
        ei, be = edge_index.detach().cpu().long(), batch_edge.detach().cpu().long()
        B = int(n_graphs)
        if bn.numel() and (bn[1:] < bn[:-1]).any() or bp.numel() and (bp[1:] < bp[:-1]).any():
            raise ValueError('phoregen_amd: batch vectors must be sorted by graph id')
        nlig = torch.bincount(bn, minlength=B)
        nph = torch.bincount(bp, minlength=B)
        self.n_graphs, self.n_lig, self.n_phore, self.n_bond = B, int(bn.numel()), int(bp.numel()), int(ei.size(1))
        self.n_ctx = self.n_lig + self.n_phore
        if int((nlig * (nlig - 1)).sum()) != self.n_bond:
            raise ValueError('phoregen_amd: the bond graph must be the fully connected directed ligand graph '
                             '(utils/sample_utils.py:40-54 / datasets/transform.py:488-501)')
        tot = nlig + nph
        g_ctx_off = torch.zeros(B + 1, dtype=torch.long)
        g_ctx_off[1:] = tot.cumsum(0)
        lig_off = torch.zeros(B + 1, dtype=torch.long)
        lig_off[1:] = nlig.cumsum(0)
        ph_off = torch.zeros(B + 1, dtype=torch.long)
        ph_off[1:] = nph.cumsum(0)
        a = torch.arange(self.n_lig)
        lig2ctx = g_ctx_off[bn] + nph[bn] + (a - lig_off[bn])
        p = torch.arange(self.n_phore)
        phore2ctx = g_ctx_off[bp] + (p - ph_off[bp])
        ctx_graph = torch.empty(self.n_ctx, dtype=torch.long)
        ctx_graph[lig2ctx] = bn
        ctx_graph[phore2ctx] = bp
        is_lig = torch.zeros(self.n_ctx, dtype=torch.uint8)
        is_lig[lig2ctx] = 1
        g_eid_off = torch.zeros(B + 1, dtype=torch.long)
        g_eid_off[1:] = (nlig * nlig).cumsum(0)
        ls, ld = ei[0] - lig_off[be], ei[1] - lig_off[be]
        if self.n_bond and ((ls < 0).any() or (ld < 0).any() or (ls >= nlig[be]).any() or (ld >= nlig[be]).any()
                            or (bn[ei[0]] != be).any()):
            raise ValueError('phoregen_amd: edge_index / batch_edge inconsistent with batch_node')
        if self.n_bond and (be[1:] < be[:-1]).any():
            raise ValueError('phoregen_amd: bond edges must be grouped by graph (batch_edge ascending)')
        # ---- internal bond order: per graph, TARGET-major (all edges k -> i of target i together, k ascending) ----
        # Inside the engine h_bond and every bond-row tensor live in this order: the rows P[k -> j] that the triplets of a source
        # atom j read (uni_denoiser.py:123-165) and the rows a target node's bond attention reads (:43-59) are then one
        # contiguous block each.  The caller's order (make_edge_data's "all a<b, then all b<a", or FeaturizeLigandBond's,
        # which already is target-major) only exists at the boundary: edge_ref[e_internal] = caller's row.
        nmax = int(nlig.max()) + 1 if B else 1
        order = torch.argsort((be * nmax + ld) * nmax + ls, stable=True) if self.n_bond else torch.zeros(0, dtype=torch.long)
        self.edge_identity = bool(torch.equal(order, torch.arange(self.n_bond)))
        ei_ref, ei = ei, ei[:, order]
        ls, ld = ls[order], ld[order]
        inv = torch.empty_like(order)
        inv[order] = torch.arange(self.n_bond)
        eid = torch.full((int(g_eid_off[-1]),), -1, dtype=torch.long)
        eid[g_eid_off[be] + ls * nlig[be] + ld] = torch.arange(self.n_bond)
        if int((eid >= 0).sum()) != self.n_bond:
            raise ValueError('phoregen_amd: duplicate bond edges')

        self.device = device
        i32 = lambda t: t.to(torch.int32).to(device)
        self.num_atoms = nlig
        self.g_ctx_off, self.g_nph, self.g_nlig, self.g_eid_off = i32(g_ctx_off), i32(nph), i32(nlig), i32(g_eid_off)
        self.eid, self.ctx_graph, self.ctx_is_lig = i32(eid), i32(ctx_graph), is_lig.to(device)
        self.lig2ctx, self.phore2ctx = i32(lig2ctx), i32(phore2ctx)
        self.lig2ctx_long = lig2ctx.to(device)
        self.phore2ctx_long = phore2ctx.to(device)
        self.bond_src, self.bond_dst = i32(lig2ctx[ei[0]]), i32(lig2ctx[ei[1]])
        self.bond_graph, self.lig_graph, self.phore_graph = i32(be), i32(bn), i32(bp)
        # one 16-byte descriptor per bond edge j->i (triplet kernel): no dependent index chain per segment
        self.bond_desc = i32(torch.stack([lig2ctx[ei[0]], ld + (ls << 16), nlig[be], g_eid_off[be]], 1).contiguous()) \
            if self.n_bond else torch.zeros(0, 4, dtype=torch.int32, device=device)
        self.g_lig_off = i32(lig_off)
        bond_off = torch.zeros(B + 1, dtype=torch.long)
        bond_off[1:] = torch.bincount(be, minlength=B).cumsum(0)
        self.g_bond_off = i32(bond_off)                 # first bond row of each graph (graph-keyed device RNG)
        self.batch_node, self.batch_edge, self.edge_index = bn.to(device), be.to(device), ei_ref.to(device)     # caller's order
        self.edge_ref, self.edge_int = i32(order), i32(inv)            # internal -> caller's row, caller's -> internal row
        self.edge_ref_long, self.edge_int_long = order.to(device), inv.to(device)

        t = hip.PgTopo()
        t.n_graphs, t.n_ctx, t.n_lig, t.n_phore, t.n_bond = B, self.n_ctx, self.n_lig, self.n_phore, self.n_bond
        t.max_nlig = int(nlig.max()) if B else 0
        t.max_gctx = int(tot.max()) if B else 0
        # cost-balanced chunks of consecutive bond edges for the 256 persistent triplet workgroups:
        # cost(edge) ~ row tiles of its ligand + a fixed per-segment part (query fold / value unfold)
        # triplet segments are visited in source-atom order (edge j->i reads the rows P[k->j], shared by all i)
        order = torch.argsort(ei[0], stable=True) if self.n_bond else torch.zeros(0, dtype=torch.long)
        self.tri_order = i32(order)
        cost = ((nlig[be[order]] + 15) // 16).double() + 1.0
        csum = torch.cat([torch.zeros(1, dtype=torch.double), cost.cumsum(0)])
        targets = torch.linspace(0, float(csum[-1]), 257, dtype=torch.double)
        self.tri_chunks = i32(torch.searchsorted(csum, targets).clamp(max=self.n_bond))
        self.tri_chunks[0], self.tri_chunks[-1] = 0, self.n_bond
        # source-atom groups of the LDS-staged triplet kernel (csrc/triplet2.hip): consecutive source atoms of one ligand whose
        # P blocks ((n-1) rows each) fit the 80 staged rows; pulled from a queue, most expensive first
        rows_cap, waves = 80, 12
        groups = []
        for gi in range(B):
            n = int(nlig[gi])
            if n < 2 or n - 1 > rows_cap or n > 96:
                continue
            A = max(1, min(rows_cap // (n - 1), n))
            for j0 in range(0, n, A):
                groups.append((gi, n, j0, min(A, n - j0)))
        # small batches (fewer than 2 groups per workgroup): a group's segments are handed out in 2 parts of whole 12-wave rounds, so that
        # the queue can level the workgroups.  (Rounds 2 - 5: up to 4 parts below 4 groups per workgroup; with the queue's tail in half-groups
        # -- below -- coarser is better: 8 / 16 / 32 graphs 1.93 / 2.87 / 4.98 -> 1.90 / 2.82 / 4.90 ms per step, profiles/r06_triplet_queue_tail.txt)
        want_parts = 1 if len(groups) >= 2 * 256 or not groups else 2

        def queue(grps):
            its = []
            for gi, n, j0, a in grps:
                tiles = (n - 2 + 15) // 16              # rows of a segment: every atom but its source and its target
                n_seg = a * (n - 1)
                rounds = (n_seg + waves - 1) // waves
                parts = max(1, min(want_parts, rounds // 2))
                r0 = 0
                for k in range(parts):
                    r1 = r0 + rounds // parts + (1 if k < rounds % parts else 0)
                    s0, s1 = r0 * waves, min(r1 * waves, n_seg)
                    its.append(((r1 - r0) * (tiles + 1.0) + 0.5, int(lig2ctx[lig_off[gi]]), n | (j0 << 8) | (a << 16), int(bond_off[gi]),
                                0 if parts == 1 else (s0 | (s1 << 16))))
                    r0 = r1
            its.sort(key=lambda r: -r[0])
            # the TAIL of the queue in half-groups (round 6): the last 128 entries -- the cheapest groups, handed out when the persistent
            # workgroups run dry -- cover half of their whole 12-wave rounds each, so the launch ends within half a group of its last workgroup
            # instead of a whole one (a group of a 40-atom ligand is 6.5 rounds = ~180 us).  Measured on the headline batch: the sub-layer alone
            # 1.957 -> 1.859 ms, the step 19.53 -> 19.25 ms; 16 graphs 2.92 -> 2.87 (the last 256: 128 is equal at 128 graphs and better below);
            # halving more of the queue, or the tail twice, loses to the repeated staging of the groups' rows (profiles/r06_triplet_queue_tail.txt)
            head, last = its[:max(len(its) - 128, 0)], its[max(len(its) - 128, 0):]
            halves = []
            for c, lig0, w1, boff, w3 in last:
                n_, a_ = w1 & 0xff, w1 >> 16
                s0, s1 = (w3 & 0xffff, w3 >> 16) if w3 else (0, a_ * (n_ - 1))
                rounds = (s1 - s0 + waves - 1) // waves
                if rounds < 2:
                    halves.append((c, lig0, w1, boff, w3))
                    continue
                mid = s0 + (rounds + 1) // 2 * waves
                halves += [(c / 2, lig0, w1, boff, s0 | (mid << 16)), (c / 2, lig0, w1, boff, mid | (s1 << 16))]
            its = head + sorted(halves, key=lambda r: -r[0])
            return torch.tensor([[r[1], r[2], r[3], r[4]] for r in its], dtype=torch.int32).reshape(-1, 4).to(device), len(its)
        usable = bool(B and int(nlig.max()) - 1 <= rows_cap and int(nlig.max()) <= 96)
        self.tri_iters, n_its = queue(groups)
        self.n_tri_iters = n_its if usable else 0
        # the same entries as two queues, by the row tiles of the ligand: the kernel is instantiated for the largest ligand a queue holds,
        # and the instance for 4 (5) tiles costs every segment ~4 % (8 %) -- a few 51+-atom ligands in a batch of smaller ones get their own
        # launch (engine: options.tri_split) and the rest run on the 3-tile instance
        small = [g_ for g_ in groups if g_[1] - 2 <= 48]
        self.tri_split = None
        if usable and small and len(small) < len(groups):
            big = [g_ for g_ in groups if g_[1] - 2 > 48]
            it_s, n_s = queue(small)
            it_b, n_b = queue(big)
            self.tri_split = dict(small=(it_s, n_s, max(g_[1] for g_ in small), torch.zeros(2, dtype=torch.int32, device=device)),
                                  big=(it_b, n_b, 0, torch.zeros(2, dtype=torch.int32, device=device)))
        self.tri_counter = torch.zeros(2, dtype=torch.int32, device=device)      # queue head + exit count; the kernel re-zeroes them
        for name in ('g_ctx_off', 'g_nph', 'g_nlig', 'g_eid_off', 'eid', 'ctx_graph', 'ctx_is_lig', 'lig2ctx',
                     'bond_src', 'bond_dst', 'bond_desc', 'g_bond_off'):
            setattr(t, name, getattr(self, name).data_ptr())
        t.edge_ref = None if self.edge_identity else self.edge_ref.data_ptr()
        self.topo = t
        self.topo_ref = C.byref(t)
        self.key = (bn.numel(), bp.numel(), ei.size(1), B)
        self._cpu_sig = (bn.clone(), bp.clone(), ei_ref.clone())     # (copies: .cpu().long() of a CPU long tensor is the caller's own storage)
        self.ws = None   # engine workspace, attached lazily

    def bwd_atom_order(self, grid):
        """Ligand atoms in the order the `grid` persistent workgroups of the triplet adjoint take them (PgSegAttnGrad.atom_order):
        sorted by cost (n - 1 segments x (ceil(n / 16) tiles + 1)) descending and dealt out in a snake, so that every workgroup gets
        the same mix (config-5 batch: heaviest workgroup 1.16 x the mean in index order, 1.01 x this way)."""
        cache = self.__dict__.setdefault('_bwd_atom_order', {})
        if grid not in cache:
            n = torch.repeat_interleave(self.num_atoms, self.num_atoms).double()
            cost = (n - 1) * (torch.ceil(n / 16) + 1)
            srt = torch.argsort(cost, descending=True, stable=True)
            idx = torch.arange(srt.numel())
            rnd, pos = idx // grid, idx % grid
            full = (rnd + 1) * grid <= srt.numel()                       # (a partial last round keeps its order)
            j = torch.where((rnd % 2 == 1) & full, grid - 1 - pos, pos)
            cache[grid] = srt[rnd * grid + j].to(torch.int32).to(self.device)
        return cache[grid]

    @staticmethod
    def _versions(tensors):
        """Version counters of the caller's index tensors, or None when one of them does not track versions (tensors created
        under torch.inference_mode()): then only the contents can tell."""
        try:
            return tuple(t._version for t in tensors)
        except RuntimeError:
            return None

    def matches(self, batch_node, batch_phore, edge_index):
        """Is this the topology the plan was built for?  The same tensor OBJECTS as the last match, unmodified since (version
        counters), answer without touching their contents; otherwise the contents are compared on the host -- for device tensors
        that is a device -> host copy which WAITS for everything enqueued before it (a training loop that calls compute_loss on the
        same batch object every step would otherwise drain the GPU once per step: 142 ms of a 258 ms step were spent in that wait).
        The plan keeps references to the three tensors that matched: an object that is held alive cannot have its storage
        recycled for a different batch of the same shapes."""
        now = (batch_node, batch_phore, edge_index)
        last = getattr(self, '_last_match', None)
        if last is not None and all(a is b for a, b in zip(now, last[0])):
            ver = self._versions(now)
            if ver is not None and ver == last[1]:
                return True
        bn, bp, ei = self._cpu_sig
        same = (batch_node.numel() == bn.numel() and batch_phore.numel() == bp.numel() and
                edge_index.shape == ei.shape and torch.equal(batch_node.cpu(), bn) and
                torch.equal(batch_phore.cpu(), bp) and torch.equal(edge_index.cpu(), ei))
        if same:
            self._last_match = (now, self._versions(now))
        return same


def make_edge_data(num_atoms, device=None):
    """Fully connected directed bond edges in the sampler's order: per graph all (a<b) pairs, then all
    reversed pairs (same contract as utils/sample_utils.py:40-54), built without a per-graph Python loop."""
    num_atoms = num_atoms.detach().cpu().long()
    device = device if device is not None else num_atoms.device
    B = num_atoms.numel()
    off = torch.zeros(B + 1, dtype=torch.long)
    off[1:] = num_atoms.cumsum(0)
    nmax = int(num_atoms.max()) if B else 0
    iu = torch.triu_indices(nmax, nmax, offset=1)                    # row-major (a<b) pairs of the largest graph
    srcs, dsts, bats = [], [], []
    # pairs of a graph with n atoms are exactly the (a<b) pairs of the nmax table with b < n, in the same order
    keep = iu[1][None, :] < num_atoms[:, None]                       # [B, P]
    g_idx, p_idx = keep.nonzero(as_tuple=True)
    a = iu[0][p_idx] + off[g_idx]
    b = iu[1][p_idx] + off[g_idx]
    half_cnt = keep.sum(1)
    # per graph: [half pairs..., reversed half pairs...]
    pos_in_g = torch.arange(g_idx.numel()) - torch.repeat_interleave(half_cnt.cumsum(0) - half_cnt, half_cnt)
    e_off = torch.zeros(B + 1, dtype=torch.long)
    e_off[1:] = (2 * half_cnt).cumsum(0)
    E = int(e_off[-1])
    src = torch.empty(E, dtype=torch.long)
    dst = torch.empty(E, dtype=torch.long)
    i1 = e_off[g_idx] + pos_in_g
    i2 = i1 + half_cnt[g_idx]
    src[i1], dst[i1] = a, b
    src[i2], dst[i2] = b, a
    batch = torch.repeat_interleave(torch.arange(B), 2 * half_cnt)
    return torch.stack([src, dst]).to(device), batch.to(device)
