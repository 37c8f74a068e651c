#!/usr/bin/env python3
"""Structural check of the engine's launch lists (advisor, round 5): every pair of launches that touch the same buffer region, at least one
of them writing, must be ORDERED -- by lane order or through the record / wait points between the lanes -- not merely far apart in time.

The engine spreads one denoiser step over four lanes (HIP streams) and, in the pipelined sampler, lets the end of one step overlap the start
of the next (`Engine.prog_step` / `prog_ahead` + the posterior kernels the caller puts on lanes 2 / 3).  Until round 6 the only evidence
that its order points are sufficient was bit-identity of repeated runs (tools/stress_*.py), which a race that timing keeps closed passes.
This tool derives, from the launch lists themselves:
  * the read and write regions of every launch: (workspace buffer, column range) from the pointer arguments and the PgGemm / PgSegAttn
    structs -- `access_of` below is the table of what every entry point of include/phoregen_hip.h reads and writes;
  * a vector clock per launch from the lane it is enqueued on and the record / wait entries (a wait refers to the latest record of its
    point in host order, as hipStreamWaitEvent does);
and reports every conflicting pair without a happens-before edge.  The pipelined loop is checked as the sequence
prog_ahead, (prog_step, the caller's launches of `_reverse_step_pipelined`, prog_ahead) x 3, so that hazards ACROSS steps are seen.

    python tools/check_schedule.py [graphs ...]        (GPU box: the lists hold device pointers; nothing is launched)
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

N_LANES = 4


class Buffers:
    """Device pointer -> (name, byte offset) over a set of named tensors."""

    def __init__(self):
        self.items = []          # (base, nbytes, name, row_bytes)

    def add(self, name, t):
        if torch.is_tensor(t) and t.is_cuda and t.numel():
            st = t.untyped_storage()
            row = t.stride(0) * t.element_size() if t.dim() >= 2 else 0
            self.items.append((st.data_ptr(), st.nbytes(), name, row))

    def add_ns(self, prefix, ns):
        for k, v in vars(ns).items():
            if torch.is_tensor(v):
                self.add(f'{prefix}.{k}', v)
            elif isinstance(v, (list, tuple)):
                for i, x in enumerate(v):
                    if torch.is_tensor(x):
                        self.add(f'{prefix}.{k}[{i}]', x)

    def finish(self):
        seen, out = set(), []
        for it in sorted(self.items, key=lambda r: (r[0], -r[1])):
            if it[0] not in seen:                 # (views of one storage: the first, largest entry names it)
                seen.add(it[0])
                out.append(it)
        self.items = out
        return self

    def add_queue(self, iters, n):
        """A triplet queue (BatchPlan.tri_iters, int32 [n, 4]): the ligands it covers, as the set of their first bond rows."""
        self.queues = getattr(self, 'queues', {})
        self.queues[iters.data_ptr()] = frozenset(int(v) for v in iters[:n, 2].cpu().tolist())

    def region(self, ptr, width=None):
        """(buffer name, first column byte, last column byte + 1) of a pointer; width = bytes per row touched (None: the whole row)."""
        if not ptr:
            return None
        for base, nbytes, name, row in self.items:
            if base <= ptr < base + nbytes:
                if not row or width is None:
                    return (name, 0, 1 << 40)
                c0 = (ptr - base) % row
                return (name, c0, c0 + width)
        return None                                # not a workspace buffer: weights, topology (constant during a run)


def _val(a):
    if a is None:
        return 0
    if hasattr(a, '_obj'):
        return a
    return a.value if hasattr(a, 'value') else int(a)


def access_of(name, args, buf):
    """(reads, writes) of one launch: lists of regions.  The table of include/phoregen_hip.h's entry points as the engine calls them."""
    R, W = [], []
    a = [_val(x) for x in args]
    reg = buf.region

    def rd(p, w=None):
        r = reg(p, w)
        if r:
            R.append(r)

    def wr(p, w=None):
        r = reg(p, w)
        if r:
            W.append(r)
    if name == 'pg_gemm':
        g = a[0]._obj
        rd(g.X, 4 * g.K1), rd(g.X2, 4 * g.K2), rd(g.add1, 4 * g.N), rd(g.add2, 4 * g.N)
        wr(g.Y, 4 * g.N)
    elif name == 'pg_seg_attn':
        s = a[1]._obj
        # modes: 0 knn-node, 1 knn-pos, 2 bond-node, 3 bond-pos, 4 triplet, 5 pharmacophore encoder.  The bond-node sub-layer reads no
        # coordinates (csrc/node_attn.hip: x only `if constexpr (KNN || POS)`), the neighbour lists / gate / direction vectors belong to the knn modes
        for p_ in ((s.x,) if s.mode != 2 else ()) + ((s.nrm, s.nbr, s.deg, s.ew) if s.mode in (0, 1) else ()) + (s.G, s.resid):
            rd(C.cast(p_, C.c_void_p).value)
        for p_ in (s.Csrc_k, s.Csrc_v, s.Cdst_k, s.Cdst_v, s.q):
            rd(C.cast(p_, C.c_void_p).value, 512)
        fused = bool(s.W2k_l) and s.mode != 5
        for p_ in (s.U, s.S, s.swn):               # fused forms: scratch of the one-pass fallback (may be written); plain form: U read, S / swn written
            v = C.cast(p_, C.c_void_p).value
            (wr if (fused or p_ is not s.U) else rd)(v)
            if fused:
                rd(v)
        out = reg(C.cast(s.out, C.c_void_p).value, 512)
        if out:
            # a staged triplet launch writes the rows of ITS queue's ligands only: two launches over disjoint queues may run beside each
            # other (engine: options.tri_overlap); the set of ligands rides on the region
            rows = getattr(buf, 'queues', {}).get(C.cast(s.tri_iters, C.c_void_p).value) if (s.mode == 4 and s.tri_iters) else None
            W.append(out if rows is None else out + (rows,))
        wr(C.cast(s.dx, C.c_void_p).value)
        if s.accumulate_dx:
            rd(C.cast(s.dx, C.c_void_p).value)
        v = C.cast(s.tri_counter, C.c_void_p).value
        rd(v), wr(v)
    elif name == 'pg_embed_ctx':
        if a[10]:                                  # features (embed_ctx_kernel reads an input only for the output it serves)
            rd(a[1]), rd(a[3]), rd(a[7]), wr(a[10])
        if a[11]:                                  # coordinates
            rd(a[2]), rd(a[8]), wr(a[11])
    elif name == 'pg_embed_bond':
        rd(a[1]), rd(a[3]), wr(a[7])
    elif name == 'pg_knn_ctx':
        rd(a[1]), wr(a[3]), wr(a[4])
    elif name == 'pg_edge_gate':
        rd(a[1]), rd(a[2]), rd(a[3]), wr(a[11])
    elif name == 'pg_knn_group_by_kind':
        rd(a[2]), wr(a[2]), rd(a[3]), rd(a[4]), wr(a[4])
    elif name == 'pg_lig_normals':
        rd(a[1]), rd(a[2]), wr(a[4])
    elif name == 'pg_bond_smear':
        rd(a[1]), wr(a[2])
    elif name == 'pg_layer_geom':
        rd(a[1]), rd(a[2]), rd(a[3]), rd(a[4]), wr(a[5]), wr(a[6]), wr(a[7])
    elif name == 'pg_apply_dx':
        rd(a[1]), rd(a[2]), rd(a[3]), wr(a[4])
    elif name == 'pg_attn_fold_query':
        rd(a[0]), wr(a[5])
    elif name == 'pg_attn_unfold_value':
        rd(a[0]), rd(a[1]), wr(a[6], 512)
    elif name == 'pg_rows_linear':
        rd(a[0], 4 * a[2]), wr(a[8], 4 * a[5])
    elif name == 'pg_atom_count':
        rd(a[0]), rd(a[1]), wr(a[6]), wr(a[7])
    else:
        raise KeyError(f'check_schedule: no access table for {name}')
    return R, W


class Timeline:
    """Launches with vector clocks; `launch` / `record` / `wait` in host order."""

    def __init__(self):
        self.clock = [[0] * N_LANES for _ in range(N_LANES)]      # per lane: what it has seen of every lane
        self.points = {}
        self.ops = []                                             # (label, lane, clock snapshot, reads, writes)

    def launch(self, label, lane, reads, writes):
        self.clock[lane][lane] += 1
        self.ops.append((label, lane, list(self.clock[lane]), reads, writes))

    def record(self, key, lane):
        self.points[key] = list(self.clock[lane])

    def wait(self, key, lane):
        if key in self.points:                                    # (a wait with no record before it in host order waits for nothing)
            self.clock[lane] = [max(a, b) for a, b in zip(self.clock[lane], self.points[key])]

    def sync_all(self):
        """A device-wide synchronisation point (the start of a run)."""
        top = [max(c[i] for c in self.clock) for i in range(N_LANES)]
        self.clock = [list(top) for _ in range(N_LANES)]

    def add_program(self, prog, buf, tag, drop=None):
        """`drop(k, what, pt, lane)` -> True removes that record / wait (tests: the check must notice a missing order point)."""
        for k, (fn, args, lane) in enumerate(prog):
            if lane >= 0:
                r, w = access_of(fn.__name__, args, buf)
                self.launch(f'{tag}[{k}] {fn.__name__}', lane, r, w)
            elif fn.kind == 'order':
                for what, pt, ln in fn.ops:
                    if drop is not None and drop(k, what, pt, ln):
                        continue
                    (self.record if what == 'record' else self.wait)((tag.split('#')[0], pt), ln)

    def hazards(self):
        out = []
        last = {}                         # buffer name -> list of op indices touching it
        for i, (label, lane, clk, R, W) in enumerate(self.ops):
            for kind, regs in (('r', R), ('w', W)):
                for (name, c0, c1, *rows) in regs:
                    for j, jkind, d0, d1, jrows in last.get(name, ()):
                        if jkind == 'r' and kind == 'r':
                            continue
                        if c1 <= d0 or d1 <= c0:
                            continue
                        if rows and jrows and not (rows[0] & jrows[0]):
                            continue                              # disjoint sets of ligands (rows) of the same columns
                        lj, lanej, clkj = self.ops[j][0], self.ops[j][1], self.ops[j][2]
                        if lanej == lane or clk[lanej] >= clkj[lanej]:
                            continue                              # j happens-before i
                        out.append((lj, lanej, jkind, label, lane, kind, name, (max(c0, d0), min(c1, d1))))
            for kind, regs in (('r', R), ('w', W)):
                for (name, c0, c1, *rows) in regs:
                    last.setdefault(name, []).append((i, kind, c0, c1, rows))
        # one line per (earlier launch, later launch, buffer)
        seen, uniq = set(), []
        for h in out:
            key = (h[0].split('] ')[-1], h[1], h[3].split('] ')[-1], h[4], h[6], h[0].split('[')[0], h[3].split('[')[0])
            if key not in seen:
                seen.add(key)
                uniq.append(h)
        return uniq


def caller_launches(tl, st, eng, buf, i, guided, tag):
    """The launches `PhoreDiff._reverse_step_pipelined` adds around the two programs (phoregen_amd/models/diffusion.py: keep in step with it):
    node posterior on lane 2, bond posterior on lane 3, [the bond_drawn point], the step counter of the next step on lane 2, `prog_ahead`,
    then on lane 0 the guidance (behind bond_drawn) and the Gaussian posterior."""
    w = eng.ws
    reg = lambda t: buf.region(t.data_ptr())
    cur = i % 2
    tl.launch(f'{tag} posterior(node)', 2, [reg(w.out_v), reg(st.log_node[cur])],
              [reg(st.log_node[1 - cur]), reg(w.in_h_node)] + ([reg(st.node_traj)] if st.node_traj is not None else []))
    tl.launch(f'{tag} posterior(edge)', 3, [reg(w.out_bond), reg(st.log_edge[cur])],
              [reg(st.log_edge[1 - cur]), reg(w.in_h_edge)] + ([reg(st.edge_traj)] if st.edge_traj is not None else []))
    if guided:
        tl.record(('caller', 'bond_drawn'), 3)
    tl.launch(f'{tag} in_t_next.fill_', 2, [], [reg(w.in_t_next)])
    tl.add_program(eng.prog_ahead, buf, f'ahead#{tag}')
    if guided:
        tl.wait(('caller', 'bond_drawn'), 0)
        tl.launch(f'{tag} guidance', 0, [reg(w.in_pos), reg(w.in_h_edge)], [reg(st.grad), reg(st.gtmp), reg(st.cnt_ws), reg(st.mean_ws)])
    x0 = w.x[eng.final_idx]
    tl.launch(f'{tag} posterior(pos)', 0, [reg(w.in_pos), reg(x0)] + ([reg(st.grad)] if guided else []),
              [reg(w.in_pos), reg(w.x[0]), reg(st.x0_buf)] + ([reg(st.pos_traj)] if st.pos_traj is not None else []))


def check_engine(model, work, guided=False, steps=3, drop_step=None):
    """Hazards of the plain forward list and of `steps` pipelined sampler steps for one batch; returns (hazards_forward, hazards_pipelined, info).
    drop_step: see Timeline.add_program (applied to `prog_step`)."""
    G = int(work['num_atoms'].numel())
    st = model.begin_sampling(work['h_phore'], work['pos_phore'], work['phore_norm'], work['batch_phore'], work['num_atoms'],
                              torch.zeros(G, 3), rng='device', seed=0, return_traj=True, num_steps=steps + 1, pipeline=True)
    eng = st.eng
    buf = Buffers()
    buf.add_ns('ws', eng.ws)
    buf.add_ns('st', st)
    buf.finish()
    buf.add_queue(eng.plan.tri_iters, eng.plan.n_tri_iters)
    for q in (eng.plan.tri_split or {}).values():
        buf.add_queue(q[0], q[1])
    tl = Timeline()
    tl.add_program(eng.prog_fwd, buf, 'fwd')
    fwd = tl.hazards()
    pipe = None
    if st.pipelined:
        tl = Timeline()
        tl.launch('init embed_ctx', 0, [], [buf.region(eng.ws.x[0].data_ptr())])
        tl.sync_all()                                          # (fork_lanes after the initial state)
        tl.launch('init in_t_next.fill_', 2, [], [buf.region(eng.ws.in_t_next.data_ptr())])
        tl.add_program(eng.prog_ahead, buf, 'ahead#init')
        for i in range(steps):
            tl.add_program(eng.prog_step, buf, f'step#{i}', drop=drop_step)
            caller_launches(tl, st, eng, buf, i, guided, f's{i}')
        pipe = tl.hazards()
    info = dict(eng=eng, graphs=G, n_bond=eng.plan.n_bond, pipelined=bool(st.pipelined), v2=bool(eng.ahead_v2), launches_fwd=sum(1 for e in eng.prog_fwd if e[2] >= 0),
                launches_step=sum(1 for e in eng.prog_step if e[2] >= 0) if st.pipelined else 0)
    return fwd, pipe, info


def fmt(h):
    return f'{h[6]} cols [{h[7][0]}, {h[7][1]}): {h[0]} (lane {h[1]}, {h[2]}) || {h[3]} (lane {h[4]}, {h[5]})'


if __name__ == '__main__':
    from bench import ligphore_workload
    from phoregen_amd.config import default_model_config
    from phoregen_amd.models.diffusion import PhoreDiff
    from phoregen_amd.weights import init_deterministic_
    sizes = [int(a) for a in sys.argv[1:]] or [2, 8, 16, 48, 72, 128]
    model = init_deterministic_(PhoreDiff(default_model_config(), 'zinc_300'), 0).eval().to('cuda')
    bad = 0
    for G in sizes:
        for guided in (False, True):
            fwd, pipe, info = check_engine(model, ligphore_workload(G), guided)
            n = len(fwd) + len(pipe or [])
            bad += n
            print(f'G={G:4d} guidance={"on " if guided else "off"}: {info["launches_fwd"]} launches per forward, {info["launches_step"]} per pipelined step '
                  f'(v2={info["v2"]}): {len(fwd)} unordered conflicting pairs in the forward list, {len(pipe) if pipe is not None else "-"} in 3 pipelined steps')
            for h in fwd + (pipe or []):
                print('   ', fmt(h))
            model._engine = None
    print('ORDERED' if bad == 0 else f'{bad} UNORDERED PAIRS')
    sys.exit(1 if bad else 0)
