"""PhoreDiff with the reference's nn.Module surface (models/diffusion.py:19-525), HIP-backed.

Same constructor, same attribute / state_dict names, same `forward` and `sample` contracts, so the
reference's sample_all.py can drive it.  `compute_loss` (training, config 5) needs backward kernels and
is not available yet: it raises instead of silently falling back to a CPU path.

Design differences that do not change results:
  * everything that is constant over the 1000 reverse steps is computed once per `sample` call:
    batch topology (compose_context sort, bond remap, triplet table), pharmacophore encoder, count heads
    (the reference recomputes them every step and discards the counts, diffusion.py:186-201,244,436);
  * one step = one pre-built list of kernel launches (phoregen_amd/engine.py), transition fused in
    csrc/posterior.hip, no host synchronisation inside the loop.
"""
import torch
from torch import nn
from torch.nn import functional as F

from . import get_denoiser_net, get_phore_encoder
from .common import GaussianSmearing, ShiftedSoftplus, TimeGaussianSmearing, get_beta_schedule
from .transition import ContigousTransition, GeneralCategoricalTransition
from .. import hip
from ..engine import Engine
from ..packing import ModelPack
from ..plan import BatchPlan, make_edge_data
from ..utils.sample_utils import sample_from_interval


class PhoreDiff(nn.Module):
    def __init__(self, config, data_name, **kwargs):
        super().__init__()
        self.config, self.data_name = config, data_name
        self.num_node_types = config.num_atom_classes
        self.num_edge_types = config.num_bond_classes
        self.bond_len_loss = config.bond_len_loss
        self.bond_diffusion = config.bond_diffusion
        self.bond_net_type = config.bond_net_type
        self.count_pred_type = config.count_pred_type
        self.max_atom, self.min_atom = 78, 4
        self.loss_weight = getattr(config, 'loss_weight', [1, 100, 100])
        self.count_factor = getattr(config, 'count_factor', 1)
        self.hp_emb_with_pos = getattr(config, 'hp_emb_with_pos', False)
        if (self.num_node_types, self.num_edge_types) != (12, 6) or not self.bond_diffusion or \
                self.bond_net_type != 'lin' or self.count_pred_type != 'boundary' or not self.hp_emb_with_pos or \
                config.diff.categorical_space != 'discrete' or config.hidden_dim != 128 or config.diff.time_dim != 10:
            raise NotImplementedError('phoregen_amd implements the configuration both shipped YAMLs use: 12 atom / 6 bond '
                                      'classes, bond_diffusion, bond_net_type=lin, count_pred_type=boundary, '
                                      'hp_emb_with_pos, discrete categorical space, hidden 128, time_dim 10')
        d = config.diff
        self.num_timesteps = d.num_timesteps
        self.categorical_space = d.categorical_space
        self.scaling = [1., 1., 1.]
        self.pos_transition = ContigousTransition(get_beta_schedule(num_timesteps=self.num_timesteps, **d.diff_pos))
        self.node_transition = GeneralCategoricalTransition(
            get_beta_schedule(num_timesteps=self.num_timesteps, **d.diff_atom), 12, init_prob=d.diff_atom.init_prob)
        self.edge_transition = GeneralCategoricalTransition(
            get_beta_schedule(num_timesteps=self.num_timesteps, **d.diff_bond), 6, init_prob=d.diff_bond.init_prob)

        H = config.hidden_dim
        self.node_embedder = nn.Linear(12, H - d.time_dim, bias=False)
        self.edge_embedder = nn.Linear(6, H - d.time_dim, bias=False)
        self.time_emb = nn.Sequential(TimeGaussianSmearing(stop=self.num_timesteps, num_gaussians=d.time_dim))
        self.phore_embedding = nn.Linear(config.phore_feat_dim, H)
        self.phore_encoder = get_phore_encoder(config.denoiser)
        assert config.denoiser.hidden_dim == H
        self.denoiser = get_denoiser_net(config.denoiser)
        self.v_inference = nn.Sequential(nn.Linear(H, H), ShiftedSoftplus(), nn.Linear(H, 12))
        self.distance_expansion = GaussianSmearing(0., 5., num_gaussians=config.denoiser.num_r_gaussian, fix_offset=False)
        self.bond_inference = nn.Sequential(nn.Linear(H, H), ShiftedSoftplus(), nn.Linear(H, 6))
        self.atom_mlp = nn.Sequential(nn.Linear(H, H * 2), nn.ReLU(), nn.Linear(H * 2, 1), nn.Sigmoid())
        self.atom_mlp_1 = nn.Sequential(nn.Linear(H, H * 2), nn.ReLU(), nn.Linear(H * 2, 1), nn.Sigmoid())
        self._pack = None
        self._pack_version = None
        self._plan = None
        self._engine = None

    # ------------------------------------------------------------------ engine plumbing
    @property
    def ex_col(self):
        return 12 if self.data_name in ('zinc_300', 'pdbbind') else 10          # diffusion.py:152-155

    def _device(self):
        return self.node_embedder.weight.device

    def packed(self):
        """Kernel-layout weights, rebuilt when parameters change (version counters) or move."""
        dev = self._device()
        if dev.type != 'cuda':
            raise RuntimeError('phoregen_amd: the model must be on an MI355X (`.to("cuda")`): the HIP path has no CPU fallback')
        ver = (str(dev),) + tuple(p._version for p in self.parameters())
        if self._pack is None or self._pack_version != ver:
            hip.lib()
            self._pack = ModelPack(self.state_dict(), self.denoiser.num_layers)
            self._pack_version = ver
            self._engine = None
        return self._pack

    def engine_for(self, plan):
        pack = self.packed()
        if self._engine is None or self._engine.plan is not plan:
            self._engine = Engine(pack, plan, knn_k=self.denoiser.k)
        return self._engine

    # ------------------------------------------------------------------ forward (diffusion.py:175-246)
    def forward(self, h_node_pert, pos_pert, batch_node, h_edge_pert, edge_index, batch_edge, time_step,
                h_phore, pos_phore, phore_norm, batch_phore):
        self.packed()
        if self._plan is None or not self._plan.matches(batch_node, batch_phore, edge_index):
            self._plan = BatchPlan(batch_node, batch_phore, edge_index, batch_edge, time_step.numel(), self._device())
        eng = self.engine_for(self._plan)
        eng.encode_phore(h_phore, pos_phore, phore_norm, self.ex_col)
        v, x0, bond = eng.forward(h_node_pert.float(), pos_pert.float(), h_edge_pert.float(), time_step)
        w = eng.ws
        return v.clone(), x0, bond.clone(), (w.count_l.clone().unsqueeze(-1), w.count_u.clone().unsqueeze(-1))

    def compute_loss(self, data):
        raise NotImplementedError('phoregen_amd: compute_loss (training, SURVEY.md 8 a19) needs the backward kernels, '
                                  'which are not built yet; there is deliberately no CPU fallback.')

    # ------------------------------------------------------------------ atom-count sampling (diffusion.py:355-387)
    @torch.no_grad()
    def sample_nodes(self, data, batch_size, device, sample_mode='uniform', normal_scale=4.0):
        ph = data['phore']
        p = ph.x.size(0)
        z = torch.zeros(0, dtype=torch.long)
        plan = BatchPlan(z, torch.zeros(p, dtype=torch.long), torch.zeros(2, 0, dtype=torch.long), z, 1, self._device())
        eng = Engine(self.packed(), plan, knn_k=self.denoiser.k)
        eng.encode_phore(ph.x.to(self._device()), ph.pos.to(self._device()), ph.norm.to(self._device()), self.ex_col)
        span = self.max_atom - self.min_atom
        lo = int((eng.ws.count_l * span + self.min_atom).round().int().item())
        hi = int((eng.ws.count_u * span + self.min_atom).round().int().item())
        return sample_from_interval(lo, hi, batch_size, mode=sample_mode, scale=normal_scale).to(device)

    # ------------------------------------------------------------------ sampler (diffusion.py:390-525)
    @torch.no_grad()
    def sample(self, data, n_graphs, device, pos_guidance_opt=None, sample_mode='uniform', normal_scale=4.0,
               rng='device', seed=None, num_atoms=None, return_traj=True, **kwargs):
        """Reference contract: returns {'pred': [logits_node, x0 + center, logits_edge],
        'traj': [node, pos, edge], 'lig_info': [num_atoms, batch, edge_index, edge_batch]}.

        rng='device': (default, like the reference on a GPU) counter-based Philox inside the transition kernels, no host
                      traffic in the loop; `seed` defaults to torch.initial_seed(), so `seed_all(seed)` still controls it.
        rng='cpu'   : noise is drawn from torch's default CPU generator in the reference's order, shape and dtype
                       and uploaded -> same seeds give the reference CPU path's draws.
        num_atoms   : optional LongTensor [n_graphs] overriding the atom-count draw (tests / benchmarks)."""
        ph = data['phore']
        if seed is None:
            seed = torch.initial_seed() & 0x7FFFFFFFFFFFFFFF
        if num_atoms is None:
            num_atoms = self.sample_nodes(data, n_graphs, device, sample_mode, normal_scale)
        p = ph.x.size(0)
        batch_phore = torch.repeat_interleave(torch.arange(n_graphs), p)
        center = data.center.to(self._device()).float()
        return self.sample_batch(ph.x.repeat(n_graphs, 1), ph.pos.repeat(n_graphs, 1), ph.norm.repeat(n_graphs, 1),
                                 batch_phore, num_atoms, center.unsqueeze(0).expand(n_graphs, 3),
                                 pos_guidance_opt=pos_guidance_opt, rng=rng, seed=seed, return_traj=return_traj,
                                 guidance_center=ph.pos[ph.x[:, self.ex_col] != 1].mean(0), **kwargs)

    @torch.no_grad()
    def sample_batch(self, h_phore, pos_phore, phore_norm, batch_phore, num_atoms, centers, pos_guidance_opt=None,
                     rng='device', seed=0, return_traj=True, guidance_center=None, num_steps=None, on_step=None):
        """Sampler over a batch of (possibly different) pharmacophores: the multi-pharmacophore entry point
        the reference lacks.  `centers` [B,3] are added back to coordinates as the reference does."""
        st = self.begin_sampling(h_phore, pos_phore, phore_norm, batch_phore, num_atoms, centers, rng=rng, seed=seed,
                                 return_traj=return_traj, num_steps=num_steps, guidance_center=guidance_center)
        T = self.num_timesteps
        for i, step in enumerate(range(T)[::-1][:st.n_steps]):
            self.reverse_step(st, i, step, pos_guidance_opt)
            if on_step is not None:
                on_step(i, step, st.eng.ws.out_v, st.x0, st.eng.ws.out_bond)
        return self.finish_sampling(st)

    # ---- sampler pieces (also used teacher-forced by the parity tests) ----
    @torch.no_grad()
    def begin_sampling(self, h_phore, pos_phore, phore_norm, batch_phore, num_atoms, centers, rng='device', seed=0,
                       return_traj=True, num_steps=None, guidance_center=None):
        dev = self._device()
        hip.lib()
        B = int(num_atoms.numel())
        num_atoms = num_atoms.detach().cpu().long()
        batch_node = torch.repeat_interleave(torch.arange(B), num_atoms)
        edge_index, batch_edge = make_edge_data(num_atoms)
        plan = BatchPlan(batch_node, batch_phore, edge_index, batch_edge, B, dev)
        eng = self.engine_for(plan)
        w = eng.ws
        eng.encode_phore(h_phore.to(dev), pos_phore.to(dev), phore_norm.to(dev), self.ex_col)
        N, E = plan.n_lig, plan.n_bond
        st = type('SamplerState', (), {})()
        st.eng, st.plan, st.num_atoms, st.N, st.E, st.B = eng, plan, num_atoms, N, E, B
        st.cpu, st.seed, st.return_traj = rng == 'cpu', seed, return_traj
        st.n_steps = self.num_timesteps if num_steps is None else num_steps
        centers = centers.to(dev).float()
        st.center_rows = centers[plan.batch_node]                                        # [N,3]
        if return_traj and not bool((centers == centers[0:1]).all()):
            raise NotImplementedError('phoregen_amd: return_traj needs one shared centre (the reference adds data.center)')
        st.c0 = centers[0].contiguous()

        # ---- init state (diffusion.py:406-408, transition.py:65-69,331-339) ----
        if st.cpu:
            pos = torch.randn([N, 3]).to(dev) - st.center_rows
            u_n = torch.rand(N, 12, dtype=torch.float64)
            u_e = torch.rand(E, 6, dtype=torch.float64)
        else:
            gen = torch.Generator(device=dev).manual_seed(seed)
            pos = torch.randn([N, 3], device=dev, generator=gen) - st.center_rows
            u_n = torch.rand(N, 12, dtype=torch.float64, device=dev, generator=gen)
            u_e = torch.rand(E, 6, dtype=torch.float64, device=dev, generator=gen)

        def init_types(tr, u):
            lp = torch.log(torch.from_numpy(tr.init_prob) + tr.eps).clamp_min(-32.).to(u.device)
            gum = -torch.log(-torch.log(u + 1e-30) + 1e-30)
            return (gum + lp.unsqueeze(0)).argmax(-1).to(dev)

        node_t, edge_t = init_types(self.node_transition, u_n), init_types(self.edge_transition, u_e)
        h_node, h_edge = F.one_hot(node_t, 12).float(), F.one_hot(edge_t, 6).float()
        st.log_node = [torch.log(h_node.clamp(min=1e-30)), torch.empty(N, 12, device=dev)]   # common.py:398-402
        st.log_edge = [torch.log(h_edge.clamp(min=1e-30)), torch.empty(E, 6, device=dev)]
        st.cur = 0
        w.in_h_node.copy_(h_node), w.in_pos.copy_(pos), w.in_h_edge.copy_(h_edge)
        st.node_traj = st.pos_traj = st.edge_traj = None
        if return_traj:
            st.node_traj = torch.zeros(st.n_steps + 1, N, 12, device=dev)
            st.pos_traj = torch.zeros(st.n_steps + 1, N, 3, device=dev)
            st.edge_traj = torch.zeros(st.n_steps + 1, E, 6, device=dev)
            st.node_traj[0], st.pos_traj[0], st.edge_traj[0] = h_node, pos, h_edge        # :424-426 (no +center)
        st.grad = torch.zeros(N, 3, device=dev)
        st.cnt_ws, st.mean_ws, st.gtmp = torch.zeros(B, device=dev), torch.zeros(B, 3, device=dev), torch.zeros(N, 3, device=dev)
        st.gc = guidance_center.to(dev).float().contiguous() if guidance_center is not None else None
        st.x0 = None
        return st

    @torch.no_grad()
    def reverse_step(self, st, i, step, pos_guidance_opt=None, draws=None):
        """One iteration of the loop at diffusion.py:432-517 on the state held in the engine workspace.
        `draws` = (u_node [N,12], u_edge [E,6], eps [N,3]) overrides the noise source (teacher-forced tests)."""
        lib, eng, plan = hip.lib(), st.eng, st.plan
        pk, w, N, E = eng.pack, eng.ws, st.N, st.E
        dev = self._device()
        tp = lambda tr: tr[i + 1].data_ptr() if tr is not None else None
        w.in_t.fill_(step)
        _, st.x0, _ = eng.forward_inplace()
        s = hip.stream_ptr()
        un = ue = eps = None
        if draws is not None:
            un, ue, eps = (d.to(dev).contiguous() for d in draws)
        elif st.cpu:                                                 # Appendix B item 5: rand, rand, then randn
            un, ue = torch.rand(N, 12).to(dev), torch.rand(E, 6).to(dev)
        cur = st.cur
        hip.check(lib.pg_posterior_categorical(
            w.out_v.data_ptr(), st.log_node[cur].data_ptr(), plan.lig_graph.data_ptr(), w.in_t.data_ptr(),
            pk.node_tab[0].data_ptr(), pk.node_tab[1].data_ptr(), N, 12, hip.ptr(un), st.seed, 0, step,
            st.log_node[1 - cur].data_ptr(), w.in_h_node.data_ptr(), tp(st.node_traj), s), 'posterior(node)')
        hip.check(lib.pg_posterior_categorical(
            w.out_bond.data_ptr(), st.log_edge[cur].data_ptr(), plan.bond_graph.data_ptr(), w.in_t.data_ptr(),
            pk.edge_tab[0].data_ptr(), pk.edge_tab[1].data_ptr(), E, 6, hip.ptr(ue), st.seed, 1, step,
            st.log_edge[1 - cur].data_ptr(), w.in_h_edge.data_ptr(), tp(st.edge_traj), s), 'posterior(edge)')
        grad = None
        if pos_guidance_opt:                                         # diffusion.py:476-502
            grad = st.grad
            grad.zero_()
            for o in pos_guidance_opt:
                atom = o['type'] == 'atom_prox'
                if not atom and o['type'] != 'center_prox':
                    continue
                hip.check(lib.pg_guidance_grad(
                    plan.topo_ref, w.in_pos.data_ptr(), w.in_h_edge.data_ptr(), plan.lig_graph.data_ptr(),
                    plan.g_lig_off.data_ptr(), int(atom), float(o.get('min_d', 1.2)), float(o.get('max_d', 2.8)),
                    int(not atom), hip.ptr(st.gc), st.cnt_ws.data_ptr(), st.mean_ws.data_ptr(), st.gtmp.data_ptr(), s),
                    'guidance')
                grad += st.gtmp
        if draws is None and st.cpu:
            eps = torch.randn(N, 3).to(dev)
        hip.check(lib.pg_posterior_position(
            w.in_pos.data_ptr(), st.x0.data_ptr(), plan.lig_graph.data_ptr(), w.in_t.data_ptr(),
            pk.pos_tab[0].data_ptr(), pk.pos_tab[1].data_ptr(), pk.pos_tab[2].data_ptr(), hip.ptr(grad), hip.ptr(eps),
            st.seed, 2, step, N, st.c0.data_ptr() if st.return_traj else None,
            w.in_pos.data_ptr(), tp(st.pos_traj), s), 'posterior(pos)')          # in place: x_t -> x_{t-1}
        st.cur = 1 - cur

    def finish_sampling(self, st):
        w, plan = st.eng.ws, st.plan
        return {'pred': [w.out_v.clone(), st.x0 + st.center_rows, w.out_bond.clone()],
                'traj': [st.node_traj, st.pos_traj, st.edge_traj],
                'lig_info': [st.num_atoms.to(self._device()), plan.batch_node, plan.edge_index, plan.batch_edge]}
