"""PhoreDiff with the reference's nn.Module surface (models/diffusion.py:19-525), HIP-backed.

Same constructor, same attribute / state_dict names, same `forward` and `sample` contracts, so the
reference's sample_all.py / train.py can drive it.  `compute_loss` (training, config 5) runs the same kernels with
their hand-written HIP adjoints (phoregen_amd/training.py).

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


def _on_model_device(fn):
    """Run `fn` with the model's GPU as the current device: the C-ABI launches go to torch's current stream of that device,
    and HIP launches use the calling thread's current device (a model on cuda:1 must not be driven from device 0)."""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *a, **k):
        dev = self._device()
        if dev.type != 'cuda':
            return fn(self, *a, **k)
        with torch.cuda.device(dev):
            return fn(self, *a, **k)
    return wrapped


class _LazyFloats(dict):
    """dict of 0-dim device tensors that reads as a dict of python floats (each value is converted on first access)."""

    def _conv(self, k):
        v = dict.__getitem__(self, k)
        if torch.is_tensor(v):
            v = v.item()
            dict.__setitem__(self, k, v)
        return v

    def __getitem__(self, k):
        return self._conv(k)

    def get(self, k, default=None):
        return self._conv(k) if k in self else default

    def items(self):
        return [(k, self._conv(k)) for k in self.keys()]

    def values(self):
        return [self._conv(k) for k in self.keys()]

    def __repr__(self):
        return repr(dict(self.items()))


class PhoreDiff(nn.Module):
    def __init__(self, config, data_name, **kwargs):
        super().__init__()
        self.config, self.data_name = config, data_name
        self.num_node_types = config.num_atom_classes
        self.num_edge_types = config.num_bond_classes
        self.bond_len_loss = config.bond_len_loss          # the loss_len term of compute_loss (diffusion.py:286-290,333,341)
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
        self._count_engine = None

    # ------------------------------------------------------------------ engine plumbing
    def __deepcopy__(self, memo):
        """copy.deepcopy(model) (EMA helpers, tests): the kernel-side caches hold raw pointers and are rebuilt lazily."""
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            object.__setattr__(new, k, None if k in ('_pack', '_pack_version', '_plan', '_engine', '_count_engine') else copy.deepcopy(v, memo))
        return new

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

    def invalidate_pack(self):
        """Drop the kernel-layout weight cache.  `packed()` notices optimizer steps and `.to()` through the parameters'
        version counters; writes through `param.data` (EMA swaps, weight surgery) do not bump them -- call this after such
        a write.  `load_state_dict` and `_apply` (`.to`, `.cuda`, `.float`) call it themselves."""
        self._pack = self._pack_version = self._engine = self._count_engine = None

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self.invalidate_pack()
        return out

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self.invalidate_pack()
        return out

    def engine_for(self, plan):
        pack = self.packed()
        if self._engine is None or self._engine.plan is not plan:
            self._engine = Engine(pack, plan, knn_k=self.denoiser.k)
        return self._engine

    # ------------------------------------------------------------------ forward (diffusion.py:175-246)
    @_on_model_device
    def forward(self, h_node_pert, pos_pert, batch_node, h_edge_pert, edge_index, batch_edge, time_step,
                h_phore, pos_phore, phore_norm, batch_phore):
        self.packed()
        if self._plan is None or not self._plan.matches(batch_node, batch_phore, edge_index):
            self._plan = BatchPlan(batch_node, batch_phore, edge_index, batch_edge, time_step.numel(), self._device())
        floats = (h_node_pert, pos_pert, h_edge_pert, h_phore, pos_phore, phore_norm)
        if torch.is_grad_enabled() and (any(p.requires_grad for p in self.parameters()) or
                                        any(torch.is_tensor(a) and a.requires_grad for a in floats)):
            # the reference's forward is differentiable (diffusion.py:175-246, used under autograd by :267): with gradients being
            # recorded the same kernels run with their hand-written HIP adjoints on the tape (phoregen_amd/training.py), so a
            # caller's own loss on these outputs reaches the parameters (and pos_pert / h_node_pert).  Under torch.no_grad()
            # (sampling) the pre-built launch list below runs instead.
            from ..training import TrainForward
            dev = self._device()
            params = {**dict(self.named_buffers()), **dict(self.named_parameters())}
            tf = TrainForward(params, self._plan, knn_k=self.denoiser.k, num_layers=self.denoiser.num_layers, ex_col=self.ex_col)
            return tf.forward(h_node_pert.to(dev), pos_pert.to(dev), h_edge_pert.to(dev), time_step.to(dev), h_phore.to(dev).float(),
                              pos_phore.to(dev).float(), phore_norm.to(dev).float(), batch_phore.to(dev))
        eng = self.engine_for(self._plan)
        eng.encode_phore(h_phore, pos_phore, phore_norm, self.ex_col)
        v, x0, bond = eng.forward(h_node_pert.float(), pos_pert.float(), h_edge_pert.float(), time_step)
        w = eng.ws
        return v.clone(), x0, bond.clone(), (w.count_l.clone().unsqueeze(-1), w.count_u.clone().unsqueeze(-1))

    # ------------------------------------------------------------------ training objective (diffusion.py:249-352)
    def sample_time(self, num_graphs, device, **kwargs):
        """diffusion.py:138-145: antithetic time steps."""
        ts = torch.randint(0, self.num_timesteps, size=(num_graphs // 2 + 1,), device=device)
        ts = torch.cat([ts, self.num_timesteps - ts - 1], dim=0)[:num_graphs]
        return ts, torch.ones_like(ts).float() / self.num_timesteps

    @_on_model_device
    def compute_loss(self, data, draws=None):
        """Reference contract: (loss with grad, dict of floats).  The denoiser forward and its adjoint run in the HIP
        kernels (phoregen_amd/training.py); the noising, the posteriors and the loss terms are elementwise tensor ops
        on the device.  `draws` (tests): dict(time_draw, pos_noise, u_node, u_edge) replaces the four random draws
        of sample_time / add_noise (transition.py:28-41,245-263) in the reference's order."""
        from ..training import TrainForward
        dev = self._device()
        if dev.type != 'cuda':
            raise RuntimeError('phoregen_amd: compute_loss runs on the MI355X HIP path only (no CPU fallback)')
        hip.lib()
        lig, e, ph = data['ligand'], data['ligand', 'ligand'], data['phore']
        B = int(data.num_graphs)
        bn, be = lig.batch.to(dev), e.f_edge_attr_batch.to(dev)
        pos0, x_cls, e_cls = lig.pos.to(dev).float(), lig.x.to(dev), e.f_edge_attr.to(dev)
        T = self.num_timesteps
        if draws is None:
            t, _ = self.sample_time(B, dev)
            eps = torch.zeros_like(pos0).normal_()
            u_n = torch.rand(x_cls.numel(), 12, device=dev)
            u_e = torch.rand(e_cls.numel(), 6, device=dev)
        else:
            ts = draws['time_draw'].to(dev)
            t = torch.cat([ts, T - ts - 1], dim=0)[:B]
            eps, u_n, u_e = draws['pos_noise'].to(dev), draws['u_node'].to(dev), draws['u_edge'].to(dev)
        ab = self.pos_transition.alphas_bar[t][bn].unsqueeze(-1)
        pos_pert = ab.sqrt() * pos0 + (1 - ab).sqrt() * eps                                  # transition.py:28-41

        def noise_cat(tr, v, K, batch, u):
            log_v0 = torch.log(F.one_hot(v, K).float().clamp(min=1e-30))                     # common.py:398-402
            q = (log_v0.exp().unsqueeze(-1) * tr.q_mats[t[batch]]).sum(1)                     # transition.py:265-271 (row-vector x [K,K] per row, elementwise: a batched 6x6 GEMM call is 100x slower)
            cls = (-torch.log(-torch.log(u + 1e-30) + 1e-30) + torch.log(q + tr.eps).clamp_min(-32.)).argmax(-1)
            oh = F.one_hot(cls, K).float()
            return oh, torch.log(oh.clamp(min=1e-30)), log_v0
        h_node, log_node_t, log_node_0 = noise_cat(self.node_transition, x_cls, 12, bn, u_n)
        h_edge, log_edge_t, log_edge_0 = noise_cat(self.edge_transition, e_cls, 6, be, u_e)

        if self._plan is None or not self._plan.matches(lig.batch, ph.batch, e.f_edge_index):
            self._plan = BatchPlan(lig.batch, ph.batch, e.f_edge_index, e.f_edge_attr_batch, B, dev)
        params = {**dict(self.named_buffers()), **dict(self.named_parameters())}
        tf = TrainForward(params, self._plan, knn_k=self.denoiser.k, num_layers=self.denoiser.num_layers, ex_col=self.ex_col)
        pred_node, pred_pos, pred_edge, (c_l, c_u) = tf.forward(h_node, pos_pert, h_edge, t, ph.x.to(dev).float(),
                                                                ph.pos.to(dev).float(), ph.norm.to(dev).float(),
                                                                ph.batch.to(dev))
        loss_pos = F.mse_loss(pred_pos, pos0) * self.loss_weight[0]

        def posterior(tr, log_v0, log_vt, batch):                                              # transition.py:285-315
            tb = t[batch]
            f1 = (log_vt.exp().unsqueeze(-1) * tr.transpopse_q_onestep_mats[tb]).sum(1)
            f2 = (log_v0.exp().unsqueeze(-1) * tr.q_mats[torch.clamp(tb - 1, min=0)]).sum(1)
            out = torch.log(f1 + tr.eps).clamp_min(-32.) + torch.log(f2 + tr.eps).clamp_min(-32.)
            out = out - torch.logsumexp(out, -1, keepdim=True)
            return torch.where((tb == 0).unsqueeze(-1), log_v0, out)

        def cat_loss(tr, pred, log_t, log_0, batch):                                           # transition.py:317-329
            post_true = posterior(tr, log_0, log_t, batch)
            post_pred = posterior(tr, F.log_softmax(pred, -1), log_t, batch)
            kl = (post_true.exp() * (post_true - post_pred)).sum(-1)
            nll = -(log_0.exp() * post_pred).sum(-1)
            m = (t == 0).float()[batch]
            return torch.mean(m * nll + (1 - m) * kl)
        loss_node = cat_loss(self.node_transition, pred_node, log_node_t, log_node_0, bn) * self.loss_weight[1]
        loss_edge = cat_loss(self.edge_transition, pred_edge, log_edge_t, log_edge_0, be) * self.loss_weight[2]
        ptr = lig.ptr.to(dev)
        true = ((ptr[1:] - ptr[:-1]).float() - self.min_atom) / (self.max_atom - self.min_atom)
        loss_count = self.compute_count_loss(true.unsqueeze(-1), (c_l, c_u))
        loss = loss_pos + loss_node + loss_edge + loss_count
        loss_len = None
        if self.bond_len_loss:                                                                 # diffusion.py:286-290,333
            bonds = getattr(e, 'edge_index', None) if not isinstance(e, dict) else e.get('edge_index')
            if bonds is None:
                raise ValueError("phoregen_amd: config.bond_len_loss is set, but the batch has no data['ligand','ligand'].edge_index "
                                 '(the bonds of the molecule; TrainBatch(..., edge_index=...))')
            src, dst = bonds.to(dev)
            true_len = torch.norm(pos0.index_select(0, src) - pos0.index_select(0, dst), dim=-1)
            pred_len = torch.norm(pred_pos.index_select(0, src) - pred_pos.index_select(0, dst), dim=-1)
            loss_len = F.mse_loss(pred_len, true_len)
            loss = loss + loss_len

        def acc(true_cls, logits, batch):                                                      # common.py:284-297
            bad = torch.zeros(B, device=dev).index_add(0, batch, (logits.argmax(-1) != true_cls).float())
            present = torch.zeros(B, device=dev).index_add(0, batch, torch.ones_like(batch, dtype=torch.float32)) > 0
            return ((bad == 0) & present).sum().float() / present.sum().clamp(min=1).float()
        # the reference returns python floats (`.item()` right here = a host sync between forward and backward);
        # same values, converted when first read, so `loss.backward()` can be enqueued while the forward still runs
        info = _LazyFloats({'loss': loss.detach(), 'loss_pos': loss_pos.detach(), 'loss_node': loss_node.detach(),
                            'loss_count': loss_count.detach(), 'loss_edge': loss_edge.detach(),
                            'node_acc': acc(x_cls, pred_node.detach(), bn), 'edge_acc': acc(e_cls, pred_edge.detach(), be),
                            **({'loss_len': loss_len.detach()} if loss_len is not None else {})})   # diffusion.py:341
        return loss, info

    def compute_count_loss(self, true_norm, pred_count, a=0.05, s=160, nd=15, epsilon=1e-12):
        """common.py:261-281 (qd_loss, mode='soft') on the normalised atom counts (diffusion.py:166-172)."""
        y_l, y_u = pred_count
        n = true_norm.shape[0]
        k_h = torch.relu(torch.sign(y_u - true_norm)) * torch.relu(torch.sign(true_norm - y_l))
        k_s = torch.sigmoid((y_u - true_norm) * s) * torch.sigmoid((true_norm - y_l) * s)
        mpiw = torch.sum((y_u - y_l) * k_h) / (torch.sum(k_h) + epsilon) * self.count_factor
        return mpiw + (torch.relu((1 - a) - torch.mean(k_s)) ** 2) * (n ** 0.5) * nd

    # ------------------------------------------------------------------ atom-count sampling (diffusion.py:355-387)
    @torch.no_grad()
    @_on_model_device
    def sample_nodes(self, data, batch_size, device, sample_mode='uniform', normal_scale=4.0):
        ph = data['phore']
        p = ph.x.size(0)
        pack = self.packed()
        eng = self._count_engine
        if eng is None or eng.pack is not pack or eng.plan.n_phore != p:
            # pharmacophore-only plan (no ligand yet); kept for the next call: sample_all.py draws counts once per batch
            z = torch.zeros(0, dtype=torch.long)
            plan = BatchPlan(z, torch.zeros(p, dtype=torch.long), torch.zeros(2, 0, dtype=torch.long), z, 1, self._device())
            eng = self._count_engine = Engine(pack, plan, knn_k=self.denoiser.k, phore_only=True)
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
                      traffic in the loop; `seed` = the Philox key, by default a fresh draw from torch's default generator
                      per call (so consecutive calls differ and `seed_all(seed)` still controls the whole run).
        rng='cpu'   : noise is drawn from torch's default CPU generator in the reference's order, shape and dtype
                       and uploaded -> same seeds give the reference CPU path's draws.
        num_atoms   : optional LongTensor [n_graphs] overriding the atom-count draw (tests / benchmarks)."""
        ph = data['phore']
        if seed is None:
            # a fresh key per call, drawn from torch's default generator: repeated sample() calls (sample_all.py's while loop)
            # get different noise like the reference's global-RNG draws do, and `seed_all(seed)` still makes a run reproducible.
            # rng='cpu' replays the reference's own CPU draws: nothing extra may be taken from that generator there.
            seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if rng == 'device' else 0
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
                     rng='device', seed=0, return_traj=True, guidance_center=None, num_steps=None, on_step=None,
                     graph_ids=None, guidance_batch=None, pipeline=True):
        """Sampler over a batch of (possibly different) pharmacophores: the multi-pharmacophore entry point
        the reference lacks.  `centers` [B,3] are added back to coordinates as the reference does.
        pipeline (device RNG, no `on_step`): the loop runs as a software pipeline over the reverse steps (`_reverse_step_pipelined`:
            same kernels on the same operands, the same trajectory bit for bit; tested).
        graph_ids [B] (device RNG): the noise of graph g is keyed by (seed, graph_ids[g]) and by positions INSIDE the graph,
            so a graph draws the same noise in any batch / shard (default: 0..B-1).
        guidance_center [3] or [B,3]: target of the `center_prox` energy (default: per graph, the mean of its non-EX
            pharmacophore nodes, diffusion.py:489-491).
        guidance_batch: the number of graphs the guidance energies average over (default B, the reference's behaviour; a
            shard of a larger logical batch passes the full batch size so that it reproduces the unsharded run)."""
        st = self.begin_sampling(h_phore, pos_phore, phore_norm, batch_phore, num_atoms, centers, rng=rng, seed=seed,
                                 return_traj=return_traj, num_steps=num_steps, guidance_center=guidance_center,
                                 graph_ids=graph_ids, guidance_batch=guidance_batch, pipeline=pipeline and on_step is None)
        T = self.num_timesteps
        for i, step in enumerate(range(T)[::-1][:st.n_steps]):
            self.reverse_step(st, i, step, pos_guidance_opt)
            if on_step is not None:
                on_step(i, step, st.eng.ws.out_v, st.x0, st.eng.ws.out_bond)
        return self.finish_sampling(st)

    # ---- sampler pieces (also used teacher-forced by the parity tests) ----
    @torch.no_grad()
    @_on_model_device
    def begin_sampling(self, h_phore, pos_phore, phore_norm, batch_phore, num_atoms, centers, rng='device', seed=0,
                       return_traj=True, num_steps=None, guidance_center=None, graph_ids=None, guidance_batch=None, pipeline=False):
        """pipeline=True: `reverse_step` must then be called for consecutive steps with nothing written to the carried state in
        between (the next step's features are embedded at the end of the step before it); the teacher-forced tests leave it off."""
        dev = self._device()
        lib = hip.lib()
        B = int(num_atoms.numel())
        num_atoms = num_atoms.detach().cpu().long()
        batch_node = torch.repeat_interleave(torch.arange(B), num_atoms)
        edge_index, batch_edge = make_edge_data(num_atoms)
        plan = BatchPlan(batch_node, batch_phore, edge_index, batch_edge, B, dev)
        eng = self.engine_for(plan)
        w, pk = eng.ws, eng.pack
        h_phore, pos_phore = h_phore.to(dev), pos_phore.to(dev)
        eng.encode_phore(h_phore, pos_phore, phore_norm.to(dev), self.ex_col)
        N, E = plan.n_lig, plan.n_bond
        st = type('SamplerState', (), {})()
        st.eng, st.plan, st.num_atoms, st.N, st.E, st.B = eng, plan, num_atoms, N, E, B
        st.cpu, st.seed, st.return_traj = rng == 'cpu', seed, return_traj
        st.n_steps = self.num_timesteps if num_steps is None else num_steps
        st.pipelined = bool(pipeline) and rng == 'device' and eng.pipelined_programs() is not None
        st.next_step = None              # (pipelined: the step whose features `prog_ahead` has embedded)
        # (pipelined: the posteriors of step s run on the side lanes, possibly while lane 0 has begun step s - 1: the step number they
        #  read is a row of a table written once)
        st.t_table = torch.arange(self.num_timesteps, dtype=torch.int64, device=dev).unsqueeze(1).expand(-1, max(B, 1)).contiguous() \
            if st.pipelined else None
        st.x0_buf = torch.empty(N, 3, device=dev) if st.pipelined else None
        st.centers = centers.to(dev).float().contiguous()                                # [B,3]
        st.center_rows = st.centers[plan.batch_node]                                     # [N,3]
        st.graph_key = (torch.arange(B) if graph_ids is None else graph_ids.detach().cpu()).to(torch.int32).to(dev)
        st.guidance_batch = int(guidance_batch) if guidance_batch else B

        # ---- init state (diffusion.py:406-408, transition.py:65-69,331-339) ----
        lp_n = torch.log(torch.from_numpy(self.node_transition.init_prob) + self.node_transition.eps).clamp_min(-32.)
        lp_e = torch.log(torch.from_numpy(self.edge_transition.init_prob) + self.edge_transition.eps).clamp_min(-32.)
        if st.cpu:
            pos = torch.randn([N, 3]).to(dev) - st.center_rows
            u_n = torch.rand(N, 12, dtype=torch.float64)
            u_e = torch.rand(E, 6, dtype=torch.float64)

            def init_types(lp, u):
                gum = -torch.log(-torch.log(u + 1e-30) + 1e-30)
                return (gum + lp.unsqueeze(0)).argmax(-1).to(dev)
            h_node = F.one_hot(init_types(lp_n, u_n), 12).float()
            h_edge = F.one_hot(init_types(lp_e, u_e), 6).float()
        else:
            # the same draws by the transition kernels themselves: Gumbel-argmax of the (normalised) prior = the kernel's t = 0
            # branch on logits = log prior; N(0,1) = the Gaussian posterior with mean 0 and sigma 1.  Counter step = T (never a
            # real step), so the initial noise is one more graph-keyed Philox stream
            s = hip.stream_ptr()
            T = self.num_timesteps
            t0, t1 = torch.zeros(B, dtype=torch.int64, device=dev), torch.ones(B, dtype=torch.int64, device=dev)
            h_node, h_edge = torch.empty(N, 12, device=dev), torch.empty(E, 6, device=dev)
            for lp, tab, rows, K, sid, rg, row0, oh in ((lp_n, pk.node_tab, N, 12, 0, plan.lig_graph, plan.g_lig_off, h_node),
                                                        (lp_e, pk.edge_tab, E, 6, 1, plan.bond_graph, plan.g_bond_off, h_edge)):
                logits = lp.float().to(dev).unsqueeze(0).expand(rows, K).contiguous()
                scratch = torch.empty(rows, K, device=dev)
                hip.check(lib.pg_posterior_categorical(
                    logits.data_ptr(), logits.data_ptr(), rg.data_ptr(), t0.data_ptr(), tab[0].data_ptr(), tab[1].data_ptr(),
                    rows, K, None, seed, sid, T, row0.data_ptr(), st.graph_key.data_ptr(), scratch.data_ptr(), oh.data_ptr(),
                    None, s), 'init types')
            ones, zx = torch.ones(2, device=dev), torch.zeros(N, 3, device=dev)
            pos = torch.empty(N, 3, device=dev)
            hip.check(lib.pg_posterior_position(
                zx.data_ptr(), zx.data_ptr(), plan.lig_graph.data_ptr(), t1.data_ptr(), ones.data_ptr(), ones.data_ptr(),
                ones.data_ptr(), None, None, seed, 2, T, N, plan.g_lig_off.data_ptr(), st.graph_key.data_ptr(), None,
                pos.data_ptr(), None, s), 'init positions')
            pos -= st.center_rows

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
        # per-graph target of the center_prox energy: given, or the mean of the graph's non-EX pharmacophore nodes
        if guidance_center is not None:
            gc = guidance_center.to(dev).float()
            st.gc = (gc.unsqueeze(0).expand(B, 3) if gc.dim() == 1 else gc).contiguous()
        else:
            # summed on the host, in node order: a device index_add_ accumulates with atomics in whatever order the hardware serves them,
            # and the centre -- hence every guided coordinate -- would differ by an ulp from run to run (round 5: found by the
            # pipelined-loop test; once per batch, a few hundred rows)
            keep = (h_phore[:, self.ex_col] != 1).float().unsqueeze(-1).cpu()
            bp = plan.phore_graph.long().cpu()
            sums = torch.zeros(B, 3).index_add_(0, bp, pos_phore.float().cpu() * keep)
            cnt = torch.zeros(B, 1).index_add_(0, bp, keep)
            st.gc = (sums / cnt).to(dev).contiguous()      # (0/0 = nan for a pharmacophore of exclusion spheres only, as in the reference)
        st.x0 = None
        if getattr(eng, '_tune', None) is not None and not st.cpu:
            self._calibrate_launch_configuration(st)
        return st

    def _calibrate_launch_configuration(self, st):
        """Small batches: `Engine.calibrate_tri_grid` times a few REAL sampler steps per candidate grid -- here, before the caller's loop, on the
        state just initialised, which is put back afterwards (the device noise is counter-based: nothing is consumed; no trajectory frame is
        written).  The loop then runs a fixed launch list with no host synchronisation in it (round-5 review, item 8)."""
        eng, w = st.eng, st.eng.ws
        saved = [t.clone() for t in (w.in_h_node, w.in_pos, w.in_h_edge, st.log_node[0], st.log_edge[0])]
        keep = (st.node_traj, st.pos_traj, st.edge_traj, st.return_traj, st.n_steps)
        st.node_traj = st.pos_traj = st.edge_traj = None
        st.return_traj, st.n_steps = False, self.num_timesteps
        T = self.num_timesteps
        try:
            eng.calibrate_tri_grid(lambda k: self.reverse_step(st, k, T - 1 - k))
        finally:
            if st.pipelined:
                eng.join_lanes((2, 3))
            st.node_traj, st.pos_traj, st.edge_traj, st.return_traj, st.n_steps = keep
            for dst, src in zip((w.in_h_node, w.in_pos, w.in_h_edge, st.log_node[0], st.log_edge[0]), saved):
                dst.copy_(src)
            st.cur, st.next_step, st.x0 = 0, None, None

    @torch.no_grad()
    @_on_model_device
    def reverse_step(self, st, i, step, pos_guidance_opt=None, draws=None):
        """One iteration of the loop at diffusion.py:432-517 on the state held in the engine workspace.
        `draws` = (u_node [N,12], u_edge [E,6], eps [N,3]) overrides the noise source (teacher-forced tests)."""
        if getattr(st, 'pipelined', False):
            if draws is not None:
                raise RuntimeError('phoregen_amd: reverse_step(draws=...) on a pipelined sampler state (begin_sampling(pipeline=True))')
            return self._reverse_step_pipelined(st, i, step, pos_guidance_opt)
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
            plan.g_lig_off.data_ptr(), st.graph_key.data_ptr(),
            st.log_node[1 - cur].data_ptr(), w.in_h_node.data_ptr(), tp(st.node_traj), s), 'posterior(node)')
        hip.check(lib.pg_posterior_categorical(
            w.out_bond.data_ptr(), st.log_edge[cur].data_ptr(), plan.bond_graph.data_ptr(), w.in_t.data_ptr(),
            pk.edge_tab[0].data_ptr(), pk.edge_tab[1].data_ptr(), E, 6, hip.ptr(ue), st.seed, 1, step,
            plan.g_bond_off.data_ptr(), st.graph_key.data_ptr(),
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
                    int(not atom), hip.ptr(st.gc), st.guidance_batch, st.cnt_ws.data_ptr(), st.mean_ws.data_ptr(),
                    st.gtmp.data_ptr(), s),
                    'guidance')
                grad += st.gtmp
        if draws is None and st.cpu:
            eps = torch.randn(N, 3).to(dev)
        hip.check(lib.pg_posterior_position(
            w.in_pos.data_ptr(), st.x0.data_ptr(), plan.lig_graph.data_ptr(), w.in_t.data_ptr(),
            pk.pos_tab[0].data_ptr(), pk.pos_tab[1].data_ptr(), pk.pos_tab[2].data_ptr(), hip.ptr(grad), hip.ptr(eps),
            st.seed, 2, step, N, plan.g_lig_off.data_ptr(), st.graph_key.data_ptr(),
            st.centers.data_ptr() if st.return_traj else None,
            w.in_pos.data_ptr(), tp(st.pos_traj), s), 'posterior(pos)')          # in place: x_t -> x_{t-1}
        st.cur = 1 - cur

    def _reverse_step_pipelined(self, st, i, step, pos_guidance_opt=None):
        """`reverse_step` as one stage of a software pipeline over the reverse steps (device RNG).  The order of a step's results is
        types first (the heads finish inside the last layer), coordinates last; the next step can use its types long before its
        coordinates exist.  So: the denoiser program ends WITHOUT joining its side lanes; the node posterior follows the node head on
        lane 2, the bond posterior the bond head on lane 3; behind them `Engine.prog_ahead` embeds the next step's features and runs layer
        0's coordinate-free products (first-layer blocks, queries, bond-node sub-layer) -- all beside the last layer's position phase,
        the Gaussian posterior and the next step's coordinate embedding / knn search on lanes 0 / 1.  Same kernels on the same operands
        as `reverse_step`: the trajectory is the same bit for bit (tests/test_gpu_parity.py)."""
        lib, eng, plan = hip.lib(), st.eng, st.plan
        pk, w, N, E = eng.pack, eng.ws, st.N, st.E
        tp = lambda tr: tr[i + 1].data_ptr() if tr is not None else None
        s2, s3 = eng.lane_stream(2), eng.lane_stream(3)
        if st.next_step is None:                       # first step: nothing was launched ahead yet
            # the initial coordinates into the denoiser's buffer (from then on the Gaussian posterior writes them there itself)
            hip.check(lib.pg_embed_ctx(plan.topo_ref, w.in_h_node.data_ptr(), w.in_pos.data_ptr(), w.in_t.data_ptr(),
                                       pk.W_node_emb.data_ptr(), pk.t_off.data_ptr(), pk.t_coeff.data_ptr(), w.hp_emb.data_ptr(),
                                       w.pos_phore.data_ptr(), plan.phore2ctx.data_ptr(), None, w.x[0].data_ptr(), hip.stream_ptr()),
                      'pg_embed_ctx')
            eng.fork_lanes((2, 3))                     # (the initial state was written on the caller's stream)
            with torch.cuda.stream(s2):
                w.in_t_next.fill_(step)
            eng._run(eng.prog_ahead)
        elif st.next_step != step:
            raise RuntimeError(f'phoregen_amd: pipelined sampler state expects step {st.next_step}, got {step}')
        tb = st.t_table.data_ptr() + step * st.t_table.stride(0) * 8        # the [B] row `step` of the table: no fill launch
        _, x0_ctx, _ = eng.step_forward()
        cur = st.cur
        hip.check(lib.pg_posterior_categorical(
            w.out_v.data_ptr(), st.log_node[cur].data_ptr(), plan.lig_graph.data_ptr(), tb,
            pk.node_tab[0].data_ptr(), pk.node_tab[1].data_ptr(), N, 12, None, st.seed, 0, step,
            plan.g_lig_off.data_ptr(), st.graph_key.data_ptr(),
            st.log_node[1 - cur].data_ptr(), w.in_h_node.data_ptr(), tp(st.node_traj), s2.cuda_stream), 'posterior(node)')
        hip.check(lib.pg_posterior_categorical(
            w.out_bond.data_ptr(), st.log_edge[cur].data_ptr(), plan.bond_graph.data_ptr(), tb,
            pk.edge_tab[0].data_ptr(), pk.edge_tab[1].data_ptr(), E, 6, None, st.seed, 1, step,
            plan.g_bond_off.data_ptr(), st.graph_key.data_ptr(),
            st.log_edge[1 - cur].data_ptr(), w.in_h_edge.data_ptr(), tp(st.edge_traj), s3.cuda_stream), 'posterior(edge)')
        last = step == 0 or i + 1 >= st.n_steps
        if pos_guidance_opt:
            # the guidance below (lane 0, on the way to the next step's coordinates) reads the bond types just drawn: it waits for THIS point
            # of lane 3, not for what `prog_ahead` puts behind it (bond embedding, bond-node rows and attention of the next step)
            if getattr(st, 'bond_drawn', None) is None:
                st.bond_drawn = torch.cuda.Event()
            st.bond_drawn.record(s3)
        if not last:
            with torch.cuda.stream(s2):
                w.in_t_next.fill_(step - 1)
            eng._run(eng.prog_ahead)
        s = hip.stream_ptr()
        grad = None
        if pos_guidance_opt:                                         # diffusion.py:476-502 (reads the bond types just drawn on lane 3)
            torch.cuda.current_stream().wait_event(st.bond_drawn)
            grad = st.grad
            grad.zero_()
            for o in pos_guidance_opt:
                atom = o['type'] == 'atom_prox'
                if not atom and o['type'] != 'center_prox':
                    continue
                hip.check(lib.pg_guidance_grad(
                    plan.topo_ref, w.in_pos.data_ptr(), w.in_h_edge.data_ptr(), plan.lig_graph.data_ptr(),
                    plan.g_lig_off.data_ptr(), int(atom), float(o.get('min_d', 1.2)), float(o.get('max_d', 2.8)),
                    int(not atom), hip.ptr(st.gc), st.guidance_batch, st.cnt_ws.data_ptr(), st.mean_ws.data_ptr(),
                    st.gtmp.data_ptr(), s), 'guidance')
                grad += st.gtmp
        hip.check(lib.pg_posterior_position_ctx(
            w.in_pos.data_ptr(), x0_ctx.data_ptr(), plan.lig2ctx.data_ptr(), plan.lig_graph.data_ptr(), tb,
            pk.pos_tab[0].data_ptr(), pk.pos_tab[1].data_ptr(), pk.pos_tab[2].data_ptr(), hip.ptr(grad), None,
            st.seed, 2, step, N, plan.g_lig_off.data_ptr(), st.graph_key.data_ptr(),
            st.centers.data_ptr() if st.return_traj else None,
            w.in_pos.data_ptr(), tp(st.pos_traj), w.x[0].data_ptr(), st.x0_buf.data_ptr(), s), 'posterior(pos)')
        st.x0 = st.x0_buf
        st.cur = 1 - cur
        st.next_step = step - 1
        if last:
            eng.join_lanes((2, 3))

    @_on_model_device
    def finish_sampling(self, st):
        w, plan = st.eng.ws, st.plan
        if getattr(st, 'pipelined', False):
            st.eng.join_lanes((2, 3))              # (out_v / out_bond / the discrete trajectories are completed on the side lanes)
        return {'pred': [w.out_v.clone(), st.x0 + st.center_rows, w.out_bond.clone()],
                'traj': [st.node_traj, st.pos_traj, st.edge_traj],
                'lig_info': [st.num_atoms.to(self._device()), plan.batch_node, plan.edge_index, plan.batch_edge]}
