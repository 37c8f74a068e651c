"""Denoiser network containers with the reference's attribute names (models/uni_denoiser.py).

`forward` of the top-level module runs the HIP engine; the sub-layers are parameter holders (their
arithmetic is fused across layer boundaries inside csrc/seg_attn.hip, so they are not callable alone).
"""
import torch
from torch import nn

from .common import MLP, AngularEncoding, GaussianSmearing


class NodeUpdateLayer(nn.Module):
    """k/v/q MLPs of uni_denoiser.py:13-38 (out_fc=False in every shipped config)."""

    def __init__(self, input_dim, hidden_dim, output_dim, n_heads, edge_feat_dim, act_fn='relu', norm=True,
                 out_fc=True, direction_match=False):
        super().__init__()
        if out_fc or act_fn != 'relu' or not norm:
            raise NotImplementedError('phoregen_amd: only x2h_out_fc=False, act_fn=relu, norm=True (shipped configs)')
        self.input_dim, self.hidden_dim, self.output_dim, self.n_heads = input_dim, hidden_dim, output_dim, n_heads
        kv = input_dim * 2 + edge_feat_dim + (9 if direction_match else 0)
        self.hk_func = MLP(kv, output_dim, hidden_dim)
        self.hv_func = MLP(kv, output_dim, hidden_dim)
        self.hq_func = MLP(input_dim, output_dim, hidden_dim)
        self.edge_feat_dim = edge_feat_dim

    def forward(self, h, edge_feat, edge_index, e_w=None):
        """uni_denoiser.py:40-72.  Callable in the configuration the reference calls on its own -- the pharmacophore
        encoder (models/__init__.py:29-35, diffusion.py:186-191: fully connected graphs, scalar distance feature);
        the instances inside the denoiser layers are fused into the layer kernels and are not callable alone."""
        if self.edge_feat_dim != 1:
            raise NotImplementedError('phoregen_amd: this NodeUpdateLayer is fused into the denoiser layer kernels; only the '
                                      'pharmacophore-encoder form (edge_feat_dim=1) is callable on its own')
        from ..engine import phore_encoder_standalone
        return phore_encoder_standalone(self, h, edge_feat, edge_index, e_w)


class BondUpdateLayer(nn.Module):
    """uni_denoiser.py:75-99 with include_h_node=True."""

    def __init__(self, input_dim, hidden_dim, output_dim, n_heads, norm=True, act_fn='relu', include_h_node=False):
        super().__init__()
        if not include_h_node:
            raise NotImplementedError('phoregen_amd: h_node_in_bond_net=True only (shipped configs)')
        self.distance_expansion = GaussianSmearing()
        self.angle_expansion = AngularEncoding()
        kv = input_dim + 20 * 2 + 13 + input_dim * 2
        self.hk_func = MLP(kv, output_dim, hidden_dim)
        self.hv_func = MLP(kv, output_dim, hidden_dim)
        self.hq_func = MLP(input_dim * 2, output_dim, hidden_dim)


class PosUpdateLayer(nn.Module):
    """uni_denoiser.py:168-185."""

    def __init__(self, input_dim, hidden_dim, output_dim, n_heads, edge_feat_dim, act_fn='relu', norm=True,
                 direction_match=False):
        super().__init__()
        kv = input_dim * 2 + edge_feat_dim + (9 if direction_match else 0)
        self.xk_func = MLP(kv, output_dim, hidden_dim)
        self.xv_func = MLP(kv, n_heads, hidden_dim)
        self.xq_func = MLP(input_dim, output_dim, hidden_dim)


class AttentionLayerO2TwoUpdateNodeGeneral(nn.Module):
    """uni_denoiser.py:212-258 (registration order kept so state_dict order matches)."""

    def __init__(self, hidden_dim, n_heads, num_r_gaussian, edge_feat_dim, act_fn='relu', norm=True, r_min=0.,
                 r_max=10., include_h_node=False, x2h_out_fc=True, direction_match=False):
        super().__init__()
        if not direction_match:
            raise NotImplementedError('phoregen_amd: direction_match=True only (shipped configs)')
        self.hidden_dim, self.n_heads = hidden_dim, n_heads
        ef = num_r_gaussian * edge_feat_dim + edge_feat_dim
        self.distance_expansion = GaussianSmearing(r_min, r_max, num_gaussians=num_r_gaussian)
        self.lin_node = nn.Linear(hidden_dim, hidden_dim)
        self.node_layer_with_edge = NodeUpdateLayer(hidden_dim, hidden_dim, hidden_dim, n_heads, ef, act_fn, norm,
                                                    out_fc=x2h_out_fc, direction_match=True)
        self.node_layer_with_bond = NodeUpdateLayer(hidden_dim, hidden_dim, hidden_dim, n_heads, hidden_dim, act_fn,
                                                    norm, out_fc=x2h_out_fc)
        self.bond_layer = BondUpdateLayer(hidden_dim, hidden_dim, hidden_dim, n_heads, norm, act_fn, include_h_node)
        self.pos_layer_with_edge = PosUpdateLayer(hidden_dim, hidden_dim, hidden_dim, n_heads, ef, act_fn, norm,
                                                  direction_match=True)
        self.pos_layer_with_bond = PosUpdateLayer(hidden_dim, hidden_dim, hidden_dim, n_heads, hidden_dim, act_fn, norm)
        self.dire_embedding = nn.Linear(3, 9)


class UniTransformerO2TwoUpdateGeneralBond(nn.Module):
    """uni_denoiser.py:301-349; forward contract of :396-430."""

    def __init__(self, num_blocks, num_layers, hidden_dim, n_heads=1, k=32, num_bond_classes=1, num_r_gaussian=50,
                 edge_feat_dim=0, act_fn='relu', norm=True, cutoff_mode='radius', use_global_ew=True, r_max=10.,
                 x2h_out_fc=True, h_node_in_bond_net=False, direction_match=False):
        super().__init__()
        if (num_blocks, hidden_dim, n_heads, num_r_gaussian, edge_feat_dim, cutoff_mode) != (1, 128, 16, 20, 4, 'knn'):
            raise NotImplementedError('phoregen_amd kernels are specialised to num_blocks=1, hidden_dim=128, n_heads=16, '
                                      'num_r_gaussian=20, edge_feat_dim=4, cutoff_mode=knn (both shipped configs)')
        if k % 4 or k > 64:
            raise NotImplementedError('phoregen_amd: knn k must be a multiple of 4, <= 64')
        self.num_blocks, self.num_layers, self.hidden_dim, self.n_heads, self.k = num_blocks, num_layers, hidden_dim, n_heads, k
        self.distance_expansion = GaussianSmearing(0., r_max, num_gaussians=num_r_gaussian)
        self.edge_pred_layer = MLP(num_r_gaussian, 1, hidden_dim)
        self.base_block = nn.ModuleList([
            AttentionLayerO2TwoUpdateNodeGeneral(hidden_dim, n_heads, num_r_gaussian, edge_feat_dim, act_fn, norm,
                                                 r_max=r_max, x2h_out_fc=x2h_out_fc, include_h_node=h_node_in_bond_net,
                                                 direction_match=direction_match)
            for _ in range(num_layers)])

    def forward(self, h, x, group_idx, bond_index, h_bond, mask_ligand, mask_ligand_atom, batch, phore_norm=None,
                return_all=False):
        """Same contract as uni_denoiser.py:396-430: ctx-ordered h [N,128], x [N,3], bond_index [2,E] in ctx
        indices, h_bond [E,128], masks, batch, phore_norm [N_phore,3] -> {'x','h','h_bond'}."""
        if group_idx is not None:
            # the reference appends two more edge-type columns then (uni_denoiser.py:386-393), which changes the first-layer widths
            # of every knn MLP; both shipped configs pass None (diffusion.py:211)
            raise NotImplementedError('phoregen_amd: group_idx must be None (edge_feat_dim=4 kernels; both shipped configs)')
        if mask_ligand_atom is not None and mask_ligand_atom is not mask_ligand and \
                not torch.equal(mask_ligand_atom.bool(), mask_ligand.bool()):
            raise NotImplementedError('phoregen_amd: mask_ligand_atom must equal mask_ligand (diffusion.py:208-216 passes one mask)')
        # like the reference's module, differentiable when gradients are being recorded (the same kernels with their HIP adjoints on the
        # autograd tape, phoregen_amd/training.py); under torch.no_grad() the pre-built launch list of the sampler runs instead
        record = torch.is_grad_enabled() and (h.requires_grad or x.requires_grad or h_bond.requires_grad or
                                              any(p.requires_grad for p in self.parameters()))
        from ..engine import denoiser_forward_standalone
        return denoiser_forward_standalone(self, h, x, bond_index, h_bond, mask_ligand, batch, phore_norm, return_all, record=record)
