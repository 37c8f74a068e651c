"""Config plumbing: an EasyDict-compatible attribute dict and the `model:` block both shipped YAMLs
share (configs/train_lig-phore.yml:1-56 == configs/train_dock-cpx-phore.yml:1-56)."""
import copy

import yaml


class AttrDict(dict):
    """Recursive attribute dict; `getattr(cfg, name, default)` works as with easydict.EasyDict."""

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            v = AttrDict(v)
        elif isinstance(v, (list, tuple)):
            v = type(v)(AttrDict(x) if isinstance(x, dict) and not isinstance(x, AttrDict) else x for x in v)
        super().__setitem__(k, v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    __setattr__ = __setitem__

    def __deepcopy__(self, memo):
        return AttrDict(copy.deepcopy(dict(self), memo))


_MODEL_YAML = """
name: diffusion
num_atom_classes: 12
num_bond_classes: 6
lig_feat_dim: 12
phore_feat_dim: 16
hidden_dim: 128
bond_diffusion: True
bond_net_type: lin
bond_len_loss: False
count_pred_type: boundary
loss_weight: [1, 100, 100]
count_factor: 1
hp_emb_with_pos: True
diff:
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
denoiser:
  name: uni_node_edge
  num_blocks: 1
  num_layers: 6
  hidden_dim: 128
  n_heads: 16
  knn: 32
  edge_feat_dim: 4
  num_r_gaussian: 20
  act_fn: relu
  norm: True
  cutoff_mode: knn
  r_max: 10.
  x2h_out_fc: False
  h_node_in_bond_net: True
  direction_match: True
"""


def default_model_config(data_name='zinc_300'):
    """`config.model` as sample_all.py hands it to PhoreDiff: phore_feat_dim 16 -> 18 for zinc_300/pdbbind
    (sample_all.py:41-43)."""
    cfg = AttrDict(yaml.safe_load(_MODEL_YAML))
    if data_name in ('zinc_300', 'pdbbind'):
        cfg.phore_feat_dim += 2
    return cfg


def load_config(path):
    with open(path) as f:
        return AttrDict(yaml.safe_load(f))
