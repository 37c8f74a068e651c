"""Training helpers with the reference's import surface (`from models.model_utils import EMA`, run/run.py:9,61,138-139).

`EMA(beta, parameters)` keeps shadow copies of the parameters; `update_model_average(model)` moves them towards the
current weights, `shadow = beta * shadow + (1 - beta) * current` (models/model_utils.py:21-42).  Here the update is one
fused multi-tensor lerp on the device instead of a Python loop over 641 tensors."""
import torch


class EMA:
    def __init__(self, beta, parameters):
        self.beta = beta
        self.shadow_params = [p.detach().clone() for p in parameters]

    @torch.no_grad()
    def update_model_average(self, current_model):
        cur = [p.detach() for p in current_model.parameters()]
        if len(cur) != len(self.shadow_params):
            raise ValueError('EMA: the model has a different number of parameters than the shadow copy')
        torch._foreach_lerp_(self.shadow_params, cur, 1.0 - self.beta)

    def update_average(self, old, new):
        return new if old is None else old * self.beta + (1 - self.beta) * new

    @torch.no_grad()
    def copy_to(self, model):
        """Load the averaged weights into `model` (evaluation with EMA weights)."""
        for p, s in zip(model.parameters(), self.shadow_params):
            p.copy_(s)

    def state_dict(self):
        return {'beta': self.beta, 'shadow_params': self.shadow_params}

    def load_state_dict(self, state_dict, device):
        self.beta = state_dict['beta']
        self.shadow_params = [t.to(device) for t in state_dict['shadow_params']]
