"""
Fused Adam over the model's flat parameter buffer: one HIP launch per step instead of one per tensor
(replaces torch.optim.Adam(model.parameters(), lr) of reference train.py:289; same defaults: betas (0.9, 0.999),
eps 1e-8, no weight decay, no amsgrad).  It is a torch.optim.Optimizer so LambdaLR (train.py:292) works unchanged.
"""
import torch

from . import _lib as L


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, model, lr=3e-4, betas=(0.9, 0.999), eps=1e-8):
        self.model = model
        model.flatten_parameters_()
        super().__init__(list(model.parameters()), dict(lr=lr, betas=betas, eps=eps))
        flat_p = model._flat[0]
        self.exp_avg = torch.zeros_like(flat_p)
        self.exp_avg_sq = torch.zeros_like(flat_p)
        self.step_count = 0

    def zero_grad(self, set_to_none=False):
        # gradients live in the flat buffer; the backward pass accumulates into it, so it is cleared here
        self.model.flatten_parameters_()
        self.model._flat[1].zero_()
        for p, g in zip(self.model._enumerate()['plist'], self.model._flat[3]):
            if p.grad is not g:
                p.grad = g

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0):
        self.model.flatten_parameters_()
        flat_p, flat_g = self.model._flat[0], self.model._flat[1]
        if self.exp_avg.device != flat_p.device or self.exp_avg.data_ptr() == 0:
            self.exp_avg = torch.zeros_like(flat_p)
            self.exp_avg_sq = torch.zeros_like(flat_p)
        g = self.param_groups[0]
        self.step_count += 1
        L.call('srvp_adam', L.ptr(flat_p), L.ptr(flat_g), L.ptr(self.exp_avg), L.ptr(self.exp_avg_sq), flat_p.numel(),
               float(g['lr']), float(g['betas'][0]), float(g['betas'][1]), float(g['eps']), self.step_count,
               float(grad_scale), L.stream())
        # parameters changed behind autograd's back: the packed bf16 weights must be refreshed
        self.model._pack_version = None
        return None

    def state_dict(self):
        sd = super().state_dict()
        sd['fused'] = dict(step=self.step_count, exp_avg=self.exp_avg, exp_avg_sq=self.exp_avg_sq)
        return sd

    def load_state_dict(self, sd):
        fused = sd.pop('fused', None)
        super().load_state_dict(sd)
        if fused is not None:
            self.step_count = fused['step']
            self.exp_avg.copy_(fused['exp_avg'])
            self.exp_avg_sq.copy_(fused['exp_avg_sq'])
