"""Fused AdamW over the model's flat parameter buffer (one kernel launch per step).

Same update rule and defaults as the optimizer the reference wires through LightningCLI
(train/main.py:13-14, train/configs/spectralv2.yaml:99-111): decoupled weight decay, no amsgrad.
"""
import torch

from . import _lib
from ._lib import ptr


class FusedAdamW:
    def __init__(self, model, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        self.model = model
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), (float(betas[0]), float(betas[1])), float(eps), float(weight_decay)
        self._m = self._v = self._step = None

    def _state(self):
        flat, grads = self.model.bind_grads()
        if self._m is None or self._m.device != flat.device or self._m.numel() != flat.numel():
            self._m = torch.zeros_like(flat)
            self._v = torch.zeros_like(flat)
            self._step = torch.zeros((), dtype=torch.int64, device=flat.device)
        return flat, grads

    def zero_grad(self, set_to_none=False):
        _, grads = self._state()
        grads.zero_()

    def step(self, grad_scale=1.0):
        flat, grads = self._state()
        _lib.require_cuda(flat)
        _lib.call("noise_adamw_step", ptr(flat), ptr(grads), ptr(self._m), ptr(self._v), ptr(self._step), flat.numel(),
                  self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay, float(grad_scale), _lib.stream(),
                  nbytes=28 * flat.numel())
        self.model.model.weights_changed()     # the kernel wrote the parameters through a raw pointer

    def state_dict(self):
        self._state()
        return {"exp_avg": self._m.clone(), "exp_avg_sq": self._v.clone(), "step": self._step.clone(),
                "lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": self.weight_decay}

    def load_state_dict(self, sd):
        self._state()
        self._m.copy_(sd["exp_avg"])
        self._v.copy_(sd["exp_avg_sq"])
        self._step.copy_(sd["step"])
        self.lr, self.betas, self.eps, self.weight_decay = sd["lr"], tuple(sd["betas"]), sd["eps"], sd["weight_decay"]
